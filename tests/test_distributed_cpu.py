"""CPU, gloo, world_size 2: the data-parallel glue of the path -- batch sharding by sample,
max-over-ranks timing (bench.py's aggregation), the bucketed gradient all-reduce (post-backward
and overlapped-with-backward forms, parameters unused on one rank included) and plain torch
DDP around the registry modules -- checked against the single-process result on the full batch.
(reference: MMDistributedDataParallel in apis/train.py:222-230)"""
import importlib
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _neck(mods):
    torch.manual_seed(0)
    return mods.OutdoorImVoxelNeck(in_channels=4, out_channels=8,
                                   norm_cfg=dict(type='GN', num_groups=2))  # per-sample norm


def _worker(rank, world, port, q, mode):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    par = importlib.import_module('depth-from-motion_amd.parallel')
    mods = importlib.import_module('depth-from-motion_amd.modules')
    importlib.import_module('depth-from-motion_amd.group_norm').allow_cpu_reference(True)
    neck = _neck(mods)
    x = torch.randn(6, 4, 5, 4, 12, generator=torch.Generator().manual_seed(1))
    lo, hi = par.shard_range(6, rank, world)
    extra = None
    if mode == 'unused':
        # a parameter only rank 0 uses: rank 1 has no gradient for it
        extra = torch.nn.Parameter(torch.ones(3))
    params = list(neck.parameters()) + ([extra] if extra is not None else [])
    info = 0
    if mode == 'ddp':
        model = torch.nn.parallel.DistributedDataParallel(neck, broadcast_buffers=False)
        out = model(x[lo:hi])[0]
        out.square().mean().backward()  # equal shards: DDP's mean over ranks == global mean
    else:
        reducer = par.GradientBucketReducer(params, bucket_bytes=4096) if mode in ('overlap', 'unused') \
            else None
        out = neck(x[lo:hi])[0]
        # DDP semantics: mean over the GLOBAL batch = average over ranks of per-rank means
        # only if shards are equal; weight by shard size to be exact
        loss = out.square().mean() * (hi - lo) * world / 6.0
        if extra is not None and rank == 0:
            loss = loss + (extra * torch.tensor([1.0, 2.0, 3.0])).sum()
        loss.backward()
        if reducer is not None:
            info = reducer.launched_during_backward
            nb = reducer.finalize()
            # a second step through the same reducer gives the same result
            for p in params:
                p.grad = None
            loss2 = neck(x[lo:hi])[0].square().mean() * (hi - lo) * world / 6.0
            if extra is not None and rank == 0:
                loss2 = loss2 + (extra * torch.tensor([1.0, 2.0, 3.0])).sum()
            loss2.backward()
            reducer.finalize()
        else:
            nb = par.allreduce_gradients(params, bucket_bytes=4096)
        info = (info, nb)
    t, every = par.gather_rank_times(1.0 + rank)
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    # plain bytes through the queue (a shared-memory tensor handle dies with this process)
    q.put((rank, (lo, hi), info, (t, every, par.max_over_ranks(1.0 + rank)), flat.detach().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def _run(mode):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda r: r[0])
    res = [r[:4] + (torch.from_numpy(r[4]),) for r in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _reference():
    mods = importlib.import_module('depth-from-motion_amd.modules')
    gn = importlib.import_module('depth-from-motion_amd.group_norm')
    prev = gn.allow_cpu_reference(True)
    try:
        neck = _neck(mods)
        x = torch.randn(6, 4, 5, 4, 12, generator=torch.Generator().manual_seed(1))
        neck(x)[0].square().mean().backward()
    finally:
        gn.allow_cpu_reference(prev)
    return torch.cat([p.grad.reshape(-1) for p in neck.parameters()])


def test_two_rank_gloo_matches_single_process():
    res = _run('post')
    assert [r[1] for r in res] == [(0, 3), (3, 6)]
    assert all(r[2][1] > 1 for r in res)           # several buckets were exercised
    # bench.py's aggregation: max over ranks is the job's time, every rank's time is kept
    assert all(r[3] == (2.0, [1.0, 2.0], 2.0) for r in res)
    assert torch.equal(res[0][4], res[1][4])       # identical gradients on both ranks
    torch.testing.assert_close(res[0][4], _reference(), rtol=1e-4, atol=1e-6)


def test_overlapped_bucket_reducer_matches_single_process():
    res = _run('overlap')
    assert all(r[2][1] > 1 for r in res)
    # buckets were handed to the collective while backward was still running
    assert all(r[2][0] >= 1 for r in res)
    assert torch.equal(res[0][4], res[1][4])
    torch.testing.assert_close(res[0][4], _reference(), rtol=1e-4, atol=1e-6)


def test_parameter_unused_on_one_rank_does_not_desync():
    """Round-1 advisor finding: buckets built from `p.grad is not None` differ between ranks
    when a parameter is unused on one of them.  Both reducers bucket over requires_grad."""
    res = _run('unused')
    assert torch.equal(res[0][4], res[1][4])
    ref = _reference()
    torch.testing.assert_close(res[0][4][:-3], ref, rtol=1e-4, atol=1e-6)
    # rank 0's gradient (1,2,3) averaged with rank 1's zeros
    torch.testing.assert_close(res[0][4][-3:], torch.tensor([0.5, 1.0, 1.5]))


def test_torch_ddp_wraps_the_modules_unchanged():
    res = _run('ddp')
    assert torch.equal(res[0][4], res[1][4])
    torch.testing.assert_close(res[0][4], _reference(), rtol=1e-4, atol=1e-6)


def test_hip_group_norm_refuses_cpu_tensors_by_default():
    gn = importlib.import_module('depth-from-motion_amd.group_norm')
    m = gn.HipGroupNorm(2, 4)
    try:
        m(torch.zeros(1, 4, 2, 2, 2))
    except RuntimeError as e:
        assert 'no CPU path' in str(e)
    else:
        raise AssertionError('HipGroupNorm ran on a CPU tensor without the test-only opt-in')


def test_shard_range_covers_the_batch():
    par = importlib.import_module('depth-from-motion_amd.parallel')
    for n, w in ((64, 8), (10, 4), (3, 8), (8, 1)):
        spans = [par.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert [par.shard_range(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]


# ---------------------------------------------------------------------------------------------
# bench.py's multi-process protocol (parallel.BenchJob), driven with a stub step under gloo:
# the code bench.py runs per rank on RCCL -- seeds / shards, the barrier-bracketed timed region with the
# MAX over ranks, the agreement on one launch shape, the whole-job value, the reducer A/B switch
# ---------------------------------------------------------------------------------------------
def _bench_worker(rank, world, port, q):
    import time
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    par = importlib.import_module('depth-from-motion_amd.parallel')
    job = par.BenchJob(rank, world)
    calls = []

    def step():  # rank 1 is the straggler
        calls.append(1)
        time.sleep(0.01 if rank == 0 else 0.03)
    job_s, every = job.timed_steps(step, 5)
    # the ranks time the two candidate launch shapes differently: rank 0 prefers 'a', rank 1 'b' strongly
    local = {'a': 1.0, 'b': 1.2} if rank == 0 else {'a': 2.0, 'b': 0.9}
    key, agreed = job.agree_fastest(local)
    # the reducer's A/B switch: a step with the exchange disabled leaves .grad local and launches nothing
    p = torch.nn.Parameter(torch.ones(4) * (rank + 1))
    red = par.GradientBucketReducer([p])
    red.enabled = False
    (p * 2.0).sum().backward()
    assert red.finalize() == 0 and red.launched_during_backward == 0
    local_grad = p.grad.clone()
    red.enabled = True
    p.grad = None
    (p * (rank + 1.0)).sum().backward()
    red.finalize()
    q.put((rank, len(calls), job_s, every, key, agreed, job.seed(7), job.shard(9), job.value(8, 5, job_s),
           local_grad.tolist(), p.grad.tolist()))
    dist.destroy_process_group()


def test_bench_job_protocol_under_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, s0, e0, k0, a0, seed0, sh0, v0, lg0, g0), (r1, n1, s1, e1, k1, a1, seed1, sh1, v1, lg1, g1) = res
    assert n0 == n1 == 5, 'exactly K timed steps on every rank'
    assert s0 == s1 == max(e0) and e0 == e1 and len(e0) == 2, 'job time = MAX over ranks, per-rank times kept'
    assert e0[1] >= 5 * 0.03 * 0.9 and e0[0] >= 5 * 0.01 * 0.9
    assert e0[0] < 0.8 * e0[1], "a rank's clock stops when ITS work is done, before the closing barrier"
    assert k0 == k1 == 'b' and a0 == a1 == {'a': 2.0, 'b': 1.2}, 'one launch shape for the whole job: MAX then min'
    assert (seed0, seed1) == (7, 1007) and (sh0, sh1) == ((0, 5), (5, 9))
    assert abs(v0 - 8 * 2 * 5 / s0) < 1e-9 and v0 == v1
    assert lg0 == [2.0] * 4 and lg1 == [2.0] * 4, 'exchange disabled: gradients stay local'
    assert g0 == g1 == [1.5] * 4, 'exchange on: the average over ranks of (1, 2)'


def test_bench_job_single_process_needs_no_process_group():
    par = importlib.import_module('depth-from-motion_amd.parallel')
    job = par.BenchJob()
    n = []
    s, every = job.timed_steps(lambda: n.append(1), 3)
    assert len(n) == 3 and every == [s] and job.agree_fastest({'x': 2.0, 'y': 1.0})[0] == 'y'
    assert job.value(8, 3, 2.0) == 12.0 and job.seed(5) == 5 and job.shard(8) == (0, 8)
    import pytest
    with pytest.raises(RuntimeError):
        par.BenchJob(0, 2)


# ---------------------------------------------------------------------------------------------
# `python bench.py --gpus N` with no launcher around it re-executes itself as N ranks
# (parallel.self_launch = tools/dist_train.sh:10-20 of the reference in one call); here: a stub script
# on gloo, two ranks, the census all-reduce that bench.py prints as `ranks_seen`
# ---------------------------------------------------------------------------------------------
_STUB = """
import importlib, json, os, sys
sys.path.insert(0, {root!r})
import torch.distributed as dist
assert os.environ['MASTER_ADDR'] == '127.0.0.1'
dist.init_process_group('gloo')
par = importlib.import_module('depth-from-motion_amd.parallel')
job = par.BenchJob(int(os.environ['RANK']), int(os.environ['WORLD_SIZE']))
s, every = job.timed_steps(lambda: None, 3)
seen = job.ranks_seen()
if job.rank == 0:
    print(json.dumps(dict(ranks_seen=seen, collective=par.collective_library(), argv=sys.argv[1:],
                          ranks=len(every), value=job.value(8, 3, max(s, 1e-9)))))
dist.destroy_process_group()
"""


def test_self_launch_spawns_the_ranks_and_the_census_counts_them(tmp_path):
    import json
    par = importlib.import_module('depth-from-motion_amd.parallel')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'stub_bench.py'
    script.write_text(_STUB.format(root=root))
    # a stale launcher environment must not leak into the children
    os.environ['WORLD_SIZE'], os.environ['RANK'] = '7', '3'
    try:
        rc, out, err = par.self_launch(str(script), ['--gpus', '2', '--steps', '3'], 2, capture=True)
    finally:
        os.environ.pop('WORLD_SIZE'), os.environ.pop('RANK')
    assert rc == 0, err[-2000:]
    line = json.loads([l for l in out.splitlines() if l.startswith('{')][-1])
    assert line['ranks_seen'] == 2 and line['ranks'] == 2 and line['collective'] == 'gloo'
    assert line['argv'] == ['--gpus', '2', '--steps', '3']


def test_bench_py_launches_itself_when_asked_for_more_than_one_gpu():
    """the branch exists and sits BEFORE the GPU assertions (source check: bench.py needs a GPU to run)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'bench.py')).read()
    i, j = src.index("'WORLD_SIZE' not in os.environ"), src.index('assert torch.cuda.is_available()')
    assert i < j and 'self_launch' in src[i:j]
    par = importlib.import_module('depth-from-motion_amd.parallel')
    assert par.BenchJob().ranks_seen() == 1 and par.collective_library() is None


# ---------------------------------------------------------------------------------------------
# the N > 1 line carries the fields of the N = 1 line: rank 0 measures the reported extras (HBM counter
# passes, CPU baseline, secondary rows) on its own AFTER the closing barrier, through BenchJob.solo(),
# while the other ranks leave
# ---------------------------------------------------------------------------------------------
def _solo_worker(rank, world, port, q):
    import time
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    par = importlib.import_module('depth-from-motion_amd.parallel')
    job = par.BenchJob(rank, world)
    job_s, every = job.timed_steps(lambda: time.sleep(0.005), 3)
    seen = job.ranks_seen()
    if rank != 0:
        # what bench.py's other ranks do after the timed region: leave
        dist.destroy_process_group()
        q.put((rank, seen, None))
        return
    time.sleep(1.0)          # rank 1 is gone by now: any collective below would hang or fail
    solo = job.solo()
    n = []
    s, ev = solo.timed_steps(lambda: n.append(1), 4)
    key, agreed = solo.agree_fastest({'a': 2.0, 'b': 1.0})
    solo.barrier()
    q.put((rank, seen, dict(world=solo.world, rank=solo.rank, steps=len(n), every=len(ev), seen=solo.ranks_seen(),
                            key=key, seed=solo.seed(3), value=solo.value(8, 4, 2.0))))
    dist.destroy_process_group()


def test_rank0_measures_the_line_extras_alone_after_the_job():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_solo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (seen, solo)) for r, seen, solo in (q.get(timeout=120) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0] == 2
    assert res[0][1] == dict(world=1, rank=0, steps=4, every=1, seen=1, key='b', seed=3, value=16.0)


def test_bench_line_has_the_same_fields_at_every_world_size():
    """bench.py decides the extras of the headline line (api_build_dfm_cost, channels_last_variant, roofline.traffic,
    secondary, cpu_baseline) from the command line and the workload only; nothing after `if rank == 0:` in run()
    may look at the world size again (round 5's line dropped traffic / cpu_baseline / secondary at N > 1)."""
    import inspect
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    assert 'world' not in inspect.signature(bench.line_extras).parameters
    tail = inspect.getsource(bench.run).split('if rank == 0:', 1)[1]
    assert 'world == 1' not in tail and 'world > 1' not in tail.split('print(json.dumps(line)')[0]
    a = types.SimpleNamespace(workload='nstar', channels_last=False, traffic_bytes=None, no_traffic=False,
                              no_secondary=False, no_cpu_baseline=False)
    ex = bench.line_extras(a, bench.WORKLOADS['nstar'], 2, {}, 2)
    assert ex == ['api_build_dfm_cost', 'channels_last_variant', 'traffic', 'secondary', 'cpu_baseline']
    # the SMI load loop (400 extra launches on rank 0 only) is a one-GPU probe
    head = inspect.getsource(bench.run).split('if rank == 0:', 1)[0]
    assert 'if world == 1 and not args.no_smi' in head
