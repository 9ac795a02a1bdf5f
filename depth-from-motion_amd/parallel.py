"""Data-parallel glue of the path: the batch shards by sample,
one process per GPU, forward has NO collective; training adds one bucketed
gradient all-reduce per step (what MMDistributedDataParallel does in the
reference, apis/train.py:222-230) over RCCL (`backend='nccl'` on ROCm),
overlapped with the rest of backward.

Two ways to get that step, both covered by tests/test_distributed_cpu.py on gloo:
  * ``torch.nn.parallel.DistributedDataParallel`` wraps the registry modules unchanged (their
    autograd Functions produce ordinary ``.grad`` tensors);
  * ``GradientBucketReducer`` below: the same bucketing with bucket sizes chosen for xGMI.

Only torch.distributed is used; the same code runs on gloo for the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(num_samples, rank, world_size):
    """Contiguous slice of a global batch owned by `rank` (remainder to the
    first ranks), e.g. 64 samples on 8 ranks -> 8 each."""
    base, extra = divmod(num_samples, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _active():
    return dist.is_available() and dist.is_initialized()


def max_over_ranks(value, device=None):
    """Slowest-rank time (bench.py's wall clock)."""
    if not _active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rank_times(value, device=None):
    """(max over ranks, [every rank's value]): the job's time is the slowest rank's; the list
    exposes stragglers (bench.py prints it as ``per_rank_ms_per_step``)."""
    if not _active():
        return float(value), [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(every, t)
    vals = [float(v.item()) for v in every]
    return max(vals), vals


def _buckets_of(params, bucket_bytes):
    """Static bucket layout over ALL trainable parameters (not only those that happen to have a
    gradient on this rank), in reverse registration order -- roughly the order backward
    produces them -- split by size and dtype.  Identical on every rank by construction."""
    buckets, cur, cur_bytes = [], [], 0
    for p in reversed([p for p in params if p.requires_grad]):
        nbytes = p.numel() * p.element_size()
        if cur and (cur_bytes + nbytes > bucket_bytes or p.dtype != cur[0].dtype):
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(p)
        cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    return buckets


def allreduce_gradients(parameters, bucket_bytes=64 << 20, average=True):
    """Sum (then average) .grad over ranks in flat buckets, after backward (no overlap).

    The bucket layout covers every parameter with ``requires_grad`` -- a parameter that got no
    gradient on this rank (unused branch, frozen-by-config norm, empty shard) contributes zeros
    and receives the other ranks' sum -- so all ranks always reduce buffers of the same size in
    the same order.  Returns the number of buckets reduced."""
    params = list(parameters)
    if not params or not _active():
        return 0
    world = dist.get_world_size()
    buckets = _buckets_of(params, bucket_bytes)
    for bucket in buckets:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                          for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= world
        off = 0
        for p in bucket:
            piece = flat[off:off + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = piece.clone()
            else:
                p.grad.copy_(piece)
            off += p.numel()
    return len(buckets)


class GradientBucketReducer:
    """Bucketed gradient all-reduce overlapped with backward (DDP semantics).

    xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is per-link
    bound: few large buckets amortise the ring latency better than DDP's 25 MB default.  The
    KITTI student (~40 MB of gradients) goes out in two 32 MB buckets: the first leaves while
    the image backbone is still in backward.

    ``reducer = GradientBucketReducer(model.parameters())`` once; every step:
    ``loss.backward(); reducer.finalize()``.  A post-accumulate hook on each parameter copies
    its finished gradient into the bucket's flat buffer; when a bucket is complete -- and all
    earlier buckets have been launched, so every rank issues the collectives in the same order
    even if their graphs differ -- its ``all_reduce`` starts asynchronously (on RCCL: on the
    process group's own stream, next to the remaining backward kernels).  ``finalize`` launches
    what is left (parameters without a gradient on this rank send zeros), waits, averages and
    writes the results back into ``.grad``.
    """

    def __init__(self, parameters, bucket_bytes=32 << 20, average=True):
        self.params = [p for p in parameters if p.requires_grad]
        self.average = average
        self.buckets = _buckets_of(self.params, bucket_bytes)
        self._where = {}
        for bi, bucket in enumerate(self.buckets):
            off = 0
            for p in bucket:
                self._where[p] = (bi, off)
                off += p.numel()
        self._flat = [None] * len(self.buckets)
        self._ready = [0] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._copied = [[] for _ in self.buckets]  # events of the gradient copies into each bucket (CUDA)
        self._zeroed = [None] * len(self.buckets)  # event behind the last zero fill of each bucket (CUDA)
        self._next = 0  # first bucket not launched yet
        self._filled = set()
        self.launched_during_backward = 0
        self.enabled = True  # False: hooks and finalize() do nothing (a step without the exchange, for A/B timing)
        for bi in range(len(self.buckets)):  # allocated and zeroed now, not inside the first hook
            self._buffer(bi)
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]

    def _buffer(self, bi):
        if self._flat[bi] is None:
            b = self.buckets[bi]
            self._flat[bi] = torch.zeros(sum(p.numel() for p in b), dtype=b[0].dtype,
                                         device=b[0].device)
            self._mark_zeroed(bi)
        return self._flat[bi]

    def _mark_zeroed(self, bi):
        """the zero fill of a bucket runs on the stream that is current HERE; a hook on another stream (the mono
        stack's backward runs on DfMBackbone's side stream) must not copy its gradient in before the fill has
        executed -- hooks wait for this event"""
        buf = self._flat[bi]
        if buf.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(buf.device))
            self._zeroed[bi] = ev

    def _launch_ready(self, final=False):
        while self._next < len(self.buckets) and (final or
                                                  self._ready[self._next] == len(self.buckets[self._next])):
            bi = self._next
            buf = self._buffer(bi)
            if self._copied[bi]:
                cur = torch.cuda.current_stream(buf.device)
                for ev in self._copied[bi]:
                    cur.wait_event(ev)
                self._copied[bi] = []
            if _active():
                self._work[bi] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
            if not final:
                self.launched_during_backward += 1
            self._next += 1

    def _hook(self, p):
        if not self.enabled:
            return
        bi, off = self._where[p]
        if p in self._filled:  # a second backward before finalize(): gradients accumulate
            raise RuntimeError('GradientBucketReducer: call finalize() after every backward()')
        self._filled.add(p)
        buf = self._buffer(bi)
        if self._zeroed[bi] is not None:
            torch.cuda.current_stream(buf.device).wait_event(self._zeroed[bi])
        buf[off:off + p.numel()].copy_(p.grad.reshape(-1))
        if p.grad.is_cuda:
            # backward nodes run on the stream their forward ran on (DfMBackbone's two stacks use two): the copy
            # above is ordered on THIS hook's stream only -- the stream that launches the bucket waits for it
            ev = torch.cuda.Event()
            ev.record()
            self._copied[bi].append(ev)
        self._ready[bi] += 1
        self._launch_ready()

    def finalize(self):
        """Launch the remaining buckets, wait for all, write the averaged sums back."""
        if not self.enabled:
            return 0
        self._launch_ready(final=True)
        world = dist.get_world_size() if _active() else 1
        for bi, bucket in enumerate(self.buckets):
            if self._work[bi] is not None:
                self._work[bi].wait()
            buf = self._buffer(bi)
            if self.average and world > 1:
                buf /= world
            off = 0
            for p in bucket:
                piece = buf[off:off + p.numel()].view_as(p)
                if p.grad is None:
                    p.grad = piece.clone()
                else:
                    p.grad.copy_(piece)
                off += p.numel()
            buf.zero_()
            self._mark_zeroed(bi)
        n = len(self.buckets)
        self._ready = [0] * n
        self._work = [None] * n
        self._next = 0
        self._filled.clear()
        return n

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


class BenchJob:
    """bench.py's multi-process protocol (one process per GPU, launched by torch.distributed.run),
    factored out of the script so that a gloo world on CPU can drive exactly the same code with a
    stub step (tests/test_distributed_cpu.py): the rank / shard bookkeeping, the barrier-bracketed
    timed region with the MAX over ranks, the agreement on one launch shape, the whole-job value.

    ``sync``: the device synchronisation (``torch.cuda.synchronize`` on a GPU; a no-op on CPU)."""

    def __init__(self, rank=0, world=1, device=None, sync=None, solo=False):
        self.rank, self.world, self.device = int(rank), int(world), device
        self._sync = sync or (lambda: None)
        self._solo = bool(solo)
        if self._solo:
            return
        if self.world > 1 and not _active():
            raise RuntimeError('BenchJob(world > 1) needs an initialised torch.distributed process group')
        if _active() and dist.get_world_size() != self.world:
            raise RuntimeError(f'process group has {dist.get_world_size()} ranks, the job says {self.world}')

    def solo(self):
        """this rank on its own, whatever process group exists: a one-rank job whose methods issue NO
        collective.  bench.py's rank 0 measures the reported extras of the line (HBM counter passes, the CPU
        baseline, the secondary rows) with it AFTER the job's closing barrier, while the other ranks leave --
        so the line carries the same fields at every N."""
        return BenchJob(0, 1, self.device, self._sync, solo=True)

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def seed(self, base):
        """shards differ: every rank draws its synthetic batch from its own generator"""
        return int(base) + 1000 * self.rank

    def shard(self, global_batch):
        """this rank's contiguous slice [lo, hi) of a global batch (strong scaling)"""
        return shard_range(global_batch, self.rank, self.world)

    def timed_steps(self, step, steps, clock=None):
        """EXACTLY ``steps`` calls of ``step`` bracketed by barrier + device sync on both sides.
        Returns (job seconds = MAX over ranks, [every rank's seconds])."""
        import time
        clock = clock or time.perf_counter
        self.barrier()
        self._sync()
        t0 = clock()
        for _ in range(int(steps)):
            step()
        self._sync()
        # the rank's clock stops when ITS device is done; the closing barrier only keeps the ranks
        # together for what follows (the job's time is the MAX over ranks either way, and a fast rank's
        # own time is no longer padded with its wait for the straggler)
        elapsed = clock() - t0
        self.barrier()
        if self._solo:
            return float(elapsed), [float(elapsed)]
        return gather_rank_times(elapsed, self.device)

    def ranks_seen(self):
        """How many ranks the collective library actually reached: an all-reduce (SUM) of a one per rank
        over the job's process group (RCCL on a GPU node).  bench.py prints it next to ``n_gpus``."""
        if self._solo or (self.world == 1 and not _active()):
            return 1
        t = torch.ones(1, dtype=torch.int32, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    def agree_fastest(self, timings):
        """``timings``: {candidate: this rank's time}.  Every rank must launch the same shape: a
        candidate's time is its MAX over ranks, the choice is the minimum of those (ties: the first
        key in the dict's order, identical on every rank).  Returns (key, {candidate: agreed time})."""
        keys = list(timings)
        vals = [float(timings[k]) for k in keys]
        if self.world > 1:
            t = torch.tensor(vals, dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            vals = [float(v) for v in t]
        agreed = dict(zip(keys, vals))
        return min(keys, key=lambda k: agreed[k]), agreed

    def value(self, units_per_rank_per_step, steps, job_seconds):
        """whole-job throughput (weak scaling: every rank processes ``units_per_rank_per_step``)"""
        return units_per_rank_per_step * self.world * steps / job_seconds


def collective_library():
    """'rccl x.y.z' when the process group runs on the nccl (= RCCL on ROCm) backend, else the backend's name"""
    if not _active():
        return None
    backend = dist.get_backend()
    if backend == 'nccl':
        try:
            v = torch.cuda.nccl.version()
            return 'rccl ' + '.'.join(str(x) for x in (v if isinstance(v, tuple) else (v,)))
        except Exception:  # noqa: BLE001 -- a version string is informational
            return 'rccl'
    return str(backend)


def free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(script, argv, nproc, port=None, extra_env=None, capture=False):
    """Run ``python script argv...`` as ``nproc`` ranks of ONE node under ``torch.distributed.run`` -- what
    ``tools/dist_train.sh:10-20`` does for the reference (``python -m torch.distributed.launch
    --nproc_per_node=$GPUS``) -- so that ``python bench.py --gpus N`` is one self-contained command.  The
    children see RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and take the ordinary per-rank path; rendezvous on
    127.0.0.1 (the container hostname may not resolve).  Returns the launcher's exit code (and its stdout
    when ``capture``)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or nproc) // int(nproc))))
    env.update(extra_env or {})
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={int(nproc)}',
           '--master-addr', '127.0.0.1', '--master-port', str(port or free_port()), script] + list(argv)
    if capture:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        return r.returncode, r.stdout, r.stderr
    return subprocess.call(cmd, env=env)
