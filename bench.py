#!/usr/bin/env python
"""bench.py -- cost-volumes/sec of the plane-sweep build on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE
JSON line on rank 0.  For N>1 the driver launches it under
``python -m torch.distributed.run`` (one rank per GPU, RCCL).

A "step" = one pass of the hot path (dfm_plane_sweep_fwd: re-block the two
feature maps + write the (B,2C,D,H,W) volume) over one batch of B=8 synthetic
KITTI-shape frame pairs that are already resident in HBM.  The batch shards
over ranks with no data-path collective (weak scaling: B=8 per GPU).

Workloads (``--workload``):
  nstar (default, the BASELINE.json metric): B=8, C=256, D=112, 94x311, bf16,
        feat_sample_factor=4, cost_sample_factor=1
  nstar_aug : SURVEY.md 8d's second N* run: the same with flip=True, crop_offset=(11,55),
        scale_factor=1.03 (the general-geometry code path of the same kernel)
  kitti : config K of configs/dfm/dfm_r34_1x8_kitti-3d-3class.py: B=8 (the
        config runs 1/GPU), C=32, 320x1280 fp32, csf=4, D=72 -> (64,72,80,320)
  kitti_nhwc : the same sweep with channels_last (NHWC) feature maps, what the channels_last
        SPPUNetNeck emits: sampled in place (dfm_plane_sweep_fwd_from_nhwc), no pack pass
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# the reference's own build_dfm_cost on PyTorch-CPU, timed in the build container by
# tools/ref_cpu_timing.py (/root/reference does not exist on the GPU box)
REF_CPU_FILE = os.path.join(ROOT, 'profiles', 'r02_reference_cpu_timing.json')


def schedule_key(opts):
    """canonical name of a launch configuration (dict of dfm_sweep_opts fields; 0 = library default)"""
    o = opts or {}
    ppl = o.get('points_per_lane') or 8
    key = 'lanes{}_ppl{}_planes{}_chunk{}_{}_cut{}'.format(
        o.get('lanes_per_workgroup') or 256, ppl, o.get('planes_per_workgroup') or 2,
        o.get('bands_per_chunk') or 1, 'serial' if o.get('pipeline') == 1 else 'pipelined',
        o.get('store_align_points') or 64)
    return key + ('_8Bstores' if ppl == 4 and o.get('pair_stores') == 2 else '')


def sweep_bench_cfg(opts):
    """the same configuration as a tools/sweep_bench argument"""
    o = opts or {}
    names = (('kernel', 'kernel'), ('lanes_per_workgroup', 'lanes'), ('lds_kib', 'lds'),
             ('blocks_per_group', 'bpg'), ('planes_per_workgroup', 'planes'), ('bands_per_chunk', 'chunk'),
             ('points_per_lane', 'ppl'), ('pipeline', 'pipe'), ('store_align_points', 'align'),
             ('pair_stores', 'pair'))
    items = [f'{short}={o[k]}' for k, short in names if o.get(k)]
    return ','.join(items) or 'default'


# secondary rows whose step the torch-free harness can replay for the counter passes: bench workload -> (harness
# workload, mode, kernel-name filter)
TRAFFIC_ROWS = {
    'sweep_bwd': ('nstar', 'bwd', ('sweep_bwd', 'gather_fit', 'fillBuffer')),
    'sweep_bwd_kitti': ('kitti', 'bwd', ('sweep_bwd', 'gather_fit', 'fillBuffer')),
    'kitti_nhwc': ('kitti', 'nhwc', ('sweep_', 'pack_')),
}


def measure_traffic(workload, opts, timeout=90):
    """HBM bytes per launch of the configuration that ran, measured NOW on this part: two separate
    rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; no trace domains beside --pmc) over the torch-free
    tools/sweep_bench, (2 x FETCH_SIZE + WRITE_SIZE) x 1024 over the launch's kernels
    (tools/pmc_traffic.py; the x2 is MI355X_MICROARCH.md's gfx950 correction for wide coalesced reads).
    None when rocprofv3 or the harness is missing or a pass fails -- never a number from another run.
    Round 6: also the secondary rows of TRAFFIC_ROWS (the plane-sweep backward at N* and at config K, the strided
    forward from NHWC maps), replayed by the harness's --mode bwd / nhwc."""
    import shutil
    import subprocess
    row = TRAFFIC_ROWS.get(workload)
    if (workload not in ('nstar', 'nstar_aug', 'kitti') and row is None) or not shutil.which('rocprofv3'):
        return None
    exe = os.path.join(ROOT, 'tools', 'sweep_bench')
    try:
        src = os.path.join(ROOT, 'tools', 'sweep_bench.cpp')
        if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
            subprocess.run(['hipcc', '--offload-arch=gfx950', '-O2', '-std=c++17', '-I', os.path.join(ROOT, 'include'),
                            src, '-L', os.path.join(ROOT, 'depth-from-motion_amd', 'lib'), '-ldfm_hip',
                            '-Wl,-rpath,$ORIGIN/../depth-from-motion_amd/lib', '-o', exe],
                           check=True, timeout=120, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import pmc_traffic
        cfg = sweep_bench_cfg(opts)
        hw, mode, names = (workload, 'fwd', None) if row is None else row
        fetch = pmc_traffic.one_pass('FETCH_SIZE', cfg, hw, '/tmp/bench_pmc', timeout, mode)
        write = pmc_traffic.one_pass('WRITE_SIZE', cfg, hw, '/tmp/bench_pmc', timeout, mode)
        # the library's kernels only (the harness also fills and checksums the volume)
        keep = pmc_traffic.is_library_kernel if names is None else (lambda k: any(t in k for t in names))
        fetch = {k: v for k, v in fetch.items() if keep(k)}
        write = {k: v for k, v in write.items() if keep(k)}
        kernels = set(fetch) | set(write)
        total = sum(2 * fetch.get(k, 0.0) + write.get(k, 0.0) for k in kernels) * 1024
        return {'hbm_bytes_per_launch': total,
                'fetch_bytes_x2': sum(2 * v for v in fetch.values()) * 1024,
                'write_bytes': sum(write.values()) * 1024,
                'how': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/sweep_bench '
                       f'{cfg}' + ('' if row is None else f' --mode {mode}') + ', this part, this run'} if total > 0 else None
    except Exception:  # a failed pass must not fail the bench
        return None


KITTI_P2 = np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791],
                     [0, 0, 1, 0.002745884], [0, 0, 0, 1]], np.float32)

# secondary workloads (other rows of SURVEY.md 8a), reported with the same JSON shape
SECONDARY = ('waymo', 'depth_head', 'f2v', 'group_norm', 'sweep_bwd', 'sweep_bwd_kitti', 'sweep_bwd_kitti_cl', 'voxel_sample', 'voxel_sample_bwd',
             'backbone', 'backbone_train', 'neck', 'dfm_neck', 'stereo_infer', 'stereo_train',
             # the same rows in the layout the bf16 NDHWC pipeline hands them (channels-last sources
             # sampled in place, channels-last results for the MFMA convolutions that follow)
             'waymo_cl', 'depth_head_bf16', 'f2v_cl', 'f2v_bwd', 'group_norm_cl')
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak of MI355X (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: B, C, H, W, D, fsf, csf, crop, dtype
    'nstar': dict(B=8, C=256, H=94, W=311, D=112, fsf=4, csf=1, crop=(0, 0), dtype='bf16',
                  dmin=2.0, dmax=59.6, flip=False, scale=1.0),
    'nstar_aug': dict(B=8, C=256, H=94, W=311, D=112, fsf=4, csf=1, crop=(11, 55), dtype='bf16',
                      dmin=2.0, dmax=59.6, flip=True, scale=1.03),
    'kitti': dict(B=8, C=32, H=320, W=1280, D=72, fsf=1, csf=4, crop=(0, 55), dtype='f32',
                  dmin=2.0, dmax=59.6, flip=False, scale=1.0),
    # N* with ONE -0 per map and sample (a value the matrix-core unpack is not exact for): the tiles that stage
    # that feature row take the VALU unpack, every other tile the matrix core (per-row flags of the pack pass;
    # rounds 3-4 flipped the whole launch); `nstar_negzero_all`: a whole channel of -0, every tile on the VALU body
    'nstar_negzero': dict(B=8, C=256, H=94, W=311, D=112, fsf=4, csf=1, crop=(0, 0), dtype='bf16',
                          dmin=2.0, dmax=59.6, flip=False, scale=1.0, plant='one'),
    'nstar_negzero_all': dict(B=8, C=256, H=94, W=311, D=112, fsf=4, csf=1, crop=(0, 0), dtype='bf16',
                              dmin=2.0, dmax=59.6, flip=False, scale=1.0, plant='channel'),
    # the same sweep fed by the channels_last (NHWC) SPPUNetNeck: maps sampled in place, no pack pass,
    # reference-layout volume out
    'kitti_nhwc': dict(B=8, C=32, H=320, W=1280, D=72, fsf=1, csf=4, crop=(0, 55), dtype='f32',
                       dmin=2.0, dmax=59.6, flip=False, scale=1.0, nhwc=True),
}


def poses(batch, seed):
    """forward t_z~U(-1.5,-0.3) m, lateral t_x~U(-0.1,0.1), yaw~U(-2,2) deg"""
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(batch):
        yaw, tx, tz = np.radians(rng.uniform(-2, 2)), rng.uniform(-0.1, 0.1), rng.uniform(-1.5, -0.3)
        c, s = np.cos(yaw), np.sin(yaw)
        out.append([[c, 0, s, tx], [0, 1, 0, 0], [-s, 0, c, tz], [0, 0, 0, 1]])
    return np.asarray(out, np.float32)


def depth_planes(num, dmin, dmax):
    return np.array([dmin + (i + 0.5) * ((dmax - dmin) / num) for i in range(num)], np.float32)


def algorithmic_bytes(w, elem):
    """s*(2*C*H_in*W_in + 2*C*D*H_out*W_out) per volume."""
    h_out, w_out = round(w['H'] / w['csf']), round(w['W'] / w['csf'])
    return elem * (2 * w['C'] * w['H'] * w['W'] + 2 * w['C'] * w['D'] * h_out * w_out)


def cpu_baseline(w, budget_s=12.0):
    """The oracle (a C port of the reference algorithm, OpenMP over rows) timed
    on this host's cores on a BOUNDED sample: one sample's volume restricted
    to a channel subset, extrapolated linearly in C (the work is exactly
    proportional to the channel count)."""
    from oracle import dfm_oracle as orc
    lib = orc.lib()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    # (the host's cores, not omp_get_max_threads(): torchrun exports OMP_NUM_THREADS=1 to its ranks)
    lib.dfm_oracle_set_threads(cores)
    rng = np.random.RandomState(0)
    P = KITTI_P2
    Pinv = torch.inverse(torch.from_numpy(P.copy())).numpy()
    T = poses(1, 2)[0]
    depths = depth_planes(w['D'], w['dmin'], w['dmax'])
    prm = orc.sweep_params(w['H'], w['W'], w['D'], w['fsf'], w['csf'], P, Pinv, T, (375, 1242),
                           w['flip'], w['crop'], w['scale'])

    def run(c_sub):
        cur = rng.randn(c_sub, w['H'], w['W']).astype(np.float32)
        prev = rng.randn(c_sub, w['H'], w['W']).astype(np.float32)
        out = np.zeros((2 * c_sub, prm.D, prm.h_out, prm.w_out), np.float32)  # pre-faulted
        t0 = time.perf_counter()
        lib.dfm_oracle_build_dfm_cost(ctypes.byref(prm), orc._vp(depths), orc._vp(cur),
                                      orc._vp(prev), ctypes.c_int(c_sub), orc._vp(out))
        return time.perf_counter() - t0

    run(2)  # warm the thread pool / page in the library
    t1 = run(4)
    c_sub = int(max(4, min(w['C'], round(4 * 3.0 / max(t1, 1e-6)))))  # ~3 s per repetition
    reps, total = 0, 0.0
    while total < budget_s and reps < 16:
        total += run(c_sub)
        reps += 1
    t = total / reps
    sec_per_volume = t * (w['C'] / c_sub)
    res = {
        'value': 1.0 / sec_per_volume,
        'unit': 'cost-volumes/s',
        'cores': cores,
        'kind': 'port',
        'sample': f'1 sample, {c_sub} of {w["C"]} channels x all D={w["D"]} planes x '
                  f'{prm.h_out}x{prm.w_out}, fp32, {reps} repetitions, {total:.1f} s of wall time on '
                  f'{cores} threads, scaled by C',
    }
    try:
        # the reference's own PyTorch-CPU build_dfm_cost (tools/ref_cpu_timing.py, build container)
        with open(REF_CPU_FILE) as f:
            res['reference_torch_cpu'] = json.load(f)
    except (OSError, ValueError):
        pass
    try:
        ts = torch_cpu_sampling(w, prm, depths, orc)
    except Exception as e:  # never fail the bench line over a reported baseline
        res['torch_cpu_grid_sample_this_box'] = {'error': repr(e)[:200]}
        return res
    # The line's CPU baseline is the REFERENCE'S whole compute through the REFERENCE'S library on this box's host
    # cores: every torch call of build_dfm_cost (dfm_backbone.py:217-314) on PyTorch-CPU, one sample at full C.
    # (/root/reference cannot travel to the GPU box, so the function's own file is timed in the build
    # container only: `reference_torch_cpu`.)  The C port of the oracle -- faster than torch on the same
    # cores -- is reported beside it.
    port = {k: res[k] for k in ('value', 'unit', 'cores', 'sample')}
    port['what'] = 'oracle/dfm_oracle.c (C restatement of the reference algorithm, OpenMP over rows)'
    out = {'value': ts['value'], 'unit': ts['unit'], 'cores': ts['threads'], 'kind': 'reference',
           'sample': ts['sample'], 'what': ts['what'], 'c_port': port}
    if 'reference_torch_cpu' in res:
        out['reference_torch_cpu'] = res['reference_torch_cpu']
    return out


def torch_cpu_sampling(w, prm, depths, orc, budget_s=14.0):
    """The reference's WHOLE ``build_dfm_cost`` (dfm_backbone.py:217-314) on PyTorch-CPU with this host's cores:
    lattice + meshgrid, un-projection / re-projection (the ~25 element-wise and matmul calls of :247-294), the
    two ``F.grid_sample`` calls and the channel ``cat`` -- every library call the reference issues, through
    ``oracle/dfm_torch_baseline.py`` (checked bit for bit against the reference-generated fixtures by the CPU
    tests; /root/reference itself does not exist on the GPU box).  ONE sample at the workload's FULL channel
    count, nothing scaled -- when the host has the memory for it (N*: 27 GB of fp32 for the two halves and
    their cat); otherwise a channel subset, scaled by C, and the line says so.  A reported baseline."""
    from oracle import dfm_torch_baseline as tb
    threads = os.cpu_count() or 1
    try:
        threads = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    torch.set_num_threads(threads)   # (torchrun exports OMP_NUM_THREADS=1 to its ranks)
    C, H, W = w['C'], w['H'], w['W']
    need = 4 * 4 * C * prm.D * prm.h_out * prm.w_out * 1.15   # a, b, cat(a, b) + slack, bytes
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 0
    c_run = C if need < 0.6 * avail else int(max(2, min(C, C * 0.6 * avail / need)))
    dt = torch.from_numpy(np.asarray(depths, np.float32))
    P = torch.from_numpy(KITTI_P2.copy())[None]
    T = torch.from_numpy(poses(1, 2))

    def run(c_sub):
        cur = torch.randn(1, c_sub, H, W)
        prev = torch.randn(1, c_sub, H, W)
        t0 = time.perf_counter()
        with torch.no_grad():
            vol = tb.build_dfm_cost(cur, prev, dt, w['fsf'], w['csf'], P, T, (375, 1242), w['flip'],
                                    tuple(w['crop']), w['scale'])
        dtm = time.perf_counter() - t0
        assert vol.shape == (1, 2 * c_sub, prm.D, prm.h_out, prm.w_out)
        return dtm
    t1 = run(min(2, c_run))   # pages the library in, warms the thread pool
    if c_run < C:
        c_run = int(max(2, min(c_run, round(2 * (budget_s / 3.0) / max(t1, 1e-6)))))
    run(c_run)                # first touch of the large allocations
    reps, total = 0, 0.0
    while reps < 2 or (total < budget_s and reps < 8):
        total += run(c_run)
        reps += 1
    sec_per_volume = total / reps * (C / c_run)
    return {'value': 1.0 / sec_per_volume, 'unit': 'cost-volumes/s', 'threads': threads,
            'what': 'the reference\'s build_dfm_cost in full (dfm_backbone.py:217-314: grid construction, '
                    'points_img2cam / points_cam2img, F.grid_sample x 2, cat) on PyTorch-CPU fp32, this host, '
                    'via oracle/dfm_torch_baseline.py (bit-identical to the reference on the golden fixtures)',
            'sample': (f'1 sample at the full C={C} x D={w["D"]} x {prm.h_out}x{prm.w_out}, nothing scaled, '
                       if c_run == C else
                       f'1 sample, {c_run} of {C} channels (host memory) x D={w["D"]} x {prm.h_out}x{prm.w_out}, '
                       'scaled by C, ') + f'{reps} repetitions, {total:.1f} s'}


def line_extras(args, w, elem, explicit, last_kernel):
    """names of the reported blocks rank 0 adds to the headline line after the timed region, in order.  A pure
    function of the command line and the workload -- NOT of the world size: `--gpus 8` prints the fields
    `--gpus 1` prints (tests/test_distributed_cpu.py holds that)."""
    ex = []
    if not args.channels_last:
        ex.append('api_build_dfm_cost')
        if w['C'] % (16 // elem) == 0:
            ex.append('channels_last_variant')
    if (args.traffic_bytes is None and not args.no_traffic and not args.channels_last and last_kernel == 2
            and not w.get('nhwc')):
        ex.append('traffic')
    if args.workload == 'nstar' and not explicit and not args.channels_last and not args.no_secondary:
        ex.append('secondary')
    if not args.no_cpu_baseline:
        ex.append('cpu_baseline')
    return ex


def rank_census(job):
    """{'ranks_seen': N, 'collective': 'rccl x.y.z'}: a SUM all-reduce of a one per rank over the job's
    process group -- the line's own evidence that RCCL reached ``n_gpus`` ranks (1 / None without a group)"""
    par = importlib.import_module('depth-from-motion_amd.parallel')
    return {'ranks_seen': job.ranks_seen(), 'collective': par.collective_library()}


def secondary(args, pkg, dev, job, emit=True):
    """Other hot-path rows on their config shapes; `value` = passes/s of the op over
    one sample batch, roofline from HIP-event step time (one fused kernel per step).
    emit=False: return the JSON line's dict instead of printing it (the default run's
    ``secondary`` block)."""
    rank, world = job.rank, job.world
    gen = torch.Generator().manual_seed(job.seed(0) // 1000)
    comm = None
    flops, dtype_name = None, None
    if args.workload in ('backbone', 'backbone_train', 'neck', 'dfm_neck'):
        # the MFMA-bound rows: whole-module forward, bf16 channels_last_3d,
        # every 3x3x3 convolution in the hand-written MFMA kernels (csrc/conv3d.hip, conv3d_g.hip)
        mods = importlib.import_module('depth-from-motion_amd.modules')
        torch.manual_seed(0)
        B, nbytes, dtype_name = 1, None, 'bf16'
        cl = torch.channels_last_3d
        if args.workload in ('backbone', 'backbone_train'):
            train = args.workload == 'backbone_train'
            m = mods.DfMBackbone(in_channels=32).to(dev).to(torch.bfloat16).train(train)
            m.downsampled_depth = pkg.prepare_depth(dict(num_bins=288, depth_min=2, depth_max=59.6,
                                                         downsample_factor=4))[0]
            m.volume_memory_format = cl
            # DFM_NO_SWEEP_FUSION=1: the materialised cost volume instead of the fused plane sweep +
            # dres0 / dres0_mono kernel (csrc/sweep_conv.hip) -- the A/B of SURVEY 8f rank 1
            m.fuse_sweep_dres0 = os.environ.get('DFM_NO_SWEEP_FUSION') != '1'
            # DFM_BACKBONE_ONE_STREAM=1: the stereo and mono stacks on one stream (the A/B of round 5's two streams)
            m.two_streams = os.environ.get('DFM_BACKBONE_ONE_STREAM') != '1'
            meta = dict(ori_cam2img=KITTI_P2, cur2prevs=torch.from_numpy(poses(1, 2 + rank)),
                        ori_shape=(375, 1242, 3), pad_shape=(320, 1280, 3), crop_offset=[0, 55], flip=False,
                        scale_factor=[1.0])
            cur = torch.randn(1, 32, 320, 1280, generator=gen).to(dev).bfloat16().requires_grad_(train)
            prev = torch.randn(1, 32, 320, 1280, generator=gen).to(dev).bfloat16().requires_grad_(train)
            if os.environ.get('DFM_FEATS_NHWC') == '1' and not train:  # the layout SPPUNetNeck emits
                cur, prev = (t.contiguous(memory_format=torch.channels_last) for t in (cur, prev))

            fwd_bwd_module, reducer = m, None
            if train and args.reducer != 'none':
                # the training step's one collective: the gradient all-reduce of
                # MMDistributedDataParallel (apis/train.py:222-230), overlapped with backward
                par = importlib.import_module('depth-from-motion_amd.parallel')
                if args.reducer == 'ddp':
                    if not torch.distributed.is_initialized():  # a one-rank group: same code path
                        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
                        os.environ.setdefault('MASTER_PORT', '29533')
                        torch.distributed.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
                    fwd_bwd_module = torch.nn.parallel.DistributedDataParallel(
                        m, device_ids=[dev.index], broadcast_buffers=False, bucket_cap_mb=32)
                else:
                    reducer = par.GradientBucketReducer(m.parameters())
                grad_bytes = sum(p.numel() * p.element_size() for p in m.parameters() if p.requires_grad)
                comm = {'reducer': args.reducer, 'gradient_bytes': grad_bytes,
                        'buckets': len(reducer.buckets) if reducer is not None else None}

            grads = []

            def fwd_bwd(module):
                # the module's own forward + backward: output gradients are fed directly, in the outputs'
                # layout and dtype (what DepthHead / FrustumToVoxel's backward hands over).  A synthetic
                # scalar loss on the channels-last bf16 outputs (round 2: .float().square().mean()) runs
                # ATen's strided elementwise kernels for 3.3 of 19 ms per step (profiles/archive/r03_c14_*).
                m.zero_grad(set_to_none=True)
                outs = module(cur, prev, [meta])
                if not grads:
                    grads.extend(torch.empty_like(o).normal_(generator=None) * 1e-3 for o in outs)
                torch.autograd.backward(list(outs), grads)

            def step():
                if train:  # forward + backward (all gradients; no optimizer): 3x the forward FLOPs
                    fwd_bwd(fwd_bwd_module)
                    if reducer is not None:
                        reducer.finalize()
                    return None
                with torch.no_grad():
                    return m(cur, prev, [meta])
            if comm is not None:
                if reducer is not None:
                    def no_exchange():
                        reducer.enabled = False
                        fwd_bwd(m)
                        reducer.enabled = True
                    comm.update(step_without_exchange=no_exchange, reducer_obj=reducer)
                else:
                    def no_exchange():   # the bare module: no DDP bookkeeping, no all-reduce
                        fwd_bwd(m)
                    comm.update(step_without_exchange=no_exchange)
            flops = (3 if train else 1) * 0.96e12  # stereo 532 G + mono 430 G per sample
            name = ('DfMBackbone forward + backward' if train else 'DfMBackbone.forward') + \
                ' config K (plane sweep + 3-D aggregation, 72x80x320, bf16 NDHWC)'
            unit = 'samples/s'
        else:
            big = args.workload == 'dfm_neck'
            m = (mods.DfMNeck(in_channels=64, out_channels=256, num_frames=2) if big else
                 mods.OutdoorImVoxelNeck(in_channels=64, out_channels=256)).to(dev).to(torch.bfloat16).eval()
            if big:
                m.two_streams = os.environ.get('DFM_BACKBONE_ONE_STREAM') != '1'  # (A/B of round 5's two streams)
            x = torch.randn(1, 128 if big else 64, 220, 300, 12, generator=gen).to(dev).bfloat16() \
                .contiguous(memory_format=cl)

            def step():
                with torch.no_grad():
                    return m(x)
            flops = 7.65e12 if big else 3.21e12
            name = ('DfMNeck' if big else 'OutdoorImVoxelNeck') + \
                '.forward config W (220x300x12 voxels, eval: BN folded into the MFMA conv epilogue, bf16 NDHWC)'
            unit = 'voxel-volumes/s'
    elif args.workload in ('stereo_infer', 'stereo_train'):
        # the WHOLE stereo path of configs/dfm/dfm_r34_1x8_kitti-3d-3class.py at config K (one 320 x 1280 frame
        # pair): SPPUNetNeck x 2 -> DfMBackbone -> DepthHead (fused into its consumers) -> FrustumToVoxel ->
        # voxel_convs -> BEVHourglass, built from the reference's own config dict (tests/golden/configs_dfm.json);
        # stereo_train: forward + dense depth loss + backward (no optimizer), bf16 fast path
        train = args.workload == 'stereo_train'
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests', 'golden', 'configs_dfm.json')) as f:
            model_cfg = dict(json.load(f)['dfm_r34_1x8_kitti-3d-3class.py']['model'])
        torch.manual_seed(0)
        B, nbytes, dtype_name = 1, None, 'bf16'
        H, W = 320, 1280
        K2 = KITTI_P2.copy()
        K2[1, 2] -= 55.0
        if train:
            path = pkg.DfMStereoPath(model_cfg).to(dev).train()
            pkg.enable_fast_path(path)
            path.fuse_depth_head = True
        else:
            path = pkg.DfMStereoPath(model_cfg).to(dev).eval().to(torch.bfloat16)
            path.backbone_stereo.volume_memory_format = torch.channels_last_3d

        def pyramid():
            lv = [torch.randn(1, c, H // s_, W // s_, generator=gen).to(dev).bfloat16()
                  for c, s_ in ((3, 1), (64, 2), (128, 4), (128, 4), (128, 4))]
            return lv if train else [t.contiguous(memory_format=torch.channels_last) for t in lv]
        cur, prev = pyramid(), pyramid()
        depth_img = (torch.rand(1, 1, H, W, generator=gen) * 60).to(dev)
        depth_img[torch.rand(1, 1, H, W, generator=gen).to(dev) < 0.93] = 0   # LiDAR: ~7 % of the pixels
        fg = (torch.rand(1, 1, H, W, generator=gen) < 0.3).float().to(dev)

        def meta():
            return dict(ori_cam2img=KITTI_P2, cam2img=K2.tolist(), cur2prevs=torch.from_numpy(poses(1, 2 + rank)),
                        ori_shape=(375, 1242, 3), pad_shape=(H, W, 3), crop_offset=[0, 55], flip=False,
                        scale_factor=[1.0])

        fwd_module, reducer = path, None
        if train and getattr(args, 'reducer', 'none') != 'none':
            # the training step's one collective (apis/train.py:222-230), overlapped with backward:
            # torch DDP around the path, or GradientBucketReducer's xGMI-sized buckets
            par = importlib.import_module('depth-from-motion_amd.parallel')
            if args.reducer == 'ddp':
                if not torch.distributed.is_initialized():  # a one-rank group: same code path
                    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
                    os.environ.setdefault('MASTER_PORT', '29534')
                    torch.distributed.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
                fwd_module = torch.nn.parallel.DistributedDataParallel(
                    path, device_ids=[dev.index], broadcast_buffers=False, bucket_cap_mb=32,
                    find_unused_parameters=True)
            else:
                reducer = par.GradientBucketReducer(path.parameters())
            comm = {'reducer': args.reducer,
                    'gradient_bytes': sum(p.numel() * p.element_size() for p in path.parameters() if p.requires_grad),
                    'buckets': len(reducer.buckets) if reducer is not None else None}

        def train_step(module):
            path.zero_grad(set_to_none=True)
            out = module(cur, prev, [meta()])
            loss = path.loss_dense_depth(out, depth_img, fg) + out['bev_feat'].float().square().mean()
            loss.backward()

        def step():
            if train:
                train_step(fwd_module)
                if reducer is not None:
                    reducer.finalize()
                return None
            with torch.no_grad():
                return path(cur, prev, [meta()])
        if comm is not None:
            if reducer is not None:
                def no_exchange():
                    reducer.enabled = False
                    train_step(path)
                    reducer.enabled = True
                comm.update(step_without_exchange=no_exchange, reducer_obj=reducer)
            else:
                comm.update(step_without_exchange=lambda: train_step(path))
        # the 3-D aggregation stacks' share of the step's arithmetic: what the fraction is quoted on
        flops = (3 if train else 1) * 0.96e12
        name = ('DfMStereoPath training step (forward + dense depth loss + backward)' if train else
                'DfMStereoPath inference') + \
            ' config K (2-D necks + plane sweep + 3-D aggregation + depth head + FrustumToVoxel + BEV hourglass, bf16; ' \
            'the fraction counts the 3-D aggregation FLOPs only)'
        unit = 'samples/s'
    elif args.workload in ('waymo', 'waymo_cl'):
        # config W: 5 views x 2 frames, 64 ch, 208x312 level-0 maps, 220x300x12 voxels, concat
        from tests.golden.make_golden import waymo_like_cameras
        B, nv, nf, C, hf, wf, nvox = 2, 5, 2, 64, 208, 312, (220, 300, 12)
        feats = torch.randn(B, nv * nf, C, hf, wf, generator=gen).to(dev)
        cl = args.workload == 'waymo_cl'
        esz = 2 if cl else 4
        if cl:  # what a channels_last bf16 image neck hands over, viewed (B, F*Nv, C, H, W)
            feats = feats.bfloat16().reshape(B * nv * nf, C, hf, wf).contiguous(
                memory_format=torch.channels_last).view(B, nv * nf, C, hf, wf)
            dtype_name = 'bf16'
        cams = waymo_like_cameras(nv, nf, 5)
        cams[:, 0, :] *= 1248 / 156.0
        cams[:, 1, :] *= 832 / 104.0
        meta = {'ori_lidar2img': [m for m in cams], 'input_shape': (832, 1248),
                'img_shape': [(832, 1248, 3)] * (nv * nf)}
        pts = pkg.voxel_centers([-35.0, -75.0, -2.0, 75.0, 75.0, 4.0], nvox).to(dev)

        def step():
            return pkg.mv_feature_transformation(
                feats, [meta] * B, nv, nf, None, nvox, 'concat', points=pts,
                memory_format=torch.channels_last_3d if cl else torch.contiguous_format)
        nbytes = B * esz * (nv * nf * C * hf * wf + C * nf * nvox[0] * nvox[1] * nvox[2])
        name = 'multi-view voxel lifting (5 views x 2 frames -> 128x220x300x12, ' + \
            ('bf16, channels-last views in place -> channels-last volume)' if cl else 'fp32)')
        unit = 'voxel-volumes/s'
    elif args.workload in ('voxel_sample', 'voxel_sample_bwd'):
        # SURVEY 8a row a10 (point_fusion.py:324-410; unused by the released configs): config K's voxel grid
        # (288 x 304 x 20, 32 channels) sampled back into its frustum (72 x 80 x 320)
        B, C = 1, 32
        vr, vs = [2.0, -30.4, -3.0, 59.6, 30.4, 1.0], [0.2, 0.2, 0.2]
        vox = torch.randn(1, C, 288, 304, 20, generator=gen).to(dev)
        ds = torch.tensor([(k + 0.5) * (57.6 / 288) + 2 for k in range(288)])
        proj = torch.from_numpy(KITTI_P2.copy())
        a = (vr, vs, ds, proj, 4, torch.tensor([1.0, 1.0]), torch.tensor([0.0, 0.0]), False, (320, 1280), (320, 1280))
        bwd = args.workload == 'voxel_sample_bwd'
        if bwd:
            vox.requires_grad_(True)
            gout = torch.randn(1, C, 72, 80, 320, generator=gen).to(dev)

        def step():
            out = pkg.voxel_sample(vox, *a, aligned=True)
            if bwd:
                vox.grad = None
                out.backward(gout)
            return out
        esz = 4
        nbytes = esz * C * (288 * 304 * 20 + 72 * 80 * 320) * (2 if bwd else 1)
        name = 'voxel_sample ' + ('forward + backward' if bwd else 'forward') + ' (32 x 288x304x20 voxels -> 72x80x320 frustum, fp32)'
        unit = 'frustum-volumes/s'
    elif args.workload in ('depth_head', 'depth_head_bf16'):
        B = 8
        x = (torch.randn(B, 1, 72, 80, 320, generator=gen) * 4).to(dev)
        esz = 4
        if args.workload == 'depth_head_bf16':
            x, esz, dtype_name = x.bfloat16(), 2, 'bf16'
        ds = torch.tensor([(k + 0.5) * (57.6 / 288) + 2 for k in range(288)])

        def step():
            return pkg.depth_head_forward(x, ds)
        nbytes = B * esz * (72 * 80 * 320 + 2 * 288 * 320 * 1280 + 320 * 1280)
        name = f'DepthHead.forward (1,72,80,320)->2x(288,320,1280)+map, {"bf16" if esz == 2 else "fp32"}'
        unit = 'depth-volumes/s'
    elif args.workload == 'sweep_bwd_kitti_cl':
        # config K's backward as the bf16 NDHWC training stack sees it: a channels-last bf16 gradient volume read
        # in place by the gather kernel (both maps), pixel-major fp32 map gradients -- the public function
        sweep = importlib.import_module('depth-from-motion_amd.plane_sweep')
        w = WORKLOADS['kitti']
        B, dtype_name = w['B'], 'bf16'
        cur = torch.empty(B, w['C'], w['H'], w['W'], dtype=torch.bfloat16, device=dev)
        desc = sweep._make_desc(cur, w['D'], w['fsf'], w['csf'], (375, 1242), False, w['crop'], 1.0)
        depths = torch.from_numpy(depth_planes(w['D'], w['dmin'], w['dmax'])).to(dev)
        P, Pinv, T = sweep.camera_matrices(torch.from_numpy(np.stack([KITTI_P2] * B)),
                                           torch.from_numpy(poses(B, 2 + rank)), B, dev)
        gout = torch.randn(B, 2 * w['C'], w['D'], desc.h_out, desc.w_out, device=dev).bfloat16().contiguous(
            memory_format=torch.channels_last_3d)
        walk, gather = [False], [os.environ.get('DFM_NO_PREV_GATHER') != '1']

        def step():
            with sweep.prev_gather(gather[0]):
                return sweep.plane_sweep_backward(desc, gout, depths, P, Pinv, T)
        nbytes = gout.numel() * 2 + 2 * B * w['C'] * w['H'] * w['W'] * 4
        name = 'plane-sweep backward (bf16 channels-last grad volume read in place -> 2 fp32 feature grads)'
        unit = 'cost-volume-grads/s'
    elif args.workload in ('sweep_bwd', 'sweep_bwd_kitti'):
        # backward of the plane sweep on the N* / K shape: grad volume -> fp32 feature grads
        sweep = importlib.import_module('depth-from-motion_amd.plane_sweep')
        w = WORKLOADS['nstar' if args.workload == 'sweep_bwd' else 'kitti']
        B = w['B']
        tdt = torch.bfloat16 if w['dtype'] == 'bf16' else torch.float32
        esz = 2 if w['dtype'] == 'bf16' else 4
        cur = torch.empty(B, w['C'], w['H'], w['W'], dtype=tdt, device=dev)
        desc = sweep._make_desc(cur, w['D'], w['fsf'], w['csf'], (375, 1242), False, w['crop'], 1.0)
        depths = torch.from_numpy(depth_planes(w['D'], w['dmin'], w['dmax'])).to(dev)
        P, Pinv, T = sweep.camera_matrices(torch.from_numpy(np.stack([KITTI_P2] * B)),
                                           torch.from_numpy(poses(B, 2 + rank)), B, dev)
        gout = torch.randn(B, 2 * w['C'], w['D'], desc.h_out, desc.w_out, device=dev, dtype=tdt)
        g_cur = torch.zeros(B, w['C'], w['H'], w['W'], device=dev)
        g_prev = torch.zeros_like(g_cur)
        lib = pkg._capi.lib()

        # strided fp32 sweeps: the cur map from the 3x3-window kernel (pixel-major gradient map), the prev map from
        # the tile kernel -- what plane_sweep_backward() does for them; everything else dfm_plane_sweep_bwd
        walk = [w['csf'] >= 1.5 and tdt == torch.float32]
        prev_only = sweep.make_opts(kernel=8)
        # ... and, since round 5, the prev map from the gather kernel (a lane per map pixel, stores instead of
        # atomics: csrc/plane_sweep_bwd_gather.hip); DFM_NO_PREV_GATHER=1 pins the tile kernel (A/B)
        gather = [walk[0] and os.environ.get('DFM_NO_PREV_GATHER') != '1']
        gws_bytes = lib.dfm_plane_sweep_bwd_prev_gather_workspace_bytes(ctypes.byref(desc))
        gws = torch.empty(max(int(gws_bytes), 256), dtype=torch.uint8, device=dev)

        dense = [os.environ.get('DFM_GATHER_DENSE') == '1']   # (experiments: both maps by the gather kernel)

        def step():
            a = (ctypes.byref(desc), gout.data_ptr(), depths.data_ptr(), P.data_ptr(), Pinv.data_ptr(), T.data_ptr())
            st = torch.cuda.current_stream(dev).cuda_stream
            if dense[0]:
                for half, gm in ((0, g_cur), (1, g_prev)):
                    pkg._capi.check(lib.dfm_plane_sweep_bwd_gather(a[0], half, a[1], 0, *a[2:], gm.data_ptr(), 0,
                                                                  gws.data_ptr(), gws_bytes, st))
                return
            g_cur.zero_()
            if not gather[0]:
                g_prev.zero_()    # (the gather kernel stores the prev map: no zero fill)
            if walk[0]:
                rc = lib.dfm_plane_sweep_bwd_cur_nhwc(*a, g_cur.data_ptr(), st)
                if rc == 0:
                    if gather[0]:
                        rc = lib.dfm_plane_sweep_bwd_prev_gather(*a, g_prev.data_ptr(), gws.data_ptr(), gws_bytes, st)
                        if rc == 0:
                            return
                        gather[0] = False
                        g_prev.zero_()
                    pkg._capi.check(lib.dfm_plane_sweep_bwd_opts(*a, g_prev.data_ptr(), g_prev.data_ptr(), st,
                                                                 ctypes.byref(prev_only)))
                    return
                walk[0] = False
            pkg._capi.check(lib.dfm_plane_sweep_bwd(*a, g_cur.data_ptr(), g_prev.data_ptr(), st))
        nbytes = gout.numel() * esz + 2 * g_cur.numel() * 4
        name = f'plane-sweep backward ({w["dtype"]} grad volume -> 2 fp32 feature grads)'
        unit = 'cost-volume-grads/s'
    elif args.workload == 'f2v_bwd':
        # FrustumToVoxel's backward as DfMStereoPath trains through it at config K: one frame's bf16 NDHWC cost
        # volume + semantic map, the depth head fused (lazy statistics), gradient of the 64-channel voxel volume
        # gathered back per frustum cell (the cell pre-pass, the gather kernel, the semantic map's reduction)
        B, C, D, H, W = 1, 32, 72, 80, 320
        dtype_name = 'bf16'
        stereo = torch.randn(B, C, D, H, W, generator=gen).to(dev).bfloat16().contiguous(
            memory_format=torch.channels_last_3d).requires_grad_(True)
        cost = (torch.randn(B, 1, D, H, W, generator=gen) * 4).to(dev).bfloat16()
        sem = torch.randn(B, C, H, W, generator=gen).to(dev).bfloat16().requires_grad_(True)
        ds = torch.tensor([(k + 0.5) * (57.6 / 288) + 2 for k in range(288)])
        zz, yy, xx = torch.meshgrid(torch.linspace(-2.9, 0.9, 20), torch.linspace(-30.3, 30.3, 304),
                                    torch.linspace(2.1, 59.5, 288), indexing='ij')
        coords = torch.stack([xx, yy, zz], -1).to(dev)
        K = KITTI_P2.copy()
        K[1, 2] -= 55.0
        metas = [{'cam2img': K.tolist(), 'pad_shape': (320, 1280, 3)}] * B
        lazy, _ = pkg.depth_head_statistics(cost, ds, 4)
        vox = pkg.frustum_to_voxel_sample(stereo, lazy, metas, sem, coords, dict(depth_min=2, depth_max=59.6))
        go = torch.randn(vox.shape, generator=gen).to(dev).bfloat16()
        if not vox.is_contiguous():
            go = go.contiguous(memory_format=torch.channels_last_3d)

        def step():
            return torch.autograd.grad(vox, [stereo, sem], go, retain_graph=True)
        # the voxel gradient read once, both input gradients written once
        nbytes = B * 2 * (2 * C * 20 * 304 * 288 + C * D * H * W + C * H * W)
        name = 'FrustumToVoxel backward (grad 64x20x304x288 -> 32x72x80x320 + 32x80x320, bf16 channels-last, fused depth head)'
        unit = 'voxel-volumes/s'
    elif args.workload in ('group_norm', 'group_norm_cl'):
        B = 8
        x = (torch.randn(B, 32, 72, 80, 320, generator=gen) + 0.5).to(dev)
        m = pkg.HipGroupNorm(32, 32).to(dev)
        esz = 4
        if args.workload == 'group_norm_cl':
            x = x.bfloat16().contiguous(memory_format=torch.channels_last_3d)
            m, esz, dtype_name = m.to(torch.bfloat16), 2, 'bf16'

        def step():
            with torch.no_grad():
                return m(x, relu=True)
        nbytes = B * esz * 2 * 32 * 72 * 80 * 320
        name = 'fused GroupNorm(32,32)+ReLU on (32,72,80,320) ' + \
            ('bf16 channels_last_3d' if esz == 2 else 'fp32') + ' (in + out bytes)'
        unit = 'volumes/s'
    else:
        B, C, D, H, W = 8, 32, 72, 80, 320
        stereo = torch.randn(B, C, D, H, W, generator=gen).to(dev)
        soft = torch.softmax(torch.randn(B, 1, 4 * D, 4 * H, 4 * W, device=dev), dim=2)
        sem = torch.randn(B, C, H, W, generator=gen).to(dev)
        zz, yy, xx = torch.meshgrid(torch.linspace(-2.9, 0.9, 20), torch.linspace(-30.3, 30.3, 304),
                                    torch.linspace(2.1, 59.5, 288), indexing='ij')
        coords = torch.stack([xx, yy, zz], -1).to(dev)
        K = KITTI_P2.copy()
        K[1, 2] -= 55.0
        metas = [{'cam2img': K.tolist(), 'pad_shape': (320, 1280, 3)}] * B
        cfg = dict(depth_min=2, depth_max=59.6)
        esz = 4
        if args.workload == 'f2v_cl':  # the NDHWC stack's cost volume, sampled in place
            stereo = stereo.bfloat16().contiguous(memory_format=torch.channels_last_3d)
            soft, sem, esz, dtype_name = soft.bfloat16(), sem.bfloat16(), 2, 'bf16'

        def step():
            return pkg.frustum_to_voxel_sample(stereo, soft, metas, sem, coords, cfg)
        nbytes = B * esz * (C * D * H * W + 288 * 320 * 1280 + C * H * W + 2 * C * 20 * 304 * 288)
        name = 'FrustumToVoxel sampling (32x72x80x320 + 288x320x1280 + 32x80x320 -> 64x20x304x288, ' + \
            ('bf16, channels-last in and out)' if esz == 2 else 'fp32)')
        unit = 'voxel-volumes/s'
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    graphed = False
    if os.environ.get('DFM_BENCH_GRAPH') == '1' and args.workload in ('backbone', 'neck', 'dfm_neck'):
        # the inference step captured ONCE into a HIP graph and replayed: the same launches on the same
        # streams, without the ~100 host-side launch calls per step (matrices staged on the device first: a
        # capture takes no host-to-device copies)
        if args.workload == 'backbone':
            importlib.import_module('depth-from-motion_amd.data_geometry').stage_geometry([meta], dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            graph_out = step()
        eager_step, step, graphed = step, graph.replay, True
        step()
        torch.cuda.synchronize()
    if comm is not None:
        # the same step WITHOUT the gradient exchange, timed first: what the all-reduce adds on top of
        # it in the headline below is the communication that backward did not hide
        no_comm_s, _ = job.timed_steps(comm.pop('step_without_exchange'), args.steps)
        comm['ms_per_step_no_exchange'] = round(no_comm_s * 1e3 / args.steps, 4)
        step()
        torch.cuda.synchronize()
        if 'reducer_obj' in comm:
            comm['reducer_obj'].launched_during_backward = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    job.barrier()
    torch.cuda.synchronize()
    e0.record()
    elapsed, every = job.timed_steps(step, args.steps)   # barrier + sync both sides, MAX over ranks
    e1.record()
    torch.cuda.synchronize()
    ms = elapsed * 1e3 / args.steps
    dev_ms = e0.elapsed_time(e1) / args.steps
    if comm is not None:
        comm['exposed_exchange_ms'] = round(ms - comm['ms_per_step_no_exchange'], 4)
        if 'reducer_obj' in comm:
            comm['buckets_launched_during_backward_per_step'] = comm.pop('reducer_obj').launched_during_backward / args.steps
    if flops is not None:
        tf = flops / (dev_ms * 1e-3) / 1e12
        roof = {'bound': 'mfma', 'achieved': round(tf, 1), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(tf / MFMA_PEAK_TFLOPS, 4), 'traffic': None, 'kernel_ms': round(dev_ms, 4),
                'algorithmic_flops_per_step': flops}
    else:
        achieved = nbytes / (dev_ms * 1e-3) / 1e9
        traffic = None  # (tools/pmc_summary.py over this command gives the counters of a secondary row)
        roof = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBPS,
                'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBPS, 4),
                'traffic': traffic, 'kernel_ms': round(dev_ms, 4),
                'algorithmic_bytes_per_launch': nbytes}
    line = {
        'metric': unit.replace('/s', '/sec'), 'value': round(B * world / (ms / 1e3), 2),
        'unit': unit, 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None,
        'dtype': dtype_name or ('bf16' if args.workload == 'sweep_bwd' else 'f32'),
        'data': 'synthetic',
        'config': {'workload': f'{args.workload}: {name}', 'global_batch': B * world,
                   'parallelism': f'dp{world}'},
        'roofline': roof, 'per_rank_ms_per_step': [round(v * 1e3 / args.steps, 4) for v in every],
        **rank_census(job),
        **({'gradient_exchange': comm} if comm is not None else {})}
    if graphed:
        line['config']['hip_graph_replay'] = True
    if args.workload.startswith('sweep_bwd'):
        # 1 scatter, 5 LDS-atomic tiles, 6 matrix product
        line['config']['bwd_kernel'] = int(pkg._capi.lib().dfm_plane_sweep_bwd_last_kernel())
        line['config']['cur_map_window_kernel'] = bool(walk[0])
        line['config']['prev_map_gather_kernel'] = bool(gather[0])
    if rank == 0 and emit:
        print(json.dumps(line), flush=True)
    if emit and torch.distributed.is_available() and torch.distributed.is_initialized() and world == 1:
        torch.distributed.destroy_process_group()   # the one-rank group of --reducer ddp
    return line


def secondary_block(pkg, sweep, dev, job, budget_s=45.0, traffic=True):
    """What else the default run witnesses, after the timed headline and never as ``value``: other rows
    of SURVEY.md 8a on their config shapes, a few steps each -- the plane-sweep backward at N*, the
    shipped config's strided sweep (config K, NHWC maps) and its backward, FrustumToVoxel's channels-last sampling,
    DfMBackbone.forward and the voxel neck.
    Each entry: value + unit, ms per step, fraction of its roofline.  A row that fails or would overrun
    the wall-clock budget is reported as skipped, not silently dropped."""
    import types
    out = {}
    t_start, t_traffic = time.perf_counter(), 0.0
    for wl in ('sweep_bwd', 'kitti_nhwc', 'sweep_bwd_kitti', 'sweep_bwd_kitti_cl', 'f2v_cl', 'f2v_bwd', 'backbone', 'neck',
               'dfm_neck', 'backbone_train', 'waymo_cl', 'nstar_negzero', 'nstar_negzero_all', 'stereo_infer',
               'stereo_train'):
        if time.perf_counter() - t_start - t_traffic > budget_s:   # (the counter passes have their own time)
            out[wl] = {'skipped': 'wall-clock budget of the secondary block spent'}
            continue
        try:
            if wl in WORKLOADS:
                line = quick_sweep_row(pkg, sweep, dev, wl, steps=10, warmup=3)
            else:
                # (the launch-heavy module rows settle later than the single-kernel ones: more steps, still < 1 s each)
                nsteps = {'backbone': 30, 'neck': 30, 'dfm_neck': 20, 'backbone_train': 20, 'stereo_infer': 20}.get(wl, 10)
                a = types.SimpleNamespace(workload=wl, steps=nsteps, warmup=5 if nsteps > 10 else 3, reducer='none')
                if wl == 'backbone':
                    os.environ.setdefault('DFM_FEATS_NHWC', '1')  # the layout SPPUNetNeck hands over
                line = secondary(a, pkg, dev, job, emit=False)
            r = line['roofline']
            out[wl] = {'what': line['config']['workload'], 'value': line['value'], 'unit': line['unit'],
                       'ms_per_step': line['ms_per_step'], 'bound': r['bound'], 'achieved': r['achieved'],
                       'roofline_unit': r['unit'], 'frac': r['frac'], 'steps': line['steps']}
            if traffic and wl in TRAFFIC_ROWS:
                # counter bytes of the row's step, measured now (two ~5 s passes over the torch-free harness)
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                tt = time.perf_counter()
                tr = measure_traffic(wl, None, timeout=60)
                t_traffic += time.perf_counter() - tt
                out[wl]['traffic'] = tr['hbm_bytes_per_launch'] if tr else None
                if tr and r.get('algorithmic_bytes_per_launch'):
                    out[wl]['traffic_over_algorithmic'] = round(tr['hbm_bytes_per_launch'] /
                                                                r['algorithmic_bytes_per_launch'], 3)
        except Exception as e:  # a secondary row must not take the headline down with it
            out[wl] = {'skipped': f'{type(e).__name__}: {e}'[:200]}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    out['wall_s'] = round(time.perf_counter() - t_start, 1)
    out['traffic_passes_s'] = round(t_traffic, 1)
    return out


def plant_special(cur, prev, w):
    """WORKLOADS[..]['plant']: -0 values in the synthetic maps ('one': one per map and sample; 'channel': a whole
    channel of them)"""
    if w.get('plant') == 'one':
        cur[:, 0, w['H'] // 2, w['W'] // 2] = -0.0
        prev[:, 1, w['H'] // 3, w['W'] // 3] = -0.0
    elif w.get('plant') == 'channel':
        cur[:, 0] = -0.0
        prev[:, 1] = -0.0


def quick_sweep_row(pkg, sweep, dev, wl, steps, warmup):
    """a forward plane sweep of WORKLOADS[wl] through the public launch, HIP events around the steps"""
    w = WORKLOADS[wl]
    tdtype = torch.bfloat16 if w['dtype'] == 'bf16' else torch.float32
    elem = 2 if w['dtype'] == 'bf16' else 4
    B = w['B']
    g = torch.Generator().manual_seed(7)
    cur = torch.randn(B, w['C'], w['H'], w['W'], generator=g).to(dev).to(tdtype)
    prev = torch.randn(B, w['C'], w['H'], w['W'], generator=g).to(dev).to(tdtype)
    plant_special(cur, prev, w)
    if w.get('nhwc'):
        cur, prev = (t.contiguous(memory_format=torch.channels_last) for t in (cur, prev))
    depths = torch.from_numpy(depth_planes(w['D'], w['dmin'], w['dmax'])).to(dev)
    desc = sweep._make_desc(cur, w['D'], w['fsf'], w['csf'], (375, 1242), w['flip'], w['crop'], w['scale'])
    P, Pinv, T = sweep.camera_matrices(torch.from_numpy(np.stack([KITTI_P2] * B)),
                                       torch.from_numpy(poses(B, 2)), B, dev)
    out = torch.empty((B, 2 * w['C'], w['D'], desc.h_out, desc.w_out), dtype=tdtype, device=dev)
    for _ in range(warmup):
        sweep.plane_sweep_forward(desc, cur, prev, depths, P, Pinv, T, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        sweep.plane_sweep_forward(desc, cur, prev, depths, P, Pinv, T, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    nbytes = algorithmic_bytes(w, elem) * B
    achieved = nbytes / (ms * 1e-3) / 1e9
    return {'value': round(B / (ms * 1e-3), 2), 'unit': 'cost-volumes/s', 'ms_per_step': round(ms, 4), 'steps': steps,
            'config': {'workload': f'{wl}: plane-sweep forward B={B} x (2C={2 * w["C"]}, D={w["D"]}, '
                                   f'{desc.h_out}x{desc.w_out}) {w["dtype"]}, csf={w["csf"]}'
                                   + (', NHWC maps sampled in place' if w.get('nhwc') else '')},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1), 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBPS, 4), 'algorithmic_bytes_per_launch': nbytes}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='nstar', choices=sorted(WORKLOADS) + list(SECONDARY))
    ap.add_argument('--kernel', type=int, default=0, help='0 auto, 1 gather, 2 LDS tiles, 3 direct tiles, 4 pixel-major taps + LDS transpose (A/B)')
    ap.add_argument('--lanes', type=int, default=0, help='LDS kernel lanes/workgroup (128|256)')
    ap.add_argument('--lds-kib', type=int, default=0, help='LDS kernel KiB/workgroup')
    ap.add_argument('--bpg', type=int, default=0, help='LDS kernel channel blocks per group')
    ap.add_argument('--planes', type=int, default=0, help='LDS kernel depth planes per workgroup')
    ap.add_argument('--band-chunk', type=int, default=0,
                    help='LDS kernel: adjacent bands scheduled back to back (default 1)')
    ap.add_argument('--ppl', type=int, default=0, help='LDS kernel: lattice points per lane (4|8)')
    ap.add_argument('--channels-last', action='store_true',
                    help='write the volume (B,D,H,W,2C) (memory_format channels_last_3d)')
    ap.add_argument('--no-autotune', action='store_true',
                    help='skip dfm_plane_sweep_autotune (workgroup schedule stays at its default)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--reducer', default='none', choices=['none', 'ddp', 'bucket'],
                    help='backbone_train / stereo_train: gradient exchange of the step (DDP, or GradientBucketReducer)')
    ap.add_argument('--traffic-bytes', type=float, default=None,
                    help='HBM bytes per launch from a separate rocprofv3 --pmc pass (default: measured in this run)')
    ap.add_argument('--no-traffic', action='store_true', help='skip the in-run rocprofv3 --pmc passes')
    ap.add_argument('--no-secondary', action='store_true', help="skip the default run's secondary rows")
    ap.add_argument('--no-smi', action='store_true', help='skip the amd-smi / rocm-smi readings')
    ap.add_argument('--pipeline', type=int, default=0, help='LDS kernel body: 1 serial, 2 pipelined')
    ap.add_argument('--store-align', type=int, default=0, help='LDS kernel: band cuts at multiples of 8|16|32|64 points')
    ap.add_argument('--pair-stores', type=int, default=0, help='4 points per lane: 1 paired 16-byte stores, 2 8-byte stores')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` is ONE command: no launcher around it -> this process becomes the
        # launcher (one rank per GPU under torch.distributed.run, rendezvous on 127.0.0.1; the reference's
        # tools/dist_train.sh:10-20 in one line) and the ranks below print the line
        par = importlib.import_module('depth-from-motion_amd.parallel')
        sys.exit(par.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    assert torch.cuda.is_available(), 'bench.py needs a GPU: there is no CPU product path'
    assert args.gpus <= torch.cuda.device_count() or world > 1, \
        f'--gpus {args.gpus} but this node shows {torch.cuda.device_count()} GPU(s)'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    pkg = importlib.import_module('depth-from-motion_amd')
    sweep = importlib.import_module('depth-from-motion_amd.plane_sweep')
    lib = pkg._capi.lib()
    # explicit launch options (A/B runs) travel with every call; none = tuned / default
    explicit = {k: v for k, v in dict(kernel=args.kernel, lanes=args.lanes, lds_kib=args.lds_kib,
                                      blocks_per_group=args.bpg, planes=args.planes,
                                      bands_per_chunk=args.band_chunk,
                                      points_per_lane=args.ppl, pipeline=args.pipeline,
                                      store_align=args.store_align, pair_stores=args.pair_stores).items() if v}
    if args.no_autotune:
        os.environ['DFM_AUTOTUNE'] = '0'
    par = importlib.import_module('depth-from-motion_amd.parallel')
    job = par.BenchJob(rank, world, dev, torch.cuda.synchronize)
    with sweep.launch_options(**explicit):
        return run(args, pkg, sweep, lib, dev, job, explicit)


def run(args, pkg, sweep, lib, dev, job, explicit):
    rank, world = job.rank, job.world
    if args.workload in SECONDARY:
        return secondary(args, pkg, dev, job)
    w = WORKLOADS[args.workload]
    tdtype = torch.bfloat16 if w['dtype'] == 'bf16' else torch.float32
    elem = 2 if w['dtype'] == 'bf16' else 4
    B = w['B']
    # synthetic inputs: cur seed 0, prev seed 1 (+rank so shards differ)
    gc = torch.Generator().manual_seed(job.seed(0))
    gp = torch.Generator().manual_seed(job.seed(1))
    cur = torch.randn(B, w['C'], w['H'], w['W'], generator=gc).to(dev).to(tdtype)
    prev = torch.randn(B, w['C'], w['H'], w['W'], generator=gp).to(dev).to(tdtype)
    plant_special(cur, prev, w)
    if w.get('nhwc'):
        cur, prev = (t.contiguous(memory_format=torch.channels_last) for t in (cur, prev))
    depths = torch.from_numpy(depth_planes(w['D'], w['dmin'], w['dmax'])).to(dev)
    desc = sweep._make_desc(cur, w['D'], w['fsf'], w['csf'], (375, 1242), w['flip'], w['crop'],
                            w['scale'])
    cam2imgs = torch.from_numpy(np.stack([KITTI_P2] * B))
    cur2prevs = torch.from_numpy(poses(B, 2 + rank))
    P, Pinv, T = sweep.camera_matrices(cam2imgs, cur2prevs, B, dev)
    if args.channels_last:
        # same values, volume laid out (B, D, H, W, 2C): torch memory_format channels_last_3d
        out = torch.empty((B, w['D'], desc.h_out, desc.w_out, 2 * w['C']), dtype=tdtype,
                          device=dev).permute(0, 4, 1, 2, 3)
    else:
        out = torch.empty((B, 2 * w['C'], w['D'], desc.h_out, desc.w_out), dtype=tdtype, device=dev)

    def step():
        sweep.plane_sweep_forward(desc, cur, prev, depths, P, Pinv, T, out=out,
                                  channels_last=args.channels_last)

    # which part did the lease land on?  Rounds 1-3 saw the same binary at 0.63 of the roofline on some
    # parts and at 0.48 on others; round 4 found the cause in the kernel's own store stream -- every band
    # cut of every channel plane left two partial 64-byte writes behind, which some parts absorb and others
    # do not (profiles/archive/r04_c1..c4_*) -- and removed it.  The probes stay as a record of the part: the tile
    # kernel's store pattern replayed as zeros, a linear fill, the shader clock under an FMA load, and what
    # the SMI tools report under load (tools/part_info.py).
    torch.cuda.synchronize()
    plane_bytes = w['D'] * desc.h_out * desc.w_out * elem
    part = None
    if not args.channels_last and plane_bytes % 16 == 0:
        def probe(fn):
            pe0, pe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            pe0.record()
            for _ in range(3):
                fn()
            pe1.record()
            torch.cuda.synchronize()
            return 3 * out.numel() * out.element_size() / (pe0.elapsed_time(pe1) * 1e-3) / 1e9
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        tile_gbps = probe(lambda: pkg._capi.check(lib.dfm_store_probe(
            ctypes.c_void_p(out.data_ptr()), B, 2 * w['C'], plane_bytes, 0, 0, st)))
        clk = torch.zeros(3, dtype=torch.int64, device=dev)
        pkg._capi.check(lib.dfm_clock_probe(ctypes.c_void_p(clk.data_ptr()), 1 << 18, st))
        torch.cuda.synchronize()
        cyc, ref = (int(v) for v in clk[:2].tolist())
        ghz = cyc / max(ref, 1) / 10.0
        part = {'tile_store_probe_gbps': round(tile_gbps, 1), 'linear_fill_gbps': round(probe(out.zero_), 1),
                'shader_clock_ghz_under_fma_load': round(ghz, 3),
                'note': "zeros in 4 KiB runs walking every channel plane (dfm_store_probe), a linear fill, the shader "
                        'clock under an FMA load on every CU (dfm_clock_probe); smi: amd-smi / rocm-smi readings, '
                        'clocks and power sampled while the sweep runs'}
        if world == 1 and not args.no_smi:   # (N > 1: rank 0 must not run 400 extra launches while the others go ahead)
            try:
                sys.path.insert(0, os.path.join(ROOT, 'tools'))
                import part_info

                def load():
                    torch.cuda.set_device(dev)
                    for _ in range(400):
                        step()
                    torch.cuda.synchronize()
                os.environ['DFM_AUTOTUNE'] = '0'
                part['smi'] = part_info.collect(load)
                os.environ.pop('DFM_AUTOTUNE', None)
            except Exception as e:
                part['smi'] = {'error': repr(e)}
    if explicit or args.channels_last or args.no_autotune or w.get('nhwc'):
        os.environ['DFM_AUTOTUNE'] = '0'  # keep the first launch from tuning by itself
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    tuned = None
    if not args.channels_last and not explicit and not args.no_autotune and not w.get('nhwc'):
        # untimed, like the warm-up: the library's choice among its candidate launch shapes /
        # workgroup orders for this shape on this part (the first dfm_plane_sweep_fwd of a volume
        # this size ran the autotuner; query what it cached, tune now if it did not)
        tuned = sweep.plane_sweep_tuning(desc) or \
            sweep.plane_sweep_autotune(desc, cur, prev, depths, P, Pinv, T, out)
    verified, check_agrees = None, None
    if tuned is not None and w['dtype'] == 'bf16':
        # untimed, informational: the library's pick (dfm_plane_sweep_autotune: 4 rounds x 2 launches per
        # candidate, median round) next to both tile shapes timed over as many launches as the timed
        # region has.  The bench runs WHAT THE LIBRARY PICKED -- what a build_dfm_cost() user gets -- and
        # reports whether this check agrees (round 2 overrode the library here).
        shapes = [dict(kernel=2, lanes=256, points_per_lane=8, bands_per_chunk=1),
                  dict(kernel=2, lanes=512, points_per_lane=4, bands_per_chunk=1)]
        shapes = {schedule_key(sweep.make_opts(**kw).as_dict()): kw for kw in shapes}
        verified = {}
        for key, kw in shapes.items():
            with sweep.launch_options(**kw):
                step()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.steps):
                    step()
                e1.record()
                torch.cuda.synchronize()
                verified[key] = round(e0.elapsed_time(e1) / args.steps, 4)
        best_key, verified = job.agree_fastest(verified)
        verified = {k: round(v, 4) for k, v in verified.items()}
        check_agrees = schedule_key(tuned) == best_key
    chosen_kw = {}
    import contextlib
    with (sweep.launch_options(**chosen_kw) if chosen_kw else contextlib.nullcontext()):
        step()
        pkg._capi.check(lib.dfm_profile_begin(args.steps))
        # barrier + synchronize on both sides; the job's time is the MAX over ranks, every rank's own
        # time is kept to expose stragglers (parallel.BenchJob: the same code runs under gloo in the tests)
        elapsed, every = job.timed_steps(step, args.steps)
    kms, klaunches = ctypes.c_double(0), ctypes.c_int(0)
    pkg._capi.check(lib.dfm_profile_end(ctypes.byref(kms), ctypes.byref(klaunches)))
    per_rank_ms = [round(v * 1e3 / args.steps, 4) for v in every]
    census = rank_census(job)   # a collective: every rank takes part
    ms_per_step = elapsed * 1e3 / args.steps
    value = job.value(B, args.steps, elapsed)

    if rank == 0:
        # everything below is rank 0 on its own, AFTER the job's closing barrier (the other ranks leave): the
        # line carries the same fields at every N (line_extras: nothing in it depends on the world size)
        extras = line_extras(args, w, elem, explicit, lib.dfm_plane_sweep_last_kernel())
        bytes_per_launch = algorithmic_bytes(w, elem) * B
        avg_kernel_ms = kms.value / max(klaunches.value, 1)
        achieved = bytes_per_launch / (avg_kernel_ms * 1e-3) / 1e9
        line = {
            'metric': 'cost-volumes/sec',
            'value': round(value, 2),
            'unit': 'cost-volumes/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': w['dtype'],
            'data': 'synthetic',
            'config': {
                'workload': f'{args.workload}: plane-sweep cost-volume build '
                            f'B={B}/GPU x (2C={2 * w["C"]}, D={w["D"]}, {desc.h_out}x{desc.w_out}), '
                            f'feats {w["C"]}x{w["H"]}x{w["W"]} {w["dtype"]}, fsf={w["fsf"]} '
                            f'csf={w["csf"]}',
                'global_batch': B * world,
                'parallelism': f'dp{world}',
                'kernel': 'sweep_cl_kernel' if args.channels_last else
                {1: 'sweep_gather_kernel', 2: 'sweep_tile_kernel<LDS>', 3: 'sweep_tile_kernel<direct>',
                 4: 'sweep_clt_kernel (pixel-major taps + LDS transpose)',
                 5: 'sweep_cltw_kernel (pixel-major taps, depth axis walked per wave)'}.get(
                    lib.dfm_plane_sweep_last_kernel(), 'none'),
                'launch': schedule_key(tuned if tuned is not None else
                                       pkg._capi.SweepOpts(**{sweep._OPT_FIELDS[k]: v for k, v in
                                                              explicit.items()}).as_dict()),
                'autotuned': tuned is not None,
                'tuning_check_ms': verified,
                'tuning_check_agrees_with_library': check_agrees,
                'volume_layout': '(B,D,H,W,2C) channels_last_3d' if args.channels_last
                else '(B,2C,D,H,W) contiguous (the reference layout)',
            },
            'roofline': {
                'bound': 'hbm',
                'achieved': round(achieved, 1),
                'peak': HBM_PEAK_GBPS,
                'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBPS, 4),
                # the same bytes over the whole timed step (pack pass + launch gaps included)
                'frac_step': round(bytes_per_launch / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                # HBM bytes per launch of the configuration that ran, from counter passes made in THIS
                # run on THIS part after the timed region (measure_traffic); null if they cannot be made
                'traffic': args.traffic_bytes,
                'kernel_ms': round(avg_kernel_ms, 4),
                'algorithmic_bytes_per_launch': bytes_per_launch,
            },
            'per_rank_ms_per_step': per_rank_ms,
            **census,
            'part': part,
        }
        if 'api_build_dfm_cost' in extras:
            # the public API (what DfMBackbone.forward calls): build_dfm_cost with device-resident
            # intrinsics / poses -- pads, inverts and packs them on the device, allocates the
            # volume, launches.  Reported beside the raw launch, never as `value`.
            k_dev, t_dev = cam2imgs.to(dev), cur2prevs.to(dev)
            del out
            torch.cuda.synchronize()

            def api_step():
                return pkg.build_dfm_cost(cur, prev, depths, w['fsf'], w['csf'], k_dev, t_dev,
                                          (375, 1242), w['flip'], w['crop'], w['scale'])
            vol = None
            for _ in range(max(args.warmup, 2)):
                vol = api_step()  # (held like in the timed loop: both 26.8 GB blocks get cached now)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                vol = api_step()
            torch.cuda.synchronize()
            ms_api = (time.perf_counter() - t0) * 1e3 / args.steps
            del vol
            line['api_build_dfm_cost'] = {
                'value': round(B / (ms_api / 1e3), 2), 'unit': 'cost-volumes/s',
                'ms_per_step': round(ms_api, 4),
                'note': 'public build_dfm_cost(): device-side pad/inverse/pack of the camera '
                        'matrices + output allocation (caching allocator) + the same launch'}
            out = None
        if 'channels_last_variant' in extras:
            # the same volume written channels-last (opt-in layout, identical values): reported
            # beside the headline, never instead of it
            out_cl = torch.empty((B, w['D'], desc.h_out, desc.w_out, 2 * w['C']), dtype=tdtype,
                                 device=dev).permute(0, 4, 1, 2, 3)
            for _ in range(args.warmup):
                sweep.plane_sweep_forward(desc, cur, prev, depths, P, Pinv, T, out=out_cl,
                                          channels_last=True)
            torch.cuda.synchronize()
            pkg._capi.check(lib.dfm_profile_begin(args.steps))
            t0 = time.perf_counter()
            for _ in range(args.steps):
                sweep.plane_sweep_forward(desc, cur, prev, depths, P, Pinv, T, out=out_cl,
                                          channels_last=True)
            torch.cuda.synchronize()
            ms_cl = (time.perf_counter() - t0) * 1e3 / args.steps
            pkg._capi.check(lib.dfm_profile_end(ctypes.byref(kms), ctypes.byref(klaunches)))
            k_cl = kms.value / max(klaunches.value, 1)
            line['channels_last_variant'] = {
                'value': round(B / (ms_cl / 1e3), 2), 'unit': 'cost-volumes/s',
                'ms_per_step': round(ms_cl, 4), 'kernel': 'sweep_cl_kernel',
                'volume_layout': '(B,D,H,W,2C) channels_last_3d, bit-identical values',
                'roofline': {'achieved': round(bytes_per_launch / (k_cl * 1e-3) / 1e9, 1),
                             'frac': round(bytes_per_launch / (k_cl * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                             'kernel_ms': round(k_cl, 4)}}
            del out_cl
        ran = tuned if tuned is not None else pkg._capi.SweepOpts(
            **{sweep._OPT_FIELDS[k]: v for k, v in explicit.items()}).as_dict()
        if 'traffic' in extras:
            out = None
            torch.cuda.empty_cache()
            tr = measure_traffic(args.workload, ran)
            if tr is not None:
                line['roofline']['traffic'] = tr['hbm_bytes_per_launch']
                line['roofline']['traffic_detail'] = tr
        if 'secondary' in extras:
            out = None
            torch.cuda.empty_cache()
            line['secondary'] = secondary_block(pkg, sweep, dev, job.solo(), traffic=not args.no_traffic)
        if 'cpu_baseline' in extras:
            line['cpu_baseline'] = cpu_baseline(w)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
