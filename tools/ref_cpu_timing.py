#!/usr/bin/env python
"""Times the REFERENCE's own ``build_dfm_cost`` (mmdet3d/models/backbones/dfm_backbone.py:217-314)
on PyTorch-CPU -- the "reference CPU neck" BASELINE.md section 3 / SURVEY.md 8d ask to report beside
the GPU number.  The function is lifted from the reference file by AST and executed unmodified
(same harness as tests/golden/make_golden.py), so this runs in the BUILD CONTAINER only
(/root/reference is not on the GPU box); the result is committed as
profiles/r02_reference_cpu_timing.json and bench.py carries it as
``cpu_baseline.reference_torch_cpu``.

  K  : config K shape (C=32, 320x1280 fp32, csf=4, D=72) -- one volume, as shipped
  N* : C_sub of the 256 channels of the north-star shape (94x311, D=112, fsf=4), scaled by C
       (the work is linear in C; the full fp32 output would be 6.7 GB per sample)
usage: python tools/ref_cpu_timing.py [--threads N] [--reps R] [--csub C]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', type=int, default=os.cpu_count())
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--csub', type=int, default=32)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r02_reference_cpu_timing.json'))
    args = ap.parse_args()
    import make_golden as mg
    import bench
    ref = mg.load_reference()
    fn = ref['build_dfm_cost']
    torch.set_num_threads(args.threads)
    K = torch.tensor(bench.KITTI_P2)[None]
    T = torch.from_numpy(bench.poses(1, 2))

    def timed(C, H, W, D, fsf, csf, crop, label):
        gen = torch.Generator().manual_seed(0)
        cur, prev = torch.randn(1, C, H, W, generator=gen), torch.randn(1, C, H, W, generator=gen)
        depths = torch.from_numpy(bench.depth_planes(D, 2.0, 59.6))
        ts = []
        for r in range(args.reps + 1):  # first repetition warms up
            t0 = time.perf_counter()
            with torch.no_grad():
                out = fn(cur, prev, depths, fsf, csf, K, T, (375, 1242), False, crop, 1.0)
            ts.append(time.perf_counter() - t0)
        print(f'{label}: out {tuple(out.shape)}  {[round(t, 3) for t in ts]} s', flush=True)
        return float(np.median(ts[1:])), tuple(out.shape)

    tk, shape_k = timed(32, 320, 1280, 72, 1, 4, (0, 55), 'config K')
    tn, shape_n = timed(args.csub, 94, 311, 112, 4, 1, (0, 0), f'N* ({args.csub} of 256 channels)')
    res = {
        'what': "the reference's own build_dfm_cost (dfm_backbone.py:217-314), PyTorch-CPU fp32, "
                'executed unmodified (AST lift, tools/ref_cpu_timing.py)',
        'host': 'build container', 'cores': args.threads, 'torch': torch.__version__,
        'config_k': {'seconds_per_volume': round(tk, 4), 'value': round(1.0 / tk, 4),
                     'unit': 'cost-volumes/s', 'output_shape': list(shape_k)},
        'nstar': {'sample': f'{args.csub} of 256 channels x all D=112 planes x 94x311, fp32, median of '
                            f'{args.reps} repetitions, scaled by 256/{args.csub}',
                  'seconds_per_volume': round(tn * 256 / args.csub, 3),
                  'value': round(1.0 / (tn * 256 / args.csub), 5), 'unit': 'cost-volumes/s',
                  'measured_seconds': round(tn, 4), 'output_shape': list(shape_n)},
    }
    with open(args.out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
