#!/usr/bin/env python
"""General MFMA Conv3d / ConvTranspose3d (csrc/conv3d_g.hip) vs torch/MIOpen (bf16, channels_last_3d)
at the hourglass (config K) and voxel-neck (config W) shapes.  GPU box.
usage: python tools/conv_g_timing.py [--only hg|neck]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cv = importlib.import_module('depth-from-motion_amd.conv3d')
torch.backends.cudnn.benchmark = True
dev = torch.device('cuda:0')

HG = [  # name, kind, cin, cout, in_size, stride, padding
    ('hg.conv1 32->64 s2', 'conv', 32, 64, (72, 80, 320), 2, 1),
    ('hg.conv2 64->64', 'conv', 64, 64, (36, 40, 160), 1, 1),
    ('hg.conv3 64->64 s2', 'conv', 64, 64, (36, 40, 160), 2, 1),
    ('hg.conv4 64->64', 'conv', 64, 64, (18, 20, 80), 1, 1),
    ('hg.conv5 T 64->64', 'convT', 64, 64, (18, 20, 80), 2, 1),
    ('hg.conv6 T 64->32', 'convT', 64, 32, (36, 40, 160), 2, 1),
]
NECK = [
    ('neck.res0 64->64', 'conv', 64, 64, (220, 300, 12), 1, 1),
    ('neck.down0 64->128 s(1,1,2)', 'conv', 64, 128, (220, 300, 12), (1, 1, 2), 1),
    ('neck.res1 128->128', 'conv', 128, 128, (220, 300, 6), 1, 1),
    ('neck.down1 128->256 s(1,1,2)', 'conv', 128, 256, (220, 300, 6), (1, 1, 2), 1),
    ('neck.res2 256->256', 'conv', 256, 256, (220, 300, 3), 1, 1),
    ('neck.out 256->256 p(1,1,0)', 'conv', 256, 256, (220, 300, 3), 1, (1, 1, 0)),
    ('dfmneck.res0 128->128', 'conv', 128, 128, (220, 300, 12), 1, 1),
    # candidates for a single-pass 64 -> 32 (today: two 32 -> 32 passes through an fp32 partial)
    ('dres0 64->32', 'conv', 64, 32, (72, 80, 320), 1, 1),
    ('voxel_convs 64->32', 'conv', 64, 32, (20, 304, 288), 1, 1),
]


GRAPH = [False]


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if GRAPH[0]:
        # device time: `iters` launches captured once into a HIP graph and replayed -- a call costs the host ~21 us
        # to enqueue (tools/host_overhead_probe.py), more than the hourglass's small layers take on the device
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (3 * iters)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    ap.add_argument('--no-miopen', action='store_true')
    ap.add_argument('--case', default='', help='only the cases whose name contains this')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--graph', action='store_true', help='device time: launches replayed from a captured HIP graph')
    args = ap.parse_args()
    GRAPH[0] = args.graph
    cases = (HG if args.only != 'neck' else []) + (NECK if args.only != 'hg' else [])
    for name, kind, cin, cout, size, stride, padding in cases:
        if args.case and args.case not in name:
            continue
        x = torch.randn(1, cin, *size, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
        if kind == 'conv':
            ref = torch.nn.Conv3d(cin, cout, 3, stride=stride, padding=padding, bias=False)
            tr = False
        else:
            ref = torch.nn.ConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False)
            tr = True
        ref = ref.to(dev).bfloat16().to(memory_format=torch.channels_last_3d)
        pk = cv.pack_conv3d_g_weights(ref.weight, cin, cout, swap=tr)
        st = (1, 1, 1) if tr else cv._triple(stride)
        pd = (1, 1, 1) if tr else cv._triple(padding)
        plan = cv.conv3d_g_plan(1, cin, cout, size, st, pd, tr)
        with torch.no_grad():
            y = cv.conv3d_g(x, pk, cout, st, pd, tr)
            vox = y.numel() // cout
            flops = 2 * 27 * cin * cout * vox / (8 if tr else 1)
            t = timeit(lambda: cv.conv3d_g(x, pk, cout, st, pd, tr), args.iters)
            line = (f'{name:30s} {str(size):16s} MFMA {t:7.3f} ms {flops / t / 1e9:7.1f} TFLOP/s '
                    f'({flops / t / 1e9 / 25:4.1f} %)  plan pfw={plan["pfw"]} tile={plan["tile"]} '
                    f'lds={plan["lds"] // 1024}K wgs={plan["workgroups"]}')
            if not args.no_miopen:
                t_mi = timeit(lambda: ref(x))
                yr = ref(x)
                err = float((y.float() - yr.float()).abs().max())
                line += f' | MIOpen {t_mi:7.3f} ms {flops / t_mi / 1e9:7.1f} TFLOP/s | max diff {err:.3g}'
            print(line, flush=True)
        del x


if __name__ == '__main__':
    main()
