#!/usr/bin/env python
"""MIOpen (through torch) timing of the path's Conv3d shapes -- the baseline a
hand-written MFMA implicit-GEMM conv has to beat."""
import time, torch
torch.backends.cudnn.benchmark = True
dev = torch.device('cuda:0')
def bench(name, mod, x, flops, iters=10):
    mod = mod.to(dev); x = x.to(dev)
    for dt in (torch.float32, torch.bfloat16):
        m = mod.to(dt); xx = x.to(dt)
        mem_fmt = [('ncdhw', torch.contiguous_format)] + ([('ndhwc', torch.channels_last_3d)] if x.dim() == 5 else [])
        for fname, fmt in mem_fmt:
            try:
                mm = m.to(memory_format=fmt); xi = xx.contiguous(memory_format=fmt)
                with torch.no_grad():
                    for _ in range(3): mm(xi)
                    torch.cuda.synchronize(); t = time.perf_counter()
                    for _ in range(iters): mm(xi)
                    torch.cuda.synchronize(); ms = (time.perf_counter() - t) * 1e3 / iters
                print(f'{name:34s} {str(dt)[6:]:9s} {fname}: {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s', flush=True)
            except Exception as e:
                print(name, dt, fname, 'failed', str(e)[:80])
V = 72 * 80 * 320
bench('dres0 Conv3d 64->32 K (72,80,320)', torch.nn.Conv3d(64, 32, 3, padding=1, bias=False), torch.randn(1, 64, 72, 80, 320), 2 * 27 * 64 * 32 * V)
bench('dres1 Conv3d 32->32 K', torch.nn.Conv3d(32, 32, 3, padding=1, bias=False), torch.randn(1, 32, 72, 80, 320), 2 * 27 * 32 * 32 * V)
bench('hg conv1 32->64 s2', torch.nn.Conv3d(32, 64, 3, stride=2, padding=1, bias=False), torch.randn(1, 32, 72, 80, 320), 2 * 27 * 32 * 64 * V / 8)
bench('hg conv2 64->64 (36,40,160)', torch.nn.Conv3d(64, 64, 3, padding=1, bias=False), torch.randn(1, 64, 36, 40, 160), 2 * 27 * 64 * 64 * V / 8)
bench('GroupNorm(32,32) K', torch.nn.GroupNorm(32, 32), torch.randn(1, 32, 72, 80, 320), 10 * 32 * V)
bench('neck Conv3d 64->64 (220,300,12)', torch.nn.Conv3d(64, 64, 3, padding=1, bias=False), torch.randn(1, 64, 220, 300, 12), 2 * 27 * 64 * 64 * 792000)
bench('neck Conv3d 256->256 (220,300,3)', torch.nn.Conv3d(256, 256, 3, padding=1, bias=False), torch.randn(1, 256, 220, 300, 3), 2 * 27 * 256 * 256 * 198000)
