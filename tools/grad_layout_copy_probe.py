#!/usr/bin/env python
"""What the autograd engine's layout copy of a small weight gradient costs (GPU box): a contiguous fp32 (A, B, 3, 3, 3)
gradient into the channels_last_3d bf16 layout of the parameter -- the copy AccumulateGrad makes when the strides of
the incoming gradient differ from the parameter's -- against forms that hand the gradient over in the right layout."""
import torch

dev = torch.device('cuda:0')
cl = torch.channels_last_3d


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for shape in ((32, 32, 3, 3, 3), (64, 64, 3, 3, 3), (128, 128, 3, 3, 3), (256, 256, 3, 3, 3)):
    A, B = shape[:2]
    w = torch.randn(shape, device=dev).bfloat16().contiguous(memory_format=cl)
    g = torch.randn(shape, device=dev)
    gb = g.bfloat16()
    print(shape, 'param strides', w.stride())
    print('   empty_like(w).copy_(g fp32 contiguous)      %7.1f us' % t(lambda: torch.empty_like(w).copy_(g)))
    print('   empty_like(w).copy_(g bf16 contiguous)      %7.1f us' % t(lambda: torch.empty_like(w).copy_(gb)))
    print('   g.to(bf16)                                  %7.1f us' % t(lambda: g.to(torch.bfloat16)))
    print('   g.view(A,B,27).transpose(1,2).contiguous()  %7.1f us' % t(lambda: g.view(A, B, 27).transpose(1, 2).contiguous()))
    print('   ... .to(bf16) viewed channels_last_3d       %7.1f us' % t(
        lambda: g.view(A, B, 27).transpose(1, 2).to(torch.bfloat16, memory_format=torch.contiguous_format)
        .view(A, 3, 3, 3, B).permute(0, 4, 1, 2, 3)))
    r = g.view(A, B, 27).transpose(1, 2).to(torch.bfloat16, memory_format=torch.contiguous_format).view(A, 3, 3, 3, B).permute(0, 4, 1, 2, 3)
    assert r.stride() == w.stride() and torch.equal(r, g.bfloat16())
