import importlib, sys, os, torch
sys.path.insert(0,'/root/repo')
mods=importlib.import_module('depth-from-motion_amd.modules')
dev=torch.device('cuda:0')
for shape in ((1,32,160,640),(1,64,80,320)):
    x=torch.randn(*shape,device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y=mods.bilinear_resize(x, scale_factor=2.0, align_corners=False)
    gy=torch.randn_like(y)
    for flag in (True, False, True, False):
        mods._BILINEAR_GATHER=flag
        for _ in range(3): torch.autograd.grad(y,x,gy,retain_graph=True)
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        import time
        t0=time.perf_counter(); e0.record()
        for _ in range(20): torch.autograd.grad(y,x,gy,retain_graph=True)
        e1.record(); t1=time.perf_counter(); torch.cuda.synchronize()
        print(shape, 'gather' if flag else 'matmul', 'device %.1f us/call, host enqueue %.1f us/call'%(e0.elapsed_time(e1)*50, (t1-t0)*5e4))
