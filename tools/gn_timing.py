"""GroupNorm(32,32)+ReLU forward/backward on the config-K cost volume: fused HIP kernels vs torch."""
import importlib, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module('depth-from-motion_amd')
dev = torch.device('cuda:0')
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for dt in (torch.float32, torch.bfloat16):
    x = (torch.randn(1, 32, 72, 80, 320, device=dev) + 0.5).to(dt).requires_grad_(True)
    m = pkg.HipGroupNorm(32, 32).to(dev)
    gy = torch.randn_like(x)
    def f():
        with torch.no_grad(): m(x, relu=True)
    def fb():
        x.grad = None; m(x, relu=True).backward(gy)
    a, b = timed(f), timed(fb)
    print(f'GroupNorm(32,32)+ReLU K volume {str(dt)[6:]:9s} fwd {a:.3f} ms  fwd+bwd {b:.3f} ms  bwd ~{b-a:.3f} ms', flush=True)
    tm = torch.nn.GroupNorm(32, 32).to(dev).to(dt)
    def tfb():
        x.grad = None; torch.relu(tm(x)).backward(gy)
    print(f'   torch GroupNorm+ReLU fwd+bwd {timed(tfb):.3f} ms', flush=True)
