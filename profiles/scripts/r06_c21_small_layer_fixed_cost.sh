# round 6, call 21: what the hourglass's small layers spend OUTSIDE their tap loops (debug build of conv3d_g.hip:
# DFM_CONV_ABLATE bit 6 skips the tap loops; 8 the stores; 1 re-staging; 2 weight loads)
mkdir -p gpurun_out/c21
export DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_convdbg.so
for ab in 0 64 72 75 0 64; do echo "== DFM_CONV_ABLATE=$ab"; DFM_CONV_ABLATE=$ab python tools/conv_g_timing.py --only hg --no-miopen --iters 30 2>/dev/null; done > gpurun_out/c21/ablation.txt 2>&1
python - <<'PY' >> gpurun_out/c21/ablation.txt 2>&1
# an empty-ish launch for scale: the same timing loop around a trivial kernel (torch add on 1 element)
import torch
x = torch.zeros(1, device='cuda')
for _ in range(5): x.add_(1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): x.add_(1)
e1.record(); torch.cuda.synchronize()
print('== a trivial launch back to back: %.1f us' % (e0.elapsed_time(e1) / 30 * 1e3))
PY
