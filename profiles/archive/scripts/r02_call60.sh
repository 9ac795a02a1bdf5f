#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c60; mkdir -p $O
cat > /tmp/neck_prof.py <<'PY'
import importlib, os, sys, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'tools'))
import path_timing as pt
pkg, dev = pt.pkg, pt.dev
model = dict(pt.cfg('dfm_r34_1x8_kitti-3d-3class.py'))
torch.manual_seed(0)
path = pkg.DfMStereoPath(model).to(dev).eval().to(torch.bfloat16)
H, W = 320, 1280
gen = torch.Generator().manual_seed(1)
feats = [torch.randn(1, c, H // s, W // s, generator=gen).to(dev).bfloat16().contiguous(memory_format=torch.channels_last) for c, s in ((3, 1), (64, 2), (128, 4), (128, 4), (128, 4))]
with torch.no_grad():
    for _ in range(12):
        path.neck(feats)
    torch.cuda.synchronize()
PY
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt60 -- python /tmp/neck_prof.py > /dev/null 2>$GRAFT_REPO_ROOT/$O/err.txt)
python - > $O/sppunet_kernel_stats.txt <<'PY'
import csv,glob
f=glob.glob('/tmp/kt60/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f'# SPPUNetNeck.forward x 12 (config K, bf16 channels_last) under rocprofv3 --kernel-trace --stats; total {tot/1e6:.2f} ms')
for r in rows[:30]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%  {r['Name'][:120]}")
PY
cat $O/sppunet_kernel_stats.txt | cut -c1-175; tail -3 $O/err.txt
