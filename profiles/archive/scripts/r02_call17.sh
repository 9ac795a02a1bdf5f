#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c17; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -25 $O/pytest_gpu.txt
timeout 200 python tools/conv_g_timing.py --only hg --no-miopen 2>&1 | grep -v "MIOpen(HIP)" > $O/conv_g_timing_hg.txt; cat $O/conv_g_timing_hg.txt
export DFM_ONLY=bf16 DFM_ONLY_FMT=cl DFM_ITERS=20
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt17 -- python $GRAFT_REPO_ROOT/tools/backbone_timing.py > $GRAFT_REPO_ROOT/$O/backbone_run.txt 2>&1)
grep -v MIOpen $O/backbone_run.txt | grep DfMBackbone
python - <<'PY' > gpurun_out/c17/backbone_kernel_stats.txt
import csv,glob
f=glob.glob('/tmp/kt17/**/*kernel_stats.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'naive_conv' not in r['Name']]
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('# DfMBackbone.forward bf16 channels_last_3d, 22 forward passes, MIOpen find off; total', round(tot/1e6,2),'ms =', round(tot/1e6/22,3), 'ms per pass')
for r in rows[:40]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:10.3f} ms {float(r['AverageNs'])/1e3:10.1f} us {100*float(r['TotalDurationNs'])/tot:6.2f}%  {r['Name'][:150]}")
PY
cat $O/backbone_kernel_stats.txt | cut -c1-180
timeout 300 python tools/backbone_timing.py 2>&1 | grep -v MIOpen > $O/backbone_plain.txt; cat $O/backbone_plain.txt
