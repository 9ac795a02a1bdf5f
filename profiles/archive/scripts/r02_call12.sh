#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c12; mkdir -p $O
export DFM_ONLY=bf16 DFM_ONLY_FMT=cl DFM_MIOPEN_FIND=1 DFM_ITERS=40
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt12 -- python $GRAFT_REPO_ROOT/tools/backbone_timing.py > $GRAFT_REPO_ROOT/$O/run.txt 2>&1)
grep -v MIOpen $O/run.txt | tail -3
python - <<'PY' > gpurun_out/c12/backbone_kernel_stats.txt
import csv,glob
f=glob.glob('/tmp/kt12/**/*kernel_stats.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'naive_conv' not in r['Name']]
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('# DfMBackbone.forward bf16 channels_last_3d, 42 forward passes + MIOpen find (naive reference kernels excluded); total', round(tot/1e6,2),'ms')
for r in rows[:32]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:10.3f} ms {float(r['AverageNs'])/1e3:10.1f} us {100*float(r['TotalDurationNs'])/tot:6.2f}%  {r['Name'][:140]}")
PY
cat $O/backbone_kernel_stats.txt
