#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c38; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt38 -- python $GRAFT_REPO_ROOT/tools/neck_train_timing.py > $GRAFT_REPO_ROOT/$O/run.txt 2>&1)
grep -v "MIOpen(HIP)\|amdgpu.ids" $O/run.txt | grep -i "Neck\|Error\|Trace" | head -12
python - <<'PY' > gpurun_out/c38/neck_train_kernel_stats.txt
import csv,glob
f=glob.glob('/tmp/kt38/**/*kernel_stats.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('# tools/neck_train_timing.py (5 training steps of OutdoorImVoxelNeck, config W, bf16 NDHWC); total', round(tot/1e6,2),'ms')
for r in rows[:24]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:10.3f} ms {float(r['AverageNs'])/1e3:10.1f} us {100*float(r['TotalDurationNs'])/tot:6.2f}%  {r['Name'][:150]}")
PY
cat $O/neck_train_kernel_stats.txt | cut -c1-190
