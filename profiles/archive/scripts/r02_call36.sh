#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c36; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt36 -- python $GRAFT_REPO_ROOT/tools/train_step_timing.py > $GRAFT_REPO_ROOT/$O/run.txt 2>&1)
grep -v "MIOpen(HIP)\|amdgpu.ids" $O/run.txt | grep -i "DfMBackbone\|Error\|Trace" | head -12
python - <<'PY' > gpurun_out/c36/train_step_kernel_stats.txt
import csv,glob
f=glob.glob('/tmp/kt36/**/*kernel_stats.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'naive_conv' not in r['Name']]
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('# tools/train_step_timing.py (5 forward-only + 5 forward+backward passes of DfMBackbone, config K, bf16 NDHWC; MIOpen naive warm-up kernels excluded); total', round(tot/1e6,2),'ms')
for r in rows[:30]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:10.3f} ms {float(r['AverageNs'])/1e3:10.1f} us {100*float(r['TotalDurationNs'])/tot:6.2f}%  {r['Name'][:150]}")
PY
cat $O/train_step_kernel_stats.txt | cut -c1-190
