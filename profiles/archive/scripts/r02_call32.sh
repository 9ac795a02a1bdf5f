#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c32; mkdir -p $O
timeout 600 python -m pytest tests/test_conv3d_gpu.py tests/test_conv3d_g_gpu.py tests/test_modules.py -m gpu -x -q 2>&1 | tail -6
(cd /tmp && export TMPDIR=/tmp && timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt32 -- python $GRAFT_REPO_ROOT/tools/train_step_timing.py > $GRAFT_REPO_ROOT/$O/run.txt 2>&1)
grep -v "MIOpen(HIP)\|amdgpu.ids" $O/run.txt | grep -i "DfMBackbone\|Error\|Trace" | head -12
python - <<'PY' > gpurun_out/c32/train_step_kernel_stats.txt
import csv,glob
f=glob.glob('/tmp/kt32/**/*kernel_stats.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('# tools/train_step_timing.py (5 forward-only + 5 forward+backward passes of DfMBackbone, config K, bf16 NDHWC, MIOpen find off); total', round(tot/1e6,2),'ms')
for r in rows[:32]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:10.3f} ms {float(r['AverageNs'])/1e3:10.1f} us {100*float(r['TotalDurationNs'])/tot:6.2f}%  {r['Name'][:150]}")
PY
cat $O/train_step_kernel_stats.txt | cut -c1-190
