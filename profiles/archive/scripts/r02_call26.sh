#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c26; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt26 -- python $GRAFT_REPO_ROOT/tools/path_timing.py mv --iters 3 > $GRAFT_REPO_ROOT/$O/run.txt 2>&1)
grep -v "MIOpen(HIP)\|amdgpu.ids" $O/run.txt | grep -i "path\|lifting\|neck\|Error\|Trace" | head
python - <<'PY' > gpurun_out/c26/mv_kernel_stats.txt
import csv,glob
f=glob.glob('/tmp/kt26/**/*kernel_stats.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('# MultiViewVoxelPath (both Waymo configs), tools/path_timing.py mv --iters 3; total', round(tot/1e6,2),'ms')
for r in rows[:25]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:10.3f} ms {float(r['AverageNs'])/1e3:10.1f} us {100*float(r['TotalDurationNs'])/tot:6.2f}%  {r['Name'][:150]}")
PY
cat $O/mv_kernel_stats.txt | cut -c1-190
