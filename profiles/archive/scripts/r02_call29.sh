#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c29; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt29 -- python $GRAFT_REPO_ROOT/tools/path_timing.py stereo --iters 5 > $GRAFT_REPO_ROOT/$O/run.txt 2>&1)
grep -v "MIOpen(HIP)\|amdgpu.ids" $O/run.txt | grep -i "path\|neck\|backbone\|head\|frustum\|hourglass\|Error\|Trace" | head -12
python - <<'PY' > gpurun_out/c29/stereo_kernel_stats.txt
import csv,glob
f=glob.glob('/tmp/kt29/**/*kernel_stats.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('# DfMStereoPath inference, tools/path_timing.py stereo --iters 5 (7 whole passes + 7 passes of each part); total', round(tot/1e6,2),'ms')
for r in rows[:45]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:10.3f} ms {float(r['AverageNs'])/1e3:10.1f} us {100*float(r['TotalDurationNs'])/tot:6.2f}%  {r['Name'][:150]}")
PY
cat $O/stereo_kernel_stats.txt | cut -c1-200
