#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c56; mkdir -p $O
timeout 900 python -m pytest tests/test_conv3d_gpu.py tests/test_group_norm.py tests/test_modules.py -x -q -m gpu 2>&1 | tail -3
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt56 -- python $GRAFT_REPO_ROOT/bench.py --workload backbone --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$O/bench_backbone_prof.json 2>/dev/null)
python - > $O/backbone_kernel_stats.txt <<'PY'
import csv,glob
f=glob.glob('/tmp/kt56/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f'# bench.py --workload backbone --steps 10 --warmup 3 under rocprofv3 --kernel-trace --stats; total {tot/1e6:.2f} ms over 13 passes')
for r in rows[:16]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%  {r['Name'][:130]}")
PY
cat $O/backbone_kernel_stats.txt | cut -c1-170
timeout 200 python bench.py --workload backbone 2>/dev/null | cut -c1-200
