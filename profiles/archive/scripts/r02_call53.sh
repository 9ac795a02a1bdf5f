#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c53; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $O/gpu_tests.txt
for w in depth_head depth_head_bf16 f2v_cl waymo_cl; do
  timeout 200 python bench.py --workload $w 2>$O/err_$w.txt > $O/bench_$w.json
  python - $O/bench_$w.json <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); r=j['roofline']
    print(f"{j['config']['workload'][:60]:60s} {j['ms_per_step']:8.3f} ms/step B={j['config']['global_batch']} {r['achieved']:8.1f} GB/s frac {r['frac']}")
except Exception as e:
    print('FAILED', sys.argv[1], e)
PY
done
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt53 -- python $GRAFT_REPO_ROOT/bench.py --workload backbone --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$O/bench_backbone.json 2>/dev/null)
python - > $O/backbone_kernel_stats.txt <<'PY'
import csv,glob
f=glob.glob('/tmp/kt53/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f'# bench.py --workload backbone --steps 10 --warmup 3 under rocprofv3 --kernel-trace --stats; total {tot/1e6:.2f} ms')
for r in rows[:28]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%  {r['Name'][:150]}")
PY
cat $O/backbone_kernel_stats.txt | cut -c1-200; cat $O/bench_backbone.json | cut -c1-300
