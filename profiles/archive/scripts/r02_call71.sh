#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c71; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt71 -- python $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/$O/bench_default_rocprof.json 2>/dev/null)
python - > $O/bench_kernel_stats.txt <<'PY'
import csv,glob
f=glob.glob('/tmp/kt71/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print('# rocprofv3 --kernel-trace --stats -- python bench.py   (default N* workload, final round-2 build)')
for r in rows[:8]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:10.3f} ms {float(r['AverageNs'])/1e3:10.1f} us {float(r['Percentage']):6.2f}%  {r['Name'][:140]}")
PY
cat $O/bench_kernel_stats.txt | cut -c1-200
python - $O/bench_default_rocprof.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j['roofline']
print('bench (under rocprof)', j['value'], j['unit'], j['ms_per_step'], 'ms', j['config'].get('launch'), 'kernel', r['kernel_ms'], 'frac', r['frac'], 'frac_step', r.get('frac_step'))
PY
