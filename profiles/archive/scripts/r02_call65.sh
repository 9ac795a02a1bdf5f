#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c65; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_err.txt; python - $O/bench_default.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j['roofline']
print('bench', j['value'], j['unit'], j['ms_per_step'], 'ms', j['config'].get('launch'), 'kernel', r['kernel_ms'], 'frac', r['frac'], 'frac_step', r.get('frac_step'), 'traffic', r.get('traffic'), 'cpu', j['cpu_baseline']['value'])
PY
tail -2 $O/bench_err.txt
