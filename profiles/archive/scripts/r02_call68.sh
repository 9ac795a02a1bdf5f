#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c68; mkdir -p $O
timeout 900 python -m pytest tests/test_modules.py tests/test_conv3d_g_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.txt
for w in neck dfm_neck; do
  timeout 300 python bench.py --workload $w > $O/bench_$w.json 2> $O/err_$w.txt; python - $O/bench_$w.json <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); r=j['roofline']
    print(j['config']['workload'][:50], j['ms_per_step'], 'ms', r['achieved'], r['unit'], 'frac', r['frac'])
except Exception as e:
    print('FAILED', e)
PY
done
