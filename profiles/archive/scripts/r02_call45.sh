#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c45; mkdir -p $O
for i in 1 2; do timeout 400 python bench.py > $O/bench_default_$i.json 2> $O/err_$i.txt; python - $O/bench_default_$i.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j['roofline']
print('run', j['value'], 'vol/s', j['ms_per_step'], 'ms', j['config'].get('launch'), j['config'].get('tuning_check_ms'), 'kernel', r['kernel_ms'], 'frac', r['frac'], 'frac_step', r.get('frac_step'), 'traffic', r.get('traffic'))
PY
tail -2 $O/err_$i.txt
done
