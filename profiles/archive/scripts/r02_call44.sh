#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c44; mkdir -p $O
for i in 1 2 3; do timeout 400 python bench.py --no-cpu-baseline > $O/bench_default_$i.json 2> $O/err_$i.txt; python - $O/bench_default_$i.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j['roofline']
print('run', sys.argv[1][-6], j['value'], 'vol/s', j['ms_per_step'], 'ms', j['config'].get('launch'), 'kernel', r['kernel_ms'], 'frac', r['frac'], 'frac_step', r.get('frac_step'), 'traffic', r.get('traffic'))
PY
done
timeout 300 python -m pytest tests/test_plane_sweep_gpu.py tests/test_nstar_shipped_gpu.py -m gpu -x -q -k "autotune or shipped" 2>&1 | tail -2
