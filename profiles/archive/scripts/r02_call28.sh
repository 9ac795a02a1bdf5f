#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c28; mkdir -p $O
timeout 900 python -m pytest tests/test_plane_sweep_gpu.py tests/test_modules.py tests/test_nstar_shipped_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for K in 3 4 0; do timeout 200 python bench.py --workload kitti --kernel $K --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_kitti_k$K.json; python - $O/bench_kitti_k$K.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j['roofline']
print(j['config'].get('kernel'), 'value', j['value'], 'ms', j['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'frac_step', r.get('frac_step'))
PY
done
