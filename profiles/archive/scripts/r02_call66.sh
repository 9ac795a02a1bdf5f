#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c66; mkdir -p $O
timeout 900 python -m pytest tests/test_plane_sweep_gpu.py -x -q -m gpu -k "channels_last or clt or kitti or fixture" 2>&1 | tail -3 | tee $O/tests.txt
for w in kitti kitti_nhwc; do
  timeout 300 python bench.py --workload $w > $O/bench_$w.json 2> $O/err_$w.txt; python - $O/bench_$w.json <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); r=j['roofline']
    print(j['config']['workload'][:40], j['value'], j['unit'], j['ms_per_step'], 'ms kernel', r['kernel_ms'], 'frac', r['frac'], 'frac_step', r.get('frac_step'), j['config'].get('kernel'))
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('bench_','err_').replace('.json','.txt')).read()[-800:])
PY
done
