#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c54; mkdir -p $O
timeout 900 python -m pytest tests/test_depth_head.py tests/test_frustum_to_voxel.py tests/test_group_norm.py tests/test_backward_gpu.py tests/test_modules.py -x -q -m gpu 2>&1 | tail -3
for w in depth_head depth_head_bf16 group_norm_cl backbone; do
  timeout 200 python bench.py --workload $w 2>$O/err_$w.txt > $O/bench_$w.json
  python - $O/bench_$w.json <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); r=j['roofline']
    print(f"{j['config']['workload'][:60]:60s} {j['ms_per_step']:8.3f} ms/step B={j['config']['global_batch']} {r['achieved']:8.1f} {r['unit']} frac {r['frac']}")
except Exception as e:
    print('FAILED', sys.argv[1], e)
PY
done
timeout 300 python tools/path_timing.py stereo --iters 5 2>&1 | tail -8 | tee $O/path_timing_stereo.txt
