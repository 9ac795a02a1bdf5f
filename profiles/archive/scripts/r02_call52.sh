#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c52; mkdir -p $O
timeout 900 python -m pytest tests/test_depth_head.py tests/test_frustum_to_voxel.py tests/test_point_sample_gpu.py tests/test_backward_gpu.py -x -q -m gpu 2>&1 | tail -3
for w in waymo waymo_cl depth_head depth_head_bf16 f2v f2v_cl group_norm group_norm_cl; do
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
tail -3 $O/err_waymo_cl.txt
