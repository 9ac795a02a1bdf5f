#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c10; mkdir -p $O
timeout 600 python -m pytest tests/test_point_sample_gpu.py tests/test_data_geometry.py tests/test_path_parity_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/tests.txt
for w in waymo_cl waymo; do timeout 200 python bench.py --workload $w > $O/bench_$w.json 2>/dev/null; tail -1 $O/bench_$w.json | cut -c1-600; done
