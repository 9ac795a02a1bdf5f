#!/bin/bash
# round 4, call 59: FrustumToVoxel PLANAR output (reference layout) with several lanes per voxel, four consecutive voxels per lane: parity + bench
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_frustum_to_voxel.py tests/test_depth_fused_training_gpu.py tests/test_modules.py -x -q -m gpu 2>&1 | grep -v Warning | tail -4 ) > gpurun_out/r04_c59_tests.txt 2>&1
run() { timeout 300 python bench.py --workload $1 --no-secondary --no-traffic --no-smi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline']['frac'])"; }
( run f2v; run f2v; run f2v_cl ) > gpurun_out/r04_c59_bench.txt 2>&1
cat gpurun_out/r04_c59_tests.txt gpurun_out/r04_c59_bench.txt
