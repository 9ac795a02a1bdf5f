#!/bin/bash
# round 4, call 51: FrustumToVoxel channels-last output, several lanes per voxel, one source at a time, two voxels per lane in flight
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_frustum_to_voxel.py tests/test_depth_fused_training_gpu.py -x -q -m gpu 2>&1 | grep -v Warning | tail -30 ) > gpurun_out/r04_c51_tests.txt 2>&1
run() { timeout 300 python bench.py --workload $1 --no-secondary --no-traffic --no-smi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline']['frac'])"; }
( run f2v_cl; run f2v_cl; run f2v ) > gpurun_out/r04_c51_bench.txt 2>&1
cat gpurun_out/r04_c51_tests.txt gpurun_out/r04_c51_bench.txt
