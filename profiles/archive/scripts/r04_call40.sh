#!/bin/bash
# round 4, call 40: the same, bench through the new entry point
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_sweep_walk_gpu.py tests/test_plane_sweep_gpu.py -x -q -m gpu -k 'walk or (backward and (clt or lds256_p2))' 2>&1 | grep -v Warning | tail -25 ) > gpurun_out/r04_c40_tests.txt 2>&1
( for i in 1 2; do timeout 300 python bench.py --workload sweep_bwd_kitti --no-secondary --no-traffic --no-smi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sweep_bwd_kitti', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('bwd_kernel'))"; done ) > gpurun_out/r04_c40_bench.txt 2>&1
cat gpurun_out/r04_c40_tests.txt gpurun_out/r04_c40_bench.txt
