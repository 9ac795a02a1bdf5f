#!/bin/bash
# round 4, call 56: LDS-atomic backward tile kernel with a column-window slab for strided sweeps (4 channels per pass, equal bands)
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_sweep_walk_gpu.py tests/test_plane_sweep_gpu.py tests/test_backward_gpu.py -x -q -m gpu -k 'walk or (backward and (clt or lds256_p2 or gather)) or test_backward' 2>&1 | grep -v Warning | tail -6 ) > gpurun_out/r04_c56_tests.txt 2>&1
( for i in 1 2; do timeout 300 python bench.py --workload sweep_bwd_kitti --no-secondary --no-traffic --no-smi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sweep_bwd_kitti', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('bwd_kernel'))"; done ) > gpurun_out/r04_c56_bench.txt 2>&1
cat gpurun_out/r04_c56_tests.txt gpurun_out/r04_c56_bench.txt
