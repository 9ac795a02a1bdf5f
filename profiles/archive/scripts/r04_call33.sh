#!/bin/bash
# round 4, call 33: the walking kernel compiled for 5 / 4 / 3 waves per SIMD (offsets and weights re-read from the LDS ring instead of held in registers)
cd /root/repo; mkdir -p gpurun_out
L=$PWD/depth-from-motion_amd/lib
run() { timeout 300 python bench.py --workload kitti_nhwc --no-secondary --no-traffic --no-smi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kitti_nhwc', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms'))"; }
( for v in "" w4 w3; do
    echo "## ${v:-release (5 waves)}"
    DFM_HIP_LIB=$L/libdfm_hip${v:+_$v}.so timeout 600 python -m pytest tests/test_sweep_walk_gpu.py -x -q -m gpu 2>&1 | tail -1
    DFM_HIP_LIB=$L/libdfm_hip${v:+_$v}.so run; DFM_HIP_LIB=$L/libdfm_hip${v:+_$v}.so run
  done
) > gpurun_out/r04_c33_waves.txt 2>&1
cat gpurun_out/r04_c33_waves.txt
