#!/bin/bash
# round 4, call 31: the walking kernel -- its own parity tests; where its time goes (no tap loads / no stores, release speed)
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_sweep_walk_gpu.py -x -q -m gpu 2>&1 | tail -4 ) > gpurun_out/r04_c31_tests.txt 2>&1
L=depth-from-motion_amd/lib
( for v in "" wnoload wnostore; do
    lib=$L/libdfm_hip${v:+_$v}.so
    echo "## ${v:-release}"
    LD_PRELOAD=$PWD/$lib timeout 300 python bench.py --workload kitti_nhwc --no-secondary --no-traffic --no-smi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kitti_nhwc', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms'))"
  done ) > gpurun_out/r04_c31_ablate.txt 2>&1
cat gpurun_out/r04_c31_tests.txt gpurun_out/r04_c31_ablate.txt
