#!/bin/bash
# round 4, call 32: where the walking kernel's time goes (no tap loads / no stores / neither, release speed), depth-chunk length
cd /root/repo; mkdir -p gpurun_out
L=$PWD/depth-from-motion_amd/lib
run() { timeout 300 python bench.py --workload kitti_nhwc --no-secondary --no-traffic --no-smi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kitti_nhwc', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms'))"; }
( for v in "" wnoload wnostore wneither; do
    echo "## ${v:-release}"; DFM_HIP_LIB=$L/libdfm_hip${v:+_$v}.so run
  done
  for c in 72 36 24 12 6; do echo "## chunk $c"; DFM_WALK_CHUNK=$c DFM_HIP_LIB=$L/libdfm_hip_wchunk.so run; done
) > gpurun_out/r04_c32_ablate.txt 2>&1
cat gpurun_out/r04_c32_ablate.txt
