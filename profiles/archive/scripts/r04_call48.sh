#!/bin/bash
# round 4, call 48: what the backward kernels' flushes cost as device-scope atomics: the same kernels with workgroup-scope
# (XCD-local L2) atomics -- NOT coherent between XCDs, measured only
cd /root/repo; mkdir -p gpurun_out
L=$PWD/depth-from-motion_amd/lib
run() { timeout 300 python bench.py --workload $1 --no-secondary --no-traffic --no-smi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('bwd_kernel'))"; }
( for v in "" l2atom; do echo "## ${v:-release (device scope)}"; export DFM_HIP_LIB=$L/libdfm_hip${v:+_$v}.so; run sweep_bwd; run sweep_bwd_kitti; done
  echo "## parity with the XCD-local variant (may fail: not coherent)"
  DFM_HIP_LIB=$L/libdfm_hip_l2atom.so timeout 600 python -m pytest tests/test_sweep_bwd_mfma_gpu.py tests/test_sweep_walk_gpu.py -q -m gpu 2>&1 | tail -3
) > gpurun_out/r04_c48_l2_atomics.txt 2>&1
cat gpurun_out/r04_c48_l2_atomics.txt
