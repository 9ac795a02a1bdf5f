#!/bin/bash
OUT=gpurun_out/r05c43; mkdir -p $OUT
timeout 900 python -m pytest tests/test_depth_head.py tests/test_depth_fused_training_gpu.py -m gpu -x -q 2>&1 | tail -3 > $OUT/tests.txt
cat $OUT/tests.txt
L=depth-from-motion_amd/lib
for rep in 1 2; do
for v in _olddh ""; do
  export DFM_HIP_LIB=$PWD/$L/libdfm_hip$v.so
  for wl in depth_head depth_head_bf16; do
  echo -n "lib$v $wl: " >> $OUT/ab.txt
  timeout 200 python bench.py --workload $wl --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('roofline',{}).get('frac'))" >> $OUT/ab.txt 2>&1
  done
done; done
unset DFM_HIP_LIB
cat $OUT/ab.txt
