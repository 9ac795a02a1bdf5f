#!/bin/bash
OUT=gpurun_out/r05c36; mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv3d_g_gpu.py -m gpu -x -q -k "weight_gradient or wgrad or train" 2>&1 | tail -4 > $OUT/tests.txt
DFM_WGRAD_CHUNK=2 timeout 900 python -m pytest tests/test_conv3d_g_gpu.py -m gpu -x -q -k "weight_gradient" 2>&1 | tail -2 >> $OUT/tests.txt
cat $OUT/tests.txt
for rep in 1 2; do
for mode in 0 1; do
  export DFM_WGRAD_WALK=$mode
  echo -n "walk=$mode backbone_train: " >> $OUT/ab.txt
  timeout 200 python bench.py --workload backbone_train --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('roofline',{}).get('frac'))" >> $OUT/ab.txt 2>&1
  echo -n "walk=$mode: " >> $OUT/ab.txt
  timeout 300 python tools/stereo_train_timing.py --dtype bf16 --iters 5 --fused-only 2>/dev/null | tail -1 >> $OUT/ab.txt
done; done
unset DFM_WGRAD_WALK
cat $OUT/ab.txt
