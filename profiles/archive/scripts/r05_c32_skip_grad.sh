#!/bin/bash
OUT=gpurun_out/r05c32; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_conv3d_gpu.py tests/test_conv3d_g_gpu.py tests/test_modules.py tests/test_path_parity_gpu.py tests/test_depth_fused_training_gpu.py tests/test_sweep_conv_gpu.py tests/test_fast_path.py -m gpu -x -q 2>&1 | tail -15 > $OUT/tests.txt
cat $OUT/tests.txt
for rep in 1 2; do
for mode in plain fused; do
  if [ $mode = plain ]; then export DFM_NO_SKIP_GRAD_FUSION=1; else unset DFM_NO_SKIP_GRAD_FUSION; fi
  echo -n "$mode backbone_train: " >> $OUT/ab.txt
  timeout 200 python bench.py --workload backbone_train --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('roofline',{}).get('frac'))" >> $OUT/ab.txt 2>&1
  echo -n "$mode: " >> $OUT/ab.txt
  timeout 300 python tools/stereo_train_timing.py --dtype bf16 --iters 5 --fused-only 2>/dev/null | tail -1 >> $OUT/ab.txt
done; done
cat $OUT/ab.txt
