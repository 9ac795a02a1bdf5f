#!/bin/bash
OUT=gpurun_out/r05c30; mkdir -p $OUT
timeout 300 python -m pytest tests/test_long_axis_gram.py tests/test_cost_gate_gpu.py -q 2>&1 | tail -2 > $OUT/tests.txt
for rep in 1 2; do
for mode in plain split; do
  if [ $mode = plain ]; then export DFM_PLAIN_WGRAD_1X1=1; else unset DFM_PLAIN_WGRAD_1X1; fi
  echo -n "$mode: " >> $OUT/ab.txt
  timeout 300 python tools/stereo_train_timing.py --dtype bf16 --iters 5 --fused-only 2>/dev/null | tail -1 >> $OUT/ab.txt
done; done
cat $OUT/tests.txt $OUT/ab.txt
