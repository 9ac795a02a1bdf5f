#!/bin/bash
OUT=gpurun_out/r05c50; mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv3d_gpu.py tests/test_modules.py -m gpu -x -q 2>&1 | tail -2 > $OUT/tests.txt; cat $OUT/tests.txt
for rep in 1 2; do
for mode in old new; do
  if [ $mode = old ]; then export DFM_CONV_OLD_CHUNK=1; else unset DFM_CONV_OLD_CHUNK; fi
  for wl in backbone backbone_train; do
    echo -n "$mode $wl: " >> $OUT/ab.txt
    timeout 200 python bench.py --workload $wl --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('roofline',{}).get('frac'))" >> $OUT/ab.txt 2>&1
  done
done; done
cat $OUT/ab.txt
