#!/bin/bash
OUT=gpurun_out/r05c27; mkdir -p $OUT
timeout 900 python -m pytest tests/test_cost_gate_gpu.py tests/test_modules.py tests/test_conv3d_gpu.py tests/test_sweep_conv_gpu.py tests/test_group_norm.py tests/test_path_parity_gpu.py tests/test_fast_path.py -m gpu -x -q 2>&1 | tail -6 > $OUT/tests.txt
cat $OUT/tests.txt
for rep in 1 2; do
for mode in old new; do
  if [ $mode = old ]; then export DFM_GN_MERGE_KERNEL=1 DFM_GATE_TORCH=1; else unset DFM_GN_MERGE_KERNEL DFM_GATE_TORCH; fi
  for wl in backbone backbone_train; do
    echo -n "$mode $wl: " >> $OUT/ab.txt
    timeout 200 python bench.py --workload $wl --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('roofline',{}).get('frac'))" >> $OUT/ab.txt 2>&1
  done
done; done
cat $OUT/ab.txt
