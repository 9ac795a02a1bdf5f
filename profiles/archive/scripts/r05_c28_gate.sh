#!/bin/bash
OUT=gpurun_out/r05c28; mkdir -p $OUT
timeout 600 python -m pytest tests/test_cost_gate_gpu.py -m gpu -q 2>&1 | grep -v "^$" | tail -40 > $OUT/tests.txt
cat $OUT/tests.txt
python tools/gate_timing.py 2>/dev/null | tee $OUT/gate_timing.txt
for rep in 1 2 3; do
for mode in old new; do
  if [ $mode = old ]; then export DFM_GATE_TORCH=1; else unset DFM_GATE_TORCH; fi
  echo -n "$mode backbone: " >> $OUT/ab.txt
  timeout 200 python bench.py --workload backbone --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('roofline',{}).get('frac'))" >> $OUT/ab.txt 2>&1
done; done
cat $OUT/ab.txt
