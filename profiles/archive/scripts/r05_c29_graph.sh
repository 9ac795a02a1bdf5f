#!/bin/bash
OUT=gpurun_out/r05c29; mkdir -p $OUT
for rep in 1 2; do
for mode in eager graph; do
  if [ $mode = graph ]; then export DFM_BENCH_GRAPH=1; else unset DFM_BENCH_GRAPH; fi
  for wl in backbone neck dfm_neck; do
    echo -n "$mode $wl: " >> $OUT/ab.txt
    timeout 200 python bench.py --workload $wl --steps 20 --warmup 5 2>$OUT/err_${mode}_$wl.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('roofline',{}).get('frac'), d['config'].get('hip_graph_replay'))" >> $OUT/ab.txt 2>&1
  done
done; done
cat $OUT/ab.txt; tail -5 $OUT/err_graph_backbone.txt
