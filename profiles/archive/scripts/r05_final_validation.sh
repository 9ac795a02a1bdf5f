#!/bin/bash
OUT=gpurun_out/r05final; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $OUT/gpu_suite.txt
cat $OUT/gpu_suite.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 3000 $OUT/bench_default.json
