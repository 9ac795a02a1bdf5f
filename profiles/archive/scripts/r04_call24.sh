#!/bin/bash
# round 4, call 24: one product then four VALU operations, strictly alternating (pinned)
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_plane_sweep_gpu.py -x -q -m gpu -k "halves or pipe256-" 2>&1 | tail -5 ) > gpurun_out/r04_c24_tests.txt 2>&1
( for i in 1 2; do
  timeout 300 tools/sweep_bench --workload nstar --rounds 3 --launches 4 unpack=1 unpack=3 unpack=2 chunk=2,unpack=3 lanes=512,ppl=4
  done ) > gpurun_out/r04_c24_halves_ab.txt 2>&1
tail -5 gpurun_out/r04_c24_tests.txt; cat gpurun_out/r04_c24_halves_ab.txt
