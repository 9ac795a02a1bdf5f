#!/bin/bash
# round 4, call 21: late stores (stores of block k issued after the barrier of block k: two block periods to be acknowledged)
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_plane_sweep_gpu.py -x -q -m gpu -k "late" 2>&1 | tail -5 ) > gpurun_out/r04_c21_tests.txt 2>&1
( for i in 1 2; do
  timeout 300 tools/sweep_bench --workload nstar --rounds 3 --launches 4 pipe=2 pipe=3 pipe=3,unpack=2 pipe=2,unpack=2 lanes=512,ppl=4,pipe=3 lanes=512,ppl=4,pipe=2
  done ) > gpurun_out/r04_c21_late_ab.txt 2>&1
tail -5 gpurun_out/r04_c21_tests.txt; cat gpurun_out/r04_c21_late_ab.txt
