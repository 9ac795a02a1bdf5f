#!/bin/bash
# round 4, call 20: matrix-core unpack of the N* forward tile kernel -- parity, then A/B against the VALU unpack
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_plane_sweep_gpu.py -x -q -m gpu -k "pipe" 2>&1 | tail -5
  timeout 600 python -m pytest tests/test_nstar_shipped_gpu.py -x -q -m gpu 2>&1 | tail -3 ) > gpurun_out/r04_c20_tests.txt 2>&1
( for i in 1 2; do
  timeout 300 tools/sweep_bench --workload nstar --rounds 3 --launches 4 unpack=2 unpack=1 lanes=512,ppl=4 chunk=2,unpack=1 chunk=2,unpack=2
  done ) > gpurun_out/r04_c20_unpack_ab.txt 2>&1
timeout 600 python bench.py --no-secondary > gpurun_out/r04_c20_bench.json 2> gpurun_out/r04_c20_bench.err
tail -5 gpurun_out/r04_c20_tests.txt; cat gpurun_out/r04_c20_unpack_ab.txt; cat gpurun_out/r04_c20_bench.json
