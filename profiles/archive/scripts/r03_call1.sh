#!/bin/bash
# round 3, call 1: the new end-to-end / layer-wise parity tests, then the whole GPU suite, then the default bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c1; mkdir -p $O
timeout 900 python -m pytest tests/test_path_parity_gpu.py -q -m gpu -x 2>&1 | tail -60 > $O/parity.txt; tail -40 $O/parity.txt
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.txt; tail -15 $O/pytest_gpu.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.json
