#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c20; mkdir -p $O
timeout 300 python -m pytest tests/test_modules.py -m gpu -x -q -k "res_module or wide" 2>&1 | tail -40 > $O/pytest_modules.txt; cat $O/pytest_modules.txt
for W in backbone neck dfm_neck; do timeout 200 python bench.py --workload $W --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_$W.json; cat $O/bench_$W.json; done
