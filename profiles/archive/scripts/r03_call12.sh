#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c12; mkdir -p $O
timeout 300 python tools/train_adds_probe.py 2>&1 | tail -16 | tee $O/train_adds.txt
timeout 200 python bench.py --workload backbone_train 2>/dev/null | tail -1 | cut -c1-300 | tee $O/bench_train.txt
