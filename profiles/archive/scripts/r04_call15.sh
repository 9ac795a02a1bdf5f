#!/bin/bash
# round 4, call 15: whole GPU suite after the fp32 split-precision path
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c15; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu --tb=short 2>&1 | grep -v "Warning\|warn\|forward_call\|^$\|Consider using\|assert float" | tail -40 | tee $O/gpu_suite.txt
