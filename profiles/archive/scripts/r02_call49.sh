#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c49; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/gpu_tests.txt
