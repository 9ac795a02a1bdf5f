#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c16; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 300 python tools/path_timing.py 2>&1 | tail -12 | tee $O/path_timing.txt
