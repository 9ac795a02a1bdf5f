#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c24; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -12 $O/pytest_gpu.txt
