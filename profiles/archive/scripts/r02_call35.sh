#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c35; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -12 $O/pytest_gpu.txt
timeout 300 python tools/train_step_timing.py 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids" | tee $O/train_step.txt
