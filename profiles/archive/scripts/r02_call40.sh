#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c40; mkdir -p $O
timeout 600 python -m pytest tests/test_conv3d_g_gpu.py tests/test_conv3d_gpu.py tests/test_modules.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/neck_train_timing.py 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids" | tee $O/neck_train.txt
timeout 300 python tools/train_step_timing.py 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids" | tee $O/train_step.txt
