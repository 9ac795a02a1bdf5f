#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c39; mkdir -p $O
timeout 600 python -m pytest tests/test_group_norm.py tests/test_modules.py tests/test_config_build.py -m gpu -x -q 2>&1 | tail -12
timeout 300 python tools/neck_train_timing.py 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids" | tee $O/neck_train.txt
