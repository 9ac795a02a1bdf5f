#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c62; mkdir -p $O
timeout 300 python tools/neck2d_timing.py 2>&1 | grep -v amdgpu.ids | tee $O/neck2d_timing.txt
