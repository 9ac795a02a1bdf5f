#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c58; mkdir -p $O
timeout 300 python tools/conv_timing.py --chunks 0,3,4,5,6,7,8,9,10,12,18 2>&1 | grep -v amdgpu.ids | tee $O/conv_timing.txt
