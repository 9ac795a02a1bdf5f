#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c46; mkdir -p $O
timeout 600 python -m pytest tests/test_conv3d_g_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/conv_g_timing.py --no-miopen 2>&1 | tee $O/conv_g_timing.txt | tail -30
