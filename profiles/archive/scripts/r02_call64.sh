#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c64; mkdir -p $O
timeout 900 python -m pytest tests/test_conv3d_g_gpu.py tests/test_modules.py -x -q -m gpu 2>&1 | tail -8 | tee $O/tests.txt
timeout 300 python tools/neck2d_timing.py 2>&1 | grep -v amdgpu.ids | tee $O/neck2d_timing.txt
timeout 300 python tools/path_timing.py stereo --iters 10 2>&1 | tail -8 | tee $O/path_timing_stereo.txt
