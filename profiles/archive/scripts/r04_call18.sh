#!/bin/bash
# round 4, call 18: depth-1 mode of the weight-gradient kernel (2-D convolutions): tests + training step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c18; mkdir -p $O
timeout 900 python -m pytest tests/test_conv3d_g_gpu.py tests/test_conv3d_gpu.py tests/test_modules.py tests/test_depth_fused_training_gpu.py -q -m gpu --tb=short 2>&1 | tail -4 | tee $O/tests.txt
timeout 300 python tools/stereo_train_timing.py --dtype bf16 --fused-only 2>&1 | tail -1 | tee $O/stereo_train_timing.txt
timeout 300 python tools/stereo_train_timing.py --dtype fp32 --fused-only --iters 3 2>&1 | tail -1 | tee -a $O/stereo_train_timing.txt
