#!/bin/bash
# round 4, call 12: fp32 models on the MFMA kernels in split precision: tests, fp32 DfMStereoPath training step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c12; mkdir -p $O
timeout 900 python -m pytest tests/test_conv3d_g_gpu.py tests/test_modules.py tests/test_conv3d_gpu.py tests/test_fast_path.py tests/test_path_parity_gpu.py tests/test_depth_fused_training_gpu.py -q -m gpu -x --tb=short 2>&1 | grep -v "Warning\|warn\|forward_call" | tail -30 | tee $O/tests.txt
timeout 600 python tools/stereo_train_timing.py --dtype fp32 --fused-only --iters 3 2>&1 | tail -2 | tee $O/stereo_train_timing_fp32.txt
