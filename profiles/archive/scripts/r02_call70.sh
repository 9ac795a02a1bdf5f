#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c70; mkdir -p $O
timeout 900 python -m pytest tests/test_conv3d_g_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
timeout 300 python tools/conv_g_timing.py --no-miopen 2>&1 | grep MFMA | tee $O/conv_g_timing.txt
timeout 300 python tools/neck2d_timing.py 2>&1 | grep -v amdgpu.ids | tee $O/neck2d_timing.txt
timeout 200 python bench.py --workload backbone 2>/dev/null | cut -c1-140
