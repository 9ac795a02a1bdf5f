#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c8; mkdir -p $O
timeout 600 python -m pytest tests/test_sweep_conv_gpu.py -q -m gpu -x 2>&1 | tail -3
timeout 300 python tools/sweep_conv_timing.py 2>&1 | tail -2 | tee $O/sweep_conv_timing.txt
for dc in 9 12 18 24 36; do DFM_DEPTH_CHUNK=$dc timeout 120 python tools/sweep_conv_timing.py 2>&1 | grep "config K" | sed "s/^/dchunk $dc: /" | tee -a $O/sweep_conv_timing.txt; done
export DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_dbg.so
timeout 200 python tools/sweep_conv_trace.py 2>&1 | grep "wave" | tee $O/trace.txt
