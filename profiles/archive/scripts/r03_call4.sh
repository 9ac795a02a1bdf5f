#!/bin/bash
# round 3, call 4: fused kernel with production spread over all four waves
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c4; mkdir -p $O
timeout 600 python -m pytest tests/test_sweep_conv_gpu.py -q -m gpu -x 2>&1 | tail -5
timeout 300 python tools/sweep_conv_timing.py 2>&1 | tail -2 > $O/sweep_conv_timing.txt; cat $O/sweep_conv_timing.txt
export DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_dbg.so
for a in 0 1 6 14; do
  echo -n "DFM_SC_ABLATE=$a: " >> $O/ablate.txt
  DFM_SC_ABLATE=$a DFM_ITERS=10 timeout 120 python tools/sweep_conv_timing.py 2>&1 | grep "config K" | sed 's/.*fused \([0-9.]* ms\).*/fused \1/' >> $O/ablate.txt
done
cat $O/ablate.txt
