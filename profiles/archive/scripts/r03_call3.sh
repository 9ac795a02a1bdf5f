#!/bin/bash
# round 3, call 3: ablations of the fused sweep + dres0 kernel (debug build) + kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c3; mkdir -p $O
export DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_dbg.so
for a in 0 1 2 4 6 7 8 14 15; do
  echo -n "DFM_SC_ABLATE=$a: " >> $O/ablate.txt
  DFM_SC_ABLATE=$a DFM_ITERS=10 timeout 120 python tools/sweep_conv_timing.py 2>&1 | grep "config K" | sed 's/.*fused \([0-9.]* ms\).*/fused \1/' >> $O/ablate.txt
done
cat $O/ablate.txt
unset DFM_HIP_LIB
timeout 600 python -m pytest tests/test_fast_path.py tests/test_modules.py tests/test_path_parity_gpu.py -q -m gpu -x 2>&1 | tail -5
