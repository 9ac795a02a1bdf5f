#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c7; mkdir -p $O
export DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_dbg.so
for a in 0 16 32 48 8 24; do echo "== DFM_SC_ABLATE=$a" | tee -a $O/trace.txt; DFM_SC_ABLATE=$a timeout 200 python tools/sweep_conv_trace.py 2>&1 | grep -A1 "wave [02]" | grep -v "^--" | tee -a $O/trace.txt; done
