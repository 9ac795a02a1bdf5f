#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c41; mkdir -p $O
echo "## slice ring (current)" > $O/wgrad_timing.txt
timeout 200 python tools/wgrad_timing.py 2>&1 | grep -v amdgpu.ids >> $O/wgrad_timing.txt
echo "## strided tiles, three slices staged per tile (previous build)" >> $O/wgrad_timing.txt
DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_prev.so timeout 200 python tools/wgrad_timing.py 2>&1 | grep -v amdgpu.ids >> $O/wgrad_timing.txt
cat $O/wgrad_timing.txt
