#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c55; mkdir -p $O
DFM_PATH_CL2D=1 timeout 300 python tools/path_timing.py stereo --iters 5 2>&1 | tail -7 | tee $O/path_timing_stereo_cl2d.txt
