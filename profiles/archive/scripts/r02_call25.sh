#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c25; mkdir -p $O
timeout 600 python tools/path_timing.py both --iters 5 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids" > $O/path_timing.txt; cat $O/path_timing.txt
