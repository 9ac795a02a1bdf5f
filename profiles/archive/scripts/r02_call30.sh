#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c30; mkdir -p $O
timeout 600 python -m pytest tests/test_modules.py tests/test_config_build.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python tools/path_timing.py stereo --iters 5 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids" > $O/path_timing_stereo.txt; cat $O/path_timing_stereo.txt
