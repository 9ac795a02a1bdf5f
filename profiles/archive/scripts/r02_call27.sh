#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c27; mkdir -p $O
timeout 600 python -m pytest tests/test_modules.py tests/test_config_build.py tests/test_point_sample_gpu.py tests/test_frustum_to_voxel.py tests/test_plane_sweep_gpu.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python tools/path_timing.py both --iters 5 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids" > $O/path_timing.txt; cat $O/path_timing.txt
