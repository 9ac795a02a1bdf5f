#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c22; mkdir -p $O
timeout 300 python -m pytest tests/test_frustum_to_voxel.py tests/test_depth_head.py tests/test_modules.py tests/test_config_build.py -m gpu -x -q 2>&1 | tail -4
timeout 200 python tools/f2v_fused_timing.py 2>&1 | grep -v amdgpu.ids > $O/f2v_fused_timing.txt; cat $O/f2v_fused_timing.txt
timeout 200 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_nstar.json; cat $O/bench_nstar.json
