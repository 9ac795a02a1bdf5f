#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c14; mkdir -p $O
timeout 600 python -m pytest tests/test_conv3d_gpu.py tests/test_group_norm.py tests/test_modules.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 200 python bench.py --workload group_norm --steps 10 --warmup 3 > $O/bench_gn.json 2>/dev/null; cat $O/bench_gn.json
export DFM_ONLY=bf16 DFM_MIOPEN_FIND=1 DFM_ITERS=20
timeout 900 python tools/backbone_timing.py 2>&1 | grep -v MIOpen > $O/backbone.txt; cat $O/backbone.txt
timeout 200 python tools/gn_timing.py > $O/gn_timing.txt 2>&1; cat $O/gn_timing.txt
