#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c13; mkdir -p $O
timeout 600 python -m pytest tests/test_conv3d_gpu.py tests/test_group_norm.py tests/test_modules.py -x -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
timeout 300 python tools/conv_timing.py --chunks 0 > $O/conv_timing.txt 2>&1; cat $O/conv_timing.txt
export DFM_ONLY=bf16 DFM_MIOPEN_FIND=1 DFM_ITERS=20
timeout 900 python tools/backbone_timing.py 2>&1 | grep -v MIOpen > $O/backbone.txt; cat $O/backbone.txt
