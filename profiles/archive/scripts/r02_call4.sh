#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c4; mkdir -p $O
timeout 600 python -m pytest tests/test_conv3d_gpu.py -x -q > $O/pytest_conv.txt 2>&1; tail -15 $O/pytest_conv.txt
timeout 300 python tools/conv_timing.py > $O/conv_timing.txt 2>&1; cat $O/conv_timing.txt
timeout 200 tools/sweep_bench --rounds 5 --launches 3 default lanes=512,ppl=4 > $O/ab.txt 2>&1; cat $O/ab.txt
