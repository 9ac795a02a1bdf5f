#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c6; mkdir -p $O
timeout 600 python -m pytest tests/test_sweep_conv_gpu.py -q -m gpu -x 2>&1 | tail -5
timeout 300 python tools/sweep_conv_timing.py 2>&1 | tail -2 | tee $O/sweep_conv_timing.txt
for dc in 9 12 18; do DFM_DEPTH_CHUNK=$dc timeout 120 python tools/sweep_conv_timing.py 2>&1 | grep "config K" | sed "s/^/dchunk $dc: /" | tee -a $O/sweep_conv_timing.txt; done
