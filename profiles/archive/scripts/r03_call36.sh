#!/bin/bash
# round 3, call 36: weight-gradient kernel with batched, unconditional staging loads
O=gpurun_out/r03c36; mkdir -p $O
timeout 900 python -m pytest tests/test_conv3d_g_gpu.py -q -k "weight_gradient" 2>&1 | tail -3 > $O/tests.txt
cat $O/tests.txt
timeout 300 python tools/wgrad_timing.py > $O/wgrad_timing.txt 2>&1
cat $O/wgrad_timing.txt
