#!/bin/bash
# round 3, call 23: matrix-product backward -- parity tests and the per-phase cycle trace
O=gpurun_out/r03c23; mkdir -p $O
timeout 900 python -m pytest tests/test_sweep_bwd_mfma_gpu.py -q 2>&1 | grep -v "^$" | grep -n "^E  \|passed\|failed\|^FAILED" | cut -c1-300 > $O/tests.txt
head -30 $O/tests.txt
DFM_HIP_LIB=$PWD/depth-from-motion_amd/lib/libdfm_hip_dbg.so timeout 300 python tools/sweep_bwd_trace.py > $O/trace.txt 2>&1
cat $O/trace.txt
