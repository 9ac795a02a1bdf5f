#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c21; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
timeout 200 python tools/f2v_fused_timing.py 2>&1 | grep -v amdgpu.ids > $O/f2v_fused_timing.txt; cat $O/f2v_fused_timing.txt
