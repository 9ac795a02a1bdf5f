#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c59; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/gpu_tests.txt
timeout 300 python tools/path_timing.py stereo --iters 5 2>&1 | tail -6 | tee $O/path_timing_stereo.txt
timeout 200 python bench.py --workload backbone 2>/dev/null | cut -c1-220
