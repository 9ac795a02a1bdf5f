#!/bin/bash
# round 3, call 2: fused sweep+dres0 kernel: parity tests, timing vs the unfused sequence, fast-path tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c2; mkdir -p $O
timeout 600 python -m pytest tests/test_sweep_conv_gpu.py -q -m gpu -x 2>&1 | tail -40 > $O/sweep_conv_tests.txt; tail -30 $O/sweep_conv_tests.txt
timeout 300 python tools/sweep_conv_timing.py > $O/sweep_conv_timing.txt 2>&1; tail -5 $O/sweep_conv_timing.txt
for dc in 8 12 15 18 24; do DFM_DEPTH_CHUNK=$dc timeout 120 python tools/sweep_conv_timing.py 2>&1 | grep "config K" | sed "s/^/dchunk $dc: /" >> $O/sweep_conv_timing.txt; done; tail -5 $O/sweep_conv_timing.txt
timeout 600 python -m pytest tests/test_fast_path.py -q -m gpu -x 2>&1 | tail -30 > $O/fast_path_tests.txt; tail -20 $O/fast_path_tests.txt
timeout 200 python bench.py --workload backbone > $O/bench_backbone.json 2>$O/bench_backbone.err; tail -1 $O/bench_backbone.json | cut -c1-400
DFM_NO_SWEEP_FUSION=1 timeout 200 python bench.py --workload backbone > $O/bench_backbone_unfused.json 2>/dev/null; tail -1 $O/bench_backbone_unfused.json | cut -c1-400
