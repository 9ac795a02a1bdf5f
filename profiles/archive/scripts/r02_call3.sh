#!/bin/bash
# GPU call 3: pack pre-pass variants; bench.py with the reworked autotuner
cd $GRAFT_REPO_ROOT
O=gpurun_out/c3; mkdir -p $O
timeout 120 tools/pack_microbench > $O/pack_microbench.txt 2>&1; cat $O/pack_microbench.txt
timeout 400 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json
timeout 300 python -m pytest tests/test_nstar_shipped_gpu.py tests/test_plane_sweep_gpu.py -q -x -k "shipped or autotune or augmented" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
