#!/bin/bash
# round 3, call 35: plane-sweep test files + backward bench with the matrix-product backward in place
O=gpurun_out/r03c35; mkdir -p $O
timeout 1500 python -m pytest tests/test_plane_sweep_gpu.py tests/test_sweep_bwd_mfma_gpu.py tests/test_path_parity_gpu.py tests/test_fast_path.py -q -x 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt
timeout 300 python bench.py --workload sweep_bwd --steps 10 --warmup 3 2>&1 | grep '^{' > $O/bench_sweep_bwd.json
cat $O/bench_sweep_bwd.json
timeout 300 python bench.py --workload backbone_train --steps 10 --warmup 3 2>&1 | grep '^{' > $O/bench_backbone_train.json
cut -c1-400 $O/bench_backbone_train.json
