#!/bin/bash
# round 4, call 49: general MFMA conv with two taps per loop trip where the accumulators leave room (no weight-register copies): parity + backbone / neck
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_conv3d_g_gpu.py -x -q -m gpu 2>&1 | tail -2 ) > gpurun_out/r04_c49_tests.txt 2>&1
run() { timeout 300 python bench.py --workload $1 --no-secondary --no-traffic --no-smi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline']['frac'])"; }
( run backbone; run backbone; run neck ) > gpurun_out/r04_c49_bench.txt 2>&1
cat gpurun_out/r04_c49_tests.txt gpurun_out/r04_c49_bench.txt
