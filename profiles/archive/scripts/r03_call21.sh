#!/bin/bash
# round 3, call 21: matrix-product backward -- parity tests, then kernel statistics of the backward bench
mkdir -p gpurun_out/r03c21
timeout 900 python -m pytest tests/test_sweep_bwd_mfma_gpu.py -q 2>&1 | grep -v "^$" | grep -n "^E  \|passed\|failed\|^FAILED" > gpurun_out/r03c21/tests.txt
head -60 gpurun_out/r03c21/tests.txt; ls -R /tmp/prof 2>/dev/null | head
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --workload sweep_bwd --steps 5 --warmup 2 > $R/gpurun_out/r03c21/bench_bwd.txt 2>&1
cd $R
grep '^{' gpurun_out/r03c21/bench_bwd.txt
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r03c21/bwd_kernel_stats.csv
head -8 gpurun_out/r03c21/bwd_kernel_stats.csv | cut -c1-200
