#!/bin/bash
# round 4, call 44: kernel times of the strided fp32 backward (cur: window kernel, prev: tile kernel), tests
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_sweep_walk_gpu.py -x -q -m gpu 2>&1 | grep -v Warning | tail -5 ) > gpurun_out/r04_c44_tests.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof44 && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof44 --output-format csv -- python /root/repo/bench.py --workload sweep_bwd_kitti --no-secondary --no-traffic --no-smi > /dev/null 2>&1
cd /root/repo
python - <<'PY' > gpurun_out/r04_c44_kernel_stats.txt
import csv,glob
for f in glob.glob('/tmp/prof44/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:8]:
        print(r['Name'][:90], r['Calls'], r['TotalDurationNs'], 'avg_us', float(r['AverageNs'])/1e3, r['Percentage'])
PY
cat gpurun_out/r04_c44_tests.txt gpurun_out/r04_c44_kernel_stats.txt
