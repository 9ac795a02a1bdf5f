#!/bin/bash
# round 4, call 57: kernel times with the column-window slab
cd /root/repo; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof44 && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof44 --output-format csv -- python /root/repo/bench.py --workload sweep_bwd_kitti --no-secondary --no-traffic --no-smi > /dev/null 2>&1
cd /root/repo
python - <<'PY' > gpurun_out/r04_c57_kernel_stats.txt
import csv,glob
for f in glob.glob('/tmp/prof44/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:8]:
        print(r['Name'][:90], r['Calls'], r['TotalDurationNs'], 'avg_us', float(r['AverageNs'])/1e3, r['Percentage'])
PY
cat gpurun_out/r04_c57_tests.txt gpurun_out/r04_c57_kernel_stats.txt
