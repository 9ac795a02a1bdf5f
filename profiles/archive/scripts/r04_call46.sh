#!/bin/bash
# round 4, call 46: what the dense pixel-major scatter costs: 4 vs 1 dword atomics per lane and tap; HBM counters of the kernel
cd /root/repo; mkdir -p gpurun_out
L=$PWD/depth-from-motion_amd/lib
run() { timeout 300 python bench.py --workload sweep_bwd_kitti --no-secondary --no-traffic --no-smi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sweep_bwd_kitti', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('bwd_kernel'))"; }
( echo "## 4 dwords"; run; echo "## 1 dword"; DFM_HIP_LIB=$L/libdfm_hip_nch1.so run
  cd /tmp; export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pmc46; timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc46 -- python /root/repo/bench.py --workload sweep_bwd_kitti --steps 3 --warmup 1 --no-secondary --no-traffic --no-smi > /dev/null 2>&1
    python - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob('/tmp/pmc46/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']=='$c': acc[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'sweep' in k: print('$c', k, 'avg KB', sum(v)/len(v), 'n', len(v))
PY
  done ) > gpurun_out/r04_c46_scatter_cost.txt 2>&1
cat gpurun_out/r04_c46_scatter_cost.txt
