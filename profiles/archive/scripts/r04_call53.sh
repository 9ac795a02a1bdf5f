#!/bin/bash
# round 4, call 53: the cur-window backward's 36 end-of-run atomics per lane with and without the sc1 (device scope) bit: kernel time, HBM writes
cd /root/repo; mkdir -p gpurun_out
L=$PWD/depth-from-motion_amd/lib
( for v in "" sc1; do
  echo "## ${v:-release (no sc1: csf 4, exclusive windows)}"
  export DFM_HIP_LIB=$L/libdfm_hip${v:+_$v}.so
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p53; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p53 --output-format csv -- python /root/repo/bench.py --workload sweep_bwd_kitti --no-secondary --no-traffic --no-smi > /dev/null 2>&1
  python - <<PY
import csv,glob
for f in glob.glob('/tmp/p53/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:3]: print(r['Name'][:70], 'avg_us', round(float(r['AverageNs'])/1e3,1))
PY
  rm -rf /tmp/p53; timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p53 -- python /root/repo/bench.py --workload sweep_bwd_kitti --steps 3 --warmup 1 --no-secondary --no-traffic --no-smi > /dev/null 2>&1
  python - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob('/tmp/p53/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']=='WRITE_SIZE': acc[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'sweep' in k: print('WRITE_SIZE', k, 'avg MB', round(sum(v)/len(v)/1024,1))
PY
  cd /root/repo
done ) > gpurun_out/r04_c53_sc1.txt 2>&1
cat gpurun_out/r04_c53_sc1.txt
