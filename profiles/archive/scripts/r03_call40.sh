#!/bin/bash
# round 3, call 40: PMC counters of the backward bench (matrix-product kernels): HBM traffic (FETCH_SIZE, WRITE_SIZE in
# separate passes) and SQ wait / LDS / MFMA counters
O=gpurun_out/r03c40; mkdir -p $O
R=$PWD
pass() {  # name, counters...
  name=$1; shift
  rm -rf /tmp/pmc_$name
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python $R/bench.py --workload sweep_bwd --steps 2 --warmup 1 > /tmp/pmc_$name.log 2>&1)
  python - "$name" >> $O/sweep_bwd_pmc.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
name = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f'/tmp/pmc_{name}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'sweep_bwd' not in k:
            continue
        tag = 'mfma_cur' if 'mfma_kernel<0>' in k else 'mfma_prev' if 'mfma_kernel<1>' in k else 'tile_prev'
        acc[tag][r['Counter_Name']].append(float(r['Counter_Value']))
for tag in sorted(acc):
    for c in sorted(acc[tag]):
        v = acc[tag][c]
        print(f'{tag:10s} {c:28s} mean per launch {sum(v) / len(v):18.1f}  ({len(v)} launches)')
PY
}
rm -f $O/sweep_bwd_pmc.txt
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
pass sq2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SALU
cat $O/sweep_bwd_pmc.txt
