#!/bin/bash
# SQ counters of the general MFMA conv on one neck shape and one hourglass shape (two passes of 8)
cd $GRAFT_REPO_ROOT
O=gpurun_out/c18; mkdir -p $O
timeout 300 python -m pytest tests/test_modules.py tests/test_group_norm.py -m gpu -x -q 2>&1 | tail -3
for CASE in "neck.res1" "hg.conv2" "hg.conv1"; do
  for PASS in 1 2; do
    if [ $PASS = 1 ]; then PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; else PMC="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"; fi
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc18 && timeout 200 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/pmc18 -- python $GRAFT_REPO_ROOT/tools/conv_g_timing.py --case "$CASE" --no-miopen --iters 3 > /tmp/pmc18.log 2>&1)
    python - "$CASE" $PASS <<'PY' >> gpurun_out/c18/conv_g_pmc.txt
import csv,glob,sys,collections
case,ps=sys.argv[1],sys.argv[2]
fs=glob.glob('/tmp/pmc18/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(float); n=collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        if 'conv3d_g_kernel' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
print(f'## {case} pass {ps}: per-launch means over', max(n.values()) if n else 0, 'launches')
for k in sorted(acc): print(f'{k:28s} {acc[k]/n[k]:16.0f}')
PY
  done
done
cat $O/conv_g_pmc.txt
