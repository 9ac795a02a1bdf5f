#!/bin/bash
# SQ counters of the general MFMA conv on its small-plan case (hourglass conv1: 32->64 stride 2, pfw = 1)
# next to a large-plan case (neck.res1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/c72; mkdir -p $O
for CASE in "hg.conv1" "neck.res1"; do
for PASS in 1 2; do
  if [ $PASS = 1 ]; then PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; else PMC="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"; fi
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc72 && timeout 200 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/pmc72 -- python $GRAFT_REPO_ROOT/tools/conv_g_timing.py --no-miopen --case "$CASE" --iters 4 > /tmp/pmc72.log 2>&1)
  python - "$CASE" $PASS <<'PY' >> gpurun_out/c72/conv_g_pmc.txt
import csv,glob,sys,collections
case,ps=sys.argv[1],sys.argv[2]
fs=glob.glob('/tmp/pmc72/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(float); n=collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        if 'conv3d_g_kernel' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
print(f'## conv3d_g_kernel, case {case}, pass {ps}: sums over', max(n.values()) if n else 0, 'launches')
for k in sorted(acc): print(f'{k:28s} {acc[k]:18.0f}')
PY
done
done
cat $O/conv_g_pmc.txt
