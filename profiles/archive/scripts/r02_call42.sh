#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c42; mkdir -p $O
for PASS in 1 2; do
  if [ $PASS = 1 ]; then PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; else PMC="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"; fi
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc42 && timeout 200 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/pmc42 -- python $GRAFT_REPO_ROOT/tools/wgrad_timing.py > /tmp/pmc42.log 2>&1)
  python - $PASS <<'PY' >> gpurun_out/c42/wgrad_pmc.txt
import csv,glob,sys,collections
ps=sys.argv[1]
fs=glob.glob('/tmp/pmc42/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(float); n=collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        if 'conv3d_wgrad_kernel' not in r['Kernel_Name']: continue
        if r['Grid_Size'] if 'Grid_Size' in r else False: pass
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
print(f'## conv3d_wgrad_kernel, all shapes of tools/wgrad_timing.py, pass {ps}: sums over', max(n.values()) if n else 0, 'launches')
for k in sorted(acc): print(f'{k:28s} {acc[k]:18.0f}')
PY
done
cat $O/wgrad_pmc.txt
