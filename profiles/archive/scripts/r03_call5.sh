#!/bin/bash
# round 3, call 5: SQ counters of the fused sweep + dres0 kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c5; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && DFM_ITERS=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt5 -- python $GRAFT_REPO_ROOT/tools/sweep_conv_timing.py > /dev/null 2>&1)
python tools/rocpd_stats.py /tmp/kt5 2>/dev/null | head -12 > $O/kernel_stats.txt || python - >> $O/kernel_stats.txt <<'PY'
import csv,glob
f=glob.glob('/tmp/kt5/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(f"{r['Calls']:>6} {float(r['AverageNs'])/1e3:10.1f} us  {r['Name'][:110]}")
PY
cat $O/kernel_stats.txt
(cd /tmp && DFM_ITERS=2 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d /tmp/pmc5 -- python $GRAFT_REPO_ROOT/tools/sweep_conv_timing.py > /dev/null 2>&1)
python tools/pmc_summary.py /tmp/pmc5 --kernel sweep_conv_kernel > $O/pmc.txt 2>&1
(cd /tmp && DFM_ITERS=2 timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d /tmp/pmc5b -- python $GRAFT_REPO_ROOT/tools/sweep_conv_timing.py > /dev/null 2>&1)
python tools/pmc_summary.py /tmp/pmc5b --kernel sweep_conv_kernel >> $O/pmc.txt 2>&1
(cd /tmp && DFM_ITERS=2 timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_I8 --output-format csv -d /tmp/pmc5c -- python $GRAFT_REPO_ROOT/tools/sweep_conv_timing.py > /dev/null 2>&1)
python tools/pmc_summary.py /tmp/pmc5c --kernel sweep_conv_kernel >> $O/pmc.txt 2>&1
cat $O/pmc.txt
