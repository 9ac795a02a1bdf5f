#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c9; mkdir -p $O
timeout 600 python -m pytest tests/test_plane_sweep_gpu.py -x -q -k "backward" 2>&1 | tail -3
export DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_dbg.so
for a in 0 7; do
  echo "== DFM_BWD_ABLATE=$a" >> $O/ablate.txt
  (cd /tmp && DFM_BWD_ABLATE=$a timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt9_$a -- python $GRAFT_REPO_ROOT/bench.py --workload sweep_bwd --steps 3 --warmup 1 > /dev/null 2>&1)
  python - $a >> $O/ablate.txt <<'PY'
import csv,glob,sys
f=glob.glob(f'/tmp/kt9_{sys.argv[1]}/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:2]:
    print(f"{r['Calls']:>6} {float(r['AverageNs'])/1e3:10.1f} us  {r['Name'][40:110]}")
PY
done
cat $O/ablate.txt
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d /tmp/pmc9 -- python $GRAFT_REPO_ROOT/bench.py --workload sweep_bwd --steps 2 --warmup 1 > /dev/null 2>&1)
python tools/pmc_summary.py /tmp/pmc9 --kernel sweep_bwd > $O/pmc_bwd.txt 2>&1; cat $O/pmc_bwd.txt
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc9b -- python $GRAFT_REPO_ROOT/bench.py --workload sweep_bwd --steps 2 --warmup 1 > /dev/null 2>&1)
python tools/pmc_summary.py /tmp/pmc9b --kernel sweep_bwd >> $O/pmc_bwd.txt 2>&1; tail -25 $O/pmc_bwd.txt
