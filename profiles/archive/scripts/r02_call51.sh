#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c51; mkdir -p $O
export DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_dbg.so
run() {  # planes rows cw
  echo "== planes=$1 rows=$2 cw=$3" >> $O/sweep.txt
  (cd /tmp && export TMPDIR=/tmp && DFM_BWD_PLANES=$1 DFM_BWD_ROWS=$2 DFM_BWD_CW=$3 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt51_$1_$2_$3 -- python $GRAFT_REPO_ROOT/bench.py --workload sweep_bwd --steps 3 --warmup 1 > /dev/null 2>&1)
  python - /tmp/kt51_$1_$2_$3 >> $O/sweep.txt <<'PY'
import csv,glob,sys
f=glob.glob(f'{sys.argv[1]}/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:2]:
    print(f"{r['Calls']:>6} {float(r['AverageNs'])/1e3:10.1f} us  {r['Name'][40:110]}")
PY
}
export DFM_BWD_ROWCAP=16; run 12 16 4; run 16 16 4; run 20 16 4; run 24 16 4; run 8 16 4; run 16 16 2; run 24 16 2; run 10 16 8; run 16 16 8
cat $O/sweep.txt
