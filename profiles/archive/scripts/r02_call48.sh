#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c48; mkdir -p $O
export DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_dbg.so
run() {  # ablate planes cw
  echo "== ablate=$1 planes=$2 cw=$3" >> $O/sweep.txt
  (cd /tmp && export TMPDIR=/tmp && DFM_BWD_ABLATE=$1 DFM_BWD_PLANES=$2 DFM_BWD_CW=$3 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt48_$1_$2_$3 -- python $GRAFT_REPO_ROOT/bench.py --workload sweep_bwd --steps 3 --warmup 1 > /dev/null 2>&1)
  python - /tmp/kt48_$1_$2_$3 >> $O/sweep.txt <<'PY'
import csv,glob,sys
f=glob.glob(f'{sys.argv[1]}/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:2]:
    print(f"{r['Calls']:>6} {float(r['AverageNs'])/1e3:10.1f} us  {r['Name'][40:110]}")
PY
}
for a in 0 1 2 4 7; do run $a 28 4; done
run 0 16 4; run 0 20 4; run 0 24 8; run 0 16 8; run 0 24 4; run 0 32 4
cat $O/sweep.txt
