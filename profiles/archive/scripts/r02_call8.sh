#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c8; mkdir -p $O
export DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_dbg.so
for a in 0 1 2 4 3 6 7; do
  echo "== DFM_BWD_ABLATE=$a" >> $O/ablate.txt
  (cd /tmp && DFM_BWD_ABLATE=$a timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt8_$a -- python $GRAFT_REPO_ROOT/bench.py --workload sweep_bwd --steps 3 --warmup 1 > /dev/null 2>&1)
  python - $a >> $O/ablate.txt <<'PY'
import csv,glob,sys
f=glob.glob(f'/tmp/kt8_{sys.argv[1]}/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:2]:
    print(f"{r['Calls']:>6} {float(r['AverageNs'])/1e3:10.1f} us  {r['Name'][40:110]}")
PY
done
cat $O/ablate.txt
