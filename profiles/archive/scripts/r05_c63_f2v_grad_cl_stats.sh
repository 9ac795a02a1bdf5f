#!/bin/bash
# kernel-level A/B of the channels-last stereo gradient (rocprofv3 --kernel-trace --stats, 7 training steps each)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05c63; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for mode in planar native; do
  if [ $mode = planar ]; then export DFM_F2V_PLANAR_GRAD=1; else unset DFM_F2V_PLANAR_GRAD; fi
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$mode -- python $GRAFT_REPO_ROOT/bench.py --workload stereo_train --steps 5 --warmup 2 > $OUT/$mode.log 2>&1
  f=$(find $OUT/$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode" >> $OUT/summary.txt
  python - "$f" >> $OUT/summary.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('   total kernel ms (7 steps)', round(tot/1e6, 3), ' per step', round(tot/7e6, 3))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs'])):
    n=r['Name']
    if any(k in n for k in ('f2v_bwd', 'CUDAFunctor_add', 'direct_copy', 'FillFunctor', 'bfloat16_copy', 'float32_copy')):
        print('   %6s %9.3f ms %9.1f us  %s' % (r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, n[:130]))
PY
  rm -rf $OUT/$mode
done
cat $OUT/summary.txt
