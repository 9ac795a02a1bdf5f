#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05c33; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for mode in plain fused; do
  if [ $mode = plain ]; then export DFM_NO_SKIP_GRAD_FUSION=1; else unset DFM_NO_SKIP_GRAD_FUSION; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$mode -- python $GRAFT_REPO_ROOT/bench.py --workload backbone_train --steps 5 --warmup 2 > $OUT/$mode.log 2>&1
  f=$(find $OUT/$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode" >> $OUT/summary.txt
  python - "$f" >> $OUT/summary.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:22]:
    print('%6s %9.3f ms %8.1f us  %s' % (r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Name'][:110]))
PY
done
cat $OUT/summary.txt
