#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05c39; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for w in 0 1; do
DFM_WGRAD_WALK=$w timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/w$w -- python $GRAFT_REPO_ROOT/tools/wgrad_timing.py > $OUT/w$w.log 2>&1
f=$(find $OUT/w$w -name "*kernel_stats.csv" | head -1)
echo "== walk=$w" >> $OUT/summary.txt
python - "$f" >> $OUT/summary.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:8]:
    print('%6s %9.3f ms %8.1f us  min %8.1f max %8.1f  %s' % (r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Name'][:90]))
PY
done
cat $OUT/summary.txt
