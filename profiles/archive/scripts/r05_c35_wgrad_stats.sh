#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05c35; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/new -- python $GRAFT_REPO_ROOT/bench.py --workload backbone_train --steps 5 --warmup 2 > $OUT/new.log 2>&1
f=$(find $OUT/new -name "*kernel_stats.csv" | head -1)
python - "$f" > $OUT/summary.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    print('%6s %9.3f ms %8.1f us  %s' % (r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Name'][:100]))
PY
cat $OUT/summary.txt
