#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py rows, summarised per kernel (calls / total / avg / min / max).
# usage (GPU box): tools/kernel_stats.sh OUTDIR tag1:"bench args" tag2:"bench args" ...   -> OUTDIR/summary.txt
OUT=$1; shift; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for spec in "$@"; do
  tag=${spec%%:*}; args=${spec#*:}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python $GRAFT_REPO_ROOT/bench.py $args > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag: rocprofv3 --kernel-trace --stats -- python bench.py $args" >> $OUT/summary.txt
  grep '^{' $OUT/$tag.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   bench line: value', d['value'], d['unit'], 'ms_per_step', d['ms_per_step'], 'roofline', {k: d['roofline'].get(k) for k in ('achieved','frac','kernel_ms')})" >> $OUT/summary.txt 2>&1
  python - "$f" >> $OUT/summary.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('   total kernel ms', round(tot/1e6, 3))
print('    calls   total ms    avg us    min us    max us     pct  name')
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:int(__import__("os").environ.get("ROWS", "28"))]:
    print('   %6s %9.3f %9.1f %9.1f %9.1f  %5.1f  %s' % (r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3,
          float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, 100*float(r['TotalDurationNs'])/tot, r['Name'][:120]))
PY
  rm -rf $OUT/$tag
done
cd $GRAFT_REPO_ROOT
