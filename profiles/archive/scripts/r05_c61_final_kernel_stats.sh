#!/bin/bash
# rocprofv3 --kernel-trace --stats of the final round-5 tree: the default bench command and the aggregation rows
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05c61; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() {  # tag, bench args...
  tag=$1; shift
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag: rocprofv3 --kernel-trace --stats -- python bench.py $*" >> $OUT/summary.txt
  grep '^{' $OUT/$tag.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   bench line: value', d['value'], d['unit'], 'ms_per_step', d['ms_per_step'], 'roofline', {k: d['roofline'].get(k) for k in ('achieved','frac','kernel_ms')})" >> $OUT/summary.txt 2>&1
  python - "$f" >> $OUT/summary.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('   total kernel ms', round(tot/1e6, 3))
print('    calls   total ms    avg us    min us    max us  name')
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:24]:
    print('   %6s %9.3f %9.1f %9.1f %9.1f  %s' % (r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3,
          float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Name'][:120]))
PY
  rm -rf $OUT/$tag
}
run default --no-secondary --no-cpu-baseline --no-traffic
run backbone --workload backbone --steps 10 --warmup 3
run backbone_train --workload backbone_train --steps 5 --warmup 2
run stereo_train --workload stereo_train --steps 5 --warmup 2
cat $OUT/summary.txt
