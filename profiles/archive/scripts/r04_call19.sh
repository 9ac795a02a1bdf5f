#!/bin/bash
# round 4, call 19: same-lease default bench line (in-run PMC traffic) + rocprofv3 kernel statistics of the tile kernel, looking for a slow part
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c19; mkdir -p $O
timeout 600 python bench.py --no-secondary 2>/dev/null | tail -1 > $O/bench_default.json; python3 -c "
import json
d=json.loads(open('$O/bench_default.json').read())
print('value', d['value'], 'frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'], d['config']['launch'], 'probe', d['part']['tile_store_probe_gbps'], 'traffic', d['roofline']['traffic'])"
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt19 -- $GRAFT_REPO_ROOT/tools/sweep_bench --rounds 6 --launches 3 pipe=1,align=8 default lanes=512,ppl=4 > $GRAFT_REPO_ROOT/$O/sweep_bench_under_rocprof.txt 2>&1)
python - > $O/tile_kernel_stats.txt <<'PY'
import csv,glob
f=glob.glob('/tmp/kt19/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print('# rocprofv3 --kernel-trace --stats -- tools/sweep_bench --rounds 6 --launches 3 pipe=1,align=8 default lanes=512,ppl=4')
print('# (pipe=1,align=8 = round 3 binary\\'s shape: <.., 256, true, 8, false>; default = <.., 256, true, 8, true>; lanes=512 = <.., 512, true, 4, true>)')
print('# calls   total ms   average us   share   kernel')
for r in rows[:12]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:12.1f} {float(r['Percentage']):6.2f}%  {r['Name'][:150]}")
PY
head -9 $O/tile_kernel_stats.txt | cut -c1-220; cat $O/sweep_bench_under_rocprof.txt | grep -v "^#" | cut -c1-120
