#!/bin/bash
# round 3, call 48: full GPU suite, default bench line, rocprofv3 kernel statistics of the default bench
# command and of the backward bench
O=gpurun_out/r03c48; mkdir -p $O
R=$PWD
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $O/gpu_suite.txt
cat $O/gpu_suite.txt
timeout 600 python bench.py 2>&1 | grep '^{' > $O/bench_default.json
cat $O/bench_default.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
stats() {  # name, bench args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python $R/bench.py "$@" > /tmp/bench_$name.txt 2>&1)
  grep '^{' /tmp/bench_$name.txt > $O/${name}_bench_line_under_profiler.json
  python - "$name" "$*" > $O/${name}_kernel_stats.txt <<'PY'
import csv, glob, sys
name, args = sys.argv[1], sys.argv[2]
f = glob.glob(f'/tmp/prof_{name}/**/*kernel_stats.csv', recursive=True)
print(f'# rocprofv3 --kernel-trace --stats -- python bench.py {args}')
print('# calls   total ms   average us   share   kernel')
for r in csv.DictReader(open(f[0])):
    print(f"{int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:11.1f} {float(r['Percentage']):6.2f}%  {r['Name'][:150]}")
PY
  head -8 $O/${name}_kernel_stats.txt | cut -c1-200
}
stats default
stats sweep_bwd --workload sweep_bwd --steps 10 --warmup 3
