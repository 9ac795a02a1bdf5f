#!/bin/bash
# round 4, call 11: the whole GPU suite, the default bench line, kernel statistics of the default command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c11; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -x --tb=short 2>&1 | tail -8 | tee $O/gpu_suite.txt
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err; tail -1 $O/bench_default.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step')}, d['config']['launch'], d['config']['tuning_check_ms'])
print(d['roofline'])
print(json.dumps(d.get('cpu_baseline'))[:900])
"
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt11 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-traffic --no-smi > /dev/null 2>&1)
python - > $O/default_kernel_stats.txt <<'PY'
import csv,glob
f=glob.glob('/tmp/kt11/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print('# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-traffic --no-smi (round 4 default: autotune, timed region, api / channels-last / secondary rows)')
print('# calls   total ms   average us   share   kernel')
for r in rows[:24]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:12.1f} {float(r['Percentage']):6.2f}%  {r['Name'][:160]}")
PY
head -12 $O/default_kernel_stats.txt | cut -c1-200
