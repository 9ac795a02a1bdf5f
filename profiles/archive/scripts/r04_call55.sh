#!/bin/bash
# round 4, call 55 (final): the whole GPU suite (serial, as the driver runs it), the default bench line (traffic + secondary
# rows measured in the run), rocprofv3 kernel statistics of the default command, smoke
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r04_c55_gpu_suite.txt
timeout 600 python bench.py > gpurun_out/r04_c55_bench.json 2> gpurun_out/r04_c55_bench.err
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_c55_smoke.txt 2>&1
( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p55; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p55 --output-format csv -- python /root/repo/bench.py --steps 10 --warmup 3 --no-secondary --no-traffic --no-smi > /dev/null 2>&1
  python - <<'PY'
import csv,glob
print('# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-secondary --no-traffic --no-smi')
print('# calls   total ms   average us   share   kernel')
for f in glob.glob('/tmp/p55/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print('%6s %10.3f %12.1f %7s%%  %s' % (r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage'], r['Name'][:150]))
PY
) > gpurun_out/r04_c55_default_kernel_stats.txt 2>&1
cat gpurun_out/r04_c55_gpu_suite.txt gpurun_out/r04_c55_smoke.txt; head -8 gpurun_out/r04_c55_default_kernel_stats.txt | cut -c1-160
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_c55_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline'].get('traffic'), d['config']['launch'])
print({k:(v.get('ms_per_step'),v.get('frac'),v.get('skipped')) for k,v in (d.get('secondary') or {}).items() if isinstance(v,dict)}, d['secondary'].get('wall_s'))
PY
