#!/bin/bash
# round 4, call 52: the whole GPU suite (serial, as the driver runs it), default bench line, smoke
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r04_c52_gpu_suite.txt
timeout 600 python bench.py > gpurun_out/r04_c52_bench.json 2> gpurun_out/r04_c52_bench.err
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_c52_smoke.txt 2>&1
cat gpurun_out/r04_c52_gpu_suite.txt gpurun_out/r04_c52_smoke.txt; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_c52_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline'].get('traffic'), d['config']['launch'])
print({k:(v.get('ms_per_step'),v.get('frac')) for k,v in (d.get('secondary') or {}).items() if isinstance(v,dict)})
PY
