#!/bin/bash
# round 3, call 38: default bench line with the clock / read probes
O=gpurun_out/r03c38; mkdir -p $O
timeout 600 python bench.py 2>&1 | grep '^{' > $O/bench_default.json
python - <<'PY'
import json
j = json.loads(open('gpurun_out/r03c38/bench_default.json').read())
print(j['value'], j['roofline']['frac'], j['roofline']['kernel_ms'], j['part'])
PY
rocm-smi --showclocks 2>/dev/null | head -20 > $O/rocm_smi_clocks.txt; head -12 $O/rocm_smi_clocks.txt
