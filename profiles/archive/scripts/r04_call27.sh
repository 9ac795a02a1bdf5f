#!/bin/bash
# round 4, call 27: parity of the final matrix-core unpack build (pinned order), N* shipped shapes, default bench line
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_plane_sweep_gpu.py -x -q -m gpu -k "pipe" 2>&1 | tail -3
  timeout 600 python -m pytest tests/test_nstar_shipped_gpu.py -x -q -m gpu 2>&1 | tail -3 ) > gpurun_out/r04_c27_tests.txt 2>&1
timeout 600 python bench.py > gpurun_out/r04_c27_bench.json 2> gpurun_out/r04_c27_bench.err
cat gpurun_out/r04_c27_tests.txt; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_c27_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['config']['launch'], d['config'].get('tuning_check_ms'))
print({k:(v.get('ms_per_step'),v.get('frac')) for k,v in (d.get('secondary') or {}).items() if isinstance(v,dict)})
PY
