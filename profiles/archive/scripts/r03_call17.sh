#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c17; mkdir -p $O
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03c17/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['config']['launch'], d['config']['tuning_check_ms'], d['config']['tuning_check_agrees_with_library'], d['part'], d['api_build_dfm_cost']['value'])
PY
tail -3 $O/bench_default.err
