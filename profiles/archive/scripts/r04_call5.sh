#!/bin/bash
# round 4, call 5: the reworked default bench line (library autotuner 4 x 2 launches, in-run PMC traffic, SMI under load,
# secondary rows), the backward tests at the tightened bar
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c5; mkdir -p $O
timeout 600 python -m pytest tests/test_sweep_bwd_mfma_gpu.py -q -m gpu -x -s 2>&1 | grep -v "^$" | tail -5 | tee $O/tests.txt
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err; tail -1 $O/bench_default.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step')}, d['config']['launch'], d['config']['tuning_check_ms'])
print(d['roofline'])
print(json.dumps(d['part'])[:1500])
print(json.dumps(d.get('secondary'))[:2500])
"
