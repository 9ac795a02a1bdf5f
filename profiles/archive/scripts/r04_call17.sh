#!/bin/bash
# round 4, call 17: is the fused-vs-materialised training test stable?  (6 repetitions) + whole suite + kernel stats / bench on this part
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c17; mkdir -p $O
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_depth_fused_training_gpu.py -q -m gpu -k fused_vs_materialised --tb=short 2>&1 | grep -E "passed|failed|assert|Error" | head -5; done | tee $O/flaky.txt
timeout 1800 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -4 | tee $O/gpu_suite.txt
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_default.json; python3 -c "
import json
d=json.loads(open('$O/bench_default.json').read())
print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['config']['launch'], d['part']['tile_store_probe_gbps'], d['roofline']['traffic'])"
