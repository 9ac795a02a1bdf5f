#!/bin/bash
# round 3, call 52: last full GPU suite + default bench line of the round on the final tree
O=gpurun_out/r03c52; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/gpu_suite.txt
cat $O/gpu_suite.txt
timeout 600 python bench.py 2>&1 | grep '^{' > $O/bench_default.json
python -c "
import json
j=json.loads(open('$O/bench_default.json').read()); print(j['value'], j['roofline']['frac'], j['part']['tile_store_probe_gbps'], j['part']['shader_clock_ghz_under_fma_load'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
