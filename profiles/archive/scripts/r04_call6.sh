#!/bin/bash
# round 4, call 6: how the volume's stores leave the CU (nt / plain / sc1 / sc0 sc1 / sc1 nt builds of the tile kernel),
# the issue cost of the blend loop's VALU instructions, the reworked default bench line and the backward tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c6; mkdir -p $O
timeout 200 tools/sweep_bench --rounds 7 --launches 3 default lanes=512,ppl=4 pipe=1 2>&1 | sed 's/^/nt      /' | tee $O/store_flavours.txt
for f in 1 2 3 4; do
  mkdir -p /tmp/sf$f && cp depth-from-motion_amd/lib/libdfm_hip_sf$f.so /tmp/sf$f/libdfm_hip.so
  LD_LIBRARY_PATH=/tmp/sf$f timeout 200 tools/sweep_bench --rounds 7 --launches 3 default lanes=512,ppl=4 pipe=1 2>&1 | grep -v "^#" | sed "s/^/flavour$f /" | tee -a $O/store_flavours.txt
done
timeout 120 tools/valu_microbench 2>&1 | tee $O/valu_microbench.txt | grep "waves/SIMD 2"
timeout 600 python -m pytest tests/test_sweep_bwd_mfma_gpu.py tests/test_data_geometry.py -q -m gpu -x -s 2>&1 | grep -v "^$" | tail -4 | tee $O/tests.txt
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err; tail -1 $O/bench_default.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step')}, d['config']['launch'], d['config']['tuning_check_ms'])
print(d['roofline'])
print(json.dumps(d.get('secondary'))[:1800])
"
