#!/bin/bash
# round 3, call 53: which part of the N* tile kernel is slow on a slow part?  debug-build ablations (1 no staging,
# 2 no stores, 4 no blend) next to the part probes of the same lease
O=gpurun_out/r03c53; mkdir -p $O
R=$PWD
D=$R/depth-from-motion_amd/lib/libdfm_hip_dbg.so
for ab in 0 1 2 4; do
  DFM_HIP_LIB=$D DFM_ABLATE=$ab DFM_AUTOTUNE=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-autotune 2>/dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ablate $ab: kernel', j['roofline']['kernel_ms'], 'ms  frac', j['roofline']['frac'], ' store probe', j['part']['tile_store_probe_gbps'], 'clock', j['part']['shader_clock_ghz_under_fma_load'])" >> $O/nstar_ablation.txt
done
cat $O/nstar_ablation.txt
