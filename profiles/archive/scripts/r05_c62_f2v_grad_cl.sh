#!/bin/bash
# FrustumToVoxel gather backward: the gradient of a channels-last cost volume written channels-last in its own type
OUT=gpurun_out/r05c62; mkdir -p $OUT
timeout 150 python -m pytest tests/test_frustum_to_voxel.py tests/test_depth_fused_training_gpu.py -q -x -m gpu 2>&1 | tail -4 > $OUT/tests.txt
cat $OUT/tests.txt
for rep in 1 2; do
for mode in planar native; do
  if [ $mode = planar ]; then export DFM_F2V_PLANAR_GRAD=1; else unset DFM_F2V_PLANAR_GRAD; fi
  timeout 100 python bench.py --workload stereo_train --steps 30 --warmup 5 2> $OUT/err_$mode.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$mode', 'ms_per_step', d['ms_per_step'])" >> $OUT/ab.txt
done
done
cat $OUT/ab.txt
