#!/bin/bash
# round 4, call 7: non-temporal vs plain stores of the lifted volumes (FrustumToVoxel, multi-view lifting), and the staged-metas test
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c7; mkdir -p $O
timeout 300 python -m pytest tests/test_data_geometry.py -q -m gpu -x 2>&1 | grep -B 40 "Error\|error" | head -80 | tee $O/tests.txt
for wl in f2v f2v_cl waymo waymo_cl; do
  for lib in nt plain; do
    if [ $lib = plain ]; then export DFM_HIP_LIB=$PWD/depth-from-motion_amd/lib/libdfm_hip_liftplain.so; else unset DFM_HIP_LIB; fi
    timeout 200 python bench.py --workload $wl 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl', '$lib', d['value'], d['unit'], d['ms_per_step'], 'ms', d['roofline']['frac'])" | tee -a $O/lift_nt_vs_plain.txt
  done
done
unset DFM_HIP_LIB
timeout 600 python -m pytest tests/test_frustum_to_voxel.py tests/test_point_sample_gpu.py -q -m gpu -x 2>&1 | tail -2 | tee -a $O/tests.txt
