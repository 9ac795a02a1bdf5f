#!/bin/bash
# round 3, call 41: strided-sweep kernel with the next stage's first gather round in flight under the flush; A/B on one box
O=gpurun_out/r03c41; mkdir -p $O
R=$PWD
timeout 900 python -m pytest tests/test_plane_sweep_gpu.py -q -x -k "clt or kitti or strided or channels_last" 2>&1 | tail -2 > $O/tests.txt; cat $O/tests.txt
for rep in 1 2; do
for v in old new; do
  L=$R/depth-from-motion_amd/lib/libdfm_hip.so; [ $v = old ] && L=$R/depth-from-motion_amd/lib/libdfm_hip_cltold.so
  for wl in kitti_nhwc kitti; do
  DFM_HIP_LIB=$L timeout 300 python bench.py --workload $wl --steps 30 --warmup 5 2>&1 | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$v $wl', j['value'], 'vol/s', j['ms_per_step'], 'ms/step kernel', j['roofline'].get('kernel_ms'), 'frac', j['roofline']['frac'])" >> $O/clt_ab.txt
  done
done
done
cat $O/clt_ab.txt
