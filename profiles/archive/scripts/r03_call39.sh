#!/bin/bash
# round 3, call 39: where the strided-sweep kernel (config K, NHWC maps) spends its time: parts compiled out
O=gpurun_out/r03c39; mkdir -p $O
R=$PWD
for v in release clt_nocur clt_noprev clt_nostore clt_nogather; do
  L=$R/depth-from-motion_amd/lib/libdfm_hip_$v.so; [ $v = release ] && L=$R/depth-from-motion_amd/lib/libdfm_hip.so
  DFM_HIP_LIB=$L timeout 300 python bench.py --workload kitti_nhwc --steps 30 --warmup 5 2>&1 | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$v', j['value'], 'vol/s', j['ms_per_step'], 'ms/step kernel', j['roofline'].get('kernel_ms'), 'frac', j['roofline']['frac'])" >> $O/clt_ablation.txt
done
cat $O/clt_ablation.txt
