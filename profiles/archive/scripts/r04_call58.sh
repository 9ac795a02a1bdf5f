#!/bin/bash
# round 4, call 58: every secondary row of bench.py with the round's last library, one line each
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/r04_c58_all_rows.txt
for w in nstar_aug kitti kitti_nhwc sweep_bwd sweep_bwd_kitti waymo waymo_cl depth_head depth_head_bf16 f2v f2v_cl group_norm group_norm_cl backbone backbone_train neck dfm_neck; do
  timeout 200 python bench.py --workload $w --no-secondary --no-traffic --no-smi 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
    print('%-16s %10.2f %-22s %8.4f ms/step  %s %8.1f %s  frac %.4f' % ('$w', d['value'], d['unit'], d['ms_per_step'], r['bound'], r['achieved'], r['unit'], r['frac']))
except Exception as e: print('$w', 'FAILED', e)
" >> gpurun_out/r04_c58_all_rows.txt
done
cat gpurun_out/r04_c58_all_rows.txt
