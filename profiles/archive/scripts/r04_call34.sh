#!/bin/bash
# round 4, call 34: HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of config K forward, walking kernel (5) vs per-plane kernel (4)
cd /root/repo; mkdir -p gpurun_out
export DFM_HIP_LIB=$PWD/depth-from-motion_amd/lib/libdfm_hip_w4.so
for k in 5 4; do
  LD_PRELOAD=$DFM_HIP_LIB timeout 300 python tools/pmc_traffic.py --workload kitti --out gpurun_out/r04_c34_kitti_traffic_kernel$k.json kernel=$k > gpurun_out/r04_c34_k$k.txt 2>&1
done
LD_PRELOAD=$DFM_HIP_LIB timeout 200 tools/sweep_bench --workload kitti --rounds 3 --launches 4 kernel=5 kernel=4 > gpurun_out/r04_c34_sweep_bench.txt 2>&1
cat gpurun_out/r04_c34_k5.txt gpurun_out/r04_c34_k4.txt gpurun_out/r04_c34_sweep_bench.txt
python - <<'PY'
import json
for k in (5,4):
    d=json.load(open(f'gpurun_out/r04_c34_kitti_traffic_kernel{k}.json'))
    for w,v in d.items():
        for key,e in v.items(): print(k, e['fetch_kb'], e['write_kb'])
PY
