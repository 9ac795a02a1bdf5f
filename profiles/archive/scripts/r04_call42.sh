#!/bin/bash
# round 4, call 42: backward walking kernel with 1 / 2 workgroups per CU (does the atomics working set fit the L2 then?)
cd /root/repo; mkdir -p gpurun_out
L=$PWD/depth-from-motion_amd/lib
run() { timeout 300 python bench.py --workload sweep_bwd_kitti --no-secondary --no-traffic --no-smi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sweep_bwd_kitti', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('bwd_kernel'))"; }
( for v in bpad1 bpad2; do echo "## ${v:-release}"; DFM_HIP_LIB=$L/libdfm_hip${v:+_$v}.so run; done ) > gpurun_out/r04_c42_ablate.txt 2>&1
cat gpurun_out/r04_c42_ablate.txt
