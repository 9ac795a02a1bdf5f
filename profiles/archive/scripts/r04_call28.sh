#!/bin/bash
# round 4, call 28: config K forward with the depth axis walked per wave (footprints cached in registers): parity, then bench
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_plane_sweep_gpu.py -x -q -m gpu -k "clt" 2>&1 | tail -4
  timeout 600 python -m pytest tests/test_plane_sweep_gpu.py -x -q -m gpu -k 'nhwc or channels_last' 2>&1 | tail -3 ) > gpurun_out/r04_c28_tests.txt 2>&1
( for i in 1 2; do
  for k in 5 4; do
    echo "## kernel $k"
    timeout 300 python bench.py --workload kitti --kernel $k --no-secondary --no-traffic --no-smi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kitti', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms'))"
  done; echo '## kitti_nhwc (walks)'; timeout 300 python bench.py --workload kitti_nhwc --no-secondary --no-traffic --no-smi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kitti_nhwc', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms'))"; done ) > gpurun_out/r04_c28_bench.txt 2>&1
cat gpurun_out/r04_c28_tests.txt gpurun_out/r04_c28_bench.txt
