#!/bin/bash
# round 4, call 35: walking kernel, the four waves of a workgroup = four depth chunks of one tile (one L1) vs four tiles of one chunk
cd /root/repo; mkdir -p gpurun_out
L=$PWD/depth-from-motion_amd/lib
run() { timeout 300 python bench.py --workload kitti_nhwc --no-secondary --no-traffic --no-smi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kitti_nhwc', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms'))"; }
export DFM_HIP_LIB=$L/libdfm_hip_wchunk.so
( echo "## tiles x4 per workgroup, chunk 24"; run
  for c in 18 12 6; do echo "## wave = chunk, chunk $c"; DFM_WALK_WAVE_CHUNKS=1 DFM_WALK_CHUNK=$c run; done
  echo "## tiles x4 per workgroup, chunk 18"; DFM_WALK_CHUNK=18 run
  echo "## wave = chunk, chunk 18 (again)"; DFM_WALK_WAVE_CHUNKS=1 DFM_WALK_CHUNK=18 run
  DFM_WALK_WAVE_CHUNKS=1 DFM_WALK_CHUNK=18 timeout 600 python -m pytest tests/test_sweep_walk_gpu.py -x -q -m gpu 2>&1 | tail -1
) > gpurun_out/r04_c35_wave_chunks.txt 2>&1
cat gpurun_out/r04_c35_wave_chunks.txt
