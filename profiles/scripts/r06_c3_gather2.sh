# round 6, call 3: the gather with staged footprints (v2) vs round 5's (DFM_GATHER_V1=1) vs round 5's loop form
mkdir -p gpurun_out/c3
(python -m pytest tests/test_sweep_walk_gpu.py tests/test_conv3d_to1n_gpu.py tests/test_backward_gpu.py tests/test_frustum_to_voxel.py -x -q -m gpu 2>&1 | tail -8) > gpurun_out/c3/tests.txt
row() { python bench.py --workload $1 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', '$1', l['ms_per_step'], l['roofline']['frac'])"; }
for i in 1 2; do
for wl in sweep_bwd_kitti sweep_bwd_kitti_cl; do
  row $wl v2
  DFM_GATHER_V1=1 row $wl v1
  DFM_GATHER_V1=1 DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_r5loops.so row $wl v1_r5loops
done
done > gpurun_out/c3/rows.txt 2>&1
for wl in stereo_train backbone_train; do
  row $wl v2; DFM_GATHER_V1=1 row $wl v1
  DFM_GATHER_V1=1 DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_r5loops.so row $wl v1_r5loops
done >> gpurun_out/c3/rows.txt 2>&1
tools/kernel_stats.sh $GRAFT_REPO_ROOT/gpurun_out/c3/ks bwd_kitti:"--workload sweep_bwd_kitti --steps 10 --warmup 3" bwd_kitti_cl:"--workload sweep_bwd_kitti_cl --steps 10 --warmup 3"
