# round 6, call 4: round-5 counted loops + the fit kernel's whole-map bound; dense gather at N* (experiment)
mkdir -p gpurun_out/c4
(python -m pytest tests/test_sweep_walk_gpu.py tests/test_backward_gpu.py tests/test_plane_sweep_gpu.py -x -q -m gpu -k "gather or walk or backward or bwd" 2>&1 | tail -6) > gpurun_out/c4/tests.txt
row() { python bench.py --workload $1 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', '$1', l['ms_per_step'], l['roofline']['frac'])"; }
for i in 1 2; do
for wl in sweep_bwd_kitti sweep_bwd_kitti_cl; do
  row $wl new
  DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_r5loops.so row $wl r5loops
done
done > gpurun_out/c4/rows.txt 2>&1
row sweep_bwd mfma >> gpurun_out/c4/rows.txt 2>&1
DFM_GATHER_DENSE=1 row sweep_bwd dense_gather >> gpurun_out/c4/rows.txt 2>&1
row stereo_train new >> gpurun_out/c4/rows.txt 2>&1
