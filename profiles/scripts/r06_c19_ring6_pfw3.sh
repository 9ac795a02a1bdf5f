# round 6, call 19: six weight buffers (five k-steps ahead) for waves of up to 3 pixel fragments (the voxel necks' layers)
# A/B: libdfm_hip_ring6pfw1.so = the same source with -DDFM_WRING6_MAXPFW=1 (only one-fragment waves)
mkdir -p gpurun_out/c19
(python -m pytest tests/test_conv3d_g_gpu.py tests/test_modules.py tests/test_path_parity_gpu.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/c19/tests.txt
(
for i in 1 2; do
echo "== ring 6 up to PFW 3"; python tools/conv_g_timing.py --no-miopen 2>/dev/null
echo "== ring 6 for PFW 1 only"; DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_ring6pfw1.so python tools/conv_g_timing.py --no-miopen 2>/dev/null
done
) > gpurun_out/c19/layers.txt 2>&1
row() { DFM_FEATS_NHWC=1 python bench.py --workload $1 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', '$1', l['ms_per_step'], l['roofline']['frac'])"; }
for i in 1 2; do for wl in neck dfm_neck backbone backbone_train; do row $wl pfw3; DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_ring6pfw1.so row $wl pfw1; done; done > gpurun_out/c19/rows.txt 2>&1
