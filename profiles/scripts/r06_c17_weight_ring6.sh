# round 6, call 17: conv3d_g_kernel FAST body, weights five k-steps ahead through six buffers for one-fragment waves (PFW = 1)
# A/B: libdfm_hip_wring3.so = the same source with -DDFM_WRING3 (three buffers, two steps ahead: round 5)
mkdir -p gpurun_out/c17
(python -m pytest tests/test_conv3d_g_gpu.py tests/test_modules.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/c17/tests.txt
(
for i in 1 2 3; do
echo "== ring 6 (PFW = 1)"; python tools/conv_g_timing.py --only hg --no-miopen 2>/dev/null
echo "== ring 3"; DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_wring3.so python tools/conv_g_timing.py --only hg --no-miopen 2>/dev/null
done
echo "== conv4 with larger tiles, ring 6 build"
for plan in 2,4,8,8 2,2,8,16 4,8,8,8; do echo "plan $plan"; DFM_CONV_G_PLAN=$plan python tools/conv_g_timing.py --only hg --no-miopen --case conv4 2>/dev/null; done
) > gpurun_out/c17/layers.txt 2>&1
row() { DFM_FEATS_NHWC=1 python bench.py --workload $1 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', '$1', l['ms_per_step'], l['roofline']['frac'])"; }
for i in 1 2; do for wl in backbone backbone_train; do row $wl ring6; DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_wring3.so row $wl ring3; done; done > gpurun_out/c17/rows.txt 2>&1
