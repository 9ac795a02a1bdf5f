# round 6, call 20: conv3d_g_kernel FAST body, activation fragments read five k-steps ahead (six buffers) for one-fragment
# waves.  A/B: libdfm_hip_qring2.so = the same source with -DDFM_QRING6_MAXPFW=0 (one step ahead everywhere: round 5)
mkdir -p gpurun_out/c20
(python -m pytest tests/test_conv3d_g_gpu.py tests/test_modules.py tests/test_path_parity_gpu.py tests/test_fast_path.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/c20/tests.txt
(
for i in 1 2 3; do
echo "== reads 5 ahead (PFW 1)"; python tools/conv_g_timing.py --only hg --no-miopen 2>/dev/null
echo "== reads 1 ahead"; DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_qring2.so python tools/conv_g_timing.py --only hg --no-miopen 2>/dev/null
done
) > gpurun_out/c20/layers.txt 2>&1
row() { DFM_FEATS_NHWC=1 python bench.py --workload $1 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', '$1', l['ms_per_step'], l['roofline']['frac'])"; }
for i in 1 2; do for wl in backbone backbone_train stereo_train; do row $wl deep; DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_qring2.so row $wl shallow; done; done > gpurun_out/c20/rows.txt 2>&1
