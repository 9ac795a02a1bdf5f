# round 6, call 18: planner with a latency floor per tap (ring-6 build); A/B against libdfm_hip_wring3.so (round-5 ring, old planner)
mkdir -p gpurun_out/c18
(python -m pytest tests/test_conv3d_g_gpu.py tests/test_modules.py tests/test_path_parity_gpu.py tests/test_fast_path.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/c18/tests.txt
(
for i in 1 2; do
echo "== new planner + ring 6"; python tools/conv_g_timing.py --no-miopen 2>/dev/null
echo "== round-5 ring, old planner"; DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_wring3.so python tools/conv_g_timing.py --no-miopen 2>/dev/null
done
) > gpurun_out/c18/layers.txt 2>&1
row() { DFM_FEATS_NHWC=1 python bench.py --workload $1 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', '$1', l['ms_per_step'], l['roofline']['frac'])"; }
for i in 1 2; do for wl in backbone backbone_train neck dfm_neck stereo_train; do row $wl new; DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_wring3.so row $wl old; done; done > gpurun_out/c18/rows.txt 2>&1
