# round 6, call 9: 16-byte output stores (the halves of a wave trade pieces of their pixel) in conv3d_g_kernel and
# conv3d_k3_c32_kernel.  A/B on the debug build of conv3d_g.hip: DFM_CONV_ABLATE=32 is the round-5 form (4 x 8 bytes).
mkdir -p gpurun_out/c9
(python -m pytest tests/test_conv3d_gpu.py tests/test_conv3d_g_gpu.py tests/test_modules.py tests/test_path_parity_gpu.py tests/test_sweep_conv_gpu.py tests/test_fast_path.py tests/test_conv3d_to1n_gpu.py -x -q -m gpu 2>&1 | tail -5) > gpurun_out/c9/tests.txt
(
echo "== release build (16-byte stores)"; python tools/conv_g_timing.py --no-miopen 2>/dev/null
export DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_convdbg.so
for ab in 0 32 0 32; do echo "== debug build DFM_CONV_ABLATE=$ab"; DFM_CONV_ABLATE=$ab python tools/conv_g_timing.py --no-miopen 2>/dev/null; done
unset DFM_HIP_LIB
) > gpurun_out/c9/conv_g_layers.txt 2>&1
row() { DFM_FEATS_NHWC=1 python bench.py --workload $1 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['ms_per_step'], l['roofline']['frac'])"; }
for wl in backbone neck dfm_neck backbone_train stereo_train backbone neck dfm_neck; do row $wl; done > gpurun_out/c9/rows.txt 2>&1
python tools/conv_timing.py > gpurun_out/c9/conv_c32.txt 2>&1
