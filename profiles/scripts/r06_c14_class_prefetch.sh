# round 6, call 14: generic tap loop of conv3d_g_kernel (transposed layers): the first weights of the next (class, chunk)
# requested at the end of the previous one.  A/B: libdfm_hip_noprefetch.so = the same source with -DDFM_NO_CLASS_PREFETCH
mkdir -p gpurun_out/c14
(python -m pytest tests/test_conv3d_g_gpu.py tests/test_modules.py tests/test_path_parity_gpu.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/c14/tests.txt
(
for i in 1 2 3; do
echo "== prefetch"; python tools/conv_g_timing.py --only hg --no-miopen 2>/dev/null
echo "== no prefetch"; DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_noprefetch.so python tools/conv_g_timing.py --only hg --no-miopen 2>/dev/null
done
) > gpurun_out/c14/layers.txt 2>&1
row() { DFM_FEATS_NHWC=1 python bench.py --workload $1 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', '$1', l['ms_per_step'], l['roofline']['frac'])"; }
for i in 1 2; do for wl in backbone backbone_train; do row $wl prefetch; DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_noprefetch.so row $wl noprefetch; done; done > gpurun_out/c14/rows.txt 2>&1
