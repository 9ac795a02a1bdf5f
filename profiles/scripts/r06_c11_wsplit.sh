# round 6, call 11: conv3d_g_kernel with its waves split 2 pixel groups x 2 channel groups (DFM_CONV_WSPLIT=1) vs 4 pixel groups
mkdir -p gpurun_out/c11
(DFM_CONV_WSPLIT=1 python -m pytest tests/test_conv3d_g_gpu.py tests/test_modules.py tests/test_path_parity_gpu.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/c11/tests_wsplit.txt
(
for i in 1 2; do
echo "== default"; python tools/conv_g_timing.py --no-miopen 2>/dev/null
echo "== DFM_CONV_WSPLIT=1"; DFM_CONV_WSPLIT=1 python tools/conv_g_timing.py --no-miopen 2>/dev/null
done
) > gpurun_out/c11/layers.txt 2>&1
row() { DFM_FEATS_NHWC=1 python bench.py --workload $1 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', '$1', l['ms_per_step'], l['roofline']['frac'])"; }
for wl in neck dfm_neck backbone backbone_train; do row $wl default; DFM_CONV_WSPLIT=1 row $wl wsplit; done > gpurun_out/c11/rows.txt 2>&1
