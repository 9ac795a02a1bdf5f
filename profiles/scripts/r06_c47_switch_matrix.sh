run() { sw="$1"; shift; echo "== $sw : $*"; env $sw timeout 900 python -m pytest "$@" -q -m gpu -p no:cacheprovider 2>&1 | tail -4 | grep -v "^$" | cut -c1-300; }
run DFM_GN_KEEP_Y=1 tests/test_group_norm.py tests/test_backward_gpu.py
run DFM_TO1_PADDED_BWD=1 tests/test_conv3d_gpu.py tests/test_conv3d_to1n_gpu.py tests/test_backward_gpu.py
run DFM_TO1_C32_FWD=1 tests/test_conv3d_gpu.py tests/test_conv3d_to1n_gpu.py
run DFM_C32_CAT=1 tests/test_conv3d_gpu.py tests/test_backward_gpu.py
run DFM_BILINEAR_MATMUL=1 tests/test_conv3d_g_gpu.py tests/test_modules.py
run DFM_BN2D_TORCH=1 tests/test_group_norm.py tests/test_modules.py
run DFM_NO_F2V_GATHER=1 tests/test_frustum_to_voxel.py tests/test_depth_fused_training_gpu.py
run DFM_NO_PREV_GATHER=1 tests/test_plane_sweep_gpu.py
run DFM_GATHER_NO_TABLE=1 tests/test_plane_sweep_gpu.py
run DFM_TRAIN_ONE_STREAM=1 tests/test_backward_gpu.py tests/test_path_parity_gpu.py
run DFM_WGRAD_COL=0 tests/test_conv3d_gpu.py
run DFM_CONV_GENERIC=1 tests/test_conv3d_g_gpu.py
run DFM_TRAIN_NCHW=1 tests/test_modules.py tests/test_backward_gpu.py
run DFM_PLAIN_WGRAD_1X1=1 tests/test_conv3d_g_gpu.py tests/test_backward_gpu.py
