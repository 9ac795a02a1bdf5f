mkdir -p gpurun_out/c2
(python -m pytest tests/test_conv3d_to1n_gpu.py tests/test_modules.py tests/test_path_parity_gpu.py tests/test_frustum_to_voxel.py tests/test_sweep_walk_gpu.py tests/test_backward_gpu.py -x -q -m gpu 2>&1 | tail -8) > gpurun_out/c2/tests.txt
for wl in backbone sweep_bwd_kitti sweep_bwd_kitti_cl backbone_train stereo_train stereo_infer; do
  python bench.py --workload $wl --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', l['ms_per_step'], l['roofline']['frac'])"
done > gpurun_out/c2/rows.txt 2>&1
for i in 1 2 3; do
DFM_FEATS_NHWC=1 python bench.py --workload backbone --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('backbone fused_pred', l['ms_per_step'])"
DFM_FEATS_NHWC=1 DFM_PRED_UNFUSED=1 python bench.py --workload backbone --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('backbone unfused_pred', l['ms_per_step'])"
done >> gpurun_out/c2/rows.txt 2>&1
(
python tools/conv_g_timing.py --only hg --no-miopen
for plan in 2,4,8,8 2,8,8,4 2,2,8,16 1,4,4,8 1,2,8,8 4,4,8,16 3,6,8,8; do echo "== conv6 plan $plan"; DFM_CONV_G_PLAN=$plan python tools/conv_g_timing.py --only hg --no-miopen --case conv6; done
for plan in 2,4,8,8 2,2,8,16 1,2,8,8 1,4,4,8 4,4,8,16 4,8,4,16 3,6,8,8; do echo "== conv2 plan $plan"; DFM_CONV_G_PLAN=$plan python tools/conv_g_timing.py --only hg --no-miopen --case conv2; done
for plan in 1,2,8,8 1,8,4,4 1,2,4,16 1,1,8,16; do echo "== conv1/conv3 plan $plan"; DFM_CONV_G_PLAN=$plan python tools/conv_g_timing.py --only hg --no-miopen --case s2; done
for plan in 1,2,8,8 1,2,4,16 2,4,8,8 2,2,8,16; do echo "== conv4/5 plan $plan"; DFM_CONV_G_PLAN=$plan python tools/conv_g_timing.py --only hg --no-miopen --case conv4; DFM_CONV_G_PLAN=$plan python tools/conv_g_timing.py --only hg --no-miopen --case conv5; done
) > gpurun_out/c2/hg_plans.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_bb -o bb -- python $GRAFT_REPO_ROOT/bench.py --workload backbone --steps 10 --warmup 3 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py /tmp/prof_bb > gpurun_out/c2/backbone_kernel_stats.txt 2>&1 || (ls -R /tmp/prof_bb | head -30 > gpurun_out/c2/backbone_kernel_stats.txt)
