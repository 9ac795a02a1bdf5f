# round 6, call 6: GroupNorm forward as two launches (the last statistics workgroup merges) vs three (DFM_GN_MERGE_KERNEL=1)
mkdir -p gpurun_out/c6
(python -m pytest tests/test_group_norm.py tests/test_modules.py tests/test_path_parity_gpu.py tests/test_conv3d_g_gpu.py tests/test_fast_path.py -x -q -m gpu 2>&1 | tail -5) > gpurun_out/c6/tests.txt
row() { DFM_FEATS_NHWC=1 python bench.py --workload $1 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', '$1', l['ms_per_step'], l['roofline']['frac'])"; }
for i in 1 2 3; do
  row backbone last_block_merge
  DFM_GN_MERGE_KERNEL=1 row backbone merge_kernel
done > gpurun_out/c6/rows.txt 2>&1
for i in 1 2; do
  row backbone_train last_block_merge; DFM_GN_MERGE_KERNEL=1 row backbone_train merge_kernel
  row dfm_neck last_block_merge; DFM_GN_MERGE_KERNEL=1 row dfm_neck merge_kernel
done >> gpurun_out/c6/rows.txt 2>&1
