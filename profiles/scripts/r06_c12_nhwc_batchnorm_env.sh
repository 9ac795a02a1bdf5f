# round 6, call 12: does torch's MIOpen BatchNorm take NHWC tensors in place when asked to (no layout copies around it)?
mkdir -p gpurun_out/c12
row() { python bench.py --workload $1 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', '$1', l['ms_per_step'])"; }
for i in 1 2; do
row stereo_train default
PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM=1 row stereo_train nhwc_bn
PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM=1 PYTORCH_MIOPEN_SUGGEST_NHWC=1 row stereo_train nhwc_bn_conv
done > gpurun_out/c12/rows.txt 2>&1
export PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM=1
tools/kernel_stats.sh $GRAFT_REPO_ROOT/gpurun_out/c12/ks stereo_train_nhwc_bn:"--workload stereo_train --steps 5 --warmup 2"
