#!/bin/bash
# round 4, call 10: NHWC training of the 2-D necks vs NCHW-between-layers (DFM_TRAIN_NCHW=1); the failing fused-vs-materialised test in full
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c10; mkdir -p $O
timeout 600 python -m pytest tests/test_depth_fused_training_gpu.py -q -m gpu -x --tb=short 2>&1 | grep -v Warning | tail -40 | tee $O/tests_fused.txt
timeout 600 python -m pytest tests/test_conv3d_g_gpu.py tests/test_modules.py tests/test_path_parity_gpu.py tests/test_fast_path.py -q -m gpu -x --tb=short 2>&1 | tail -12 | tee $O/tests.txt
echo "## NHWC training" | tee $O/stereo_train_timing.txt
timeout 300 python tools/stereo_train_timing.py --dtype bf16 --fused-only 2>&1 | tail -1 | tee -a $O/stereo_train_timing.txt
echo "## DFM_TRAIN_NCHW=1" | tee -a $O/stereo_train_timing.txt
DFM_TRAIN_NCHW=1 timeout 300 python tools/stereo_train_timing.py --dtype bf16 --fused-only 2>&1 | tail -1 | tee -a $O/stereo_train_timing.txt
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt10 -- python $GRAFT_REPO_ROOT/tools/stereo_train_timing.py --dtype bf16 --iters 3 --fused-only > /dev/null 2>&1)
python - > $O/stereo_train_kernel_stats.txt <<'PY'
import csv,glob
f=glob.glob('/tmp/kt10/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f'# rocprofv3 --kernel-trace --stats -- python tools/stereo_train_timing.py --dtype bf16 --iters 3 --fused-only (NHWC training); total kernel time {tot/1e6:.2f} ms (5 steps)')
for r in rows[:40]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%  {r['Name'][:150]}")
PY
head -32 $O/stereo_train_kernel_stats.txt | cut -c1-190
