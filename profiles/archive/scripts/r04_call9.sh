#!/bin/bash
# round 4, call 9: 2-D MFMA convolutions with autograd recording + matrix-product bilinear backward: tests, training step timing and kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c9; mkdir -p $O
timeout 600 python -m pytest tests/test_conv3d_g_gpu.py tests/test_modules.py tests/test_depth_fused_training_gpu.py -q -m gpu -x 2>&1 | tail -15 | tee $O/tests.txt
timeout 300 python tools/stereo_train_timing.py --dtype bf16 2>&1 | tail -3 | tee $O/stereo_train_timing.txt
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt9 -- python $GRAFT_REPO_ROOT/tools/stereo_train_timing.py --dtype bf16 --iters 3 > /dev/null 2>&1)
python - > $O/stereo_train_kernel_stats.txt <<'PY'
import csv,glob
f=glob.glob('/tmp/kt9/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f'# rocprofv3 --kernel-trace --stats -- python tools/stereo_train_timing.py --dtype bf16 --iters 3; total kernel time {tot/1e6:.2f} ms (10 steps)')
for r in rows[:40]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%  {r['Name'][:150]}")
PY
head -42 $O/stereo_train_kernel_stats.txt | cut -c1-200
