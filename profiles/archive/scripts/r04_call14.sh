#!/bin/bash
# round 4, call 14: fp32 split precision, 3 pieces / 6 launches, To1 + 2-D convolutions: tests, fp32 training step + kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c14; mkdir -p $O
timeout 1200 python -m pytest tests/test_conv3d_g_gpu.py tests/test_modules.py tests/test_conv3d_gpu.py tests/test_fast_path.py tests/test_path_parity_gpu.py tests/test_depth_fused_training_gpu.py -q -m gpu --tb=short 2>&1 | grep -v "Warning\|warn\|forward_call\|^$" | tail -40 | tee $O/tests.txt
timeout 600 python tools/stereo_train_timing.py --dtype fp32 --fused-only --iters 3 2>&1 | tail -1 | tee $O/stereo_train_timing_fp32.txt
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt14 -- python $GRAFT_REPO_ROOT/tools/stereo_train_timing.py --dtype fp32 --iters 2 --fused-only > /dev/null 2>&1)
python - > $O/stereo_train_kernel_stats_fp32.txt <<'PY'
import csv,glob
f=glob.glob('/tmp/kt14/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f'# rocprofv3 --kernel-trace --stats -- python tools/stereo_train_timing.py --dtype fp32 --iters 2 --fused-only (split precision: 3 pieces); total kernel time {tot/1e6:.2f} ms (4 steps)')
for r in rows[:30]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%  {r['Name'][:150]}")
PY
head -24 $O/stereo_train_kernel_stats_fp32.txt | cut -c1-200
