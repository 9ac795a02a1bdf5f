#!/bin/bash
# round 4, call 13: the failing wide-modules test in full; kernel statistics of the fp32 training step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c13; mkdir -p $O
timeout 600 python -m pytest tests/test_modules.py -q -m gpu -x --tb=long -k wide_modules 2>&1 | grep -v "Warning\|warn\|forward_call" | head -150 > $O/tests.txt; grep -n "Error\|assert\|Mismatch\|Max abs\|test_modules.py:" $O/tests.txt | head -20
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt13 -- python $GRAFT_REPO_ROOT/tools/stereo_train_timing.py --dtype fp32 --iters 2 --fused-only > /dev/null 2>&1)
python - > $O/stereo_train_kernel_stats_fp32.txt <<'PY'
import csv,glob
f=glob.glob('/tmp/kt13/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f'# rocprofv3 --kernel-trace --stats -- python tools/stereo_train_timing.py --dtype fp32 --iters 2 --fused-only; total kernel time {tot/1e6:.2f} ms')
for r in rows[:30]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%  {r['Name'][:150]}")
PY
head -26 $O/stereo_train_kernel_stats_fp32.txt | cut -c1-200
