#!/bin/bash
# round 4, call 8: where do the 41 ms of a DfMStereoPath bf16 training step go (kernel stats), staged-metas test
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c8; mkdir -p $O
timeout 300 python -m pytest tests/test_data_geometry.py -q -m gpu -x 2>&1 | tail -3 | tee $O/tests.txt
timeout 300 python tools/stereo_train_timing.py --dtype bf16 2>&1 | tail -3 | tee $O/stereo_train_timing.txt
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt8 -- python $GRAFT_REPO_ROOT/tools/stereo_train_timing.py --dtype bf16 --iters 3 > /dev/null 2>&1)
python - > $O/stereo_train_kernel_stats.txt <<'PY'
import csv,glob
f=glob.glob('/tmp/kt8/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f'# rocprofv3 --kernel-trace --stats -- python tools/stereo_train_timing.py --dtype bf16 --iters 3; total kernel time {tot/1e6:.2f} ms')
for r in rows[:45]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%  {r['Name'][:150]}")
PY
head -48 $O/stereo_train_kernel_stats.txt
