#!/bin/bash
# round 3, call 9: the fused sweep + dres0 kernel as shipped: tests, timing, kernel stats, backbone bench fused / unfused
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c9; mkdir -p $O
timeout 600 python -m pytest tests/test_sweep_conv_gpu.py tests/test_fast_path.py -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
timeout 300 python tools/sweep_conv_timing.py 2>&1 | tail -2 | tee $O/sweep_conv_timing.txt
DFM_BATCH=4 timeout 300 python tools/sweep_conv_timing.py 2>&1 | tail -2 | tee -a $O/sweep_conv_timing.txt
for v in "" 1; do
  DFM_FEATS_NHWC=1 DFM_NO_SWEEP_FUSION=$v timeout 200 python bench.py --workload backbone > $O/bench_backbone_nofusion$v.json 2>/dev/null; tail -1 $O/bench_backbone_nofusion$v.json | cut -c1-330
done
export TMPDIR=/tmp
(cd /tmp && DFM_FEATS_NHWC=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt9 -- python $GRAFT_REPO_ROOT/bench.py --workload backbone --steps 20 --warmup 3 > /dev/null 2>&1)
python - > $O/backbone_kernel_stats.txt <<'PY'
import csv,glob
f=glob.glob('/tmp/kt9/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f'# bench.py --workload backbone --steps 20 --warmup 3 (DFM_FEATS_NHWC=1), fused sweep+dres0; total kernel time {tot/1e6:.2f} ms over 23 passes = {tot/23e6:.3f} ms per pass')
for r in rows[:22]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%  {r['Name'][:120]}")
PY
head -16 $O/backbone_kernel_stats.txt
