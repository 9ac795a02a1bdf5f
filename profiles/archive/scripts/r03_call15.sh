#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c15; mkdir -p $O
timeout 900 python -m pytest tests/test_plane_sweep_gpu.py -q -m gpu -x -k "channels_last" 2>&1 | tail -3 | tee $O/tests.txt
timeout 200 python bench.py --workload backbone_train 2>/dev/null | grep '^{' | tail -1 | tee $O/bench_train.json | cut -c1-300
timeout 200 python bench.py --workload backbone_train --reducer bucket 2>/dev/null | grep '^{' | tail -1 | tee $O/bench_train_bucket.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('gradient_exchange'))"
timeout 200 python bench.py --workload backbone_train --reducer ddp 2>/dev/null | grep '^{' | tail -1 | tee $O/bench_train_ddp.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('gradient_exchange'))"
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt15 -- python $GRAFT_REPO_ROOT/bench.py --workload backbone_train --steps 10 --warmup 2 > /dev/null 2>&1)
python - > $O/train_kernel_stats.txt <<'PY'
import csv,glob
f=glob.glob('/tmp/kt15/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f'# bench.py --workload backbone_train --steps 10 --warmup 2 (13 fwd+bwd passes of DfMBackbone, config K, bf16 NDHWC, output gradients fed directly); total kernel time {tot/1e6:.2f} ms = {tot/13e6:.3f} ms per pass')
for r in rows[:24]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%  {r['Name'][:130]}")
PY
head -22 $O/train_kernel_stats.txt | cut -c1-170
