#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c14; mkdir -p $O
timeout 200 python bench.py --workload backbone_train --reducer ddp --steps 5 > $O/ddp.out 2> $O/ddp.err; tail -5 $O/ddp.err; tail -1 $O/ddp.out | cut -c1-200
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt14 -- python $GRAFT_REPO_ROOT/bench.py --workload backbone_train --steps 10 --warmup 2 > /dev/null 2>&1)
python - > $O/train_kernel_stats.txt <<'PY'
import csv,glob
f=glob.glob('/tmp/kt14/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f'# bench.py --workload backbone_train --steps 10 --warmup 2 (13 fwd+bwd passes of DfMBackbone, config K, bf16 NDHWC); total kernel time {tot/1e6:.2f} ms = {tot/13e6:.3f} ms per pass')
for r in rows[:26]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%  {r['Name'][:130]}")
PY
head -28 $O/train_kernel_stats.txt | cut -c1-175
