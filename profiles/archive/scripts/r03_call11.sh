#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c11; mkdir -p $O
timeout 900 python -m pytest tests/test_plane_sweep_gpu.py tests/test_nstar_shipped_gpu.py -q -m gpu -x -k "backward or grad or bwd" 2>&1 | tail -4 | tee $O/tests.txt
for w in sweep_bwd sweep_bwd_kitti; do timeout 300 python bench.py --workload $w > $O/bench_$w.json 2>/dev/null; tail -1 $O/bench_$w.json | cut -c1-420; done
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt11 -- python $GRAFT_REPO_ROOT/bench.py --workload sweep_bwd --steps 4 --warmup 1 > /dev/null 2>&1)
python - <<'PY' | tee $O/bwd_kernel_stats.txt
import csv,glob
f=glob.glob('/tmp/kt11/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print(f"{r['Calls']:>6} {float(r['AverageNs'])/1e3:10.1f} us  {r['Name'][:110]}")
PY
