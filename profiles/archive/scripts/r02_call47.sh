#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c47; mkdir -p $O
timeout 600 python -m pytest tests/test_plane_sweep_gpu.py -x -q -m gpu -k "backward or bwd or grad" 2>&1 | tail -3
for w in sweep_bwd sweep_bwd_kitti; do timeout 300 python bench.py --workload $w 2>$O/err_$w.txt | tee $O/bench_$w.json | cut -c1-400; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bwd -- python $GRAFT_REPO_ROOT/bench.py --workload sweep_bwd --steps 3 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/c47/prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r['Name'][:90], r['Calls'], r['AverageNs'])
PY
