#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c7; mkdir -p $O
timeout 900 python -m pytest tests/test_plane_sweep_gpu.py tests/test_point_sample_gpu.py -x -q -k "backward" > $O/pytest_bwd.txt 2>&1; tail -12 $O/pytest_bwd.txt
timeout 300 python bench.py --workload sweep_bwd --steps 5 --warmup 2 > $O/bench_bwd.json 2> $O/bench_bwd.err; cat $O/bench_bwd.json; tail -2 $O/bench_bwd.err
timeout 300 python bench.py --workload sweep_bwd_kitti --steps 5 --warmup 2 > $O/bench_bwd_kitti.json 2>> $O/bench_bwd.err; cat $O/bench_bwd_kitti.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt7 -- python $GRAFT_REPO_ROOT/bench.py --workload sweep_bwd --steps 3 --warmup 1 > /dev/null 2>&1)
python - <<'PY' > gpurun_out/c7/bwd_kernel_stats.txt
import csv,glob
f=glob.glob('/tmp/kt7/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:10.3f} ms {float(r['AverageNs'])/1e3:10.1f} us {float(r['Percentage']):6.2f}%  {r['Name'][:120]}")
PY
cat $O/bwd_kernel_stats.txt
