#!/bin/bash
# final-state evidence of the headline in ONE lease: full GPU suite, kernel trace of the two tile-kernel
# shapes (torch-free harness), PMC traffic of the autotuner's candidates, the bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/c43; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
for cfg in default lanes=512,ppl=4; do
  k=$(echo $cfg | tr ',=' '__')
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$k -- $GRAFT_REPO_ROOT/tools/sweep_bench --rounds 4 --launches 3 $cfg > $GRAFT_REPO_ROOT/$O/kt_$k.log 2>&1)
  f=$(find /tmp/kt_$k -name '*kernel_stats.csv' | head -1)
  echo "== $cfg" >> $O/kernel_stats.txt; cat "$f" >> $O/kernel_stats.txt
done
cat $O/kernel_stats.txt | cut -c1-160
timeout 600 python tools/pmc_traffic.py --out $O/r02_nstar_traffic.json > $O/pmc.txt 2>&1
tail -8 $O/pmc.txt
cp $O/r02_nstar_traffic.json profiles/r02_nstar_traffic.json 2>/dev/null
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json
timeout 300 python bench.py --workload nstar_aug --no-cpu-baseline > $O/bench_aug.json 2> $O/bench_aug.err; cat $O/bench_aug.json | cut -c1-400
timeout 300 python bench.py --workload kitti --no-cpu-baseline > $O/bench_kitti.json 2> $O/bench_kitti.err; cat $O/bench_kitti.json | cut -c1-400
