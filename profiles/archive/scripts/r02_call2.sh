#!/bin/bash
# GPU call 2: full GPU suite (new tests), kernel trace of the two tile-kernel shapes, more A/B,
# PMC traffic of the autotuner's candidates, bench lines -- all in one lease.
cd $GRAFT_REPO_ROOT
O=gpurun_out/c2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -8 $O/pytest.txt
for cfg in default lanes=512,ppl=4; do
  k=$(echo $cfg | tr ',=' '__')
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$k -- $GRAFT_REPO_ROOT/tools/sweep_bench --rounds 4 --launches 3 $cfg > $GRAFT_REPO_ROOT/$O/kt_$k.log 2>&1)
  f=$(find /tmp/kt_$k -name '*kernel_stats.csv' | head -1)
  echo "== $cfg" >> $O/kernel_stats.txt; cat "$f" >> $O/kernel_stats.txt
done
cat $O/kernel_stats.txt
timeout 400 tools/sweep_bench --rounds 7 --launches 3 \
  default lanes=512,ppl=4 \
  lanes=512,ppl=4,chunk=2 lanes=512,ppl=4,chunk=4 lanes=512,ppl=4,chunk=8 \
  lanes=512,ppl=4,planes=4 lanes=512,ppl=4,planes=4,chunk=2 lanes=512,ppl=4,planes=4,chunk=4 \
  lanes=512,ppl=4,planes=1,lds=76 lanes=512,ppl=4,planes=1,lds=76,chunk=4 \
  lanes=256,ppl=4,planes=1,lds=38 lanes=256,ppl=4,planes=2,lds=38 lanes=256,ppl=4,planes=2,lds=38,chunk=4 \
  lanes=256,ppl=4,planes=1,lds=38,chunk=4 lanes=512,ppl=4,lds=64 \
  > $O/ab_nstar.txt 2>&1
cat $O/ab_nstar.txt
timeout 900 python tools/pmc_traffic.py --out $O/r02_nstar_traffic.json > $O/pmc.txt 2>&1
cat $O/pmc.txt
mkdir -p profiles; cp $O/r02_nstar_traffic.json profiles/r02_nstar_traffic.json 2>/dev/null
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json
timeout 300 python bench.py --workload nstar_aug --no-cpu-baseline > $O/bench_aug.json 2> $O/bench_aug.err; cat $O/bench_aug.json
timeout 300 python bench.py --workload kitti --no-cpu-baseline > $O/bench_kitti.json 2> $O/bench_kitti.err; cat $O/bench_kitti.json
