#!/bin/bash
# GPU call 1: full GPU test suite, A/B of tile-kernel launch shapes, default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/c1/pytest.txt
tail -5 gpurun_out/c1/pytest.txt
timeout 300 tools/sweep_bench --rounds 7 --launches 3 \
  default chunk=15 chunk=29 \
  lanes=512,ppl=4 lanes=512,ppl=4,chunk=15 lanes=512,ppl=4,chunk=29 \
  lanes=512,ppl=4,lds=76 \
  lanes=1024,ppl=4,planes=2,lds=60 lanes=1024,ppl=4,planes=2,lds=60,chunk=8 lanes=1024,ppl=4,planes=2,lds=76,chunk=15 \
  lanes=1024,ppl=4,planes=4,lds=52 lanes=1024,ppl=4,planes=4,lds=52,chunk=8 \
  lanes=512,planes=2 lanes=512,planes=4 \
  > gpurun_out/c1/ab_nstar.txt 2>&1
cat gpurun_out/c1/ab_nstar.txt
timeout 200 tools/sweep_bench --workload nstar_aug --rounds 5 --launches 3 default lanes=512,ppl=4 > gpurun_out/c1/ab_nstar_aug.txt 2>&1
cat gpurun_out/c1/ab_nstar_aug.txt
timeout 400 python bench.py > gpurun_out/c1/bench_default.json 2> gpurun_out/c1/bench_default.err
cat gpurun_out/c1/bench_default.json
tail -3 gpurun_out/c1/bench_default.err
