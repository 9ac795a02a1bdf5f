#!/bin/bash
# round 4, call 16: odd depth groups walk the channel blocks backwards (experiment build) vs the shipped order; test_fast_path
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c16; mkdir -p $O
timeout 200 tools/sweep_bench --rounds 9 --launches 3 default lanes=512,ppl=4 2>&1 | sed 's/^/shipped /' | tee $O/rev_odd.txt
mkdir -p /tmp/rev && cp depth-from-motion_amd/lib/libdfm_hip_rev.so /tmp/rev/libdfm_hip.so
LD_LIBRARY_PATH=/tmp/rev timeout 200 tools/sweep_bench --rounds 9 --launches 3 default lanes=512,ppl=4 2>&1 | grep -v "^#" | sed 's/^/rev_odd /' | tee -a $O/rev_odd.txt
timeout 200 tools/sweep_bench --rounds 9 --launches 3 default lanes=512,ppl=4 2>&1 | grep -v "^#" | sed 's/^/shipped /' | tee -a $O/rev_odd.txt
timeout 600 python -m pytest tests/test_fast_path.py -q -m gpu --tb=short 2>&1 | tail -3 | tee $O/tests.txt
