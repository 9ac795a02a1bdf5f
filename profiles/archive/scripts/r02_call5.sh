#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c5; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 200 tools/sweep_bench --rounds 7 --launches 3 default lanes=512,ppl=4 lanes=512,ppl=4,lds=64 > $O/ab.txt 2>&1; cat $O/ab.txt
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt5 -- $GRAFT_REPO_ROOT/tools/sweep_bench --rounds 4 --launches 3 lanes=512,ppl=4 > /dev/null 2>&1)
cat $(find /tmp/kt5 -name '*kernel_stats.csv' | head -1) | cut -c1-60,200-400 > $O/kernel_stats_v4.txt; cat $O/kernel_stats_v4.txt
timeout 600 python tools/backbone_timing.py > $O/backbone.txt 2>&1; cat $O/backbone.txt
DFM_NO_MFMA_CONV=1 timeout 600 python tools/backbone_timing.py > $O/backbone_nomfma.txt 2>&1; cat $O/backbone_nomfma.txt
