#!/bin/bash
# round 4, call 3: band cuts aligned inside a plane (plane boundaries stay 16-byte cuts, patch pass unchanged), the pipelined
# body with 78 KiB instead of 80 KiB - 128 B of LDS (does a second workgroup fit the CU now?), ablations, SMI under load
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c3; mkdir -p $O
timeout 900 python -m pytest tests/test_plane_sweep_gpu.py tests/test_nstar_shipped_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee $O/tests.txt
timeout 300 tools/sweep_bench --rounds 9 --launches 3 pipe=1 pipe=1,align=32 pipe=1,align=64 pipe=2 pipe=2,align=32 pipe=2,align=64 lanes=512,ppl=4,pipe=2,align=32 lanes=512,ppl=4,pipe=1,align=32 lanes=512,ppl=4,pipe=2 2>&1 | tee $O/ab.txt
mkdir -p /tmp/dbg && cp depth-from-motion_amd/lib/libdfm_hip_dbg.so /tmp/dbg/libdfm_hip.so
for ab in 5 2 3 1; do
  echo "## DFM_ABLATE=$ab (1 no staging, 2 no volume stores, 4 no taps/blend)" | tee -a $O/ablate.txt
  LD_LIBRARY_PATH=/tmp/dbg DFM_ABLATE=$ab timeout 200 tools/sweep_bench --rounds 5 --launches 3 pipe=1 pipe=1,align=32 pipe=2 pipe=2,align=32 lanes=512,ppl=4,pipe=2,align=32 lanes=512,ppl=4,pipe=1,align=32 2>&1 | grep -v "^#" | tee -a $O/ablate.txt
done
python tools/part_info.py --load tools/sweep_bench --rounds 60 --launches 3 pipe=1 > $O/part_info.json 2>&1; cat $O/part_info.json | tr -d '\n ' | cut -c1-1500; echo
