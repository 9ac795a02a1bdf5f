#!/bin/bash
# round 4, call 2: does the alignment of the store runs matter?  tile boundaries at multiples of 8 / 32 / 64 lattice points
# (16 / 64 / 128 bytes), serial and pipelined body, stores-only ablation (DFM_ABLATE=5) per alignment, SMI readings under load
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c2; mkdir -p $O
timeout 900 python -m pytest tests/test_plane_sweep_gpu.py tests/test_nstar_shipped_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee $O/tests.txt
timeout 300 tools/sweep_bench --rounds 9 --launches 3 pipe=1 pipe=1,align=32 pipe=1,align=64 pipe=2 pipe=2,align=32 pipe=2,align=64 lanes=512,ppl=4,pipe=2,align=64 lanes=512,ppl=4,pipe=1,align=64 2>&1 | tee $O/ab.txt
mkdir -p /tmp/dbg && cp depth-from-motion_amd/lib/libdfm_hip_dbg.so /tmp/dbg/libdfm_hip.so
for ab in 5 4 2; do
  echo "## DFM_ABLATE=$ab (1 no staging, 2 no volume stores, 4 no taps/blend)" | tee -a $O/ablate.txt
  LD_LIBRARY_PATH=/tmp/dbg DFM_ABLATE=$ab timeout 200 tools/sweep_bench --rounds 5 --launches 3 pipe=1 pipe=1,align=32 pipe=1,align=64 pipe=2,align=64 planes=1,pipe=1,align=64 2>&1 | grep -v "^#" | tee -a $O/ablate.txt
done
python tools/part_info.py --load tools/sweep_bench --rounds 40 --launches 3 pipe=1 > $O/part_info.json 2>&1; cat $O/part_info.json | tr -d '\n ' | cut -c1-1200; echo
