#!/bin/bash
# round 4, call 4: 4 points per lane with paired 16-byte stores; bands_per_chunk / planes on top of aligned cuts;
# backward with hi+lo bf16 weight terms: parity at the tightened bar and N* timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c4; mkdir -p $O
timeout 900 python -m pytest tests/test_plane_sweep_gpu.py tests/test_nstar_shipped_gpu.py tests/test_sweep_bwd_mfma_gpu.py tests/test_backward_gpu.py -q -m gpu -x -s 2>&1 | grep -v "^$" | tail -6 | tee $O/tests.txt
timeout 300 tools/sweep_bench --rounds 9 --launches 3 pipe=1,align=8 default lanes=512,ppl=4 lanes=512,ppl=4,pair=2 lanes=512,ppl=4,pipe=1 chunk=2 chunk=4 lanes=512,ppl=4,chunk=2 lanes=512,ppl=4,chunk=4 lanes=512,ppl=4,planes=4 lanes=1024,ppl=4,planes=4 lanes=512,ppl=4,align=32 2>&1 | tee $O/ab.txt
mkdir -p /tmp/dbg && cp depth-from-motion_amd/lib/libdfm_hip_dbg.so /tmp/dbg/libdfm_hip.so
for ab in 5 2; do
  echo "## DFM_ABLATE=$ab (1 no staging, 2 no volume stores, 4 no taps/blend)" | tee -a $O/ablate.txt
  LD_LIBRARY_PATH=/tmp/dbg DFM_ABLATE=$ab timeout 200 tools/sweep_bench --rounds 5 --launches 3 default lanes=512,ppl=4 lanes=512,ppl=4,pair=2 chunk=4 lanes=512,ppl=4,chunk=4 2>&1 | grep -v "^#" | tee -a $O/ablate.txt
done
for wl in sweep_bwd sweep_bwd_kitti; do timeout 300 python bench.py --workload $wl 2>/dev/null | tail -1 > $O/bench_$wl.json; cut -c1-700 $O/bench_$wl.json; done
