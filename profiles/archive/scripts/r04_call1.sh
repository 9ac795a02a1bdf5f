#!/bin/bash
# round 4, call 1: the pipelined body of the N* tile kernel -- parity (kernel-mode matrix + full-size N*), same-lease A/B
# against the serial body, ablations of both (debug build), and what the SMI tools say about the part
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c1; mkdir -p $O
timeout 900 python -m pytest tests/test_plane_sweep_gpu.py tests/test_nstar_shipped_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/tests.txt
timeout 300 tools/sweep_bench --rounds 9 --launches 3 pipe=1 pipe=2 lanes=512,ppl=4,pipe=1 lanes=512,ppl=4,pipe=2 planes=4,pipe=2 2>&1 | tee $O/ab.txt
mkdir -p /tmp/dbg && cp depth-from-motion_amd/lib/libdfm_hip_dbg.so /tmp/dbg/libdfm_hip.so
for ab in 0 1 2 4 3; do
  echo "## DFM_ABLATE=$ab (1 no staging, 2 no volume stores, 4 no taps/blend)" | tee -a $O/ablate.txt
  LD_LIBRARY_PATH=/tmp/dbg DFM_ABLATE=$ab timeout 200 tools/sweep_bench --rounds 5 --launches 3 pipe=1 pipe=2 lanes=512,ppl=4,pipe=2 2>&1 | grep -v "^#" | tee -a $O/ablate.txt
done
( rocm-smi --showclocks --showpower --showmemuse --showcomputepartition --showmemorypartition --showperflevel --showmaxpower 2>&1 | head -80 ) > $O/rocm_smi.txt
( amd-smi static 2>&1 | head -250 ) > $O/amd_smi_static.txt
( amd-smi metric 2>&1 | head -250 ) > $O/amd_smi_metric.txt
ls /sys/class/drm/ > $O/sysfs.txt 2>&1; for f in /sys/class/drm/card*/device/{pp_dpm_mclk,pp_dpm_sclk,pp_dpm_fclk,current_compute_partition,current_memory_partition,power_dpm_force_performance_level,mem_info_vram_total}; do echo "== $f"; cat $f 2>&1; done >> $O/sysfs.txt
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_default.json; cut -c1-900 $O/bench_default.json
