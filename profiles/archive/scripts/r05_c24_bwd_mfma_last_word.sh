#!/bin/bash
# the eighth element from the neighbouring lane instead of a fifth load per fragment: tests, same-box A/B, TA counters
OUT=gpurun_out/r05c24; mkdir -p $OUT
timeout 600 python -m pytest tests/test_sweep_bwd_mfma_gpu.py tests/test_backward_gpu.py -m gpu -x -q 2>&1 | tail -5 > $OUT/tests.txt
L=depth-from-motion_amd/lib
for rep in 1 2; do
for v in _base ""; do
  echo "== lib$v" >> $OUT/ab.txt
  DFM_HIP_LIB=$PWD/$L/libdfm_hip$v.so timeout 120 python bench.py --workload sweep_bwd --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('roofline',{}).get('frac'))" >> $OUT/ab.txt 2>&1
done; done
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_UTCL1_REQUEST TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr"; do
  tag=$(echo $set | tr ' ' '+' | cut -c1-60)
  for v in _base ""; do
  (cd /tmp && export TMPDIR=/tmp && DFM_HIP_LIB=$GRAFT_REPO_ROOT/$L/libdfm_hip$v.so timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc${v}_$tag -- python $GRAFT_REPO_ROOT/bench.py --workload sweep_bwd --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$OUT/pmc${v}_$tag.log 2>&1)
  done
done
for d in $OUT/pmc*/; do echo "#### $d"; python tools/pmc_summary.py $d --kernel bwd_mfma; done > $OUT/pmc_summary.txt 2>&1
