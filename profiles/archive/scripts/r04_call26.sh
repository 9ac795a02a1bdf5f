#!/bin/bash
# round 4, call 26: do the two workgroups of a CU run in lockstep?  First-generation workgroups in an odd wave slot
# start N x 256 clocks late (a block period is ~7000 clocks)
cd /root/repo; mkdir -p gpurun_out
( for i in 1 2; do for n in 0 4 8 14 20 28; do
    echo "## sleep $n x 256 clocks"
    DFM_PHASE_SLEEP=$n timeout 300 tools/sweep_bench --workload nstar --rounds 3 --launches 4 unpack=1 unpack=2 | grep -v "^#"
  done; done
  echo "## all generations (first_gen = everything), sleep 14"
  DFM_FIRST_GEN=100000000 DFM_PHASE_SLEEP=14 timeout 300 tools/sweep_bench --workload nstar --rounds 3 --launches 4 unpack=1 unpack=2 | grep -v "^#"
) > gpurun_out/r04_c26_phase.txt 2>&1
cat gpurun_out/r04_c26_phase.txt
