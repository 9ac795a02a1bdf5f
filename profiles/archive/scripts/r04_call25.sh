#!/bin/bash
# round 4, call 25: the matrix-core unpack as isolated builds (build_variant): order left to hipcc (m2) or pinned (m1),
# with the half-pass body / the late-store loop compiled in or not -- which of the 5.16..5.7 ms of calls 20-24 is code, which is the box
cd /root/repo; mkdir -p gpurun_out
L=depth-from-motion_amd/lib
( for i in 1; do
  for v in "" m2 m2late m1late m2halves m1halves; do
    lib=$L/libdfm_hip${v:+_$v}.so
    echo "## variant ${v:-release(m1)}"
    LD_PRELOAD=$PWD/$lib timeout 300 tools/sweep_bench --workload nstar --rounds 3 --launches 4 unpack=1 unpack=2 | grep -v "^#"
    case "$v" in *halves) LD_PRELOAD=$PWD/$lib timeout 300 tools/sweep_bench --workload nstar --rounds 3 --launches 4 unpack=3 | grep -v "^#";; esac
    case "$v" in *late) DFM_LATE=1 LD_PRELOAD=$PWD/$lib timeout 300 tools/sweep_bench --workload nstar --rounds 3 --launches 4 unpack=1 | grep -v "^#" | sed 's/^/late: /';; esac
  done; done ) > gpurun_out/r04_c25_variants.txt 2>&1
cat gpurun_out/r04_c25_variants.txt
