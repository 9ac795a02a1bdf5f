#!/bin/bash
# round 4, call 54: N* forward with every workgroup starting its channel-block loop at a different block (the stores
# of the resident workgroups spread over all channel planes instead of marching through them together)
cd /root/repo; mkdir -p gpurun_out
L=$PWD/depth-from-motion_amd/lib
( for i in 1 2; do for v in "" rot; do echo "## ${v:-release}"; LD_PRELOAD=$L/libdfm_hip${v:+_$v}.so timeout 300 tools/sweep_bench --workload nstar --rounds 3 --launches 4 unpack=1 unpack=2 lanes=512,ppl=4 | grep -v "^#"; done; done ) > gpurun_out/r04_c54_rotate.txt 2>&1
cat gpurun_out/r04_c54_rotate.txt
