# round 6, call 8: what the output stores of conv3d_g_kernel cost (debug build of conv3d_g.hip: DFM_CONV_ABLATE bit 3 = no
# stores, bit 4 = lane-contiguous stores, bit 0 = stage only the first chunk, bit 1 = no weight loads)
mkdir -p gpurun_out/c8
export DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_convdbg.so
for ab in 0 8 16 1 2 9 11; do
  echo "== DFM_CONV_ABLATE=$ab"
  DFM_CONV_ABLATE=$ab python tools/conv_g_timing.py --only hg --no-miopen 2>/dev/null
  DFM_CONV_ABLATE=$ab python tools/conv_g_timing.py --only neck --no-miopen --case res0 2>/dev/null
done > gpurun_out/c8/ablation.txt 2>&1
unset DFM_HIP_LIB
tools/kernel_stats.sh $GRAFT_REPO_ROOT/gpurun_out/c8/ks backbone_one_stream:"--workload backbone --steps 10 --warmup 3"
