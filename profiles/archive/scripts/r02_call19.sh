#!/bin/bash
# ablations of the general MFMA conv (debug-hook build): what the staging, the weight loads and the
# LDS bank conflicts cost
cd $GRAFT_REPO_ROOT
O=gpurun_out/c19; mkdir -p $O
export DFM_HIP_LIB=$GRAFT_REPO_ROOT/depth-from-motion_amd/lib/libdfm_hip_dbg.so
for AB in 0 1 2 4 7; do
  echo "## DFM_CONV_ABLATE=$AB (1: stage first chunk only, 2: weights of the first tap only, 4: conflict-free LDS reads)" >> $O/conv_g_ablation.txt
  DFM_CONV_ABLATE=$AB timeout 120 python tools/conv_g_timing.py --no-miopen --iters 10 2>&1 | grep -E "neck.res1|neck.res2|hg.conv2|hg.conv1|hg.conv6|dfmneck" >> $O/conv_g_ablation.txt
done
cat $O/conv_g_ablation.txt | cut -c1-120
