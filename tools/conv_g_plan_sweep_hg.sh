cd $GRAFT_REPO_ROOT
run() { DFM_CONV_G_PLAN=$2 python tools/conv_g_timing.py --no-miopen --graph --iters 10 --case "$1" 2>&1 | grep MFMA | cut -c1-150 | sed "s/^/plan=$2  /"; }
for c in "hg.conv1" "hg.conv2" "hg.conv3" "hg.conv5" "hg.conv6"; do
  echo "== $c"; python tools/conv_g_timing.py --no-miopen --graph --iters 10 --case "$c" 2>&1 | grep MFMA | cut -c1-150
  for p in 1,8,4,4 1,4,8,4 1,4,4,8 1,2,8,8 1,2,4,16 1,4,2,16 1,8,2,8 2,8,4,8 2,4,8,8 2,4,4,16 2,8,8,4 2,2,8,16 4,8,8,8 4,4,8,16 4,8,4,16 4,4,4,32 4,2,8,32 3,4,6,16 3,6,4,16 3,4,12,8 3,2,12,16; do run "$c" $p; done
done
