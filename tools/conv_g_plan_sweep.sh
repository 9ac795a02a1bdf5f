cd $GRAFT_REPO_ROOT
run() { DFM_CONV_G_PLAN=$2 python tools/conv_g_timing.py --no-miopen --graph --iters 10 --case "$1" 2>&1 | grep MFMA | cut -c1-150 | sed "s/^/plan=$2  /"; }
echo "== neck.res0 (220,300,12) 64->64"
python tools/conv_g_timing.py --no-miopen --graph --iters 10 --case "neck.res0" 2>&1 | grep MFMA | cut -c1-150
for p in 4,8,16,4 4,16,8,4 4,4,32,4 4,8,8,8 3,8,8,6 3,4,16,6 3,8,4,12 3,4,8,12 3,2,16,12 2,4,16,4 2,8,8,4 2,4,8,8 4,2,32,8 4,16,4,8 4,4,16,8; do run "neck.res0" $p; done
echo "== neck.res1 (220,300,6) 128->128"
python tools/conv_g_timing.py --no-miopen --graph --iters 10 --case "neck.res1" 2>&1 | grep MFMA | cut -c1-150
for p in 4,8,16,4 4,16,8,4 3,8,8,6 3,8,16,3 3,4,16,6 3,16,4,6 4,8,32,2 4,4,32,4 2,8,16,2 4,16,16,2 3,4,32,3 3,16,8,3; do run "neck.res1" $p; done
echo "== neck.res2 (220,300,3) 256->256"
python tools/conv_g_timing.py --no-miopen --graph --iters 10 --case "neck.res2" 2>&1 | grep MFMA | cut -c1-150
for p in 3,8,16,3 3,16,8,3 3,4,32,3 4,16,32,1 4,8,64,1 3,8,48,1 3,16,24,1 2,8,32,1 4,32,16,1 3,32,4,3; do run "neck.res2" $p; done
