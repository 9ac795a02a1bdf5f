# round 6, call 5: full GPU suite; GroupNorm backward without zero fills, depth-pool kernel: stereo_train / backbone_train
mkdir -p gpurun_out/c5
(python -m pytest tests -x -q -m gpu 2>&1 | tail -6) > gpurun_out/c5/tests.txt
row() { python bench.py --workload $1 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['ms_per_step'], l['roofline']['frac'])"; }
for wl in stereo_train backbone_train stereo_infer backbone stereo_train backbone_train; do row $wl; done > gpurun_out/c5/rows.txt 2>&1
tools/kernel_stats.sh $GRAFT_REPO_ROOT/gpurun_out/c5/ks stereo_train:"--workload stereo_train --steps 5 --warmup 2" backbone_train:"--workload backbone_train --steps 5 --warmup 2"
