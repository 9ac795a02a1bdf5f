# round 6, call 7: DfMBackbone.forward as a timeline (two HIP streams): busy / overlapped / idle time per step
mkdir -p gpurun_out/c7; cd /tmp; export TMPDIR=/tmp
for mode in two one; do
  rm -rf /tmp/tl_$mode
  if [ $mode = one ]; then export DFM_BACKBONE_ONE_STREAM=1; else unset DFM_BACKBONE_ONE_STREAM; fi
  DFM_FEATS_NHWC=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$mode -- python $GRAFT_REPO_ROOT/bench.py --workload backbone --steps 12 --warmup 3 > /tmp/tl_$mode.log 2>&1
  echo "== $mode stream(s): $(grep '^{' /tmp/tl_$mode.log | tail -1 | python -c 'import sys,json; l=json.loads(sys.stdin.read()); print(l["ms_per_step"])') ms per step under the tracer"
  python $GRAFT_REPO_ROOT/tools/timeline_gaps.py /tmp/tl_$mode 10
done > $GRAFT_REPO_ROOT/gpurun_out/c7/timeline.txt 2>&1
cd $GRAFT_REPO_ROOT
