#!/bin/bash
# same-box A/B of build variants of sweep_bwd_mfma_kernel + translation / latency counters of the shipped one
OUT=gpurun_out/r05c23; mkdir -p $OUT
L=depth-from-motion_amd/lib
for rep in 1 2; do
for v in base stag pf2 pf3s; do
  echo "== $v" >> $OUT/ab.txt
  DFM_HIP_LIB=$PWD/$L/libdfm_hip_$v.so timeout 120 python bench.py --workload sweep_bwd --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('roofline',{}).get('frac'))" >> $OUT/ab.txt 2>&1
done; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$OUT/avail.txt 2>&1
grep -o -i "TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*LATENCY[A-Z0-9_]*\|UTCL[A-Z0-9_]*" $GRAFT_REPO_ROOT/$OUT/avail.txt | sort -u > $GRAFT_REPO_ROOT/$OUT/avail_tcp.txt
cd $GRAFT_REPO_ROOT
for set in "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST" "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES" "TCP_TA_TCP_STATE_READ TCP_GATE_EN1 TCP_TCP_TA_DATA_STALL_CYCLES" "TCP_UTCL1_PERMISSION_MISS TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_MULTI_MISS"; do
  tag=$(echo $set | tr ' ' '+' | cut -c1-60)
  (cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$tag -- python $GRAFT_REPO_ROOT/bench.py --workload sweep_bwd --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$OUT/pmc_$tag.log 2>&1)
done
python - <<'PY' > gpurun_out/r05c23/pmc_summary.txt 2>&1
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r05c23/pmc_*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
    print(f)
    for k,v in acc.items():
        if 'bwd_mfma' in k or 'sweep_bwd' in k:
            print('  ',k,{c:(x/ n[(k,c)]) for c,x in v.items()})
PY
