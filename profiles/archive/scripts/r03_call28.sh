#!/bin/bash
# round 3, call 28: matrix-product backward, tile pairs per workgroup + chunked footprint tables
O=gpurun_out/r03c28; mkdir -p $O
timeout 900 python -m pytest tests/test_sweep_bwd_mfma_gpu.py -q 2>&1 | grep -v "^$" | grep -n "^E  \|passed\|failed\|^FAILED" | cut -c1-300 > $O/tests.txt
head -30 $O/tests.txt
R=$PWD
D=DFM_HIP_LIB=$R/depth-from-motion_amd/lib/libdfm_hip_dbg.so
env $D timeout 300 python tools/sweep_bwd_trace.py > $O/trace.txt 2>&1
cat $O/trace.txt
stats() {  # name, env...
  name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && export TMPDIR=/tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python $R/bench.py --workload sweep_bwd --steps 4 --warmup 1 > /tmp/bench_$name.txt 2>&1)
  python - "$name" >> $O/kernel_ms.txt <<'PY'
import csv, glob, sys
name = sys.argv[1]
f = glob.glob(f'/tmp/prof_{name}/**/*kernel_stats.csv', recursive=True)
out = [name]
if f:
    for r in csv.DictReader(open(f[0])):
        n = r['Name']
        if 'sweep_bwd' in n:
            tag = 'mfma_cur' if 'mfma_kernel<0>' in n else 'mfma_prev' if 'mfma_kernel<1>' in n else 'tile_cur' if ', 0>' in n else 'tile_prev'
            out.append(f"{tag} {float(r['AverageNs'])/1e6:.3f} ms")
print('  '.join(out))
PY
}
rm -f $O/kernel_ms.txt
stats release A=1
stats dbg $D
stats noload $D DFM_BWD_ABLATE=16
stats noproduce $D DFM_BWD_ABLATE=128
stats noload_noproduce $D DFM_BWD_ABLATE=144
cat $O/kernel_ms.txt
