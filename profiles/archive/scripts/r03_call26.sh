#!/bin/bash
# round 3, call 26: is the matrix-product backward memory-bound?  loads / producer / MFMA switched off
O=gpurun_out/r03c26; mkdir -p $O
R=$PWD
D=DFM_HIP_LIB=$R/depth-from-motion_amd/lib/libdfm_hip_dbg.so
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
T="DFM_BWD_THR_X=1e9 DFM_BWD_THR_Y=1e9"
stats all $D $T
stats noload $D $T DFM_BWD_ABLATE=16
stats noproduce $D $T DFM_BWD_ABLATE=128
stats noload_noproduce $D $T DFM_BWD_ABLATE=144
stats noload_noproduce_nomfma $D $T DFM_BWD_ABLATE=176
stats nomfma $D $T DFM_BWD_ABLATE=32
cat $O/kernel_ms.txt
