#!/bin/bash
# usage: ab_sweep_bwd.sh <out dir> <lib suffix> [<lib suffix> ...]   ("" = the shipped library); alternating runs
OUT=$1; shift; mkdir -p $OUT
L=depth-from-motion_amd/lib
for rep in 1 2; do
for v in "$@"; do
  [ "$v" = "-" ] && v=""
  echo "== lib$v" >> $OUT/ab.txt
  DFM_HIP_LIB=$PWD/$L/libdfm_hip$v.so timeout 120 python bench.py --workload ${WL:-sweep_bwd} --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('roofline',{}).get('frac'))" >> $OUT/ab.txt 2>&1
done; done
cat $OUT/ab.txt
