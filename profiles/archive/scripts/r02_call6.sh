#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c6; mkdir -p $O
export DFM_ONLY=bf16
timeout 900 env DFM_MIOPEN_FIND=1 python tools/backbone_timing.py 2>&1 | grep -v "MIOpen" > $O/backbone_find_mfma.txt; cat $O/backbone_find_mfma.txt
timeout 900 env DFM_MIOPEN_FIND=1 DFM_NO_MFMA_CONV=1 python tools/backbone_timing.py 2>&1 | grep -v "MIOpen" > $O/backbone_find_nomfma.txt; cat $O/backbone_find_nomfma.txt
(cd /tmp && timeout 900 env DFM_MIOPEN_FIND=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt6 -- python $GRAFT_REPO_ROOT/tools/backbone_timing.py > /dev/null 2>&1)
python - <<'PY' > gpurun_out/c6/backbone_kernel_stats.txt
import csv,glob
f=glob.glob('/tmp/kt6/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:40]:
    print(f"{r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:10.3f} ms {float(r['AverageNs'])/1e3:10.1f} us {float(r['Percentage']):6.2f}%  {r['Name'][:150]}")
PY
cat $O/backbone_kernel_stats.txt
timeout 300 python -m pytest tests/test_conv3d_gpu.py -x -q 2>&1 | tail -3
