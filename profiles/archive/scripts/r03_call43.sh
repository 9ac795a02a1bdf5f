#!/bin/bash
# round 3, call 43: what takes 2.1 s in a DfMStereoPath training step at config K?  kernel statistics
O=gpurun_out/r03c43; mkdir -p $O
R=$PWD
python - <<'PY'
import re
p='tools/stereo_train_timing.py'
s=open(p).read()
s=s.replace("    for fuse in (False, True):","    for fuse in (True,):")
open('/tmp/stt.py','w').write(s.replace("ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))","ROOT = '%s'" % __import__('os').getcwd()))
PY
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_st -- python /tmp/stt.py --dtype bf16 --iters 1 > /tmp/st.log 2>&1)
tail -2 /tmp/st.log
python - > $O/stereo_train_kernel_stats.txt <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_st/**/*kernel_stats.csv', recursive=True)
print('# rocprofv3 --kernel-trace --stats -- python tools/stereo_train_timing.py --dtype bf16 --iters 1 (fused head only; 3 steps)')
for r in list(csv.DictReader(open(f[0])))[:25]:
    print(f"{int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.3f} ms {float(r['AverageNs'])/1e3:11.1f} us {float(r['Percentage']):6.2f}%  {r['Name'][:140]}")
PY
head -16 $O/stereo_train_kernel_stats.txt | cut -c1-220
