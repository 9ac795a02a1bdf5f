#!/usr/bin/env python
"""Where a launch-bound pass spends its wall time: from a rocprofv3 --kernel-trace CSV, the steady-state steps of a
bench.py row as a timeline -- per HIP queue the busy time, the union of both queues, the idle gaps, and the kernels
on either side of the largest gaps.
usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d OUT -- python bench.py --workload backbone ...
                 python tools/timeline_gaps.py OUT [steps]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d, steps=10):
    f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0'), r['Kernel_Name'])
                 for r in rows), key=lambda t: t[0])
    # the timed region: the last `steps` occurrences of the step's first kernel
    marker = os.environ.get('TIMELINE_MARKER')   # (a kernel that runs once per step; default: the sweep's first launch)
    first = [i for i, k in enumerate(ks) if (marker in k[3] if marker else ('sweep_conv_kernel' in k[3] or 'camera_prepare' in k[3]))]
    if len(first) < steps + 1:
        print('not enough steps in the trace', len(first))
        return
    i0, i1 = first[-steps - 1], first[-1]
    seg = ks[i0:i1]
    t0, t1 = seg[0][0], ks[i1][0]
    print(f'{steps} steps, {len(seg)} kernels, {(t1 - t0) / steps / 1e3:.1f} us per step wall (first kernel to first kernel)')
    by_q = defaultdict(list)
    for s, e, q, n in seg:
        by_q[q].append((s, e, n))
    for q, v in by_q.items():
        busy = sum(e - s for s, e, _ in v)
        print(f'  queue {q}: {len(v) / steps:.1f} kernels / step, busy {busy / steps / 1e3:.1f} us / step')
    # union of busy intervals
    iv = sorted((s, e) for s, e, _, _ in seg)
    union, cs, ce = 0, iv[0][0], iv[0][1]
    gaps = []
    for s, e in iv[1:]:
        if s > ce:
            union += ce - cs
            gaps.append((s - ce, ce))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    union += ce - cs
    print(f'  any queue busy: {union / steps / 1e3:.1f} us / step; idle (no kernel on the device): '
          f'{(t1 - t0 - union) / steps / 1e3:.1f} us / step in {len(gaps) / steps:.1f} gaps / step')
    both = 0
    qs = list(by_q)
    if len(qs) >= 2:
        a, b = sorted(by_q[qs[0]]), sorted(by_q[qs[1]])
        j = 0
        for s, e, _ in a:
            while j < len(b) and b[j][1] <= s:
                j += 1
            k = j
            while k < len(b) and b[k][0] < e:
                both += min(e, b[k][1]) - max(s, b[k][0])
                k += 1
        print(f'  both queues busy at once: {both / steps / 1e3:.1f} us / step')
    # per kernel name: mean duration when it runs alone vs overlapped is beyond a CSV; list the step's kernels instead
    agg = defaultdict(lambda: [0, 0])
    for s, e, q, n in seg:
        agg[n[:70]][0] += 1
        agg[n[:70]][1] += e - s
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f'    {c / steps:5.1f} x {t / c / 1e3:7.1f} us = {t / steps / 1e3:7.1f} us / step  {n}')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10)
