#!/usr/bin/env python
"""One steady-state step of a bench.py row as a list: start offset (us), duration (us), HIP queue, kernel -- from a
rocprofv3 --kernel-trace CSV (see tools/timeline_gaps.py for the totals).  usage: python tools/timeline_dump.py OUT [marker]"""
import csv
import glob
import os
import sys


def main(d, marker='sweep_conv_kernel'):
    f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0'), r['Kernel_Name'])
                 for r in rows), key=lambda t: t[0])
    first = [i for i, k in enumerate(ks) if marker in k[3]]
    i0, i1 = first[-2], first[-1]
    # a step starts with the 2-D necks, BEFORE the marker: back up to the previous step's last kernel + 1 by the
    # largest gap inside the window
    seg = ks[i0:i1]
    t0 = seg[0][0]
    for s, e, q, n in seg:
        short = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').replace('void at::native::', 'at::')
        print('%9.1f %8.1f  q%s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, short[:80]))


if __name__ == '__main__':
    main(*sys.argv[1:])
