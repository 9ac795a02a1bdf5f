#!/usr/bin/env python
"""Per-kernel mean of rocprofv3 --pmc counters from *_counter_collection.csv files.
usage: python tools/pmc_summary.py dir_or_csv [...] [--kernel substr]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(argv):
    kern = None
    paths = []
    it = iter(argv)
    for a in it:
        if a == '--kernel':
            kern = next(it)
        else:
            paths.append(a)
    files = []
    for p in paths:
        files += glob.glob(os.path.join(p, '**', '*counter_collection.csv'), recursive=True) \
            if os.path.isdir(p) else [p]
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(files):
        for row in csv.DictReader(open(f)):
            name = row['Kernel_Name']
            if kern and kern not in name:
                continue
            acc[name[:70]][row['Counter_Name']].append(float(row['Counter_Value']))
    for name, ctrs in acc.items():
        print(f'## {name}')
        for c, vals in sorted(ctrs.items()):
            # one row per dispatch (per dimension instance rows are summed by dispatch upstream)
            print(f'  {c:28s} n={len(vals):4d} mean={sum(vals) / len(vals):18.1f} sum={sum(vals):20.1f}')


if __name__ == '__main__':
    main(sys.argv[1:])
