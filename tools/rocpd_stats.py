#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace the way `--stats` CSV does:
per-kernel calls / total / average / min / max (ns) and percentage.
usage: python tools/rocpd_stats.py results.db > profiles/xxx_kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        'select name, count(*), sum(duration), avg(duration), min(duration), max(duration), '
        'max(vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) '
        'from kernels group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f'# source: {path}')
    print(f'{"calls":>6} {"total_ns":>14} {"avg_ns":>12} {"min_ns":>12} {"max_ns":>12} {"pct":>6} '
          f'{"vgpr":>5} {"lds":>7}  name')
    for name, n, tot, avg, mn, mx, vg, lds, gx, wx in rows:
        print(f'{n:6d} {tot:14d} {avg:12.0f} {mn:12d} {mx:12d} {100.0 * tot / total:6.2f} '
              f'{vg or 0:5d} {lds or 0:7d}  {name[:110]}')


if __name__ == '__main__':
    main(sys.argv[1])
