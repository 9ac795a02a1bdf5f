#!/usr/bin/env python
"""HBM traffic per launch of the plane-sweep forward from rocprofv3 PMC passes.

For every launch configuration given (the autotuner's candidates by default) this runs TWO
rocprofv3 passes -- ``--pmc FETCH_SIZE`` and ``--pmc WRITE_SIZE`` (they do not fit one pass on
gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots"; no trace domains beside --pmc) -- over the
torch-free harness ``tools/sweep_bench`` and writes

    {workload: {schedule_key: {"fetch_kb": .., "write_kb": .., "hbm_bytes_per_launch": ..,
                               "tile_kernel_bytes": .., "kernel_ms": ..}}}

hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over the kernels of one launch (pack,
tile, spill, patch): FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (the guide's
HBM section), WRITE_SIZE is taken as is.  bench.py reads the file (profiles/archive/r02_nstar_traffic.json)
to fill ``roofline.traffic`` for the configuration that actually ran.

usage (GPU box): python tools/pmc_traffic.py --out gpurun_out/r02_nstar_traffic.json [--workload nstar]
                 [cfg ...]       cfg as for tools/sweep_bench, e.g. lanes=512,ppl=4,chunk=15
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_CFGS = ['chunk=1', 'chunk=15', 'lanes=512,ppl=4,chunk=1', 'lanes=512,ppl=4,chunk=15']


def key_of(cfg):
    kv = dict(item.split('=') for item in cfg.split(',')) if cfg != 'default' else {}
    return 'lanes{}_ppl{}_planes{}_chunk{}'.format(kv.get('lanes', 256), kv.get('ppl', 8),
                                                   kv.get('planes', 2), kv.get('chunk', 1))


def is_library_kernel(name):
    """the plane sweep's own kernels (tools/sweep_bench also fills and checksums the volume)"""
    return any(t in name for t in ('sweep_', 'pack_blocked', 'pack_pixel_major', 'gather_fit'))


def one_pass(counter, cfg, workload, scratch, timeout, mode='fwd'):
    d = os.path.join(scratch, f'{counter}_{mode}_{key_of(cfg)}')
    shutil.rmtree(d, ignore_errors=True)
    cmd = ['rocprofv3', '--pmc', counter, '--output-format', 'csv', '-d', d, '--',
           os.path.join(ROOT, 'tools', 'sweep_bench'), '--workload', workload, '--mode', mode, '--rounds', '1',
           '--launches', '2', cfg]
    subprocess.run(cmd, check=True, timeout=timeout, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR='/tmp'), cwd='/tmp')
    per_kernel = defaultdict(list)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] == counter:
                per_kernel[row['Kernel_Name']].append(float(row['Counter_Value']))
    if mode != 'fwd':
        # a step of these rows is several kernels, some launched more than once: total per kernel name over the
        # 3 steps the harness ran (1 warm-up + 1 round x 2 launches), per step
        return {k: sum(v) / 3.0 for k, v in per_kernel.items()}
    return {k: sum(v) / len(v) for k, v in per_kernel.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--workload', default='nstar')
    ap.add_argument('--scratch', default='/tmp/pmc_traffic')
    ap.add_argument('--timeout', type=int, default=150)
    ap.add_argument('cfgs', nargs='*')
    args = ap.parse_args()
    cfgs = args.cfgs or DEFAULT_CFGS
    try:
        with open(args.out) as f:
            result = json.load(f)
    except (OSError, ValueError):
        result = {}
    res_w = result.setdefault(args.workload, {})
    for cfg in cfgs:
        try:
            fetch = one_pass('FETCH_SIZE', cfg, args.workload, args.scratch, args.timeout)
            write = one_pass('WRITE_SIZE', cfg, args.workload, args.scratch, args.timeout)
        except (subprocess.SubprocessError, OSError) as e:  # a pass that aborts must not lose the others
            print(f'{cfg}: pass failed: {e}', file=sys.stderr)
            continue
        fetch = {k: v for k, v in fetch.items() if is_library_kernel(k)}
        write = {k: v for k, v in write.items() if is_library_kernel(k)}
        kernels = sorted(set(fetch) | set(write))
        total = sum(2 * fetch.get(k, 0.0) + write.get(k, 0.0) for k in kernels) * 1024
        tile_bytes = sum(2 * fetch.get(k, 0.0) + write.get(k, 0.0) for k in kernels
                         if 'sweep_tile_kernel' in k) * 1024
        res_w[key_of(cfg)] = {
            'cfg': cfg,
            'fetch_kb': {k[:60]: round(v, 1) for k, v in fetch.items()},
            'write_kb': {k[:60]: round(v, 1) for k, v in write.items()},
            'tile_kernel_bytes': tile_bytes,
            'hbm_bytes_per_launch': total,
            'formula': '(2*FETCH_SIZE + WRITE_SIZE)*1024 over pack + tile + spill + patch kernels',
        }
        print(f'{key_of(cfg):40s} hbm {total / 1e9:8.3f} GB/launch  (tile kernel {tile_bytes / 1e9:.3f} GB)',
              flush=True)
        with open(args.out, 'w') as f:
            json.dump(result, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
