"""What the SMI tools say about the part a process landed on (bench.py's ``part.smi``).

The same binary runs the N* forward at 0.63 of the HBM roofline on some MI355X parts and at 0.48 on
others (profiles/archive/r03_c53_*); the store and clock probes of the library bracket the difference, they
do not name it.  This records what can be read without privileges -- VRAM vendor, VBIOS part number,
memory / fabric / shader clocks UNDER LOAD, socket power and cap, partition modes, throttle state --
so that every bench line says which kind of part produced it and slow parts can be told apart by
something other than their speed.

``collect(load)``: ``load`` (optional) is a callable that keeps the GPU busy for about a second; the
clocks and the power are sampled while it runs (idle clocks say nothing).  Every field is best effort:
a missing tool or an unparsable section leaves the field out; nothing here can fail a bench run.
"""
import json
import shutil
import subprocess
import sys
import threading
import time


def _run(cmd, timeout=20):
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout)
        return r.stdout.decode('utf-8', 'replace')
    except Exception:  # tool missing, timed out, not permitted
        return ''


def parse_indented(text):
    """amd-smi's default text output ("KEY: value" lines, nesting by indentation) -> nested dict"""
    root = {}
    stack = [(-1, root)]
    for raw in text.splitlines():
        if not raw.strip() or ':' not in raw:
            continue
        indent = len(raw) - len(raw.lstrip())
        key, _, val = raw.strip().partition(':')
        key, val = key.strip(), val.strip()
        while stack and stack[-1][0] >= indent:
            stack.pop()
        if not stack:
            stack = [(-1, root)]
        parent = stack[-1][1]
        if val == '':
            node = {}
            parent[key] = node
            stack.append((indent, node))
        else:
            parent[key] = val
    return root


def _find(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _mhz(node):
    """{'CLK': '2000 MHz', ...} or '2000 MHz' -> 2000"""
    v = node.get('CLK') if isinstance(node, dict) else node
    try:
        return int(str(v).split()[0])
    except Exception:
        return None


def static_info():
    out = {}
    if not shutil.which('amd-smi'):
        return out
    d = parse_indented(_run(['amd-smi', 'static', '-g', '0']))
    g = d.get('GPU', d) if isinstance(d.get('GPU'), dict) else d
    for name, path in (('market_name', ('ASIC', 'MARKET_NAME')), ('vbios_part_number', ('VBIOS', 'PART_NUMBER')),
                       ('vbios_version', ('VBIOS', 'VERSION')), ('vram_vendor', ('VRAM', 'VENDOR')),
                       ('vram_type', ('VRAM', 'TYPE')), ('vram_max_bandwidth', ('VRAM', 'MAX_BANDWIDTH')),
                       ('socket_power_limit', ('LIMIT', 'PPT0', 'SOCKET_POWER_LIMIT')),
                       ('max_power_limit', ('LIMIT', 'PPT0', 'MAX_POWER_LIMIT')),
                       ('compute_partition', ('PARTITION', 'COMPUTE_PARTITION')),
                       ('memory_partition', ('PARTITION', 'MEMORY_PARTITION'))):
        v = _find(g, *path)
        if v is None:  # layouts differ between amd-smi versions: search by the last key
            v = _search(g, path[-1])
        if v is not None:
            out[name] = v
    return out


def _search(d, key):
    if isinstance(d, dict):
        if key in d and not isinstance(d[key], dict):
            return d[key]
        for v in d.values():
            r = _search(v, key)
            if r is not None:
                return r
    return None


def metric_sample():
    """one sample of clocks / power / temperatures / throttle state"""
    out = {}
    if not shutil.which('amd-smi'):
        return out
    d = parse_indented(_run(['amd-smi', 'metric', '-g', '0']))
    g = d.get('GPU', d) if isinstance(d.get('GPU'), dict) else d
    clk = g.get('CLOCK', {}) if isinstance(g.get('CLOCK'), dict) else {}
    gfx = [_mhz(v) for k, v in clk.items() if k.startswith('GFX')]
    gfx = [v for v in gfx if v]
    if gfx:
        out['gfx_clk_mhz_min_max'] = [min(gfx), max(gfx)]
    for name, key in (('mem_clk_mhz', 'MEM_0'), ('fclk_mhz', 'FCLK_0'), ('socclk_mhz', 'SOCCLK_0')):
        if key in clk:
            out[name] = _mhz(clk[key])
    for name, path in (('socket_power', ('POWER', 'SOCKET_POWER')), ('hotspot_temp', ('TEMPERATURE', 'HOTSPOT')),
                       ('mem_temp', ('TEMPERATURE', 'MEM')), ('umc_activity', ('USAGE', 'UMC_ACTIVITY')),
                       ('gfx_activity', ('USAGE', 'GFX_ACTIVITY')),
                       ('ppt_violation', ('THROTTLE', 'PPT_VIOLATION_STATUS')),
                       ('hbm_thermal_violation', ('THROTTLE', 'HBM_THERMAL_VIOLATION_STATUS')),
                       ('socket_thermal_violation', ('THROTTLE', 'SOCKET_THERMAL_VIOLATION_STATUS')),
                       ('ppt_accumulated', ('THROTTLE', 'PPT_ACCUMULATED')),
                       ('perf_level', ('PERF_LEVEL',))):
        v = _find(g, *path)
        if v is not None:
            out[name] = v
    return out


def partitions_rocm_smi():
    out = {}
    if not shutil.which('rocm-smi'):
        return out
    txt = _run(['rocm-smi', '--showcomputepartition', '--showmemorypartition', '--showmaxpower'])
    for line in txt.splitlines():
        for name, tag in (('compute_partition', 'Compute Partition:'), ('memory_partition', 'Memory Partition:'),
                          ('max_package_power_w', 'Max Graphics Package Power (W):')):
            if tag in line:
                out[name] = line.split(tag)[1].strip()
    return out


def collect(load=None, samples=3):
    info = {}
    try:
        info.update(partitions_rocm_smi())
        info.update(static_info())
        under_load = []
        if load is not None:
            th = threading.Thread(target=load)
            th.start()
            # the load may spend seconds allocating before its first launch: sample until it ends and
            # keep the busiest samples (idle clocks say nothing)
            while th.is_alive() and len(under_load) < 24:
                s = metric_sample()
                if s:
                    under_load.append(s)
            th.join()
        if under_load:
            def busy(s):
                c = s.get('gfx_clk_mhz_min_max') or [0, 0]
                return c[1]
            under_load.sort(key=busy)
            info['under_load'] = dict(under_load[-1])
            info['under_load']['gfx_clk_mhz_samples'] = [s.get('gfx_clk_mhz_min_max') for s in under_load[-samples:]]
            info['under_load']['fclk_mhz_samples'] = sorted({s.get('fclk_mhz') for s in under_load if s.get('fclk_mhz')})
            info['under_load']['mem_clk_mhz_samples'] = sorted({s.get('mem_clk_mhz') for s in under_load if s.get('mem_clk_mhz')})
        else:
            info['idle'] = metric_sample()
    except Exception as e:  # never fail the caller
        info['error'] = repr(e)
    return info


if __name__ == '__main__':
    load = None
    if len(sys.argv) > 1 and sys.argv[1] == '--load':
        cmd = sys.argv[2:]
        load = lambda: subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)  # noqa: E731
    print(json.dumps(collect(load), indent=1))
