"""Build-time check of the machine code around ASYNCHRONOUS loads issued from inline asm.

Two idioms in csrc/ write registers behind the compiler's back:

  * the depth-walking sweep (plane_sweep_cl.hip: sweep_cltw_kernel) keeps its taps in registers WRITTEN BY
    MASKED BUFFER LOADS (`s_and_saveexec / buffer_load_dwordx4 / s_mov exec`: only the lanes whose footprint
    moved load); the data is there after the counted `s_waitcnt vmcnt(4)` at the top of the next plane;
  * the MFMA convolutions, the LDS tile sweep and the fused sweep + conv kernel read their LDS fragments with
    `ds_read_b128` from inline asm (hipcc's own LDS reads would wait for the LDS-DMA of the NEXT block with
    vmcnt(0) first) and wait with counted `s_waitcnt lgkmcnt(N)` statements, one round behind.

hipcc believes the value exists when the asm statement ends.  Between such a load and the wait that covers it
it must therefore not read, copy, spill or re-allocate the destination registers -- and nothing in the source
can force that.  Round 4's "two taps per trip" variant of conv3d_g_kernel is the worked example (DESIGN.md
6c): its two loop exits met in front of the last tap, hipcc resolved the phi of the in-flight fragment with
`v_mov_b64 v[120:121], v[84:85]` / `v[122:123], v[86:87]` placed BEFORE the `s_waitcnt lgkmcnt`, and the MFMAs
read the copy: stale whenever the LDS had not answered yet -- run-dependent results.  This checker finds that
copy in the disassembly (tests/test_async_asm_check.py keeps the variant as a negative).

The kernels are bit-exact with the compiler this repository is developed with; the library is compiled on
the user's machine by whatever hipcc is installed there, so build.py runs this check on every object it
produces.  A failing sweep_cltw_kernel is compiled out (-DDFM_WALK_UNVERIFIED: the per-plane kernel takes its
calls, same bits); any other failure fails the build -- loudly, instead of a library that is sometimes wrong.

The checks, on `llvm-objdump -d` of the gfx950 code object, per kernel:
  * masked loads (sweep_cltw_kernel only): no scratch traffic; along EVERY control-flow path from a masked
    `buffer_load_dwordx4`, no instruction names a destination register -- except another masked load into the
    same registers -- until an `s_waitcnt vmcnt(N)` with at least N vector-memory operations issued after the
    load on that path (vmcnt counts loads and stores in order on gfx9), or the wave ends;
  * LDS reads (every kernel): along every path from a `ds_read_*`, no instruction names a destination register
    until an `s_waitcnt lgkmcnt(N)` with at least N LDS operations issued after the read on that path (LDS
    operations return in order; scalar loads share the counter but may return early, so they are not counted)
    -- compiler-generated reads pass trivially, their wait follows.

`check(disassembly_text)` / `check_lds(...)` return findings (empty = verified); `check_object(path)` runs
both over every kernel of a hipcc -c object.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get('LLVM_OBJDUMP', '/opt/rocm/lib/llvm/bin/llvm-objdump')
KERNEL = 'sweep_cltw_kernel'

_INS = re.compile(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):')
_VREG = re.compile(r'\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]')
_VMEM = re.compile(r'^(buffer_|global_|flat_|scratch_)(load|store|atomic)')


def _regs(text):
    """vector / accumulator registers named in an operand string, as ('v', 12) ..."""
    out = set()
    for m in _VREG.finditer(text):
        if m.group(1) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), k) for k in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def _fmt(regs):
    return ','.join('%s%d' % r for r in sorted(regs))


def parse_kernel(disassembly, kernel=KERNEL):
    """[(addr, mnemonic, operands)] of the kernel's body, or None if the symbol is absent"""
    lines = disassembly.splitlines()
    start = None
    for i, l in enumerate(lines):
        if re.match(r'^[0-9a-fA-F]+ <.*' + re.escape(kernel) + r'.*>:\s*$', l):
            start = i + 1
            break
    if start is None:
        return None
    ins = []
    for l in lines[start:]:
        if re.match(r'^[0-9a-fA-F]+ <', l):
            break
        m = _INS.match(l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return ins


def _branch_target(mn, ops, addr):
    """absolute address a branch goes to (simm16 is in dwords, relative to the next instruction)"""
    m = re.match(r'^(-?\d+)', ops.strip())
    if not m:
        return None
    off = int(m.group(1))
    if off >= 32768:
        off -= 65536
    return addr + 4 + 4 * off


def check(disassembly, kernel=KERNEL, max_states=2_000_000):
    ins = parse_kernel(disassembly, kernel)
    if ins is None:
        return ['kernel %s not found in the disassembly' % kernel]
    findings = []
    index = {a: i for i, (a, _, _) in enumerate(ins)}
    if any(mn.startswith('scratch_') for _, mn, _ in ins):
        findings.append('scratch instructions in the kernel: a tap register may have been spilled')
    # masked loads: buffer_load_dwordx4 directly after s_and_saveexec_b64 and before s_mov_b64 exec
    masked = []
    for i, (a, mn, ops) in enumerate(ins):
        if mn == 'buffer_load_dwordx4' and i > 0 and ins[i - 1][1] == 's_and_saveexec_b64' and \
                i + 1 < len(ins) and ins[i + 1][1] == 's_mov_b64' and ins[i + 1][2].startswith('exec'):
            dst = _regs(ops.split(',')[0])
            masked.append((i, frozenset(dst)))
    if not masked:
        findings.append('no masked tap loads found: the kernel is not the one this check was written for')
        return findings
    masked_at = dict(masked)
    states = 0
    for i0, dst in masked:
        # depth-first over (instruction, vector-memory operations issued since the load), capped
        seen = set()
        stack = [(i0 + 1, 0)]
        bad = None
        while stack and bad is None:
            i, after = stack.pop()
            while True:
                states += 1
                if states > max_states:
                    return findings + ['path exploration exceeded its budget']
                if i >= len(ins) or (i, after) in seen:
                    break
                seen.add((i, after))
                a, mn, ops = ins[i]
                if mn == 's_endpgm':
                    break
                if mn == 's_waitcnt':
                    m = re.search(r'vmcnt\((\d+)\)', ops)
                    if m and after >= int(m.group(1)):
                        break  # the load has returned on this path
                    i += 1
                    continue
                if mn in ('s_branch',) or mn.startswith('s_cbranch'):
                    t = _branch_target(mn, ops, a)
                    ti = index.get(t)
                    if ti is None:
                        bad = 'branch at %#x leaves the kernel' % a
                        break
                    if mn == 's_branch':
                        i = ti
                    else:
                        stack.append((ti, after))
                        i += 1
                    continue
                touched = _regs(ops) & dst
                if touched:
                    if i in masked_at and masked_at[i] == dst:
                        break  # the same tap is loaded again (its own check starts there)
                    bad = '%s %s at %#x names %s while the masked load at %#x into it is in flight' % (
                        mn, ops, a, _fmt(touched), ins[i0][0])
                    break
                if _VMEM.match(mn):
                    after = min(after + 1, 64)
                i += 1
        if bad:
            findings.append(bad)
    return findings


def kernels(disassembly):
    """{symbol: [(addr, mnemonic, operands)]} of every function in the disassembly"""
    out, name = {}, None
    for l in disassembly.splitlines():
        m = re.match(r'^[0-9a-fA-F]+ <(.*)>:\s*$', l)
        if m:
            name = m.group(1)
            out[name] = []
            continue
        m = _INS.match(l)
        if m and name is not None:
            out[name].append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def check_lds(ins, max_states=20_000_000):
    """findings for the LDS reads of one kernel (see the module docstring)"""
    index = {a: i for i, (a, _, _) in enumerate(ins)}
    findings, states = [], 0
    for i0, (a0, mn0, ops0) in enumerate(ins):
        if not mn0.startswith('ds_read'):
            continue
        dst = frozenset(_regs(ops0.split(',')[0]))
        seen, stack, bad = set(), [(i0 + 1, 0)], None
        while stack and bad is None:
            i, after = stack.pop()
            while True:
                states += 1
                if states > max_states:
                    return findings + ['path exploration exceeded its budget']
                if i >= len(ins) or (i, after) in seen:
                    break
                seen.add((i, after))
                a, mn, ops = ins[i]
                if mn == 's_endpgm':
                    break
                if mn == 's_waitcnt':
                    m = re.search(r'lgkmcnt\((\d+)\)', ops)
                    if m and after >= int(m.group(1)):
                        break  # the read has returned on this path
                    i += 1
                    continue
                if mn == 's_branch' or mn.startswith('s_cbranch'):
                    ti = index.get(_branch_target(mn, ops, a))
                    if ti is None:
                        bad = 'branch at %#x leaves the kernel' % a
                        break
                    if mn == 's_branch':
                        i = ti
                    else:
                        stack.append((ti, after))
                        i += 1
                    continue
                touched = _regs(ops) & dst
                if touched:
                    bad = '%s %s at %#x names %s while the %s at %#x into it is in flight' % (
                        mn, ops, a, _fmt(touched), mn0, a0)
                    break
                if mn.startswith('ds_'):
                    after = min(after + 1, 64)
                i += 1
        if bad:
            findings.append(bad)
    return findings


def disassemble_object(obj):
    """gfx950 disassembly of a `hipcc -c` object (the code object is unbundled into a scratch directory)"""
    tmp = tempfile.mkdtemp(prefix='dfm_walkchk_')
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, '--offloading', local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       check=True, cwd=tmp)
        cos = [f for f in os.listdir(tmp) if 'amdgcn' in f]
        if not cos:
            raise RuntimeError('no amdgcn code object in ' + obj)
        return subprocess.run([OBJDUMP, '-d', os.path.join(tmp, cos[0])], stdout=subprocess.PIPE, check=True,
                              text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def check_object(obj):
    """the masked-load check of sweep_cltw_kernel (what build.py's -DDFM_WALK_UNVERIFIED fallback hangs on)"""
    try:
        return check(disassemble_object(obj))
    except (OSError, subprocess.SubprocessError, RuntimeError) as e:
        return ['could not disassemble %s: %r' % (obj, e)]


def check_object_lds(obj, text=None):
    """{kernel: findings} of the LDS-read check over every kernel of the object (only kernels with findings)"""
    try:
        text = text or disassemble_object(obj)
    except (OSError, subprocess.SubprocessError, RuntimeError) as e:
        return {'?': ['could not disassemble %s: %r' % (obj, e)]}
    out = {}
    for name, ins in kernels(text).items():
        f = check_lds(ins)
        if f:
            out[name] = f
    return out


if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, '..', 'depth-from-motion_amd', 'lib', 'obj',
                                                             'plane_sweep_cl.o')
    if 'plane_sweep_cl' in os.path.basename(obj):
        res = check_object(obj)
        print('\n'.join(res) if res else 'sweep_cltw_kernel: verified (no scratch, no tap register named between a '
                                        'masked load and the wait that covers it)')
    else:
        res = []
    lds = check_object_lds(obj)
    for k, f in lds.items():
        print(k[:100] + ':')
        print('\n'.join('    ' + x for x in f[:8]))
    if not lds:
        print('LDS reads: verified in every kernel of', os.path.basename(obj))
    sys.exit(1 if (res or lds) else 0)
