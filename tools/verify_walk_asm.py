"""Build-time check of sweep_cltw_kernel's machine code (csrc/plane_sweep_cl.hip).

The depth-walking sweep keeps its taps in registers that are WRITTEN BY ASYNCHRONOUS LOADS issued from
inline asm (`s_and_saveexec / buffer_load_dwordx4 / s_mov exec`: only the lanes whose footprint moved
load), and the data is only there after the counted `s_waitcnt vmcnt(4)` at the top of the next plane.
hipcc does not know that: between a load and the wait that covers it, it must not read, copy, spill or
re-allocate a tap register.  The kernel is bit-exact with the compiler this repository is developed
with; the library is compiled on the user's machine by whatever hipcc is installed there, so build.py
runs this check on the object it has just produced and, when it fails, compiles plane_sweep_cl.hip
again with -DDFM_WALK_UNVERIFIED (the per-plane kernel then takes every call: same bits, slower).

The check, on `llvm-objdump -d` of the gfx950 code object:
  * no scratch traffic in the kernel (no `scratch_` instruction: a spilled tap would be a stale tap);
  * for every masked load (a `buffer_load_dwordx4` between `s_and_saveexec_b64` and `s_mov_b64 exec`),
    along EVERY control-flow path from it, no instruction names one of its destination registers --
    except another masked load into the same registers -- until an `s_waitcnt vmcnt(N)` is reached with
    at least N vector-memory operations issued after the load on that path (vmcnt counts loads and
    stores in order on gfx9: the load has then returned), or the wave ends.

`check(disassembly_text)` returns a list of findings (empty = verified); `check_object(path)` extracts
the code object from a hipcc -c object first.
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
_VREG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')
_VMEM = re.compile(r'^(buffer_|global_|flat_|scratch_)(load|store|atomic)')


def _regs(text):
    out = set()
    for m in _VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


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
                    bad = '%s %s at %#x names v%s while the masked load at %#x into it is in flight' % (
                        mn, ops, a, sorted(touched), ins[i0][0])
                    break
                if _VMEM.match(mn):
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
    try:
        return check(disassemble_object(obj))
    except (OSError, subprocess.SubprocessError, RuntimeError) as e:
        return ['could not disassemble %s: %r' % (obj, e)]


if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, '..', 'depth-from-motion_amd', 'lib', 'obj',
                                                             'plane_sweep_cl.o')
    res = check_object(obj)
    print('\n'.join(res) if res else 'sweep_cltw_kernel: verified (no scratch, no tap register named between a '
                                    'masked load and the wait that covers it)')
    sys.exit(1 if res else 0)
