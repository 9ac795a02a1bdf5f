"""Registers written behind the compiler's back: the depth-walking sweep's masked buffer loads and the
inline-asm `ds_read_b128` fragments of the MFMA / tile kernels.  build.py checks in the machine code of
every build that hipcc left those registers alone between a load and the wait that covers it
(tools/verify_async_asm.py).  Here: the shipped build verifies, and the checker catches what it is there to
catch -- including the register copy that made round 4's two-taps conv loop run-dependent."""
import importlib
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _checker():
    spec = importlib.util.spec_from_file_location('verify_async_asm', os.path.join(ROOT, 'tools', 'verify_async_asm.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _disassembly():
    build = importlib.import_module('depth-from-motion_amd.build')
    build.build_hip()
    return _checker().disassemble_object(os.path.join(build.LIB_DIR, 'obj', 'plane_sweep_cl.o')), build


def test_shipped_walking_kernel_passes_its_disassembly_check():
    text, build = _disassembly()
    chk = _checker()
    assert chk.check(text) == []
    assert build.walk_kernel_check() == 'verified'
    ins = chk.parse_kernel(text)
    # the kernel this check was written for: 2 maps x (16 first-plane + 16 per-plane) masked tap loads
    masked = [i for i, (a, mn, ops) in enumerate(ins) if mn == 'buffer_load_dwordx4' and ins[i - 1][1] == 's_and_saveexec_b64']
    assert len(masked) == 64


def test_checker_flags_a_copy_a_spill_and_a_missing_wait():
    text, _ = _disassembly()
    chk = _checker()
    lines = text.splitlines()
    # the first per-plane masked load of the kernel (the 17th masked load: the first 16 are the first plane's,
    # followed by a vmcnt(0))
    start = next(i for i, l in enumerate(lines) if re.match(r'^[0-9a-f]+ <.*sweep_cltw_kernel.*>:', l))
    loads = [i for i in range(start, len(lines)) if 'buffer_load_dwordx4' in lines[i] and 's_and_saveexec_b64' in lines[i - 1]]
    at = loads[16]
    dst = re.search(r'buffer_load_dwordx4 v\[(\d+):(\d+)\]', lines[at])
    lo = int(dst.group(1))
    addr = int(re.search(r'//\s*([0-9A-Fa-f]+):', lines[at + 1]).group(1), 16)
    # (1) a register copy of a tap between the load and the wait -- what a different allocation could insert
    copy = '\tv_mov_b32_e32 v200, v%d                                      // %012X: 00000000' % (lo, addr)
    bad = lines[:at + 2] + [copy] + lines[at + 2:]
    # (the inserted line re-uses an address: branch targets elsewhere are unaffected)
    found = chk.check('\n'.join(bad))
    assert found and 'v_mov_b32_e32' in found[0] and 'in flight' in found[0]
    # (2) scratch traffic anywhere in the kernel
    spill = '\tscratch_store_dwordx4 off, v[%d:%d], s33                     // %012X: 00000000' % (lo, lo + 3, addr)
    found = chk.check('\n'.join(lines[:at + 2] + [spill] + lines[at + 2:]))
    assert any('scratch' in f for f in found)
    # (3) the counted wait at the top of the plane loop weakened to vmcnt(20): the loads are not covered
    weak = [l.replace('s_waitcnt vmcnt(4)', 's_waitcnt vmcnt(20)') for l in lines]
    assert chk.check('\n'.join(weak))
    # a kernel that is not there
    assert chk.check(text, kernel='no_such_kernel')


def test_every_shipped_object_passes_the_lds_read_check():
    import glob
    build = importlib.import_module('depth-from-motion_amd.build')
    build.build_hip()
    chk = _checker()
    objs = sorted(glob.glob(os.path.join(build.LIB_DIR, 'obj', '*.o')))
    assert len(objs) >= 13
    for obj in objs:
        assert chk.check_object_lds(obj) == {}, obj


def test_lds_check_flags_the_copy_that_broke_the_two_taps_conv_loop():
    """Round 4's `step(wa, wb); step(wb, wa)` variant of conv3d_g_kernel gave run-dependent results: its two
    loop exits met in front of the last tap and hipcc resolved the phi of the IN-FLIGHT activation fragment
    with a register copy placed before the `s_waitcnt lgkmcnt` (v_mov_b64 v[120:121], v[84:85]; the MFMAs then
    read v[120:123]).  Re-create that instruction in the shipped kernel's disassembly: the checker must object."""
    build = importlib.import_module('depth-from-motion_amd.build')
    build.build_hip()
    chk = _checker()
    text = chk.disassemble_object(os.path.join(build.LIB_DIR, 'obj', 'conv3d_g.o'))
    ks = chk.kernels(text)
    name = next(k for k in ks if 'conv3d_g_kernelILi2ELi1ELb0' in k)
    ins = list(ks[name])
    assert chk.check_lds(ins) == []
    i = next(i for i, (a, mn, ops) in enumerate(ins) if mn == 'ds_read_b128')
    lo = int(re.search(r'v\[(\d+):', ins[i][2]).group(1))
    ins.insert(i + 1, (ins[i][0], 'v_mov_b64_e32', 'v[200:201], v[%d:%d]' % (lo, lo + 1)))
    found = chk.check_lds(ins)
    assert found and 'v_mov_b64_e32' in found[0] and 'in flight' in found[0]
