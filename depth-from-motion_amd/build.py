"""Ahead-of-time build of lib/libdfm_hip.so with hipcc for gfx950.

No JIT, no torch.utils.cpp_extension: the library is a plain C-ABI shared
object (include/dfm_hip.h) that ctypes loads; the built .so stays in-tree so
it travels to the GPU box with the repo snapshot.

Every csrc/*.hip is its own translation unit: objects are compiled in parallel
(only the stale ones unless ``force``) into lib/obj/ and linked into the .so.
"""
import contextlib
import fcntl
import glob
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, 'csrc')
LIB_DIR = os.path.join(_HERE, 'lib')
LIB = os.path.join(LIB_DIR, 'libdfm_hip.so')

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# -ffp-contract=off is part of the numerics contract (csrc/dfm_common.h)
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
         '-fvisibility=hidden', '-Wall', '-Wno-unused-function']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _headers():
    return glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(ROOT, 'include', '*.h'))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


@contextlib.contextmanager
def _build_lock():
    """one builder at a time per checkout: torchrun ranks / pytest-xdist workers that find a stale
    library wait for the first one instead of compiling and linking the same files concurrently"""
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, '.build.lock'), 'w') as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def _flags_tag(flags):
    """objects are keyed on what they were compiled WITH as well as when: changing FLAGS or HIPCC
    rebuilds them (object staleness used to be mtime-only)"""
    try:
        ver = subprocess.run([HIPCC, '--version'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                             timeout=30).stdout
    except (OSError, subprocess.SubprocessError):
        ver = b''
    return hashlib.sha1(' '.join(flags).encode() + b'|' + HIPCC.encode() + b'|' + ver).hexdigest()[:16]


def _checker():
    import importlib.util
    spec = importlib.util.spec_from_file_location('verify_async_asm', os.path.join(ROOT, 'tools', 'verify_async_asm.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _verify_walk(obj):
    return _checker().check_object(obj)


class AsyncAsmCheckError(RuntimeError):
    """this hipcc placed a register copy / use under an in-flight inline-asm LDS read: the kernel would be
    wrong whenever the LDS answers late (tools/verify_async_asm.py).  The build refuses the object."""


def _verify_lds(obj):
    """every kernel of the object: no register named between an inline-asm ds_read and the counted wait that
    covers it (the hazard that made round 4's two-taps conv loop run-dependent)"""
    bad = _checker().check_object_lds(obj)
    if bad:
        k, f = next(iter(bad.items()))
        raise AsyncAsmCheckError('%s: %d kernel(s) fail the LDS-read check, e.g. %s: %s' % (
            os.path.basename(obj), len(bad), k[:80], f[0][:200]))


def walk_kernel_check():
    """what the last build's disassembly check of sweep_cltw_kernel said: 'verified', or its findings"""
    try:
        return open(os.path.join(LIB_DIR, 'obj', 'walk_kernel_check.txt')).read().strip()
    except OSError:
        return None


def build_hip(force=False, verbose=False, debug_hooks=False, out=None, jobs=None):
    """Compile every csrc/*.hip into lib/libdfm_hip.so. Returns the path.

    debug_hooks=True adds -DDFM_DEBUG_HOOKS: the DFM_ABLATE switches and the
    per-phase s_memtime trace of the tile kernel (tools/trace_phases.py).  They
    are compiled out by default -- even never-taken runtime branches in the
    blend loop changed hipcc's schedule by up to 25 % (profiles/archive/r01_v8_*)."""
    out = out or LIB
    if not force and out == LIB and not _stale():
        return LIB
    with _build_lock():
        if not force and out == LIB and not _stale():  # another process built it while this one waited
            return LIB
        return _build_locked(force, verbose, debug_hooks, out, jobs)


def _build_locked(force, verbose, debug_hooks, out, jobs):
    obj_dir = os.path.join(LIB_DIR, 'obj_dbg' if debug_hooks else 'obj')
    os.makedirs(obj_dir, exist_ok=True)
    flags = FLAGS + (['-DDFM_DEBUG_HOOKS'] if debug_hooks else []) + \
        ['-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]
    tag = _flags_tag(flags)
    tag_file = os.path.join(obj_dir, '.flags')
    try:
        same_flags = open(tag_file).read().strip() == tag
    except OSError:
        same_flags = False
    newest_header = max(os.path.getmtime(h) for h in _headers())
    objs, todo = [], []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.splitext(os.path.basename(src))[0] + '.o')
        objs.append(obj)
        if force or not same_flags or not os.path.exists(obj) or \
                os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            todo.append((obj, [HIPCC] + flags + ['-c', src, '-o', obj + '.tmp']))

    def run(cmd):
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)

    def compile_one(item):
        obj, cmd = item
        run(cmd)
        if os.path.basename(obj) == 'plane_sweep_cl.o':
            # the depth-walking sweep's taps are written by asynchronous loads issued from inline asm: check
            # in the machine code that THIS hipcc left the tap registers alone between a load and its wait
            # (tools/verify_walk_asm.py); if not, build the file without that kernel (the per-plane kernel
            # takes its calls, same bits)
            findings = _verify_walk(obj + '.tmp')
            with open(os.path.join(obj_dir, 'walk_kernel_check.txt'), 'w') as f:
                f.write('\n'.join(findings) if findings else 'verified')
                f.write('\n')
            if findings:
                import warnings
                warnings.warn('sweep_cltw_kernel failed its disassembly check with this hipcc (%s ...): '
                              'building plane_sweep_cl.hip with -DDFM_WALK_UNVERIFIED' % findings[0][:120],
                              RuntimeWarning)
                run(cmd[:1] + ['-DDFM_WALK_UNVERIFIED'] + cmd[1:])
        _verify_lds(obj + '.tmp')
        os.replace(obj + '.tmp', obj)  # a reader never sees a half-written object

    with ThreadPoolExecutor(max_workers=jobs or min(len(todo) or 1, os.cpu_count() or 4)) as pool:
        list(pool.map(compile_one, todo))
    with open(tag_file, 'w') as f:
        f.write(tag + '\n')
    run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out + '.tmp'])
    os.replace(out + '.tmp', out)
    return out


def build_variant(tag, extra_flags, only):
    """lib/libdfm_hip_<tag>.so: the release objects with the translation units named in ``only``
    (file names under csrc/) recompiled with ``extra_flags`` -- experiments at release speed, e.g.
    ``build_variant('noload', ['-DDFM_BM_ABLATE=1'], ['plane_sweep_bwd_mfma.hip'])``."""
    build_hip()
    obj_dir = os.path.join(LIB_DIR, 'obj_' + tag)
    os.makedirs(obj_dir, exist_ok=True)
    flags = FLAGS + list(extra_flags) + ['-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]
    objs = []
    for src in sources():
        base = os.path.splitext(os.path.basename(src))[0] + '.o'
        if os.path.basename(src) in only:
            obj = os.path.join(obj_dir, base)
            subprocess.check_call([HIPCC] + flags + ['-c', src, '-o', obj])
        else:
            obj = os.path.join(LIB_DIR, 'obj', base)
        objs.append(obj)
    out = os.path.join(LIB_DIR, 'libdfm_hip_%s.so' % tag)
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out])
    return out


if __name__ == '__main__':
    print(build_hip(force=True, verbose=True))
