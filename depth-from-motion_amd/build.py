"""Ahead-of-time build of lib/libdfm_hip.so with hipcc for gfx950.

No JIT, no torch.utils.cpp_extension: the library is a plain C-ABI shared
object (include/dfm_hip.h) that ctypes loads; the built .so stays in-tree so
it travels to the GPU box with the repo snapshot.
"""
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, 'csrc')
LIB_DIR = os.path.join(_HERE, 'lib')
LIB = os.path.join(LIB_DIR, 'libdfm_hip.so')

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# -ffp-contract=off is part of the numerics contract (csrc/dfm_common.h)
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
         '-fvisibility=hidden', '-Wall', '-Wno-unused-function']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + \
        glob.glob(os.path.join(ROOT, 'include', '*.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False, debug_hooks=False, out=None):
    """Compile every csrc/*.hip into lib/libdfm_hip.so. Returns the path.

    debug_hooks=True adds -DDFM_DEBUG_HOOKS: the DFM_ABLATE switches and the
    per-phase s_memtime trace of the tile kernel (tools/trace_phases.py).  They
    are compiled out by default -- even never-taken runtime branches in the
    blend loop changed hipcc's schedule by up to 25 % (profiles/r01_v8_*)."""
    out = out or LIB
    if not force and out == LIB and not _stale():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [HIPCC] + FLAGS + (['-DDFM_DEBUG_HOOKS'] if debug_hooks else []) + \
        ['-I' + os.path.join(ROOT, 'include'), '-I' + CSRC] + sources() + ['-o', out]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == '__main__':
    print(build_hip(force=True, verbose=True))
