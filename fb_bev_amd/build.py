"""Build libfbbev_hip.so (gfx950) in-tree with hipcc.  `python -m fb_bev_amd.build [--force]`.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the .so is
git-ignored but travels to the GPU box with the working-tree snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libfbbev_hip.so')
SOURCES = ['capi.hip', 'capi_train.hip']      # two translation units: the training-path entries recompile in seconds
def _headers():
    """every header the translation unit can include: a header-only edit must trigger a rebuild (a hard-coded list
    here once went stale and a measurement was taken on an old binary)"""
    import glob
    return sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(CSRC, 'hip_rt', 'rt.h'),
                                                          os.path.join(HERE, '..', 'include', 'fbbev.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics',
         '-fno-gpu-rdc', '-Wall', '-Wno-unused-function',
         '-I' + os.path.join(CSRC, 'hip_rt')]


def hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (needed to build libfbbev_hip.so for gfx950)')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link the shared library."""
    hdrs = _headers()
    objdir = os.path.join(HERE, 'csrc', '_obj')
    os.makedirs(objdir, exist_ok=True)
    objs = []
    cc = None
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cc = cc or hipcc()
            cmd = [cc] + FLAGS + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            subprocess.check_call(cmd, cwd=CSRC)
    if force or _stale(LIB, objs):
        cc = cc or hipcc()
        cmd = [cc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
