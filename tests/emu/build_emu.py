"""Build tests/emu/_build/libfbbev_emu.so: fb_bev_amd/csrc/capi.hip compiled as plain C++ against
tests/emu/rt.h (CPU fiber emulation of the device runtime).  Test infrastructure only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'fb_bev_amd', 'csrc')
OUT = os.path.join(HERE, '_build', 'libfbbev_emu.so')


def build():
    if os.environ.get('FBBEV_EMU_LIB'):             # e.g. an AddressSanitizer build (tools/emu_asan.sh)
        return os.environ['FBBEV_EMU_LIB']
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.h', '.hip'))]
    deps += [os.path.join(HERE, 'rt.h'), os.path.join(ROOT, 'include', 'fbbev.h')]
    if os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ['g++', '-O1', '-g', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-x', 'c++',
           '-I', HERE, os.path.join(CSRC, 'capi.hip'), '-o', OUT]
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build())
