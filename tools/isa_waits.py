#!/usr/bin/env python3
"""Memory-operation / wait skeleton of a compiled gfx950 kernel, read from the code object inside libfbbev_hip.so (no GPU):
runs of global loads / stores, LDS reads / writes, MFMAs and VALU between the s_waitcnt, s_barrier and branch instructions.
What it is for: an `s_waitcnt vmcnt(0)` right behind a prefetch, a load that the compiler sank under a branch, a burst that
an in-order counter makes an earlier consumer wait for -- the things that cost the history kernels 30 %.
    python tools/isa_waits.py                       one line per kernel: instructions, vmcnt(0) waits
    python tools/isa_waits.py <substr> [<substr>..]  skeletons of the kernels whose mangled name contains every substring
                                                     (at most 4; --all lifts the cap)
    --lib PATH                                       another HIP binary instead of fb_bev_amd/libfbbev_hip.so"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'


def disassemble(lib=None):
    lib = lib or os.path.join(ROOT, 'fb_bev_amd', 'libfbbev_hip.so')
    d = tempfile.mkdtemp(prefix='fbbev_isa_')
    fat, obj = os.path.join(d, 'fat.bin'), os.path.join(d, 'code.o')
    subprocess.check_call(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', lib, fat])
    subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + fat,
                           '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + obj], stderr=subprocess.DEVNULL)
    return subprocess.check_output([os.path.join(LLVM, 'llvm-objdump'), '-d', obj], text=True)


def skeleton(body):
    out, prev = [], None
    for line in body:
        m = re.match(r'\s+(\w+)\s+(.*?)\s*//', line)
        if not m:
            continue
        op, args = m.group(1), m.group(2)
        if op.startswith('v_mfma'): key = 'mfma'
        elif op.startswith(('global_load', 'buffer_load', 'flat_load')): key = 'gload'
        elif op.startswith(('global_store', 'buffer_store', 'flat_store')): key = 'gstore'
        elif op.startswith(('global_atomic', 'buffer_atomic', 'flat_atomic')): key = 'gatomic'
        elif op.startswith('ds_read') or op.startswith('ds_load'): key = 'lds_rd'
        elif op.startswith('ds_write') or op.startswith('ds_store'): key = 'lds_wr'
        elif op.startswith('ds_'): key = 'lds_atomic'
        elif op.startswith('s_load') or op.startswith('s_buffer_load'): key = 'sload'
        elif op == 's_waitcnt': key = 'WAIT ' + args
        elif op.startswith('s_cbranch') or op in ('s_barrier', 's_endpgm', 's_branch'): key = op
        elif op.startswith('v_'): key = 'valu'
        else: continue
        if key == prev and not key.startswith(('WAIT', 's_')):
            out[-1][1] += 1
        else:
            out.append([key, 1])
        prev = key
    return ' | '.join(f'{k} x{n}' if n > 1 else k for k, n in out)


def main():
    args = sys.argv[1:]
    lib = None
    if '--lib' in args:                                     # any HIP binary with a .hip_fatbin section (e.g. tools/micro/*)
        i = args.index('--lib')
        lib = args[i + 1]
        del args[i:i + 2]
    pats = [a for a in args if not a.startswith('--')]
    asm = disassemble(lib)
    shown = 0
    for m in re.finditer(r'^[0-9a-f]+ <(\S+)>:\n(.*?)(?=\n\n|\Z)', asm, re.S | re.M):
        name, body = m.group(1), m.group(2).splitlines()
        if pats and not all(p in name for p in pats):
            continue
        sk = skeleton(body)
        print(f'== {name}  ({len(body)} instructions; vmcnt(0) waits: {sk.count("WAIT vmcnt(0)")})')
        if pats and (shown < 4 or '--all' in sys.argv):
            print(sk)
            shown += 1


if __name__ == '__main__':
    main()
