#!/usr/bin/env python3
"""Register / scratch / LDS budget of every gfx950 kernel in libfbbev_hip.so, read from the code object's metadata
(no GPU, no recompile): objcopy the .hip_fatbin section, unbundle the gfx950 code object, parse `llvm-readelf --notes`.

    python tools/kernel_resources.py [out.json]      -> {kernel: {vgpr, agpr, sgpr, lds, scratch, vgpr_spills, sgpr_spills}}

A guard for the CPU suite (tests/test_kernel_resources.py): no kernel spills to scratch, and the hot kernels keep the
register budgets their occupancy was tuned for.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get('FBBEV_LLVM_BIN', '/opt/rocm/lib/llvm/bin')


def kernel_resources(lib=None):
    lib = lib or os.path.join(ROOT, 'fb_bev_amd', 'libfbbev_hip.so')
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, 'fat.bin'), os.path.join(tmp, 'k.co')
        subprocess.check_call(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', lib, fat])
        # one offload bundle per translation unit of the library (capi.hip, capi_train.hip), back to back in the section
        blob, magic = open(fat, 'rb').read(), b'__CLANG_OFFLOAD_BUNDLE__'
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        notes = ''
        for k, st in enumerate(starts):
            part = os.path.join(tmp, f'fat{k}.bin')
            with open(part, 'wb') as f:
                f.write(blob[st:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + part,
                                   '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co])
            notes += subprocess.check_output([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], text=True) + '\n'
    keys = {'.vgpr_count': 'vgpr', '.agpr_count': 'agpr', '.sgpr_count': 'sgpr', '.group_segment_fixed_size': 'lds',
            '.private_segment_fixed_size': 'scratch', '.vgpr_spill_count': 'vgpr_spills', '.sgpr_spill_count': 'sgpr_spills',
            '.max_flat_workgroup_size': 'max_wg'}
    out, cur = {}, None
    for line in notes.splitlines():
        line = line.strip()
        if line.startswith('- '):                                   # next kernel entry of amdhsa.kernels
            if cur and 'name' in cur:
                out[cur.pop('name')] = cur
            cur = {}
            line = line[2:].strip()
        if cur is None:
            continue
        m = re.match(r'(\.\w+):\s+(\S+)', line)
        if not m:
            continue
        if m.group(1) == '.name':
            cur['name'] = m.group(2)
        elif m.group(1) in keys:
            cur[keys[m.group(1)]] = int(m.group(2))
    if cur and 'name' in cur:
        out[cur.pop('name')] = cur
    return {k: v for k, v in out.items() if 'vgpr' in v}


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1].startswith('-'):
        raise SystemExit(f'{sys.argv[1]!r} looks like an option; this tool takes one optional OUTPUT PATH (a past `--all` / `--help` call '
                         'left files of those names in the repo root)')
    res = kernel_resources()
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], 'w'), indent=0, sort_keys=True)
    worst = sorted(res.items(), key=lambda kv: -(kv[1]['vgpr'] + kv[1].get('agpr', 0)))[:12]
    for name, r in worst:
        print(f"{r['vgpr'] + r.get('agpr', 0):4d} regs  lds {r.get('lds', 0):6d}  scratch {r.get('scratch', 0)}  {name[:90]}")
    print(len(res), 'kernels;', sum(1 for r in res.values() if r.get('scratch', 0) or r.get('vgpr_spills', 0)), 'with scratch / spills')
