#!/usr/bin/env python3
"""Loops that are chains of single memory round trips: a backward branch whose body holds global loads AND an `s_waitcnt vmcnt(0)`
(every iteration waits for its own loads before the next iteration's are issued).  Round 5 found the row kernels' weight staging,
their epilogues' parameter loads, the DA sampler's phase A and the Z-mean's staging in this form (profiles/r05_exp_weight_staging.md):
at two workgroups per CU such a loop costs one full round trip per iteration.  Reads the gfx950 code object in libfbbev_hip.so (no GPU).

    python tools/isa_chains.py                  every kernel: (instructions in the loop body, global loads, LDS writes) per such loop
    python tools/isa_chains.py <substr> ...     only kernels whose mangled name contains every substring
    --max-body N                                loops of at most N instructions (default 80: staging / epilogue loops, not main loops)"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_waits as IW


def chains(lib=None, max_body=80):
    txt = IW.disassemble(lib)
    kern, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r'^([0-9a-f]+) <(.*)>:', line)
        if m:
            cur = m.group(2); kern[cur] = []; continue
        if cur is None:
            continue
        m = re.match(r'\s+(\S+)\s+(.*?)\s*//\s*([0-9A-F]+):', line)
        if m:
            kern[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    out = {}
    for k, ins in kern.items():
        addr = {a: i for i, (a, _, _) in enumerate(ins)}
        found = []
        for i, (a, op, args) in enumerate(ins):
            if not (op.startswith('s_cbranch') or op == 's_branch'):
                continue
            try:
                off = int(args.split()[-1])
            except ValueError:
                continue
            if off > 32767:
                off -= 65536
            tgt = a + 4 + off * 4
            if tgt < a and tgt in addr:
                body = ins[addr[tgt]:i + 1]
                loads = sum(o.startswith(('global_load', 'buffer_load', 'flat_load')) for _, o, _ in body)
                wait0 = any(o == 's_waitcnt' and 'vmcnt(0)' in ar for _, o, ar in body)
                lds_w = sum(o.startswith('ds_write') for _, o, _ in body)
                if loads and wait0 and len(body) <= max_body:
                    found.append((len(body), loads, lds_w))
        if found:
            out[k] = sorted(set(found))
    return out


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    mb = 80
    if '--max-body' in sys.argv:
        mb = int(sys.argv[sys.argv.index('--max-body') + 1]); args = [a for a in args if a != str(mb)]
    for k, v in sorted(chains(max_body=mb).items()):
        if all(a in k for a in args):
            print(k[:100], v)
