#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from the passes of tools/pmc_mfma.sh.
    python tools/pmc_mfma_summary.py DIR OUT.json [TAIL_FRACTION]
TAIL_FRACTION (default 0.5): only the last fraction of each pass's dispatches counts -- the first calls of a vendor-library
convolution run MIOpen's solver search (every candidate once, the naive reference kernels included), which is not the steady state.
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs): busy cycles are per SIMD, summed over the
chip (16 per 16x16x32 bf16 instruction: SQ_INSTS_MFMA x 16 == the counter on k_history_conv_bf16x3); GRBM_GUI_ACTIVE comes back
SUMMED over the 8 XCDs on this stack (82.2 M for a 4.83 ms kernel = 8 x 2.13 GHz), hence the / 8.  Launch-averaged;
share_of_gpu_time from the kernel trace of the same pass (End - Start)."""
import collections, csv, glob, json, os, sys

CUS, SIMDS, XCDS = 256, 4, 8


def short(name):
    return name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')[:90]


def main():
    d, out = sys.argv[1], sys.argv[2]
    tail = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        rows_ = list(csv.DictReader(open(f)))
        last = max(int(r['Dispatch_Id']) for r in rows_) if rows_ else 0
        for r in rows_:
            if int(r['Dispatch_Id']) >= last * (1.0 - tail):
                ctr[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    traces = sorted(glob.glob(os.path.join(d, 'p1', '**', '*kernel_trace.csv'), recursive=True)) or \
        sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True))[:1]
    for f in traces:
        rows_ = list(csv.DictReader(open(f)))
        for i, r in enumerate(rows_):
            if i >= len(rows_) * (1.0 - tail):
                dur[short(r['Kernel_Name'])].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    total = sum(sum(v) for v in dur.values()) or 1.0
    rows = []
    busy_all = act_all = 0.0
    for k, c in ctr.items():
        rec = {'kernel': k, 'launches': max(len(v) for v in c.values())}
        for name, vals in c.items():
            rec[name] = round(sum(vals) / len(vals), 1)
        if k in dur:
            rec['avg_us'] = round(sum(dur[k]) / len(dur[k]) / 1e3, 2)
            rec['share_of_gpu_time'] = round(sum(dur[k]) / total, 4)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c and sum(c['GRBM_GUI_ACTIVE']) > 0:
            busy, act = sum(c['SQ_VALU_MFMA_BUSY_CYCLES']), sum(c['GRBM_GUI_ACTIVE'])
            rec['mfma_util'] = round(busy / (act / XCDS * CUS * SIMDS), 4)
            busy_all += busy; act_all += act
        rows.append(rec)
    rows.sort(key=lambda r: -r.get('share_of_gpu_time', 0))
    res = {'dispatches_counted': f'last {tail:.0%} of each pass (steady state: after the vendor library\'s solver search)',
           'overall_mfma_util': round(busy_all / (act_all / XCDS * CUS * SIMDS), 4) if act_all else None,
           'formula': 'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs), summed over the counted launches of a kernel',
           'kernels': rows}
    json.dump(res, open(out, 'w'), indent=1)
    print('overall', res['overall_mfma_util'])
    for r in rows[:14]:
        print(f"{r['kernel'][:60]:60s} n={r['launches']:4d} share={r.get('share_of_gpu_time')} us={r.get('avg_us')} mfma_util={r.get('mfma_util')}")


if __name__ == '__main__':
    main()
