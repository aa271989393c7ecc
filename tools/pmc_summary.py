#!/usr/bin/env python3
"""Per-kernel means of every counter found under a directory of rocprofv3 --pmc passes -> one JSON (launch-averaged).
    python tools/pmc_summary.py DIR OUT.json [name filter ...]"""
import collections, csv, glob, json, os, sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    keep = sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]
            if keep and not any(k in name for k in keep):
                continue
            agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    res = {}
    for k, v in agg.items():
        rec = {c: round(sum(x) / len(x), 1) for c, x in v.items()}
        rec['launches'] = max(len(x) for x in v.values())
        w = rec.get('SQ_WAVE_CYCLES')
        if w:
            for a, b in (('SQ_WAIT_ANY', 'frac_parked_waitcnt_barrier'), ('SQ_WAIT_INST_ANY', 'frac_issue_stall'),
                         ('SQ_ACTIVE_INST_ANY', 'frac_issuing')):
                if a in rec:
                    rec[b] = round(rec[a] / w, 3)
        if 'TCC_HIT_sum' in rec and 'TCC_MISS_sum' in rec and rec['TCC_HIT_sum'] + rec['TCC_MISS_sum'] > 0:
            rec['L2_hit_rate'] = round(rec['TCC_HIT_sum'] / (rec['TCC_HIT_sum'] + rec['TCC_MISS_sum']), 4)
        if 'TCP_TOTAL_CACHE_ACCESSES_sum' in rec and 'TCP_TCC_READ_REQ_sum' in rec and rec['TCP_TOTAL_CACHE_ACCESSES_sum'] > 0:
            rec['L1_requests_forwarded_to_L2_frac'] = round((rec['TCP_TCC_READ_REQ_sum'] + rec.get('TCP_TCC_WRITE_REQ_sum', 0)) /
                                                            rec['TCP_TOTAL_CACHE_ACCESSES_sum'], 4)
        if 'TCC_EA0_RDREQ_sum' in rec:
            rec['L2_fabric_read_bytes_64B_per_req'] = rec['TCC_EA0_RDREQ_sum'] * 64
        if 'FETCH_SIZE' in rec:
            rec['FETCH_SIZE_bytes_raw_KiB_units'] = rec['FETCH_SIZE'] * 1024
        if 'WRITE_SIZE' in rec:
            rec['WRITE_SIZE_bytes'] = rec['WRITE_SIZE'] * 1024
        res[k] = rec
    json.dump(res, open(out, 'w'), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:12]:
        print(k[:50], {a: v[a] for a in ('launches', 'frac_parked_waitcnt_barrier', 'frac_issue_stall', 'frac_issuing', 'L2_hit_rate',
                                         'L1_requests_forwarded_to_L2_frac', 'L2_fabric_read_bytes_64B_per_req', 'WRITE_SIZE_bytes') if a in v})


if __name__ == '__main__':
    main()
