#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json (per-launch HBM
bytes of the dominant kernel), following MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are
in KiB-units of the L2's memory-side requests, collected in SEPARATE passes; on gfx950 FETCH_SIZE
under-counts wide coalesced streaming reads by exactly 2x, so the read side is doubled (upper bound
for this kernel, whose reads are 16-32 B/lane gathers)."""
import csv
import glob
import json
import os
import sys


def mean_counter(d, counter, needle):
    vals = []
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and needle in r['Kernel_Name']:
                vals.append(float(r['Counter_Value']))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    out_dir, key = sys.argv[1], sys.argv[2]
    needle = sys.argv[3] if len(sys.argv) > 3 else 'k_pool_fwd_dense2'
    fetch, nf = mean_counter(os.path.join(out_dir, 'prof_fetch'), 'FETCH_SIZE', needle)
    write, nw = mean_counter(os.path.join(out_dir, 'prof_write'), 'WRITE_SIZE', needle)
    rec = {'kernel': needle, 'FETCH_SIZE_KiB_mean': fetch, 'WRITE_SIZE_KiB_mean': write, 'launches': [nf, nw]}
    if fetch is not None and write is not None:
        rec['fetch_bytes_raw'] = fetch * 1024
        rec['fetch_bytes_corrected_x2'] = 2 * fetch * 1024
        rec['write_bytes'] = write * 1024
        rec['hbm_bytes_per_launch'] = 2 * fetch * 1024 + write * 1024
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'pmc_traffic.json')
    allrec = json.load(open(path)) if os.path.exists(path) else {}
    allrec[key] = rec
    json.dump(allrec, open(os.path.join(out_dir, 'pmc_traffic.json'), 'w'), indent=1)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
