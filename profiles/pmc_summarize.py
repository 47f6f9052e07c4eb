#!/usr/bin/env python3
"""usage: pmc_summarize.py <fetch counter_collection.csv> <write counter_collection.csv> [key prefix] > profiles/rNN_pmc_traffic.json
FABRIC bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB and count the L2's memory-side
requests - Infinity-Cache hits included, so this is traffic on the fabric, not HBM traffic proper; on gfx950 FETCH_SIZE reports
exactly half of a wide (16 B/lane) coalesced read stream (MI355X_MICROARCH.md §HBM).  (Round 4: the keys were called hbm_* before.)"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get('Counter_Name') != counter:
            continue
        name = r['Kernel_Name']
        m = re.match(r'_ZN12_GLOBAL__N_1\d+(k_\w+?)I(.*?)EEv', name)          # names the profiler left mangled (__bf16 / _Float16)
        if m:
            args = re.findall(r'Li(\d+)E|Lb([01])E|(DF16_|DF16b|f)', m.group(2))
            name = '%s<%s>' % (m.group(1), ', '.join(a or ('true' if b == '1' else 'false' if b else
                                                            {'DF16_': '_Float16', 'DF16b': '__bf16', 'f': 'float'}[c]) for a, b, c in args))
        name = re.sub(r'\(anonymous namespace\)::', '', name)
        name = re.sub(r'^void ', '', name)
        name = re.sub(r'\(.*', '', name)
        acc[name].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    f = per_kernel(sys.argv[1], 'FETCH_SIZE')
    w = per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {}
    prefix = sys.argv[3] if len(sys.argv) > 3 else ''
    for k in sorted(set(f) & set(w)):
        if not k.startswith('k_'):
            continue
        out[prefix + k] = {'FETCH_SIZE_KiB': f[k], 'WRITE_SIZE_KiB': w[k],
                           'fabric_read_MB_corrected': 2 * f[k] * 1024 / 1e6, 'fabric_write_MB': w[k] * 1024 / 1e6,
                           'fabric_MB_per_launch': (2 * f[k] + w[k]) * 1024 / 1e6}
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
