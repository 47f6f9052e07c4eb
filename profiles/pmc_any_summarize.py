#!/usr/bin/env python3
"""usage: pmc_any_summarize.py <counter_collection.csv> ... : mean of every counter per kernel (GEMM kernels), last 3 launches."""
import collections
import csv
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = re.sub(r'\(.*', '', re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', r['Kernel_Name'])))
        m = re.match(r'_ZN12_GLOBAL__N_1\d+(k_\w+?)I(.*?)EEv', name)
        if m:
            name = m.group(1) + '<' + m.group(2)[:28] + '>'
        if 'gemm' not in name:
            continue
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
        acc[name]['_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for name, cs in acc.items():
    print(name)
    for c, v in sorted(cs.items()):
        v = v[-3:]
        print('   %-32s %.4g' % (c, sum(v) / len(v)))
