#!/usr/bin/env python3
"""Main-queue timeline of one profiled step (rocprofv3 kernel trace): python profiles/probes/tail_of_step.py <run_kernel_trace.csv> [first|last N]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_adam4' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0, q_main = int(rows[a]['End_Timestamp']), rows[b]['Queue_Id']
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
out, prev = [], t0
for r in rows[a + 1:b + 1]:
    if r['Queue_Id'] != q_main:
        continue
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'\(.*', '', re.sub(r'\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1\d+', '', r['Kernel_Name']))[:44]
    out.append(((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, name))
    prev = e
print('%d main-queue kernels, span %.1f us, kernel time %.1f us, gaps %.1f us' % (len(out), (int(rows[b]['End_Timestamp']) - t0) / 1e3, sum(o[1] for o in out), sum(max(0, o[2]) for o in out)))
for o in (out[:n] + [None] + out[-n:]):
    print('   ...' if o is None else '%8.1f %7.1f gap %6.1f  %s' % o)
