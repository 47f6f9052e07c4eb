#!/usr/bin/env python3
"""usage: pmc_gemm_summarize.py <pass1 counter_collection.csv> <pass2 counter_collection.csv> > profiles/rNN_pmc_gemm.md"""
import collections
import csv
import re
import sys


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name']
        m = re.match(r'_ZN12_GLOBAL__N_1\d+(k_\w+?)I(.*?)EEv', name)
        if m:
            args = re.findall(r'Li(\d+)E|Lb([01])E|(DF16_|DF16b|f)', m.group(2))
            name = '%s<%s>' % (m.group(1), ', '.join(a or ('true' if b == '1' else 'false' if b else
                                                            {'DF16_': '_Float16', 'DF16b': '__bf16', 'f': 'float'}[c]) for a, b, c in args))
        name = re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', name))
        name = re.sub(r'\(.*', '', name)
        if 'gemm' not in name:
            continue
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[name].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return acc, dur


def main():
    a1, d1 = load(sys.argv[1])
    a2, _ = load(sys.argv[2])
    mean = lambda v: sum(v) / len(v) if v else float('nan')
    print('| kernel | us (profiled) | MFMA busy cycles | CU busy cycles | MFMA utilisation | VALU insts | LDS insts | wave cycles waiting % |')
    print('|---|---|---|---|---|---|---|---|')
    for k in sorted(a1):
        c1, c2 = a1[k], a2.get(k, {})
        busy, cu = mean(c1.get('SQ_VALU_MFMA_BUSY_CYCLES', [])), mean(c1.get('SQ_BUSY_CU_CYCLES', []))
        wait, wave = mean(c2.get('SQ_WAIT_ANY', [])), mean(c2.get('SQ_WAVE_CYCLES', []))
        # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD (4 per CU); SQ_BUSY_CU_CYCLES per CU
        util = busy / (4.0 * cu) if cu else float('nan')
        print('| `%s` | %.1f | %.3g | %.3g | %.0f %% | %.3g | %.3g | %.0f |' % (
            k, mean(d1[k]), busy, cu, 100 * util, mean(c2.get('SQ_INSTS_VALU', [])), mean(c2.get('SQ_INSTS_LDS', [])),
            100 * wait / wave if wave else float('nan')))


if __name__ == '__main__':
    main()
