#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` run into a small markdown table.
usage: python profiles/summarize.py <..._kernel_stats.csv> [steps] > profiles/rNN_xxx.md"""
import collections
import csv
import re
import sys


def short(n):
    m = re.match(r'_ZN12_GLOBAL__N_1\d+(k_\w+?)I(.*?)EEv', n)        # names the profiler left mangled (_Float16 args)
    if m:
        args = re.findall(r'Li(\d+)E|Lb([01])E|(DF16_|DF16b|f)', m.group(2))
        pretty = [a or ('true' if b == '1' else 'false' if b else {'DF16_': '_Float16', 'DF16b': '__bf16', 'f': 'float'}[c])
                  for a, b, c in args]
        return '%s<%s>' % (m.group(1), ', '.join(pretty))
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    if n.startswith('Cijk_'):
        m = re.search(r'(Cijk_\w+?)_S_B.*?(MT\d+x\d+x\d+)', n)
        return 'hipBLASLt ' + (m.group(1) + ' ' + m.group(2) if m else n[:40])
    if 'rocprim' in n:
        m = re.search(r'detail::(\w+)', n)
        return 'rocprim ' + (re.sub(r'.*wrapped_(\w+?)_config.*', r'\1', n)[:40] if 'wrapped_' in n else (m.group(1) if m else ''))
    n = re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', n)
    n = n.replace('at::native::', 'aten ')
    return n[:100]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    tot = sum(int(r['TotalDurationNs']) for r in rows)
    cat = collections.OrderedDict()
    for r in rows:
        c = cat.setdefault(short(r['Name']), [0, 0])
        c[0] += int(r['Calls'])
        c[1] += int(r['TotalDurationNs'])
    print('total GPU kernel time: %.2f ms over %d kernel launches%s\n' % (
        tot / 1e6, sum(v[0] for v in cat.values()), (' (%d profiled steps incl. warm-up)' % steps) if steps else ''))
    groups = collections.OrderedDict([('stin (hand-written HIP)', 0), ('hipBLASLt GEMM', 0), ('rocprim (plan sort)', 0), ('aten elementwise/other', 0)])
    for k, v in cat.items():
        g = 'hipBLASLt GEMM' if k.startswith('hipBLASLt') else 'rocprim (plan sort)' if k.startswith('rocprim') else \
            'stin (hand-written HIP)' if k.startswith('k_') else 'aten elementwise/other'
        groups[g] += v[1]
    print('| group | ms | % |\n|---|---|---|')
    for g, t in groups.items():
        print('| %s | %.2f | %.1f |' % (g, t / 1e6, 100.0 * t / tot))
    print('\n| % | total ms | calls | avg us | kernel |\n|---|---|---|---|---|')
    for k, v in sorted(cat.items(), key=lambda kv: -kv[1][1])[:40]:
        print('| %.2f | %.2f | %d | %.1f | `%s` |' % (100.0 * v[1] / tot, v[1] / 1e6, v[0], v[1] / v[0] / 1e3, k))


if __name__ == '__main__':
    main()
