"""GPU timeline of a rocprofv3 --kernel-trace csv: busy time (union of kernel intervals over all queues), idle time
between kernels, and which kernels follow the idle gaps - for the last `--steps` Adam launches (one per training step).
usage: python profiles/gaps.py <kernel_trace.csv> [--steps 8]"""
import argparse
import collections
import csv
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--marker', default='k_adam')   # (matches k_adam and k_adam4: one launch per step)
    ap.add_argument('--context', type=float, default=0.0, help='print the kernels around every idle gap longer than this many us (last step)')
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.trace)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    assert len(marks) > a.steps, 'not enough %s launches (%d)' % (a.marker, len(marks))
    lo, hi = marks[-a.steps - 1] + 1, marks[-1] + 1
    seg = rows[lo:hi]
    span = seg[-1][1] - seg[0][0]
    busy, gaps, cur_end = 0, collections.Counter(), seg[0][0]
    gap_n = collections.Counter()
    ksum = 0
    for s, e, name, q in seg:
        ksum += e - s
        if s > cur_end:
            short = re.sub(r'^void |\(anonymous namespace\)::|at::native::', '', name)
            short = re.sub(r'\(.*', '', short)[:60]
            gaps[short] += s - cur_end
            gap_n[short] += 1
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
    n = a.steps
    print('steps %d: span %.3f ms/step, GPU busy (union) %.3f ms/step, idle %.3f ms/step, sum of kernel durations %.3f ms/step, '
          'overlapped %.3f ms/step, launches %.0f/step' % (n, span / n / 1e6, busy / n / 1e6, (span - busy) / n / 1e6, ksum / n / 1e6,
                                                        (ksum - busy) / n / 1e6, len(seg) / n))
    print('idle time by the kernel that follows the gap (ms/step, gaps/step, avg us):')
    for k, v in gaps.most_common(14):
        print('  %-62s %.3f  %5.1f  %.1f' % (k, v / n / 1e6, gap_n[k] / n, v / gap_n[k] / 1e3))
    if a.context > 0:
        context(rows, marks[-2] + 1, marks[-1] + 1, a.context)


def context(rows, lo, hi, min_us):
    def nm(n):
        n = re.sub(r'^void |\(anonymous namespace\)::|at::native::', '', n)
        return re.sub(r'\(.*', '', n)[:48]
    seg = rows[lo:hi]
    t0, cur_end = seg[0][0], seg[0][0]
    for i, (s, e, name, q) in enumerate(seg):
        if s > cur_end and (s - cur_end) / 1e3 >= min_us:
            print('--- gap %.1f us at +%.3f ms' % ((s - cur_end) / 1e3, (cur_end - t0) / 1e6))
            for j in range(max(0, i - 4), min(len(seg), i + 3)):
                ss, ee, n2, q2 = seg[j]
                print('   %s +%.3f ms  %7.1f us  q%-3s %s' % ('>>' if j == i else '  ', (ss - t0) / 1e6, (ee - ss) / 1e3, q2, nm(n2)))
        cur_end = max(cur_end, e)


if __name__ == '__main__':
    main()
