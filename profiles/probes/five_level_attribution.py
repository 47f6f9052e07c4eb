#!/usr/bin/env python3
"""Round 6, review item 3: WHERE does the 5-level network's extra gradient distance come from?

tests/test_five_level.py measured the HIP path 1.40e-3 (relative L2 over all 67 M weight gradients) from an fp64 run of the
oracle where the fp32 CPU oracle is 0.93e-3.  This probe runs the same network / mesh / seeds with three matrix-core settings
(the precision switches are read at import, so each is a child process):

    shipped   STIN_GEMM_FWD=4 (fp16x3)  STIN_GEMM_BWD=2 (bf16x3)
    bwd_exact STIN_GEMM_FWD=4           STIN_GEMM_BWD=0 (v_mfma_f32_32x32x2_f32 on unsplit operands)
    fwd_x6    STIN_GEMM_FWD=3 (bf16x6)  STIN_GEMM_BWD=2
    fwd_exact STIN_GEMM_FWD=0           STIN_GEMM_BWD=2
    bwd_x6    STIN_GEMM_FWD=4           STIN_GEMM_BWD=3 (bf16x6: 24-bit products)
    exact     STIN_GEMM_FWD=0           STIN_GEMM_BWD=0

and prints, per setting: forward max-abs vs fp64, gradient rel-L2 vs fp64 (all tensors), the same restricted to the K >= 1024
layers (the 2048-wide bottleneck, reference models/surfacetextureinpaintingnet.py:316-338) and to everything else, and the number
of entries beyond 1e-3 of the gradient scale (the "decision flip" count of tests/_golden.grad_flip_report).

    python profiles/probes/five_level_attribution.py [--md out.md]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SETTINGS = [('shipped (fp16x3 fwd, bf16x3 bwd)', {}),
            ('bwd exact fp32', {'STIN_GEMM_BWD': '0'}),
            ('bwd bf16x6', {'STIN_GEMM_BWD': '3'}),
            ('fwd bf16x6 (24-bit products)', {'STIN_GEMM_FWD': '3'}),
            ('fwd exact fp32 only', {'STIN_GEMM_FWD': '0'}),
            ('fwd + bwd exact fp32', {'STIN_GEMM_FWD': '0', 'STIN_GEMM_BWD': '0'})]


def child():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch
    import test_five_level as T
    ref = T._oracle_run()[0]
    out64, g64 = T._truth64()
    r = T._hip_run(False)
    names = [k for k, _ in ref.named_parameters()]
    cpu = [p.grad for p in ref.parameters()]
    scale = max(float(t.abs().max()) for t in g64)

    def part(grads, pick):
        num = sum(float((g.double() - t).pow(2).sum()) for g, t, k in zip(grads, g64, names) if pick(t))
        den = sum(float(t.pow(2).sum()) for t, k in zip(g64, names) if pick(t))
        return (num / den) ** 0.5 if den > 0 else None

    wide = lambda t: t.dim() == 2 and t.shape[1] >= 1024          # noqa: E731 - weights whose products reduce over K >= 1024
    rest = lambda t: not wide(t)                                 # noqa: E731

    def beyond(grads):
        return sum(int(((g.double() - t).abs() > 1e-3 * scale).sum()) for g, t in zip(grads, g64))
    # per-tensor: the ten tensors that contribute most to the squared error
    contrib = sorted(((float((g.double() - t).pow(2).sum()), k, tuple(t.shape)) for g, t, k in zip(r['grads'], g64, names)), reverse=True)[:6]
    tot = sum(float((g.double() - t).pow(2).sum()) for g, t in zip(r['grads'], g64))
    print('RESULT ' + json.dumps({
        'fwd_max_vs_fp64': float((r['out'].double() - out64).abs().max()),
        'hip': {'all': T._rel_l2(r['grads'], g64), 'k_ge_1024': part(r['grads'], wide), 'rest': part(r['grads'], rest), 'beyond': beyond(r['grads'])},
        'cpu': {'all': T._rel_l2(cpu, g64), 'k_ge_1024': part(cpu, wide), 'rest': part(cpu, rest), 'beyond': beyond(cpu)},
        'total_entries': sum(t.numel() for t in g64),
        'top': [{'share': c / tot, 'name': k, 'shape': s} for c, k, s in contrib]}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--md')
    ap.add_argument('--child', action='store_true')
    args = ap.parse_args()
    if args.child:
        return child()
    rows = []
    for name, env in SETTINGS:
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=e, capture_output=True, text=True, timeout=1500)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')]
        if r.returncode != 0 or not line:
            rows.append((name, None, r.stderr[-400:]))
            continue
        rows.append((name, json.loads(line[-1][7:]), ''))
    out = ['# Five-level network (67 M parameters, 30 k-vertex mesh): weight-gradient distance from the fp64 run of the oracle, by matrix-core setting', '',
           'rel-L2 over all weight gradients / over the weights with K >= 1024 columns / over the rest; `beyond` = entries further than 1e-3 of the '
           'gradient scale from the fp64 value (of %s).' % (rows[0][1]['total_entries'] if rows[0][1] else '?'), '',
           '| setting | fwd max-abs vs fp64 | grad rel-L2 all | K >= 1024 | rest | beyond 1e-3 |', '|---|---|---|---|---|---|']
    cpu_done = False
    for name, d, err in rows:
        if d is None:
            out.append('| %s | failed: %s |' % (name, err.replace('\n', ' ')[-200:]))
            continue
        if not cpu_done:
            c = d['cpu']
            out.append('| fp32 CPU oracle (torch addmm) | - | %.3e | %.3e | %.3e | %d |' % (c['all'], c['k_ge_1024'], c['rest'], c['beyond']))
            cpu_done = True
        h = d['hip']
        out.append('| HIP, %s | %.2e | %.3e | %.3e | %.3e | %d |' % (name, d['fwd_max_vs_fp64'], h['all'], h['k_ge_1024'], h['rest'], h['beyond']))
    for name, d, err in rows:
        if d is not None:
            out += ['', 'Largest shares of the squared error, %s: ' % name + '; '.join('%s %s %.0f %%' % (t['name'], tuple(t['shape']), 100 * t['share']) for t in d['top'])]
    text = '\n'.join(out) + '\n'
    print(text)
    if args.md:
        open(args.md, 'w').write(text)


if __name__ == '__main__':
    main()
