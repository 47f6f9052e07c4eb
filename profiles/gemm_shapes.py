#!/usr/bin/env python3
"""Per-shape GEMM table over the exact NT / TN shapes one training step of the headline config issues (200 704 / 60 211 /
18 063 vertices, pre-split weight operands as the block calls use them), random data, variants interleaved in ONE
process (rounds x variants; median and min reported).  Variants are environment switches the C library re-reads on every
call (STIN_NT_STRIP=0/1 ...), so both arms run the same binary on the same device in the same minute.

    python profiles/gemm_shapes.py [--rounds 7] [--md profiles/r02_gemm_shapes.md] [--variants STIN_NT_STRIP=0,STIN_NT_STRIP=1]
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surface_texture_inpainting_net_amd import functional as SF  # noqa: E402

N0, N1, N2 = 200704, 60211, 18063
# (rows, Cin (padded), Cout, has_shortcut, count per step, compact trans-inv layout: only the network's first block - Y = [B | S])
BLOCKS = [(N0, 12, 64, True, 1, True), (N1, 64, 128, True, 1, False), (N2, 128, 256, True, 1, False), (N2, 256, 256, False, 9, False),
          (N1, 256, 128, True, 1, False), (N0, 128, 64, True, 1, False), (N0, 64, 64, False, 1, False)]
HBM_COPY = 6.29e12
MFMA_PEAK = 2.5e15
dev = torch.device('cuda:0')


def time_once(f, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--inner', type=int, default=5)
    ap.add_argument('--md', default=None)
    ap.add_argument('--variants', default='FRAG=0,FRAG=1', help='comma list of A+B=.. settings: FRAG=0/1 picks the weight layout / NT '
                    'kernel (tiled vs resident strip) in Python, everything else is an environment variable the C library re-reads')
    ap.add_argument('--only', default=None, help='nt or tn')
    ap.add_argument('--shape', default=None, help='only this M,Nc,K')
    ap.add_argument('--cold', action='store_true', help='evict L2 / Infinity Cache before every timed group (default: operands warm, '
                    'as in the network where the previous kernel has just written them)')
    args = ap.parse_args()
    variants = []
    for v in args.variants.split(','):
        variants.append(dict(kv.split('=') for kv in v.split('+')) if v else {})
    jobs = []          # (name, M, Nc, K, count, fn)
    keep = []
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)      # 512 MB: evicts L2 and the Infinity Cache between launches
    for (M, cin, cout, sc, cnt, ti) in BLOCKS:
        H = 2 * cout
        yw = (H if ti else 2 * H) + (cout if sc else 0)
        x = torch.randn(M, cin, device=dev)
        wcat = torch.randn(yw, cin, device=dev) * 0.05
        w2 = torch.randn(cout, H, device=dev) * 0.05
        hE = torch.rand(M, H + 4, device=dev)
        dagg = torch.randn(M, cout, device=dev)
        dY = torch.randn(M, yw, device=dev)
        FR = 0x400
        ops = {}
        for fr in (0, FR):
            ops[fr] = dict(wcat=SF.split_weights(wcat, SF.PREC_FWD | fr), w2=SF.split_weights(w2, SF.PREC_FWD | fr),
                           wcatT=SF.split_weights(wcat.t().contiguous(), SF.PREC_BWD | fr),
                           w2T=SF.split_weights(w2.t().contiguous(), SF.PREC_BWD | fr))
        bias = torch.randn(yw, device=dev)
        b2 = torch.randn(cout, device=dev)
        pf, pb = SF.PREC_FWD | SF.GEMM_W_PRESPLIT, SF.PREC_BWD | SF.GEMM_W_PRESPLIT
        oY, oagg, odh, odx = (torch.empty(M, c, device=dev) for c in (yw, cout, H, cin))
        keep.append((x, ops, hE, dagg, dY, bias, b2, oY, oagg, odh, odx))

        def fr():
            return FR if os.environ.get('FRAG', '1') != '0' else 0
        if args.only in (None, 'nt'):
            jobs.append(('nt fwd Y', M, yw, cin, cnt, lambda x=x, w=ops, b=bias, o=oY, p=pf: SF.gemm_nt(x, w[fr()]['wcat'], b, out=o, precision=p | fr())))
            jobs.append(('nt fwd agg', M, cout, H, cnt,
                         lambda a=hE, w=ops, b=b2, o=oagg, p=pf, H=H: SF.gemm_nt(a[:, :H], w[fr()]['w2'], b, out=o, row_mask=a[:, H], precision=p | fr())))
            jobs.append(('nt bwd dhE', M, H, cout, cnt, lambda a=dagg, w=ops, o=odh, p=pb: SF.gemm_nt(a, w[fr()]['w2T'], out=o, precision=p | fr())))
            if cin > 12:
                jobs.append(('nt bwd dx', M, cin, yw, cnt, lambda a=dY, w=ops, o=odx, p=pb: SF.gemm_nt(a, w[fr()]['wcatT'], out=o, precision=p | fr())))
        if args.only in (None, 'tn'):
            jobs.append(('tn dW2', M, cout, H, cnt,
                         lambda g=dagg, a=hE, H=H: SF.gemm_tn(g, a[:, :H], ones_column=True, row_weight=a[:, H], precision=SF.PREC_BWD)))
            jobs.append(('tn dWcat', M, yw, cin, cnt, lambda g=dY, a=x: SF.gemm_tn(g, a, ones_column=True, precision=SF.PREC_BWD)))
    rows = []
    allkeys = {k for v in variants for k in v}
    if args.shape:
        want = tuple(int(t) for t in args.shape.split(','))
        jobs = [j for j in jobs if (j[1], j[2], j[3]) == want][:1]

    def setenv(v):
        for k in allkeys:
            os.environ.pop(k, None)
        os.environ.update(v)

    for name, M, Nc, K, cnt, fn in jobs:
        times = [[] for _ in variants]
        for v in variants:                       # warm-up of every variant (also the one-time kernel attribute calls)
            setenv(v)
            fn()
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for i, v in enumerate(variants):
                setenv(v)
                if args.cold:
                    flush.zero_()
                times[i].append(time_once(fn, args.inner))
        bytes_min = 4.0 * (M * Nc + M * K + Nc * K)
        bound = max(bytes_min / HBM_COPY, 3 * 2.0 * M * Nc * K / MFMA_PEAK)
        rows.append((name, M, Nc, K, cnt, bound, [(statistics.median(t), min(t)) for t in times]))
    hdr = '| GEMM | M | Nc | K | per step | roofline bound us | ' + ' | '.join(
        '%s med / min us (frac)' % ('+'.join('%s=%s' % kv for kv in v.items()) or 'default') for v in variants) + ' |'
    lines = [hdr, '|' + '---|' * (6 + len(variants))]
    tot = [0.0] * len(variants)
    totb = 0.0
    for name, M, Nc, K, cnt, bound, ts in rows:
        cells = ' | '.join('%.1f / %.1f (%.2f)' % (med * 1e6, mn * 1e6, bound / med) for med, mn in ts)
        lines.append('| %s | %d | %d | %d | %d | %.1f | %s |' % (name, M, Nc, K, cnt, bound * 1e6, cells))
        for i, (med, _) in enumerate(ts):
            tot[i] += med * cnt
        totb += bound * cnt
    lines.append('| **sum per step** | | | | | %.0f | %s |' % (totb * 1e6, ' | '.join('%.0f us (%.2f)' % (t * 1e6, totb / t) for t in tot)))
    text = '\n'.join(lines)
    print(text)
    if args.md:
        with open(args.md, 'w') as f:
            f.write('# GEMM shapes of one headline training step (fp32 storage, split-16-bit MFMA, pre-split weights; random data; '
                    'median / min of %d interleaved rounds x %d launches)\n\n'
                    'roofline bound = max(min HBM bytes / 6.29 TB/s, 3 x 2MNK / 2.5 PF); frac = bound / median\n\n' % (args.rounds, args.inner))
            f.write(text + '\n')


if __name__ == '__main__':
    main()
