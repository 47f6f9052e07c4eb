#!/usr/bin/env python3
"""Round 6 experiment: NT products of the 18 063-row level on pre-split operands staged by LDS-DMA (profiles/probes/planes/planes.hip)
against the shipped split kernels (fragment-order pre-split weights, operands split inside the K loop): bit-identity and stand-alone time.

    (build first, in this directory:
     hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -fno-slp-vectorize -ffp-contract=off -o libplanes_probe.so planes.hip)
    python profiles/probes/planes/nt_planes_probe.py [--stamps] [--md out.md]
"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
from surface_texture_inpainting_net_amd import functional as SF  # noqa: E402

lib = ctypes.CDLL(os.path.join(HERE, 'libplanes_probe.so'))
c_p, c_i64, c_int, c_f = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
lib.planes_split_rows.argtypes = [c_p, c_i64, c_i64, c_int, c_int, c_f, c_p, c_p]
lib.planes_gemm_nt.argtypes = [c_p, c_i64, c_p, c_i64, c_p, c_p, c_i64, c_p, c_i64, c_i64, c_int, c_int, c_p, c_i64, c_int, c_f, c_int, c_int, c_p, c_p]
dev = torch.device('cuda:0')
stamps_ptr = [0]


def stamp_report(run, tag, nblocks):
    """One stamped launch: per wave 0 / wave 4 of every block, s_memtime at start (0), prologue issued (1), after chunk t (2 + t),
    loop end (60), epilogue issued (61), stores drained (62) -> medians over the blocks, in shader cycles (s_memtime)
    ."""
    buf = torch.zeros(nblocks * 8 * 64, dtype=torch.int64, device=dev)
    stamps_ptr[0] = buf.data_ptr()
    run()
    torch.cuda.synchronize()
    stamps_ptr[0] = 0
    st = buf.view(nblocks, 8, 64).cpu()
    live = st[:, 0, 0] > 0
    st = st[live]
    t0 = st[:, :, 0].min()
    out = []
    for w in (0, 4):
        s = st[:, w, :]
        chunks = [int(i) for i in range(2, 59) if (s[:, i] > 0).all()]
        def med(a):
            return float(a.double().median())
        pro = med(s[:, 1] - s[:, 0])
        first = med(s[:, 2] - s[:, 1]) if chunks else 0.0
        per = med((s[:, chunks[-1]] - s[:, 2]).double() / max(1, len(chunks) - 1)) if len(chunks) > 1 else 0.0
        epi = med(s[:, 61] - s[:, 60])
        drain = med(s[:, 62] - s[:, 61])
        total = med(s[:, 62] - s[:, 0])
        start_skew = med(s[:, 0] - t0)
        out.append('wave %d: start skew %.0f, prologue %.0f, first chunk %.0f, per later chunk %.0f (x%d), epilogue issue %.0f, drain %.0f, total %.0f ticks'
                   % (w, start_skew, pro, first, per, max(0, len(chunks) - 1), epi, drain, total))
    span = float((st[:, :, 62].max() - t0))
    print('[stamps] %s: %d live blocks, first start -> last drain %.0f shader cycles\n  ' % (tag, int(live.sum()), span) + '\n  '.join(out), flush=True)


def stream():
    return torch.cuda.current_stream().cuda_stream


def split_rows(x, f16, scale):
    m, k = x.shape
    out = torch.empty(m, k, dtype=torch.float32, device=dev)          # same bytes / pitch as the fp32 rows
    rc = lib.planes_split_rows(x.data_ptr(), x.stride(0), m, k, int(f16), scale, out.data_ptr(), stream())
    assert rc == 0, rc
    return out


def ptr(t):
    return 0 if t is None else t.data_ptr()


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
    ap.add_argument('--md')
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--inner', type=int, default=10)
    ap.add_argument('--m', type=int, default=18063)
    ap.add_argument('--stamps', action='store_true')
    args = ap.parse_args()
    M = args.m
    g = torch.Generator().manual_seed(0)
    FR = 0x400
    # name, Nc, K, f16, bias, mask, residual, configs
    shapes = [('fwd Y', 1024, 256, True, True, False, False, [5422]),
              ('fwd agg', 256, 512, True, True, True, False, [3214, 3213, 2123]),
              ('bwd dhE', 512, 256, False, False, False, False, [5412, 5413, 3223, 3222]),
              ('bwd dx', 256, 1024, False, False, False, True, [3214, 3213, 2123])]
    lines = ['| product | M x Nc x K | shipped us (med / min) | ' + 'split rows + LDS-DMA: cfg epi: us (med / min) [bit-identical]' + ' |', '|---|---|---|---|']
    for name, Nc, K, f16, has_b, has_m, has_r, cfgs in shapes:
        A = torch.randn(M, K, generator=g).to(dev)
        W = (torch.randn(Nc, K, generator=g) * 0.05).to(dev)
        bias = torch.randn(Nc, generator=g).to(dev) if has_b else None
        mask = (torch.rand(M, 4, generator=g) < 0.9).float().to(dev) if has_m else None
        res = torch.randn(M, Nc, generator=g).to(dev) if has_r else None
        prec = SF.GEMM_F16X3 if f16 else SF.GEMM_BF16X3
        asc, wsc = (8.0, 64.0) if f16 else (1.0, 1.0)
        want = SF.gemm_nt(A, W, bias, row_mask=None if mask is None else mask[:, 0], precision=prec, residual=res)
        wfr = SF.split_weights(W, prec | FR)
        out_ref = torch.empty(M, Nc, device=dev)

        def shipped():
            SF.gemm_nt(A, wfr, bias, out=out_ref, row_mask=None if mask is None else mask[:, 0], precision=prec | SF.GEMM_W_PRESPLIT | FR, residual=res)
        shipped()
        assert torch.equal(out_ref, want), 'shipped kernels disagree with each other?'
        Ap, Wp = split_rows(A, f16, asc), split_rows(W, f16, wsc)
        torch.cuda.synchronize()
        out = torch.empty(M, Nc, device=dev)
        variants = []
        for cfg in cfgs:
            for epi in (0, 1):
                def run(cfg=cfg, epi=epi):
                    rc = lib.planes_gemm_nt(Ap.data_ptr(), K * 4, Wp.data_ptr(), K * 4, ptr(bias), ptr(mask), 4 if mask is not None else 0, ptr(res),
                                            Nc if res is not None else 0, M, Nc, K, out.data_ptr(), Nc, int(f16), 1.0 / (asc * wsc), cfg, epi, stream(), stamps_ptr[0])
                    assert rc == 0, (rc, cfg)
                out.zero_()
                run()
                torch.cuda.synchronize()
                same = torch.equal(out, want)
                err = float((out - want).abs().max())
                variants.append((cfg, epi, run, same, err))
                if args.stamps:
                    stamp_report(run, '%s cfg %d epi %d' % (name, cfg, epi), 4096)
        fns = [shipped] + [v[2] for v in variants]
        times = [[] for _ in fns]
        for f in fns:
            f()
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for i, f in enumerate(fns):
                times[i].append(time_once(f, args.inner))
        cell = lambda t: '%.1f / %.1f' % (statistics.median(t) * 1e6, min(t) * 1e6)   # noqa: E731
        cells = '; '.join('%d e%d: %s [%s]' % (v[0], v[1], cell(times[i + 1]), 'yes' if v[3] else 'NO max-abs %.2e' % v[4]) for i, v in enumerate(variants))
        lines.append('| %s | %d x %d x %d | %s | %s |' % (name, M, Nc, K, cell(times[0]), cells))
        print(lines[-1], flush=True)
    # the split itself as its own pass (what a producer epilogue would absorb)
    x = torch.randn(M, 256, device=dev)
    buf = torch.empty(M, 256, device=dev)
    t = time_once(lambda: lib.planes_split_rows(x.data_ptr(), 256, M, 256, 1, 8.0, buf.data_ptr(), stream()), 20)
    lines.append('')
    lines.append('stand-alone split pass, %d x 256: %.1f us' % (M, t * 1e6))
    text = '\n'.join(lines)
    print(text)
    if args.md:
        open(args.md, 'w').write(text + '\n')


if __name__ == '__main__':
    main()
