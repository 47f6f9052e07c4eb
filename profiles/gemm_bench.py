#!/usr/bin/env python3
"""GEMM-only microbenchmark over the exact (shape, role, count) list one training step of the headline config issues
(200 704 / 60 211 / 18 063 vertices): prints ms per step and fp32-equivalent TFLOP/s per role.  Tuning aid."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surface_texture_inpainting_net_amd import functional as SF  # noqa: E402

N0, N1, N2 = 200704, 60211, 18063
# (rows, Cin, Cout, has_shortcut, count)
BLOCKS = [(N0, 10, 64, True, 1), (N1, 64, 128, True, 1), (N2, 128, 256, True, 1), (N2, 256, 256, False, 9),
          (N1, 256, 128, True, 1), (N0, 128, 64, True, 1), (N0, 64, 64, False, 1)]
dev = torch.device('cuda:0')


def timeit(f, n=10):
    for _ in range(2):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


tot = {'fwd_nt': [0.0, 0.0], 'bwd_nt': [0.0, 0.0], 'bwd_tn': [0.0, 0.0]}
for (M, cin, cout, sc, cnt) in BLOCKS:
    H = 2 * cout
    yw = 2 * H + (cout if sc else 0)
    x = torch.randn(M, cin, device=dev)
    wcat = torch.randn(yw, cin, device=dev)
    w2 = torch.randn(cout, H, device=dev)
    hE = torch.randn(M, H + 4, device=dev)
    dagg = torch.randn(M, cout, device=dev)
    dY = torch.randn(M, yw, device=dev)
    wcatT = wcat.t().contiguous()
    w2T = w2.t().contiguous()
    jobs = [('fwd_nt', 2.0 * M * yw * cin, lambda: SF.gemm_nt(x, wcat, None, precision=SF.PREC_FWD)),
            ('fwd_nt', 2.0 * M * cout * H, lambda: SF.gemm_nt(hE[:, :H], w2, None, row_mask=hE[:, H], precision=SF.PREC_FWD)),
            ('bwd_nt', 2.0 * M * H * cout, lambda: SF.gemm_nt(dagg, w2T, precision=SF.PREC_BWD)),
            ('bwd_nt', 2.0 * M * cin * yw, lambda: SF.gemm_nt(dY, wcatT, precision=SF.PREC_BWD)),
            ('bwd_tn', 2.0 * M * cout * H, lambda: SF.gemm_tn(dagg, hE[:, :H], ones_column=True, row_weight=hE[:, H], precision=SF.PREC_BWD)),
            ('bwd_tn', 2.0 * M * yw * cin, lambda: SF.gemm_tn(dY, x, ones_column=True, precision=SF.PREC_BWD))]
    for role, flops, fn in jobs:
        t = timeit(fn)
        tot[role][0] += t * cnt
        tot[role][1] += flops * cnt
ms = sum(v[0] for v in tot.values()) * 1e3
fl = sum(v[1] for v in tot.values())
print('GEMM total %.3f ms/step, %.1f GFLOP, %.1f TF fp32-equivalent' % (ms, fl / 1e9, fl / ms / 1e9))
for k, v in tot.items():
    print('  %-7s %.3f ms  %.1f TF' % (k, v[0] * 1e3, v[1] / v[0] / 1e12))
