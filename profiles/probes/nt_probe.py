#!/usr/bin/env python3
"""Ad-hoc NT timing at arbitrary shapes: SHAPES="M,Nc,K;M,Nc,K" PREC=bwd|fwd python profiles/probes/nt_probe.py (median of rounds)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import functional as SF  # noqa: E402

shapes = [tuple(int(v) for v in s.split(',')) for s in os.environ.get('SHAPES', '18063,256,1024').split(';')]
variants = [dict(kv.split('=') for kv in v.split('+')) if v else {} for v in os.environ.get('VARIANTS', '').split(',')]
allkeys = {k for v in variants for k in v}
FR = 0x400


def t_once(f, n=5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (M, Nc, K) in shapes:
    A = torch.randn(M, K, device='cuda')
    W = torch.randn(Nc, K, device='cuda') * 0.05
    out = torch.empty(M, Nc, device='cuda')
    for pname, prec in (('bf16x3', SF.GEMM_BF16X3), ('f16x3', SF.PREC_FWD)):
        Wp = SF.split_weights(W, prec | FR)
        f = lambda: SF.gemm_nt(A, Wp, out=out, precision=prec | SF.GEMM_W_PRESPLIT | FR)
        res = []
        for v in variants:
            for k in allkeys:
                os.environ.pop(k, None)
            os.environ.update(v)
            f()
        torch.cuda.synchronize()
        ts = [[] for _ in variants]
        for _ in range(7):
            for i, v in enumerate(variants):
                for k in allkeys:
                    os.environ.pop(k, None)
                os.environ.update(v)
                ts[i].append(t_once(f))
        print(M, Nc, K, pname, ' | '.join('%s: %.1f/%.1f' % ('+'.join('%s=%s' % kv for kv in v.items()) or 'default', statistics.median(t), min(t))
                                          for v, t in zip(variants, ts)), flush=True)
