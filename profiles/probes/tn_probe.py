#!/usr/bin/env python3
"""Ad-hoc TN (weight-gradient product) timing: SHAPES="M,Nc,K;..." python profiles/probes/tn_probe.py  (median of 7 x 5 launches, incl. the slab fold)"""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import functional as SF
shapes = [tuple(int(v) for v in s.split(',')) for s in os.environ.get('SHAPES', '200704,320,12').split(';')]
def t_once(f, n=5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M, Nc, K in shapes:
    G = torch.randn(M, Nc, device='cuda'); X = torch.randn(M, K, device='cuda')
    f = lambda: SF.gemm_tn(G, X, ones_column=True, precision=SF.PREC_BWD)
    f(); torch.cuda.synchronize()
    ts = [t_once(f) for _ in range(7)]
    ref = torch.cat([X, torch.ones(M, 1, device='cuda')], 1).double()
    err = float((f().double() - G.double().t() @ ref).abs().max() / (G.double().t() @ ref).abs().max())
    print(M, Nc, K, 'median %.1f us min %.1f  rel err vs fp64 %.2e' % (statistics.median(ts), min(ts), err), flush=True)
