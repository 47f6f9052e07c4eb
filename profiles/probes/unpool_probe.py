#!/usr/bin/env python3
"""Pool / unpool kernels of the headline hierarchy stand-alone (us): max pool fwd / bwd, unpool (gather) fwd, unpool bwd (segment sum)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import functional as SF
from surface_texture_inpainting_net_amd.plan import PoolMap
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
dev = torch.device('cuda:0')
def t(f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
s = make_synthetic_mesh(200_000, 3, seed=0).to(dev)
nv = [int(v) for v in s.num_vertices.reshape(-1)]
bad = torch.zeros(1, dtype=torch.int32, device=dev)
for lvl, C in ((1, 64), (2, 128)):
    pm = PoolMap(s['hierarchy_trace_index_%d' % lvl], nv[lvl - 1], nv[lvl], bad)
    xf = torch.randn(nv[lvl - 1], C, device=dev, requires_grad=True)
    xc = torch.randn(nv[lvl], 2 * C, device=dev, requires_grad=True)
    y = SF.PoolMaxFn.apply(xf, pm); g = torch.randn_like(y)
    u = SF.UnpoolFn.apply(xc, pm); gu = torch.randn_like(u)
    ch = pm.children
    print('  direct: segment_sum %.1f us, gather_rows %.1f us' % (t(lambda: SF.segment_sum(gu, ch.rowptr, ch.col, pm.n_coarse), 50), t(lambda: SF.gather_rows(xc.detach(), pm.trace), 50)), flush=True)
    print('level %d -> %d (%d -> %d rows): pool max fwd C=%d %.1f us, bwd %.1f us | unpool fwd C=%d %.1f us, bwd (segment sum) %.1f us' % (
        lvl - 1, lvl, nv[lvl - 1], nv[lvl], C, t(lambda: SF.PoolMaxFn.apply(xf, pm)), t(lambda: torch.autograd.grad(y, xf, g, retain_graph=True)),
        2 * C, t(lambda: SF.UnpoolFn.apply(xc, pm)), t(lambda: torch.autograd.grad(u, xc, gu, retain_graph=True))), flush=True)
