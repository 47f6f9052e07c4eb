#!/usr/bin/env python3
"""fp32 forward edge kernel (with mask + indicator) at the headline level-0 shape, the Delaunay mesh, 1 M vertices and the two coarser levels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import functional as SF
from surface_texture_inpainting_net_amd.plan import EdgeSet
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
dev = torch.device('cuda:0')
def t(f, n=15):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for n0, H, irr in [(200_000, 128, False), (200_000, 128, True), (1_000_000, 128, False), (60_000, 256, False), (18_063, 512, False)]:
    s = make_synthetic_mesh(n0, 1, seed=0, dilations=(), irregular=irr)
    ei = s.edge_index.to(dev)
    N, E = s.x.shape[0], ei.shape[1]
    edges = EdgeSet(ei, N, torch.zeros(1, dtype=torch.int32, device=dev))
    Y = torch.randn(N, 2 * H, device=dev)
    out = torch.empty(N, H + 4, device=dev)
    mask = torch.empty(E * (H // 32), dtype=torch.int32, device=dev)
    g = torch.randn(N, H, device=dev); dY = torch.empty(N, 2 * H, device=dev)
    nb = (E * H + 2 * N * H) * 4 + 4 * E + 4 * (N + 1)
    us = t(lambda: SF.edge_relu_mean_fwd(Y[:, :H], Y[:, H:], edges.by_dst, out, indicator=True, mask=mask))
    ub = t(lambda: SF.edge_relu_mean_bwd_mask(g, mask, edges, dY[:, :H], dY[:, H:]))
    print('N=%d E=%d H=%d %s: fwd %.1f us (%.2f TB/s, %.3f of 8)   bwd pair %.1f us' % (N, E, H, 'delaunay' if irr else 'regular', us, nb / us / 1e6, nb / us / 8e6, ub), flush=True)
