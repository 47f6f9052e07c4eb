#!/usr/bin/env python3
"""fp32 edge kernels at the bottleneck shape (18 063 x 512, plus its dilated sets): rows in flight (STIN_EDGE_U512 / STIN_EDGE_US512)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import functional as SF
from surface_texture_inpainting_net_amd.plan import EdgeSet
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
dev = torch.device('cuda:0')
def t(f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for n0, irr in ((18_063, False), (18_063, True), (60_000, False)):
    H = 512
    s = make_synthetic_mesh(n0, 1, seed=0, dilations=(), irregular=irr)
    ei = s.edge_index.to(dev)
    N, E = s.x.shape[0], ei.shape[1]
    edges = EdgeSet(ei, N, torch.zeros(1, dtype=torch.int32, device=dev))
    Y = torch.randn(N, 2 * H, device=dev)
    out = torch.empty(N, H + 4, device=dev)
    mask = torch.empty(E * (H // 32), dtype=torch.int32, device=dev)
    g = torch.randn(N, H, device=dev); dY = torch.empty(N, 2 * H, device=dev)
    res = []
    for u in (0, 3, 4, 6):
        os.environ['STIN_EDGE_U512'] = str(u)
        res.append('U%d %.1f' % (u or 2, t(lambda: SF.edge_relu_mean_fwd(Y[:, :H], Y[:, H:], edges.by_dst, out, indicator=True, mask=mask))))
    os.environ['STIN_EDGE_U512'] = '0'
    rb = []
    for u in (0, 2, 3, 4):
        os.environ['STIN_EDGE_US512'] = str(u)
        rb.append('US%d %.1f' % (u or 1, t(lambda: SF.edge_relu_mean_bwd_mask(g, mask, edges, dY[:, :H], dY[:, H:]))))
    os.environ['STIN_EDGE_US512'] = '0'
    print('N=%d E=%d H=512 %s: fwd us %s | bwd pair us %s' % (N, E, 'delaunay' if irr else 'regular', '  '.join(res), '  '.join(rb)), flush=True)
