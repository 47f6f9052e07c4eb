#!/usr/bin/env python3
"""bf16 edge forward kernel: rows in flight (STIN_EDGE8_U) per row width, regular and Delaunay meshes.  us and TB/s (algorithmic)."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import functional as SF
from surface_texture_inpainting_net_amd.plan import EdgeSet
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
dev = torch.device('cuda:0')
def t(f, n=10):
    for _ in range(2): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
cases = [(200_000, 128, False), (200_000, 128, True), (1_000_000, 128, False), (60_000, 256, False), (18_063, 512, False), (8_100, 2048, False), (4_352, 1024, False)]
if os.environ.get('CASES'):
    cases = [cases[int(i)] for i in os.environ['CASES'].split(',')]
for n0, H, irr in cases:
    s = make_synthetic_mesh(n0, 1, seed=0, dilations=(), irregular=irr)
    ei = s.edge_index.to(dev)
    N, E = s.x.shape[0], ei.shape[1]
    edges = EdgeSet(ei, N, torch.zeros(1, dtype=torch.int32, device=dev))
    Y = torch.randn(N, 2 * H, device=dev).bfloat16()
    out = torch.empty(N, H + 8, device=dev, dtype=torch.bfloat16)
    mask = torch.empty(E * (H // 32), dtype=torch.int32, device=dev)
    nbytes = (E * H + 2 * N * H) * 2 + 4 * E + 4 * (N + 1)
    res = []
    for u in (1, 2, 3, 4, 6):
        os.environ['STIN_EDGE8_U'] = str(u) * 5
        res.append((u, t(lambda: SF.edge_relu_mean_fwd(Y[:, :H], Y[:, H:], edges.by_dst, out, indicator=True, mask=mask))))
    print('N=%d E=%d H=%d %s: ' % (N, E, H, 'delaunay' if irr else 'regular') + '  '.join('U%d %.1f us (%.2f TB/s)' % (u, us, nbytes / us / 1e6) for u, us in res), flush=True)
