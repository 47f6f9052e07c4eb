#!/usr/bin/env python3
"""A/B: the two mask-backward edge kernels as separate launches vs one launch (dB blocks then dA blocks), headline shapes."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import functional as SF  # noqa: E402
from surface_texture_inpainting_net_amd.plan import EdgeSet  # noqa: E402
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh  # noqa: E402

dev = torch.device('cuda:0')
s = make_synthetic_mesh(200_000, 3, seed=0).to(dev)
bad = torch.zeros(1, dtype=torch.int32, device=dev)
nv = [int(v) for v in s.num_vertices.view(-1)]
levels = [(s.edge_index, nv[0], 128), (s['hierarchy_edge_index_1'], nv[1], 256), (s['hierarchy_edge_index_2'], nv[2], 512)]


def t(f, n=30):
    for _ in range(3):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for ei, n, H in levels:
    es = EdgeSet(ei, n, bad)
    e = ei.shape[1]
    Y = torch.randn(n, 2 * H, device=dev)
    out = torch.empty(n, H + 4, device=dev)
    mask = torch.zeros(e * (H // 32), dtype=torch.int32, device=dev)
    SF.edge_relu_mean_fwd(Y[:, :H], Y[:, H:], es.by_dst, out, indicator=True, mask=mask)
    G = torch.randn(n, H, device=dev)
    dY = torch.empty(n, 2 * H + H // 2, device=dev)
    for r in range(2):
        tab = t(lambda: (SF.edge_relu_mean_bwd_dst_mask(G, mask, es.by_dst, dY[:, :H]), SF.edge_relu_mean_bwd_src_mask(G, mask, es, dY[:, H:2 * H])))
        tp = t(lambda: SF.edge_relu_mean_bwd_mask(G, mask, es, dY[:, :H], dY[:, H:2 * H]))
        print('N=%d E=%d H=%d: two launches %.1f us, one launch %.1f us' % (n, e, H, tab, tp))
