"""Edge-stage kernels stand-alone on every edge set of a synthetic hierarchy (config 5 by default): us and algorithmic GB/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import functional as SF
from surface_texture_inpainting_net_amd.plan import EdgeSet
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
N0, LV = int(os.environ.get('N0', 1000000)), int(os.environ.get('LV', 5))
DT = torch.bfloat16 if os.environ.get('DT', 'bf16') == 'bf16' else torch.float32
s = make_synthetic_mesh(N0, LV, seed=0)
nv = s.num_vertices.reshape(-1).tolist()
dev = 'cuda:0'
bad = torch.zeros(1, dtype=torch.int32, device=dev)
def timed(fn, it=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it
es = 2 if DT == torch.bfloat16 else 4
for k in s.keys():
    if 'edge_index' not in k: continue
    lvl = 0 if k == 'edge_index' else int(k.rsplit('_', 1)[1])
    n, H = nv[lvl], 128 * 2 ** lvl
    ei = s[k].to(dev)
    E = ei.shape[1]
    edges = EdgeSet(ei, n, bad)
    pad = 8 if DT == torch.bfloat16 else 4
    Y = torch.randn(n, 2 * H, device=dev).to(DT)
    out = torch.empty(n, H + pad, device=dev, dtype=DT)
    mask = torch.empty(max(E, 1) * (H // 32), dtype=torch.int32, device=dev)
    g = torch.randn(n, H, device=dev).to(DT)
    dY = torch.empty(n, 2 * H, device=dev, dtype=DT)
    tf = timed(lambda: SF.edge_relu_mean_fwd(Y[:, :H], Y[:, H:], edges.by_dst, out, indicator=True, mask=mask))
    tb = timed(lambda: SF.edge_relu_mean_bwd_mask(g, mask, edges, dY[:, :H], dY[:, H:]))
    bf = (E * H + 2 * n * H) * es + 4 * E + 4 * (n + 1)
    bb = E * H // 8 + 2 * n * H * es + 4 * (n + 1) + E * H * es + E * H // 8 + 12 * E + n * H * es + 4 * (n + 1)
    print('%-32s N=%7d E=%8d H=%4d  fwd %7.1f us %5.0f GB/s   bwd %7.1f us %5.0f GB/s' % (k, n, E, H, tf, bf / tf / 1e3, tb, bb / tb / 1e3), flush=True)
