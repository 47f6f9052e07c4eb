#!/usr/bin/env python3
"""Launch the roofline kernels a few times each at the headline shapes, for rocprofv3 PMC passes:

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 profiles/pmc_kernels.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 profiles/pmc_kernels.py

(separate passes: FETCH_SIZE and WRITE_SIZE do not fit one TCC pass - MI355X_MICROARCH.md "rocprofv3 PMC slots").
profiles/pmc_summarize.py turns the two counter CSVs into profiles/rNN_pmc_traffic.json."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surface_texture_inpainting_net_amd import functional as SF  # noqa: E402
from surface_texture_inpainting_net_amd.plan import build_csr, plan_for  # noqa: E402
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh  # noqa: E402

dev = torch.device('cuda:0')
# PMC_LIVE=1 (bench.py's live traffic leg): the level-0 forward edge kernel, 5 launches on the headline mesh, then 4 launches at
# the 1 M-vertex / 6 M-edge size of `hbm_honest` - one profiler start for both figures; bench.py separates them by launch order
if os.environ.get('PMC_LIVE', '0') == '1':
    from surface_texture_inpainting_net_amd.plan import EdgeSet  # noqa: E402
    H = 128
    s = make_synthetic_mesh(200_000, 1, seed=0, dilations=()).to(dev)
    e = plan_for(s).edges('edge_index', 0)
    n = s.x.shape[0]
    A, B = (torch.randn(n, H, device=dev) for _ in range(2))
    out4 = torch.empty(n, H + 4, device=dev)
    mask = torch.empty(e.n_edges * (H // 32), dtype=torch.int32, device=dev)
    for _ in range(5):
        SF.edge_relu_mean_fwd(A, B, e.by_dst, out4, indicator=True, mask=mask)
    torch.cuda.synchronize()
    del A, B, out4, mask, e, s
    NV = 1_000_000
    ei = torch.randint(0, NV, (2, 6 * NV), generator=torch.Generator().manual_seed(1)).to(dev)
    e = EdgeSet(ei, NV, torch.zeros(1, dtype=torch.int32, device=dev))
    A, B = (torch.randn(NV, H, device=dev) for _ in range(2))
    out4 = torch.empty(NV, H + 4, device=dev)
    mask = torch.empty(6 * NV * (H // 32), dtype=torch.int32, device=dev)
    for _ in range(4):
        SF.edge_relu_mean_fwd(A, B, e.by_dst, out4, indicator=True, mask=mask)
    torch.cuda.synchronize()
    sys.exit(0)
# PMC_VERTICES=1000000: the 1 M-vertex / 6 M-edge size of bench.py's `hbm_honest` (gathered operand 512 MB > Infinity Cache), forward
# and one-launch backward only
NV = int(os.environ.get('PMC_VERTICES', '200000'))
if NV != 200_000:
    g0 = torch.Generator().manual_seed(1)
    from surface_texture_inpainting_net_amd.plan import EdgeSet  # noqa: E402
    ei = torch.randint(0, NV, (2, 6 * NV), generator=g0).to(dev)
    e = EdgeSet(ei, NV, torch.zeros(1, dtype=torch.int32, device=dev))
    H = 128
    A, B, G = (torch.randn(NV, H, device=dev) for _ in range(3))
    out, out2 = torch.empty(NV, H + 4, device=dev), torch.empty(NV, H, device=dev)
    mask = torch.empty(6 * NV * (H // 32), dtype=torch.int32, device=dev)
    for _ in range(4):
        SF.edge_relu_mean_fwd(A, B, e.by_dst, out, indicator=True, mask=mask)
        SF.edge_relu_mean_bwd_mask(G, mask, e, out[:, :H], out2)
    torch.cuda.synchronize()
    # (round 6) the bf16-storage twins at the same size (BASELINE config 5's level 0: k_edge_fwd8 / k_edge_bwd_mask_pair8)
    A16, B16, G16 = A.bfloat16(), B.bfloat16(), G.bfloat16()
    del A, B, G, out, out2
    out16, out16b = torch.empty(NV, H + 8, dtype=torch.bfloat16, device=dev), torch.empty(NV, H, dtype=torch.bfloat16, device=dev)
    for _ in range(4):
        SF.edge_relu_mean_fwd(A16, B16, e.by_dst, out16, indicator=True, mask=mask)
        SF.edge_relu_mean_bwd_mask(G16, mask, e, out16[:, :H], out16b)
    torch.cuda.synchronize()
    sys.exit(0)
# PMC_ONLY_FWD=1: the level-0 forward edge kernel of the headline mesh only (what bench.py's live `roofline.traffic` leg profiles)
ONLY_FWD = os.environ.get('PMC_ONLY_FWD', '0') == '1'
s = make_synthetic_mesh(200_000, 1, seed=0, dilations=())
if os.environ.get('PMC_LOCALITY', '0') == '1':
    # (round 6) the same mesh with its vertices renumbered by locality - what loader.SceneLoader's resident plans run on: forward,
    # the one-launch backward and the compact trans-inv backward only
    from surface_texture_inpainting_net_amd.synthetic import renumber_by_locality  # noqa: E402
    s = renumber_by_locality(s)[0].to(dev)
    e = plan_for(s).edges('edge_index', 0)
    n, H = s.x.shape[0], 128
    A, B, G = (torch.randn(n, H, device=dev) for _ in range(3))
    out, out2 = torch.empty(n, H, device=dev), torch.empty(n, H, device=dev)
    mask = torch.empty(e.n_edges * (H // 32), dtype=torch.int32, device=dev)
    for _ in range(5):
        SF.edge_relu_mean_fwd(A, B, e.by_dst, out, mask=mask)
        SF.edge_relu_mean_bwd_mask(G, mask, e, out, out2)
        SF.edge_relu_mean_bwd_mask_ti(G, mask, e, out)
    torch.cuda.synchronize()
    sys.exit(0)
s = s.to(dev)
plan = plan_for(s)
e = plan.edges('edge_index', 0)
n, H = s.x.shape[0], 128
A, B, G = (torch.randn(n, H, device=dev) for _ in range(3))
out = torch.empty(n, H, device=dev)
out2 = torch.empty(n, H, device=dev)
mask = torch.empty(e.n_edges * (H // 32), dtype=torch.int32, device=dev)
if ONLY_FWD:
    out4 = torch.empty(n, H + 4, device=dev)
    for _ in range(5):
        SF.edge_relu_mean_fwd(A, B, e.by_dst, out4, indicator=True, mask=mask)
    torch.cuda.synchronize()
    sys.exit(0)
for _ in range(5):
    SF.edge_relu_mean_fwd(A, B, e.by_dst, out, mask=mask)          # as the training step runs it (writes the mask)
    SF.edge_relu_mean_bwd_dst_mask(G, mask, e.by_dst, out)
    SF.edge_relu_mean_bwd_src_mask(G, mask, e, out)
    SF.edge_relu_mean_bwd_mask(G, mask, e, out, out2)               # round 2: both halves in one launch (what the step runs)
    SF.edge_relu_mean_fwd_ti(None, B, e.by_dst, out, mask=mask)      # round 6: the first block's compact trans-inv pair
    SF.edge_relu_mean_bwd_mask_ti(G, mask, e, out2)
    SF.edge_relu_mean_bwd_dst(A, B, G, e.by_dst, out)               # recompute forms (STIN_EDGE_MASK=0)
    SF.edge_relu_mean_bwd_src(A, B, G, e.inv_deg, e.by_src, out)
# bf16-storage twins of the training-step kernels
A16, B16, G16 = A.bfloat16(), B.bfloat16(), G.bfloat16()
out16 = torch.empty(n, H, dtype=torch.bfloat16, device=dev)
out16b = torch.empty(n, H, dtype=torch.bfloat16, device=dev)
for _ in range(5):
    SF.edge_relu_mean_fwd(A16, B16, e.by_dst, out16, mask=mask)
    SF.edge_relu_mean_bwd_dst_mask(G16, mask, e.by_dst, out16)
    SF.edge_relu_mean_bwd_src_mask(G16, mask, e, out16)
    SF.edge_relu_mean_bwd_mask(G16, mask, e, out16, out16b)
# standalone scatter-add: src[E, 64] -> out[N, 64], index in arbitrary edge order
g = torch.Generator().manual_seed(0)
idx = torch.randint(0, 200_000, (1_200_000,), generator=g).to(dev)
src = torch.randn(1_200_000, 64, device=dev)
bad = torch.zeros(1, dtype=torch.int32, device=dev)
csr = build_csr(idx, None, 200_000, 1_200_000, bad, want_perm=True)
for _ in range(5):
    SF.segment_sum(src, csr.rowptr, csr.perm, 200_000)
torch.cuda.synchronize()
