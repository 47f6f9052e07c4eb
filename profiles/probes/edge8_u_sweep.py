#!/usr/bin/env python3
"""Round 6, review item 4: rows in flight (U) of the bf16-storage level-0 edge kernels (k_edge_fwd8 / k_edge_bwd_mask_pair8, H = 128)
at the HBM-served size (1 M vertices / 6 M edges) and at the cache-resident headline size.  STIN_EDGE8_U / STIN_EDGE8_US are re-read
per call (digits = U for H 128, 256, 512, 1024, 2048)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import functional as SF  # noqa: E402
from surface_texture_inpainting_net_amd.plan import EdgeSet  # noqa: E402

dev = torch.device('cuda:0')
H = 128


def timed(f, n=10):
    for _ in range(2):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for NV in (200_704, 1_000_000):
    ei = torch.randint(0, NV, (2, 6 * NV), generator=torch.Generator().manual_seed(1)).to(dev)
    e = EdgeSet(ei, NV, torch.zeros(1, dtype=torch.int32, device=dev))
    A, B, G = (torch.randn(NV, H, device=dev).bfloat16() for _ in range(3))
    out, out2 = torch.empty(NV, H + 8, dtype=torch.bfloat16, device=dev), torch.empty(NV, H, dtype=torch.bfloat16, device=dev)
    mask = torch.empty(6 * NV * (H // 32), dtype=torch.int32, device=dev)
    fb = (6 * NV * H + 2 * NV * H) * 2 + 4 * 6 * NV + 4 * (NV + 1)
    for u in (1, 2, 3, 4, 6):
        os.environ['STIN_EDGE8_U'] = '%d2222' % u
        t = timed(lambda: SF.edge_relu_mean_fwd(A, B, e.by_dst, out, indicator=True, mask=mask))
        print('N %8d fwd8  U %d: %7.1f us  %.3f of 8 TB/s' % (NV, u, t, fb / t / 1e6 / 8000.0), flush=True)
    os.environ.pop('STIN_EDGE8_U')
    for u in (1, 2, 3, 4, 6):
        os.environ['STIN_EDGE8_US'] = '%d2222' % u
        t = timed(lambda: SF.edge_relu_mean_bwd_mask(G, mask, e, out[:, :H], out2))
        print('N %8d pair8 US %d: %7.1f us' % (NV, u, t), flush=True)
    os.environ.pop('STIN_EDGE8_US')
    del ei, e, A, B, G, out, out2, mask
