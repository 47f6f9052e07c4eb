#!/usr/bin/env python3
"""GEMM kernels at one headline shape for rocprofv3 SQ-counter passes (diagnostics)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surface_texture_inpainting_net_amd import functional as SF
M, Nc, K = (int(v) for v in os.environ.get('SHAPE', '18063,1024,256').split(','))
A = torch.randn(M, K, device='cuda'); W = torch.randn(Nc, K, device='cuda'); G = torch.randn(M, Nc, device='cuda')
for _ in range(3):
    for p in (2, 3, 0):
        SF.gemm_nt(A, W, None, precision=p)
    SF.gemm_tn(G, A, ones_column=True, precision=2)
torch.cuda.synchronize()
