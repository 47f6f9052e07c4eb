#!/usr/bin/env python3
"""Tiled vs all-columns NT kernel at one shape for rocprofv3 --pmc passes (SHAPE=M,Nc,K, default 18063,256,1024)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surface_texture_inpainting_net_amd import functional as SF  # noqa: E402

M, Nc, K = (int(v) for v in os.environ.get('SHAPE', '18063,256,1024').split(','))
A = torch.randn(M, K, device='cuda')
W = torch.randn(Nc, K, device='cuda') * 0.05
out = torch.empty(M, Nc, device='cuda')
for fr in (0, 0x400):
    Wp = SF.split_weights(W, SF.GEMM_BF16X3 | fr)
    for _ in range(4):
        SF.gemm_nt(A, Wp, out=out, precision=SF.GEMM_BF16X3 | SF.GEMM_W_PRESPLIT | fr)
torch.cuda.synchronize()
