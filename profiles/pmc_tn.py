#!/usr/bin/env python3
"""One TN (weight-gradient) product for rocprofv3 --pmc passes: four-wave kernel then the producer / consumer kernel
(SHAPE=M,Nc,K, default 18063,1024,256)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surface_texture_inpainting_net_amd import functional as SF  # noqa: E402

M, Nc, K = (int(v) for v in os.environ.get('SHAPE', '18063,1024,256').split(','))
G = torch.randn(M, Nc, device='cuda')
X = torch.randn(M, K, device='cuda')
for ws in ('0', '1'):
    os.environ['STIN_TN_WS'] = ws
    for _ in range(4):
        SF.gemm_tn(G, X, ones_column=True, precision=SF.GEMM_BF16X3)
torch.cuda.synchronize()
