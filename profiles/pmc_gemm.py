#!/usr/bin/env python3
"""The GEMM kernels the training step actually runs, at one headline shape, for rocprofv3 SQ-counter passes:

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv ...   (pass 1)
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv ...                   (pass 2)

SHAPE=M,Nc,K (default 18063,1024,256).  profiles/pmc_gemm_summarize.py condenses the two CSVs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surface_texture_inpainting_net_amd import functional as SF  # noqa: E402

M, Nc, K = (int(v) for v in os.environ.get('SHAPE', '18063,1024,256').split(','))
A = torch.randn(M, K, device='cuda')
W = torch.randn(Nc, K, device='cuda') * 0.05
G = torch.randn(M, Nc, device='cuda')
Wf = SF.split_weights(W, SF.GEMM_F16X3)
Wb = SF.split_weights(W, SF.GEMM_BF16X3)
A16, G16 = A.bfloat16(), G.bfloat16()
for _ in range(3):
    SF.gemm_nt(A, Wf, None, precision=SF.GEMM_F16X3 | SF.GEMM_W_PRESPLIT)       # forward GEMMs of the fp32 path
    SF.gemm_nt(A, Wb, None, precision=SF.GEMM_BF16X3 | SF.GEMM_W_PRESPLIT)      # dgrad
    SF.gemm_tn(G, A, ones_column=True, precision=SF.GEMM_BF16X3)                # wgrad
    SF.gemm_nt(A, W, None, precision=SF.GEMM_F32)                               # exact fp32 MFMA chain (reference point)
    SF.gemm_nt(A16, W, None)                                                    # bf16-storage family
    SF.gemm_tn(G16, A16, ones_column=True)
torch.cuda.synchronize()
