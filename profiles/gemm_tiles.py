"""Tile sweep of the nt GEMM kernels over the shapes of the shipped 3-D network (tuning aid):
STIN_NT_TILE = 1 (128x128), 2 (128x64), 3 (64x64) vs the built-in rule (0).  Usage on the GPU box:
    python profiles/gemm_tiles.py > gpurun_out/gemm_tiles.log"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surface_texture_inpainting_net_amd import functional as SF   # noqa: E402

DEV = 'cuda:0'
SHAPES = [  # (M, Nc, K)
    (200704, 320, 16), (200704, 64, 128), (200704, 128, 64), (200704, 320, 128), (200704, 128, 320), (200704, 256, 64),
    (200704, 64, 256), (200704, 64, 64),
    (60211, 640, 64), (60211, 128, 256), (60211, 256, 128), (60211, 64, 640), (60211, 640, 256), (60211, 256, 640),
    (18063, 1280, 128), (18063, 256, 512), (18063, 512, 256), (18063, 128, 1280), (18063, 1024, 256), (18063, 256, 1024)]


def tm(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


for mode in ('bf16', 'f16x3', 'bf16x3'):
    print('---', mode, ' us per tile setting [rule, 128x128, 128x64, 64x64]')
    for (M, Nc, K) in SHAPES:
        A = torch.randn(M, K, device=DEV)
        W = torch.randn(Nc, K, device=DEV) * 0.05
        if mode == 'bf16':
            A = A.to(torch.bfloat16)
            out = torch.empty(M, Nc, device=DEV, dtype=torch.bfloat16)
            f = lambda: SF.gemm_nt(A, W, out=out)
        else:
            out = torch.empty(M, Nc, device=DEV)
            prec = SF.GEMM_F16X3 if mode == 'f16x3' else SF.GEMM_BF16X3
            f = lambda: SF.gemm_nt(A, W, out=out, precision=prec)
        row = []
        for tile in (0, 1, 2, 3):
            os.environ['STIN_NT_TILE'] = str(tile)
            row.append(tm(f))
        os.environ['STIN_NT_TILE'] = '0'
        best = min(range(1, 4), key=lambda i: row[i])
        print((M, Nc, K), ' '.join('%.1f' % r for r in row), ' best', ('128x128', '128x64', '64x64')[best - 1],
              '%.2fx vs rule' % (row[0] / row[best]))
