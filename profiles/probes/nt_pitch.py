"""Does the row pitch of A matter?  M x K fp32 with K = 1024 has a 4 KB pitch: every block reads the same 256-byte column window
of its rows at the same time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import functional as SF
dev = torch.device('cuda:0')
M, Nc, K = 18063, 256, 1024
FR = 0x400
W = torch.randn(Nc, K, device=dev) * 0.05
def t(f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for pad in (0, 16, 64, 96):
    big = torch.randn(M, K + pad, device=dev)
    A = big[:, :K]
    out = torch.empty(M, Nc, device=dev)
    for fr in (0, FR):
        Wp = SF.split_weights(W, SF.PREC_BWD | fr)
        us = t(lambda: SF.gemm_nt(A, Wp, out=out, precision=SF.PREC_BWD | SF.GEMM_W_PRESPLIT | fr))
        print('pitch %d B  %s kernel: %.1f us' % ((K + pad) * 4, 'wide' if fr else 'tiled', us))
# host cost of one call
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): SF.gemm_nt(A, Wp, out=out, precision=SF.PREC_BWD | SF.GEMM_W_PRESPLIT | FR)
t1 = time.perf_counter(); torch.cuda.synchronize()
print('host enqueue per call %.1f us' % ((t1 - t0) / 200 * 1e6))
