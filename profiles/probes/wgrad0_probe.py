#!/usr/bin/env python3
"""The first block's weight-gradient call (200 704 x [320 x 12 skinny + 64 x 128], ~900 row chunks) stand-alone: us per call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import functional as SF, _lib
lib = _lib.load()
DEV = 'cuda:0'
for (N, Cin, Cp, H, Cout, sc, ti) in ((200704, 10, 12, 128, 64, True, 1), (200704, 64, 64, 128, 64, False, 0), (18063, 256, 256, 512, 256, False, 0)):
    Yw = 2 * H + (Cout if sc else 0)
    dagg, hE, dY, x = (torch.randn(N, Cout, device=DEV), torch.randn(N, H + 4, device=DEV), torch.randn(N, Yw, device=DEV), torch.randn(N, Cp, device=DEV))
    outs = [torch.empty(H, Cin if ti else 2 * Cin, device=DEV), torch.empty(H, device=DEV), torch.empty(Cout, H, device=DEV), torch.empty(Cout, device=DEV),
            torch.empty(Cout, Cin, device=DEV) if sc else None, torch.empty(Cout, device=DEV) if sc else None]
    ws_bytes = lib.stin_edgeconv_wgrad_workspace_bytes(N, Cp, H, Cout, int(sc))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    def f():
        SF._call('stin_edgeconv_wgrad', 0, SF._ptr(dagg), Cout, SF._ptr(hE), hE.stride(0), SF._ptr(dY), Yw, SF._ptr(x), Cp, N, Cin, Cp, H, Cout,
                 int(sc), ti, SF.GEMM_BF16X3, SF._ptr(outs[0]), SF._ptr(outs[1]), SF._ptr(outs[2]), SF._ptr(outs[3]), SF._ptr(outs[4]), SF._ptr(outs[5]),
                 SF._ptr(ws), ws_bytes, SF._stream(x))
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    print('N=%d Cin=%d H=%d Cout=%d: %.1f us per call' % (N, Cin, H, Cout, a.elapsed_time(b) / 20 * 1e3), flush=True)
