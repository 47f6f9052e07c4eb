"""Stand-alone timing of the weight-gradient (TN) products: four-wave kernel (STIN_TN_WS=0) vs the producer / consumer kernel
of csrc/stin_wgrad.hip (STIN_TN_WS=1, STIN_TN_WS_PRIO = 0 | 1 | 2), and the block's whole weight-gradient work as round 2 ran
it (2 x stin_gemm_tn_f32 + unpack = 5 launches) vs stin_edgeconv_wgrad (2 launches).
    python profiles/tn_ws_bench.py [--md out.md]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surface_texture_inpainting_net_amd import _lib, functional as SF  # noqa: E402

DEV = 'cuda:0'
TN_SHAPES = [(18063, 256, 512), (18063, 1024, 256), (18063, 1280, 128), (60211, 128, 256), (60211, 640, 256), (200704, 320, 128)]
BLOCKS = [(18063, 256, 256, False), (18063, 128, 256, True), (60211, 64, 128, True), (60211, 256, 128, True), (200704, 128, 64, True)]


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--md', default=None)
    args = ap.parse_args()
    lib = _lib.load()
    lines = ['# TN (weight-gradient) products stand-alone, us per call incl. the slab reduction (random data, 20 calls back to back)', '',
             '| M | Nc | K | four-wave | ws prio 0 | ws prio 1 (consumers) | ws prio 2 (producers) |', '|---|---|---|---|---|---|---|']
    for (M, Nc, K) in TN_SHAPES:
        G, X = torch.randn(M, Nc, device=DEV), torch.randn(M, K, device=DEV)
        row = []
        for ws, prio in (('0', '0'), ('1', '0'), ('1', '1'), ('1', '2')):
            os.environ['STIN_TN_WS'], os.environ['STIN_TN_WS_PRIO'] = ws, prio
            row.append(timed(lambda: SF.gemm_tn(G, X, ones_column=True, precision=SF.GEMM_BF16X3)))
        lines.append('| %d | %d | %d | %s |' % (M, Nc, K, ' | '.join('%.1f' % t for t in row)))
        print(lines[-1], flush=True)
    lines += ['', '# all weight gradients of one block: round-2 sequence (2 x gemm_tn + unpack, 5 launches) vs stin_edgeconv_wgrad (2 launches)', '',
              '| N | Cin | Cout | shortcut | round-2 sequence, four-wave | round-2 sequence, ws kernel | stin_edgeconv_wgrad |', '|---|---|---|---|---|---|---|']
    for (N, Cin, Cout, sc) in BLOCKS:
        H, Cp = 2 * Cout, Cin
        Yw = 2 * H + (Cout if sc else 0)
        dagg, hE, dY, x = (torch.randn(N, Cout, device=DEV), torch.randn(N, H + 4, device=DEV), torch.randn(N, Yw, device=DEV),
                           torch.randn(N, Cp, device=DEV))
        outs = [torch.empty(H, 2 * Cin, device=DEV), torch.empty(H, device=DEV), torch.empty(Cout, H, device=DEV), torch.empty(Cout, device=DEV),
                torch.empty(Cout, Cin, device=DEV) if sc else None, torch.empty(Cout, device=DEV) if sc else None]

        def old():
            dw2b = SF.gemm_tn(dagg, hE[:, :H], ones_column=True, row_weight=hE[:, H], precision=SF.GEMM_BF16X3)
            dwb = SF.gemm_tn(dY, x, ones_column=True, precision=SF.GEMM_BF16X3)
            SF._call('stin_edgeconv_unpack_grads_f32', SF._ptr(dwb), SF._ptr(dw2b), Cin, Cp, H, Cout, int(sc), 0, SF._ptr(outs[0]),
                     SF._ptr(outs[1]), SF._ptr(outs[4]), SF._ptr(outs[5]), SF._ptr(outs[2]), SF._ptr(outs[3]), SF._stream(x))

        ws_bytes = lib.stin_edgeconv_wgrad_workspace_bytes(N, Cp, H, Cout, int(sc))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)

        def new():
            SF._call('stin_edgeconv_wgrad', 0, SF._ptr(dagg), Cout, SF._ptr(hE), hE.stride(0), SF._ptr(dY), Yw, SF._ptr(x), Cp, N, Cin, Cp,
                     H, Cout, int(sc), 0, SF.GEMM_BF16X3, SF._ptr(outs[0]), SF._ptr(outs[1]), SF._ptr(outs[2]), SF._ptr(outs[3]),
                     SF._ptr(outs[4]), SF._ptr(outs[5]), SF._ptr(ws), ws_bytes, SF._stream(x))

        os.environ['STIN_TN_WS_PRIO'] = '1'
        os.environ['STIN_TN_WS'] = '0'
        t0 = timed(old)
        os.environ['STIN_TN_WS'] = '1'
        t1, t2 = timed(old), timed(new)
        lines.append('| %d | %d | %d | %s | %.1f | %.1f | %.1f |' % (N, Cin, Cout, sc, t0, t1, t2))
        print(lines[-1], flush=True)
    if args.md:
        open(args.md, 'w').write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
