"""Stand-alone timing of stin_gemm_nt_stream_f32 (plain fp32 weights) against stin_gemm_nt_f32 on the tiled kernels (STIN_NT_STREAM=0)
and on the pre-split / fragment-ordered weights of the inpainting net's blocks: python profiles/probes/stream_probe.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from surface_texture_inpainting_net_amd import functional as SF


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    shapes = [(200704, 64, 128), (200704, 128, 64), (200704, 256, 64), (200704, 64, 256), (200704, 128, 256), (200704, 320, 128),
              (60211, 128, 256), (60211, 256, 128), (60211, 640, 64), (60211, 640, 256), (18063, 256, 128), (1200642, 64, 128), (1200642, 128, 64)]
    for m, nc, k in shapes:
        g = torch.Generator().manual_seed(1)
        A = (torch.randn(m, k, generator=g) * 0.7).to('cuda:0')
        W = (torch.randn(nc, k, generator=g) * 0.1).to('cuda:0')
        out = torch.empty(m, nc, device='cuda:0')
        st = SF._stream(A)
        res = []
        for prec in (SF.PREC_FWD, SF.PREC_BWD):
            os.environ['STIN_NT_STREAM'] = '0'
            t_tiled = timeit(lambda: SF.gemm_nt(A, W, precision=prec))
            t_pack = None
            try:
                pw = SF.pack_weight(W, prec) if hasattr(SF, 'pack_weight') else None
            except Exception:
                pw = None
            ts = {}
            for nt in ('2', '4'):
                os.environ['STIN_NT_STREAM_NT'] = nt
                try:
                    ts[nt] = timeit(lambda: SF._call('stin_gemm_nt_stream_f32', SF._ptr(A), k, SF._ptr(W), k, None, None, None, None, m, nc, k,
                                                     SF._ptr(out), nc, int(prec), st))
                except Exception as e:
                    ts[nt] = float('nan')
            res.append('tiled %.0f, stream NT=2 %.0f, NT=4 %.0f' % (t_tiled, ts['2'], ts['4']))
        mb = (m * (k + nc) * 4) / 1e6
        print('%8d x %4d x %4d (%.0f MB, %.0f us at 6.29 TB/s): fwd %s | bwd %s' % (m, nc, k, mb, mb / 6.29, res[0], res[1]), flush=True)


if __name__ == '__main__':
    main()
