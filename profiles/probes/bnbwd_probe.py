"""Stand-alone timing of SingleConvMeshNet's per-edge backward products: the three-launch route (stin_gemm_nt_f32 +
stin_colreduce_f32(DOT_BN_RELU) + stin_bn_act_bwd_f32) against stin_gemm_nt_bn_bwd_{stats,apply}_f32.
python profiles/probes/bnbwd_probe.py [E h2 cout]"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from surface_texture_inpainting_net_amd import _lib
from surface_texture_inpainting_net_amd import functional as SF
from surface_texture_inpainting_net_amd.singleconvmeshnet import _all_rows


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
    shapes = [(1_200_642, 128, 64), (361_000, 256, 128), (108_000, 512, 256)]
    if len(sys.argv) == 4:
        shapes = [tuple(int(v) for v in sys.argv[1:4])]
    lib = _lib.load()
    for e, h2, cout in shapes:
        g = torch.Generator().manual_seed(1)
        dm = (torch.randn(e, cout, generator=g) * 0.3).to('cuda:0')
        w2T = (torch.randn(h2, cout, generator=g) * 0.1).to('cuda:0')
        pre = (torch.randn(e, h2, generator=g) * 1.1 + 0.2).to('cuda:0')
        gb = torch.stack([torch.rand(h2, generator=g) + 0.5, torch.randn(h2, generator=g) * 0.3]).to('cuda:0')
        ge = _all_rows(e, pre.device)
        mean, rstd = SF.colreduce(SF.RED_MOMENTS, pre, ge, ge.ptr_sum, eps=1e-5)
        st = SF._stream(pre)
        dh = torch.empty(e, h2, device='cuda:0')
        out = torch.empty(e, h2, device='cuda:0')
        res = {}
        res['gemm_nt'] = timeit(lambda: SF.gemm_nt(dm, w2T, precision=SF.PREC_BWD))
        dh = SF.gemm_nt(dm, w2T, precision=SF.PREC_BWD)
        res['colreduce'] = timeit(lambda: SF.colreduce(SF.RED_DOT_BN_RELU, pre, ge, ge.ptr_sum, gout=dh, mean=mean, rstd=rstd, coef=gb))
        P0, Q0 = SF.colreduce(SF.RED_DOT_BN_RELU, pre, ge, ge.ptr_sum, gout=dh, mean=mean, rstd=rstd, coef=gb)
        res['bn_bwd'] = timeit(lambda: SF._call('stin_bn_act_bwd_f32', SF._ptr(pre), h2, SF._ptr(dh), h2, SF._ptr(mean), SF._ptr(rstd),
                                                SF._ptr(gb[0]), SF._ptr(gb[1]), SF._ptr(P0), SF._ptr(Q0), 1.0 / e, e, h2, 1, SF._ptr(out), h2, st))
        line = '%d x %d x %d: three launches %.0f + %.0f + %.0f = %.0f us' % (e, h2, cout, res['gemm_nt'], res['colreduce'], res['bn_bwd'],
                                                                                  res['gemm_nt'] + res['colreduce'] + res['bn_bwd'])
        print(line, flush=True)
        args = (SF._ptr(dm), cout, SF._ptr(w2T), cout, SF._ptr(pre), h2, SF._ptr(mean), SF._ptr(rstd), SF._ptr(gb[0]), SF._ptr(gb[1]))
        for bn in ['2', '4']:
            for occ in os.environ.get('PROBE_OCC', '1,2').split(','):
                os.environ['STIN_NT_STREAM_NT'] = bn
                os.environ['STIN_NT_STREAM_BPC'] = occ
                groups = int(lib.stin_gemm_nt_bn_bwd_groups(e, h2, cout, int(SF.PREC_BWD)))
                partial = torch.empty(groups, 2, h2, dtype=torch.float64, device='cuda:0')
                pq = torch.empty(2, h2, device='cuda:0')
                t0 = timeit(lambda: SF._call('stin_gemm_nt_bn_bwd_stats_f32', *args, e, h2, cout, int(SF.PREC_BWD), SF._ptr(partial),
                                             partial.numel() * 8, SF._ptr(pq), st))
                t1 = timeit(lambda: SF._call('stin_gemm_nt_bn_bwd_apply_f32', *args, SF._ptr(pq), 1.0 / e, e, h2, cout, SF._ptr(out), h2,
                                             int(SF.PREC_BWD), st))
                print('   %s column tiles, <= %s blocks / CU (groups %d): stats (+ fold) %.0f us, apply %.0f us, sum %.0f' % (bn, occ, groups, t0, t1, t0 + t1),
                      flush=True)


if __name__ == '__main__':
    main()
