"""Training-step time of SingleConvMeshNet on one MI355X: 200 k vertices, filters 64/128/256,
2 propagation steps, fp32, Adam.  Usage: python profiles/scmn_bench.py [--vertices N] [--steps K]
(under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--vertices', type=int, default=200_000)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--torch-adam', action='store_true', help='torch.optim.Adam (multi-tensor framework kernels) instead of the '
                    "package's TrainStep (flat gradient bucket + one HIP Adam launch) - rounds 1-4 measured this way")
    a = ap.parse_args()
    from surface_texture_inpainting_net_amd.singleconvmeshnet import SingleConvMeshNet
    from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
    torch.manual_seed(0)
    dev = torch.device('cuda:0')
    net = SingleConvMeshNet(10, 2, [64, 128, 256], num_classes=21).to(dev)
    s = make_synthetic_mesh(a.vertices, 3, seed=4, dilations=()).to(dev)
    tgt = torch.randn(s.x.shape[0], 21, device=dev)
    if a.torch_adam:
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = (net(s) - tgt).square().mean()
            loss.backward()
            opt.step()
            return loss
    else:
        # the package's own step around the model (train_step.TrainStep(loss_fn=...): what the segmentation trainer's loop maps to -
        # gradients written into one flat bucket, Adam as one HIP launch); same loss as above
        from surface_texture_inpainting_net_amd.train_step import TrainStep
        ts = TrainStep(net, lr=1e-3, amsgrad=False, loss_fn=lambda m, smp: (m(smp) - tgt).square().mean())

        def step():
            return ts(s)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    out = {'model': 'SingleConvMeshNet', 'optimizer': 'torch.optim.Adam' if a.torch_adam else 'TrainStep (flat bucket + HIP Adam)',
           'vertices': int(s.x.shape[0]), 'edges': int(s.edge_index.shape[1]),
           'ms_per_step': round(ms, 3), 'vertices_per_s': round(s.x.shape[0] / ms * 1e3),
           'peak_gb': round(torch.cuda.max_memory_allocated() / 2**30, 2), 'loss': float(loss)}
    # roofline of the level-0 kernels (HIP events around the launches of one more step; algorithmic bytes: every gathered row
    # charged once per edge, int32 indices) and of the per-edge GEMM (executed 16-bit MFMA flops, 3 per fp32 product)
    from surface_texture_inpainting_net_amd import functional as SF
    names = ['stin_gather_add_rows_f32', 'stin_gather_add_rows_stats_f32', 'stin_segment_sum_f32', 'stin_segment_mean_stats_f32',
             'stin_gemm_nt_bn_f32', 'stin_gemm_tn_bn_f32', 'stin_gemm_nt_bn_bwd_stats_f32', 'stin_gemm_nt_bn_bwd_apply_f32']
    SF.KernelTimer.start(names, max_records=4000)
    step()
    times = SF.KernelTimer.stop()
    n0, e0 = int(s.x.shape[0]), int(s.edge_index.shape[1])
    roof = {}
    for (name, tag), ts in times.items():
        t = sum(ts) / len(ts)
        if name in ('stin_gather_add_rows_f32', 'stin_gather_add_rows_stats_f32') and tag and tag[0] == e0:
            e, h = tag
            nbytes = 3 * e * h * 4 + 8 * e                 # SURVEY 8(d) convention: both gathered rows charged once per EDGE
            # (round 6) ... but the two gathered operands are [N, H] matrices (102 MB each at 200 k vertices): Infinity-Cache resident, so
            # the per-edge convention counts bytes that never reach HBM and the rate it gives can exceed the HBM peak.  The fraction
            # below uses the COMPULSORY bytes - each operand row once, every output row once, the indices - which is what the HBM has to
            # move; the per-edge figure is kept as a cache-side rate without a fraction.  Fabric bytes (PMC): profiles/r06_scmn.md.
            comp = e * h * 4 + 2 * n0 * h * 4 + 8 * e
            roof['%s[E=%d,H=%d]' % (name, e, h)] = {'avg_us': t * 1e6, 'launches': len(ts), 'algorithmic_MB_per_edge_convention': nbytes / 1e6,
                                                            'GBps_per_edge_convention_cache_side': nbytes / t / 1e9, 'compulsory_MB': comp / 1e6,
                                                            'GBps': comp / t / 1e9, 'frac_of_hbm_peak': comp / t / 1e9 / 8000.0}
        if name in ('stin_segment_sum_f32', 'stin_segment_mean_stats_f32') and tag and tag[0] == e0 and tag[1] == n0:
            e, n, c = tag
            nbytes = e * c * 4 + n * c * 4 + 4 * e + 4 * (n + 1)
            roof['%s[E=%d,N=%d,C=%d]' % (name, e, n, c)] = {'avg_us': t * 1e6, 'launches': len(ts), 'algorithmic_MB': nbytes / 1e6,
                                                                 'GBps': nbytes / t / 1e9, 'frac_of_hbm_peak': nbytes / t / 1e9 / 8000.0}
        if name in ('stin_gemm_nt_bn_f32', 'stin_gemm_tn_bn_f32') and tag and tag[0] == e0:
            m, nc, k = tag
            flops = 2.0 * m * nc * k * 3
            minb = 4.0 * (m * k + m * nc)
            roof['%s[M=%d,Nc=%d,K=%d]' % (name, m, nc, k)] = {
                'avg_us': t * 1e6, 'launches': len(ts), 'mfma_TFLOPs_executed': flops / t / 1e12, 'frac_of_16bit_mfma_peak': flops / t / 1e12 / 2500.0,
                'min_bytes_MB': minb / 1e6, 'GBps_min_traffic': minb / t / 1e9, 'frac_of_hbm_peak': minb / t / 1e9 / 8000.0,
                'bound': 'hbm (operand bytes): the product of an [E, 2 cout] by a [cout, 2 cout] matrix moves 4 (K + Nc) bytes per 6 K Nc flops'}
        if name in ('stin_gemm_nt_bn_bwd_stats_f32', 'stin_gemm_nt_bn_bwd_apply_f32') and tag and tag[0] == e0:
            m, nc, k = tag                       # A [M, K] and X [M, Nc] read (+ dx [M, Nc] written by `apply`); the statistics fold rides in `stats`
            flops = 2.0 * m * nc * k * 3
            minb = 4.0 * (m * k + m * nc * (2 if name.endswith('apply_f32') else 1))
            roof['%s[M=%d,Nc=%d,K=%d]' % (name, m, nc, k)] = {
                'avg_us': t * 1e6, 'launches': len(ts), 'mfma_TFLOPs_executed': flops / t / 1e12, 'min_bytes_MB': minb / 1e6,
                'GBps_min_traffic': minb / t / 1e9, 'frac_of_hbm_peak': minb / t / 1e9 / 8000.0, 'bound': 'hbm (operand bytes)'}
    out['roofline'] = roof
    print(json.dumps(out))


if __name__ == '__main__':
    main()
