#!/usr/bin/env python3
"""Training curves (GPU): the same initial weights and the same data through 200 Adam steps in fp32 storage (shipped GEMM
precision), fp32 storage with exact-fp32 GEMMs (the control: how far two fp32 evaluation orders drift apart) and bf16 storage.

    python tests/tools/bf16_training_curve.py [--steps 200] [--vertices 20000]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import functional as SF  # noqa: E402
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S  # noqa: E402
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh  # noqa: E402
from surface_texture_inpainting_net_amd.train_step import TrainStep  # noqa: E402

CFG = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=9,
           n_levels=2, pooling_type='max', dilations=[1, 1, 1, 2, 4, 8, 16, 1, 1], checkpoint_bottleneck=True)


def learnable_samples(vertices, n_masks=4, seed=11, levels=3):
    """One mesh, a SMOOTH colour field over it (a texture a network can actually inpaint) and n_masks different hole masks."""
    base = make_synthetic_mesh(vertices, levels, seed=seed, dilations=(2, 4, 8, 16))
    p = base.x[:, 6:9] * 1.5
    rgb = torch.stack([torch.sin(6.0 * p[:, 0] + 2.0 * p[:, 1]), torch.cos(5.0 * p[:, 1] - 3.0 * p[:, 2]),
                       torch.sin(4.0 * (p[:, 0] + p[:, 2]))], 1) * 0.8
    out = []
    rng = np.random.default_rng(seed)
    n = base.x.shape[0]
    for _ in range(n_masks):
        # holes = balls around random centres (contiguous regions, as the reference's masks are)
        centres = p[rng.integers(0, n, size=12)]
        hole = (torch.cdist(p, centres).min(1).values < 0.08)
        known = (~hole).to(base.x.dtype)[:, None]
        x = base.x.clone()
        x[:, :3] = rgb * known
        x[:, 9:10] = known
        s = type(base)(**{k: base[k] for k in base.keys()})
        s['x'], s['color'], s['mask'] = x, rgb.to(base.x.dtype), hole.long()[:, None]
        out.append(s)
    return out


def learnable_crop_batches(n_batches=4, levels=4, seed=21):
    """BASELINE config 3's shape: batches of 8 unequal crops (1.5-4.5 k vertices each), `levels` graph levels, smooth colour
    fields and hole masks as in learnable_samples."""
    from surface_texture_inpainting_net_amd.data import collate
    rng = np.random.default_rng(seed)
    out = []
    for b in range(n_batches):
        crops = []
        for i in range(8):
            n = int(rng.integers(1500, 4500))
            base = make_synthetic_mesh(n, levels, seed=seed * 100 + b * 8 + i, dilations=(2, 4, 8, 16))
            p = base.x[:, 6:9] * 1.5
            rgb = torch.stack([torch.sin(6.0 * p[:, 0] + 2.0 * p[:, 1]), torch.cos(5.0 * p[:, 1] - 3.0 * p[:, 2]),
                               torch.sin(4.0 * (p[:, 0] + p[:, 2]))], 1) * 0.8
            centres = p[rng.integers(0, base.x.shape[0], size=4)]
            hole = (torch.cdist(p, centres).min(1).values < 0.12)
            known = (~hole).to(base.x.dtype)[:, None]
            x = base.x.clone()
            x[:, :3] = rgb * known
            x[:, 9:10] = known
            s = type(base)(**{k: base[k] for k in base.keys()})
            s['x'], s['color'], s['mask'] = x, rgb.to(base.x.dtype), hole.long()[:, None]
            crops.append(s)
        out.append(collate(crops))
    return out


def curve(samples, steps, mode, seed=3, lr=2e-4, cfg=None):
    torch.manual_seed(seed)
    net = S.define_G(**(cfg or CFG)).to('cuda:0')
    old = SF.PREC_FWD, SF.PREC_BWD
    if mode == 'bf16':
        net.set_activation_dtype(torch.bfloat16)
    elif mode == 'f32-exact-gemm':
        SF.PREC_FWD, SF.PREC_BWD = SF.GEMM_F32, SF.GEMM_F32
    try:
        step = TrainStep(net, lr=lr)
        losses = [step(samples[i % len(samples)]) for i in range(steps)]
        return torch.stack([l.detach().float().reshape(()) for l in losses]).cpu()
    finally:
        SF.PREC_FWD, SF.PREC_BWD = old


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--vertices', type=int, default=20000)
    ap.add_argument('--lr', type=float, default=2e-4)
    ap.add_argument('--seeds', type=int, default=1)
    ap.add_argument('--window', type=int, default=50)
    args = ap.parse_args()
    samples = [s.to('cuda:0') for s in learnable_samples(args.vertices)]
    print('mesh: %d vertices, masked fraction %s' % (samples[0].x.shape[0], ['%.3f' % float((s.mask > 0).float().mean()) for s in samples]))
    w = args.window
    for seed in range(3, 3 + args.seeds):
        curves = {m: curve(samples, args.steps, m, seed=seed, lr=args.lr) for m in ('f32', 'f32-exact-gemm', 'bf16')}
        print('init seed %d' % seed)
        print('%-8s' % 'step' + ''.join('%18s' % m for m in curves))
        for a in range(0, args.steps, w):
            print('%3d-%-4d' % (a, a + w - 1) + ''.join('%18.5f' % float(c[a:a + w].mean()) for c in curves.values()))
        ref = float(curves['f32'][-w:].mean())
        for m, c in curves.items():
            print('%-16s last-%d mean %.5f  (%+.2f %% vs f32)   mean log-loss over the run %.4f' % (
                m, w, float(c[-w:].mean()), 100.0 * (float(c[-w:].mean()) - ref) / ref, float(c.log().mean())))

if __name__ == '__main__':
    main()
