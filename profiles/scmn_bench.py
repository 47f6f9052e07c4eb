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
    a = ap.parse_args()
    from surface_texture_inpainting_net_amd.singleconvmeshnet import SingleConvMeshNet
    from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
    torch.manual_seed(0)
    dev = torch.device('cuda:0')
    net = SingleConvMeshNet(10, 2, [64, 128, 256], num_classes=21).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    s = make_synthetic_mesh(a.vertices, 3, seed=4, dilations=()).to(dev)
    tgt = torch.randn(s.x.shape[0], 21, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = (net(s) - tgt).square().mean()
        loss.backward()
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    print(json.dumps({'model': 'SingleConvMeshNet', 'vertices': int(s.x.shape[0]), 'edges': int(s.edge_index.shape[1]),
                      'ms_per_step': round(ms, 3), 'vertices_per_s': round(s.x.shape[0] / ms * 1e3),
                      'peak_gb': round(torch.cuda.max_memory_allocated() / 2**30, 2), 'loss': float(loss)}))


if __name__ == '__main__':
    main()
