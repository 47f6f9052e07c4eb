#!/usr/bin/env python3
"""Does an evaluation pass (model.eval(), torch.no_grad()) between training steps slow the steps after it?  (round 6: the bench's
vertex_locality leg read 20 ms right behind the first version of the inference leg)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from surface_texture_inpainting_net_amd import functional as SF  # noqa: E402
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S  # noqa: E402
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh, renumber_by_locality  # noqa: E402
from surface_texture_inpainting_net_amd.train_step import TrainStep  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(49)
net = S.define_G(**bench.CONFIG_3D).to(dev)
step = TrainStep(net, lr=7e-5, amsgrad=True, freeze_gc=True)
sample = make_synthetic_mesh(200_000, 3, seed=0).to(dev)
other = renumber_by_locality(make_synthetic_mesh(200_000, 3, seed=0))[0].to(dev)


def timed(tag, fn, n=10):
    torch.cuda.synchronize()
    c0 = SF.NetFn.calls
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print('%-46s %.2f ms / call, NetFn calls %d, reserved %.2f GB, allocated %.2f GB' % (
        tag, (time.perf_counter() - t) / n * 1e3, SF.NetFn.calls - c0, torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_allocated() / 2**30), flush=True)


def train(s):
    s._plan_cache = None
    step(s)


def evaluate(s):
    with torch.no_grad():
        net(s)


timed('train (warm-up)', lambda: train(sample))
timed('train', lambda: train(sample))
net.eval()
timed('eval, no_grad', lambda: evaluate(sample))
net.train()
timed('train right after eval', lambda: train(sample))
timed('train, other sample', lambda: train(other))
net.eval()
timed('eval, no_grad', lambda: evaluate(sample))
net.train()
timed('train, other sample right after eval', lambda: train(other))
timed('train, other sample', lambda: train(other))
