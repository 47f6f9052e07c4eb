#!/usr/bin/env python3
"""What unequal scene sizes cost a synchronous data-parallel step, and what size-balanced sharding buys (round 5, review item 8a).

No multi-GPU node is available to the builder, and eight ranks time-sharing ONE GPU say nothing about per-rank step times, so
this is a measured MODEL: the headline step is timed on this GPU at several scene sizes (the only measured input), a line is
fitted through (vertices, ms), and the makespan of an epoch - sum over steps of the MAX over the 8 ranks - is evaluated for
(a) 8 unequal scenes of 150-200 k vertices in one step (BASELINE config 4 as the reference trains it), (b) an epoch of a
ScanNet-like size distribution under torch's DistributedSampler order and under loader.shard_indices(sizes=...).

    python profiles/straggler_model.py > profiles/rNN_straggler.json"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIG_3D  # noqa: E402
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S  # noqa: E402
from surface_texture_inpainting_net_amd.loader import shard_indices  # noqa: E402
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh  # noqa: E402
from surface_texture_inpainting_net_amd.train_step import TrainStep  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(49)
net = S.define_G(**CONFIG_3D).to(dev)
step = TrainStep(net, lr=7e-5, amsgrad=True)
points = []
for n in (100_000, 150_000, 175_000, 200_000, 250_000, 300_000):
    s = make_synthetic_mesh(n, 3, seed=n % 7).to(dev)
    for _ in range(4):
        s._plan_cache = None
        step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        s._plan_cache = None
        step(s)
    torch.cuda.synchronize()
    points.append((int(s.x.shape[0]), (time.perf_counter() - t0) / 10 * 1e3))
step.finish()
xs, ys = torch.tensor([p[0] for p in points], dtype=torch.float64), torch.tensor([p[1] for p in points], dtype=torch.float64)
A = torch.stack([xs, torch.ones_like(xs)], 1)
slope, icpt = torch.linalg.lstsq(A, ys[:, None]).solution.view(-1).tolist()
t = lambda n: slope * n + icpt                                     # noqa: E731 - ms per step of a scene of n vertices

world = 8
one_step = [150_000 + (50_000 * r) // (world - 1) for r in range(world)]
ts = [t(n) for n in one_step]
out = {'measured_ms_per_step': [{'vertices': n, 'ms': round(ms, 3)} for n, ms in points],
       'fit': {'ms_per_vertex': slope, 'intercept_ms': icpt, 'max_residual_ms': float((A @ torch.tensor([slope, icpt], dtype=torch.float64) - ys).abs().max())},
       'one_step_of_8_unequal_scenes_150k_200k': {'vertices': one_step, 'step_ms': max(ts), 'mean_rank_ms': sum(ts) / world,
                                                   'max_over_mean': max(ts) / (sum(ts) / world),
                                                   'vertices_per_s_8_gpus': sum(one_step) / (max(ts) * 1e-3),
                                                   'vertices_per_s_if_balanced': sum(one_step) / (sum(ts) / world * 1e-3)}}
# an epoch of a ScanNet-like size distribution (1 201 training scenes, log-normal around 150 k vertices, clipped to 50-400 k)
g = torch.Generator().manual_seed(0)
sizes = (torch.randn(1201, generator=g) * 0.45 + 11.9).exp().clamp(50_000, 400_000).round().tolist()
for name, kw in (('distributed_sampler_order', {}), ('size_balanced_sharding', {'sizes': sizes})):
    per_rank = [shard_indices(len(sizes), epoch=0, seed=0, rank=r, world_size=world, **kw) for r in range(world)]
    steps = list(zip(*per_rank))
    makespan = sum(max(t(sizes[i]) for i in st) for st in steps)
    work = sum(sum(t(sizes[i]) for i in st) for st in steps) / world
    out[name] = {'steps': len(steps), 'epoch_ms': makespan, 'mean_rank_busy_ms': work, 'max_over_mean': makespan / work,
                 'vertices_per_s_8_gpus': sum(sizes[i] for st in steps for i in st) / (makespan * 1e-3)}
out['note'] = ('model: t(n) fitted to the measured single-GPU steps above; a synchronous step costs max over ranks of t(n_rank) (gradient '
               'all-reduce not included: 16.8 MB over xGMI); size distribution of the epoch is synthetic (log-normal, 50-400 k vertices)')
print(json.dumps(out, indent=1))
