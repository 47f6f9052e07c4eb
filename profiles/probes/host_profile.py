#!/usr/bin/env python3
"""Where the HOST time of a step goes (cProfile over 20 steps, top functions by own and by cumulative time):
   python profiles/probes/host_profile.py [--crops 8 --levels 4 --dtype bf16 | --vertices 200000]"""
import argparse, cProfile, os, pstats, sys, time, io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import CONFIG_3D
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
from surface_texture_inpainting_net_amd.train_step import TrainStep
ap = argparse.ArgumentParser()
ap.add_argument('--crops', type=int, default=0); ap.add_argument('--levels', type=int, default=3)
ap.add_argument('--vertices', type=int, default=200000); ap.add_argument('--dtype', default='f32'); ap.add_argument('--top', type=int, default=28)
a = ap.parse_args()
dev = torch.device('cuda:0')
torch.manual_seed(49)
cfg = dict(CONFIG_3D)
if a.levels != 3: cfg['n_levels'] = a.levels - 1
net = S.define_G(**cfg).to(dev)
if a.dtype == 'bf16': net.set_activation_dtype(torch.bfloat16)
step = TrainStep(net, lr=7e-5, amsgrad=True, freeze_gc=True)
if a.crops:
    from surface_texture_inpainting_net_amd.data import collate
    sizes = [12_000 + (16_000 * i) // max(a.crops - 1, 1) for i in range(a.crops)]
    sample = collate([make_synthetic_mesh(n, a.levels, seed=i) for i, n in enumerate(sizes)]).to(dev)
else:
    sample = make_synthetic_mesh(a.vertices, a.levels, seed=0).to(dev)
pending = [None]
def one():
    sample._plan_cache = pending[0]
    pending[0] = net.build_plan(sample, inputs_ready=True)
    return step(sample)
for _ in range(8): one()
torch.cuda.synchronize()
import gc; gc.collect(); gc.disable()
t0 = time.perf_counter(); c0 = time.thread_time()
for _ in range(20): one()
t1 = time.perf_counter(); c1 = time.thread_time()
torch.cuda.synchronize(); t2 = time.perf_counter()
print('unprofiled: host enqueue %.2f ms/step (thread cpu %.2f), step %.2f ms' % ((t1 - t0) / 20 * 1e3, (c1 - c0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20): one()
pr.disable()
torch.cuda.synchronize()
for key in ('tottime', 'cumulative'):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(a.top)
    lines = s.getvalue().splitlines()
    i = next(k for k, l in enumerate(lines) if l.strip().startswith('ncalls'))
    print('--- by %s (20 steps) ---' % key); print('\n'.join(l[:190] for l in lines[i:i + a.top + 1]))
