#!/usr/bin/env python3
"""Which framework (aten) ops launch GPU kernels inside one headline training step, and from where: torch.profiler with Python
stacks over one step after warm-up.   python profiles/probes/aten_ops.py [--vertices N] [--crops K --levels L --dtype bf16]"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import CONFIG_3D  # noqa: E402
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S  # noqa: E402
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh  # noqa: E402
from surface_texture_inpainting_net_amd.train_step import TrainStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--vertices', type=int, default=200_000)
ap.add_argument('--crops', type=int, default=0)
ap.add_argument('--levels', type=int, default=3)
ap.add_argument('--dtype', default='f32')
a = ap.parse_args()
dev = torch.device('cuda:0')
torch.manual_seed(49)
cfg = dict(CONFIG_3D)
if a.levels != 3:
    cfg['n_levels'] = a.levels - 1
net = S.define_G(**cfg).to(dev)
if a.dtype == 'bf16':
    net.set_activation_dtype(torch.bfloat16)
step = TrainStep(net, lr=7e-5, amsgrad=True)
if a.crops:
    from surface_texture_inpainting_net_amd.data import collate  # noqa: E402
    sizes = [12_000 + (16_000 * i) // max(a.crops - 1, 1) for i in range(a.crops)]
    sample = collate([make_synthetic_mesh(n, a.levels, seed=i) for i, n in enumerate(sizes)]).to(dev)
else:
    sample = make_synthetic_mesh(a.vertices, a.levels, seed=0).to(dev)
pending = None
def one():
    global pending
    sample._plan_cache = pending
    pending = net.build_plan(sample, inputs_ready=True)
    return step(sample)
for _ in range(6):
    one()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode


class Trace(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.hits = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        out = func(*args, **(kwargs or {}))
        big = any(torch.is_tensor(a) and a.is_cuda for a in list(args) + [out])
        if big and not any(t in name for t in ('view', 'as_strided', 'detach', 'slice', 'select', 'alias', 'empty', 't.default', 'expand', 'unsqueeze', 'squeeze', 'transpose', 'record_stream', 'is_pinned', 'permute', 'reshape', 'unbind', 'split')):
            fr = [f for f in traceback.extract_stack() if 'surface_texture' in f.filename or 'bench' in f.filename]
            where = ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in fr[-3:][::-1])
            self.hits[(name, where)] += 1
        return out


# (the backward pass runs in autograd's thread: dispatch modes are thread-local, so the forward and the optimizer are what this
# sees; backward-side framework ops are listed by the profiler pass below)
# backward in the calling thread, so that the dispatch mode (thread-local) sees the backward-side framework ops too
with torch.autograd.set_multithreading_enabled(False):
    with Trace() as tr:
        one()
torch.cuda.synchronize()
for (name, where), n in tr.hits.most_common(40):
    print('%3d  %-34s %s' % (n, name, where))
print('---- profiler (whole step incl. backward):')
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    one()
    torch.cuda.synchronize()
rows = []
for ka in prof.key_averages(group_by_stack_n=12):
    dt = getattr(ka, 'self_device_time_total', None)
    if dt is None:
        dt = getattr(ka, 'self_cuda_time_total', 0)
    if not ka.key.startswith('aten::') or dt <= 0:
        continue
    frames = [f.strip() for f in (ka.stack or []) if ('surface_texture' in f or 'bench.py' in f or '_aten_ops' in f)]
    rows.append((ka.count, ka.key, dt, (frames[0] if frames else (ka.stack[0].strip() if ka.stack else '?'))[-120:]))
for n, name, dt, where in sorted(rows, key=lambda r: -r[0]):
    print('%3d  %-26s %7.1f us  %s' % (n, name, dt, where))
