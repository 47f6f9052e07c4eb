#!/usr/bin/env python3
"""Which framework (aten) ops launch GPU kernels inside one SingleConvMeshNet training step (profiles/scmn_bench.py's step), and from
where (dispatch-mode trace with the backward pass in the calling thread)."""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch.utils._python_dispatch import TorchDispatchMode
from surface_texture_inpainting_net_amd.singleconvmeshnet import SingleConvMeshNet
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
torch.manual_seed(0)
dev = torch.device('cuda:0')
net = SingleConvMeshNet(10, 2, [64, 128, 256], num_classes=21).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
s = make_synthetic_mesh(200_000, 3, seed=4, dilations=()).to(dev)
tgt = torch.randn(s.x.shape[0], 21, device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = (net(s) - tgt).square().mean()
    loss.backward()
    opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize()
SKIP = ('view', 'as_strided', 'detach', 'slice', 'select', 'alias', 'empty', 't.default', 'expand', 'unsqueeze', 'squeeze', 'transpose',
        'record_stream', 'is_pinned', 'permute', 'reshape', 'unbind', 'split')
class Trace(TorchDispatchMode):
    def __init__(self):
        super().__init__(); self.hits = collections.Counter()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        flat = list(args) + ([out] if torch.is_tensor(out) else list(out) if isinstance(out, (list, tuple)) else [])
        if any(torch.is_tensor(a) and a.is_cuda for a in flat) and not any(t in name for t in SKIP):
            fr = [f for f in traceback.extract_stack() if 'surface_texture' in f.filename]
            where = ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in fr[-2:][::-1]) or 'step (loss / optimizer)'
            self.hits[(name, where)] += 1
        return out
with torch.autograd.set_multithreading_enabled(False):
    with Trace() as tr:
        step()
torch.cuda.synchronize()
print('framework ops with device tensors in one step: %d' % sum(tr.hits.values()))
for (name, where), n in tr.hits.most_common(60):
    print('%3d  %-34s %s' % (n, name, where))
