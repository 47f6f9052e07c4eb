"""Per-(entry point, shape) HIP-event timings of one SingleConvMeshNet training step (all levels): python profiles/probes/scmn_shapes.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from surface_texture_inpainting_net_amd import _lib
from surface_texture_inpainting_net_amd import functional as SF
from surface_texture_inpainting_net_amd.singleconvmeshnet import SingleConvMeshNet
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
from surface_texture_inpainting_net_amd.train_step import TrainStep


def main():
    torch.manual_seed(0)
    s = make_synthetic_mesh(200_000, 3, seed=0, dilations=()).to('cuda:0')
    net = SingleConvMeshNet(10, 2, [64, 128, 256], num_classes=5).to('cuda:0')
    tgt = torch.randn(s.x.shape[0], 5, device='cuda:0')
    ts = TrainStep(net, lr=1e-3, amsgrad=False, loss_fn=lambda m, smp: (m(smp) - tgt).square().mean())
    for _ in range(3):
        ts(s)
    torch.cuda.synchronize()
    names = [n for n in _lib.SIGNATURES if n.endswith('_f32') or n.endswith('_i64')]
    SF.KernelTimer.start(names, max_records=8000)
    ts(s)
    times = SF.KernelTimer.stop()
    rows = sorted(((sum(v), len(v), k) for k, v in times.items()), reverse=True)
    tot = sum(r[0] for r in rows)
    print('bracketed launches: %.2f ms' % (tot * 1e3))
    for t, n, (name, tag) in rows[:60]:
        print('%8.1f us  x%-3d %7.1f avg  %s %s' % (t * 1e6, n, t / n * 1e6, name, tag))


if __name__ == '__main__':
    main()
