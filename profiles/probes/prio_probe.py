"""Does running the compute stream at high HIP priority protect the critical path from the side-stream work?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import surface_texture_inpainting_net_amd  # noqa
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
from surface_texture_inpainting_net_amd.train_step import TrainStep
CFG = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=9,
           n_levels=2, pooling_type='max', dilations=[1, 1, 1, 2, 4, 8, 16, 1, 1], checkpoint_bottleneck=True)
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else None)
torch.manual_seed(0)
net = S.define_G(**CFG).to('cuda:0')
step = TrainStep(net, lr=7e-5)
s = make_synthetic_mesh(200000, 3, seed=0).to('cuda:0')
def run(stream, n=30):
    pend = [None]
    def one():
        s._plan_cache = pend[0]
        pend[0] = net.build_plan(s)
        return step(s)
    with torch.cuda.stream(stream):
        for _ in range(5): one()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n): l = one()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3, float(l)
for name, st in (('default stream', torch.cuda.current_stream()), ('high-priority stream', torch.cuda.Stream(priority=-1)),
                 ('default stream', torch.cuda.current_stream()), ('high-priority stream', torch.cuda.Stream(priority=-1))):
    print(name, '%.3f ms/step loss %.6f' % run(st))
