"""Where does the bf16-storage error come from?  Per-block relative error of the bf16 network against the fp32 one."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
CFG = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=9,
           n_levels=2, pooling_type='max', dilations=[1, 1, 1, 2, 4, 8, 16, 1, 1], checkpoint_bottleneck=True)
torch.manual_seed(49)
net = S.define_G(**CFG).to('cuda:0')
s = make_synthetic_mesh(20000, 3, seed=3).to('cuda:0')
outs = {}
def hook(name):
    def f(m, i, o):
        outs.setdefault(name, []).append(o.detach().float().clone())
    return f
for n, m in net.named_modules():
    if isinstance(m, S.GraphResnetBlock):
        m.register_forward_hook(hook(n))
with torch.no_grad():
    a = net(s)
    net.set_activation_dtype(torch.bfloat16)
    b = net(s)
for n, (x, y) in outs.items():
    print('%-22s rel-L2 %.3e  max-abs %.3e (|x| max %.2f rms %.2f)' % (n, float((x - y).norm() / x.norm()), float((x - y).abs().max()), float(x.abs().max()), float(x.pow(2).mean().sqrt())))
print('output max-abs', float((a - b.float()).abs().max()))
