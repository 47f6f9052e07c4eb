"""Host profile of the bench's own step at the headline size (plan prefetch one step ahead + TrainStep), cumulative view."""
import cProfile, pstats, os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
from surface_texture_inpainting_net_amd.train_step import TrainStep
CFG = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=9,
           n_levels=2, pooling_type='max', dilations=[1, 1, 1, 2, 4, 8, 16, 1, 1], checkpoint_bottleneck=True)
torch.manual_seed(49)
net = S.define_G(**CFG).to('cuda:0')
net.plan_validation = 'deferred'
step = TrainStep(net)
s = make_synthetic_mesh(int(os.environ.get('NV', 200000)), 3, seed=0).to('cuda:0')
pend = [None]
def one():
    s._plan_cache = pend[0]
    pend[0] = net.build_plan(s)
    return step(s)
for _ in range(5): one()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(30): one()
te = time.perf_counter() - t
torch.cuda.synchronize()
print('enqueue %.2f ms/step, total %.2f ms/step' % (te / 30 * 1e3, (time.perf_counter() - t) / 30 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(30): one()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(40)
