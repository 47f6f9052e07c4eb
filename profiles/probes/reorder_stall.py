import os, sys, time, torch, gc
sys.path.insert(0, os.getcwd())
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S, plan as P
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
from surface_texture_inpainting_net_amd.train_step import TrainStep
NV=int(os.environ.get('NV',1000000)); LV=int(os.environ.get('LEVELS',5))
CFG = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=9, n_levels=LV-1, pooling_type='max', dilations=[1, 1, 1, 2, 4, 8, 16, 1, 1], checkpoint_bottleneck=True)
torch.manual_seed(49)
net = S.define_G(**CFG).to('cuda:0'); net.set_activation_dtype(torch.bfloat16)
step = TrainStep(net)
s = make_synthetic_mesh(NV, LV, seed=0).to('cuda:0')
pend=[None]
tb=[]; ts=[]
def one():
    s._plan_cache = pend[0]
    t0=time.perf_counter(); pend[0] = net.build_plan(s); t1=time.perf_counter()
    step(s); t2=time.perf_counter()
    tb.append(t1-t0); ts.append(t2-t1)
for _ in range(4): one()
torch.cuda.synchronize(); tb.clear(); ts.clear()
t=time.perf_counter()
for _ in range(6): one()
torch.cuda.synchronize()
print('reorder', P.REORDER, 'total %.1f ms/step' % ((time.perf_counter()-t)/6*1e3), 'build_plan host ms', [round(x*1e3,1) for x in tb], 'step host ms', [round(x*1e3,1) for x in ts])
print(torch.cuda.memory_stats()['num_alloc_retries'], torch.cuda.memory_stats()['num_device_alloc'] if 'num_device_alloc' in torch.cuda.memory_stats() else '', round(torch.cuda.memory_reserved()/2**30,1), 'GiB reserved')
