"""Tuning sweep of the segment sum: rows in flight U, non-temporal loads, lanes per row (G = C / 4 / VPL).
(a) the standalone scatter-add of the metric (E = 1.2 M random rows of 256 B -> N = 200 k); (b) the unpool backward shapes of
the headline step (children lists of a synthetic hierarchy, C = 128 / 256)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from surface_texture_inpainting_net_amd import functional as SF
from surface_texture_inpainting_net_amd.plan import PoolMap
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
dev = torch.device('cuda:0')
def timed(fn, it=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it
for k in ('STIN_SEG_U', 'STIN_SEG_NT', 'STIN_SEG_VPL'): os.environ.pop(k, None)
print('scatter-add default', round(bench.scatter_add_standalone(dev)['us'], 1))
for vpl in (1, 2, 4):
    for nt in (0, 1):
        row = []
        for u in (1, 2, 3, 4, 6):
            os.environ.update(STIN_SEG_U=str(u), STIN_SEG_NT=str(nt), STIN_SEG_VPL=str(vpl))
            row.append('%5.1f' % bench.scatter_add_standalone(dev)['us'])
        print('scatter-add C=64 VPL', vpl, 'NT', nt, 'U=1,2,3,4,6:', ' '.join(row), flush=True)
s = make_synthetic_mesh(200_000, 3, seed=0, dilations=()).to(dev)
nv = s.num_vertices.reshape(-1).tolist()
bad = torch.zeros(1, dtype=torch.int32, device=dev)
for lvl, C in ((1, 128), (2, 256)):
    pool = PoolMap(s['hierarchy_trace_index_%d' % lvl], nv[lvl - 1], nv[lvl], bad)
    g = torch.randn(nv[lvl - 1], C, device=dev)
    fn = lambda: SF.segment_sum(g, pool.children.rowptr, pool.children.col, pool.n_coarse, mean=False)
    for k in ('STIN_SEG_U', 'STIN_SEG_NT', 'STIN_SEG_VPL'): os.environ.pop(k, None)
    print('unpool-bwd level', lvl, 'C', C, 'default', round(timed(fn), 1))
    for vpl in (1, 2, 4):
        for nt in (0, 1):
            row = []
            for u in (1, 2, 3, 4, 6):
                os.environ.update(STIN_SEG_U=str(u), STIN_SEG_NT=str(nt), STIN_SEG_VPL=str(vpl))
                row.append('%5.1f' % timed(fn))
            print('   C', C, 'VPL', vpl, 'NT', nt, 'U=1,2,3,4,6:', ' '.join(row), flush=True)
