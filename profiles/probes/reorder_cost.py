"""GPU cost of the plan's vertex renumbering (HIP path), piece by piece, HIP-event timed.   NV=200000 LEVELS=3 python profiles/probes/reorder_cost.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import plan as P
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
dev = 'cuda:0'
NV, LV = int(os.environ.get('NV', 200000)), int(os.environ.get('LEVELS', 3))
s = make_synthetic_mesh(NV, LV, seed=0).to(dev)
P.REORDER_MIN = 0
last = LV - 1
edges = [('edge_index', 0)] + [('hierarchy_edge_index_%d' % l, l) for l in range(1, LV)] + [('hierarchy_dil_%d_edge_index_%d' % (d, last), last) for d in (2, 4, 8, 16)]
pools = list(range(1, LV))


def t(f, n=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def order_only():
    pl = P.GraphPlan(s, positions=(6, 9))
    pl._ensure_order()
    return pl


def full():
    pl = P.GraphPlan(s, positions=(6, 9))
    pl.ensure(edges, pools)


P.REORDER = True
print('NV %d levels %d: order only %.0f us' % (NV, LV, t(order_only)))
pl = order_only()
items = [('e', s.edge_index if k == 'edge_index' else s[k], l) for k, l in edges] + [('p', s['hierarchy_trace_index_%d' % l], l) for l in pools]
print('relabel of every index array %.0f us' % t(lambda: pl._relabel_many(items)))
print('plan with renumbering %.0f us' % t(full))
P.REORDER = False
print('plan without %.0f us' % t(full))
