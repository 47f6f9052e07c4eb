#!/usr/bin/env python3
"""How long a rocPRIM radix sort of plan-build size takes (through stin_coalesce_pairs_i64: 64-bit keys, all 64 bits) vs the shipped
counting-sort plan build of the headline scene."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import preprocessing as P
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import CONFIG_3D
dev = torch.device('cuda:0')
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for E, n in ((1_200_000, 200_704), (2_100_000, 200_704), (4_200_000, 400_000)):
    ei = torch.randint(0, n, (2, E), device=dev)
    print('coalesce (64-bit radix sort + unique) of %d pairs: %.1f us' % (E, t(lambda: P.coalesce(ei, n))), flush=True)
net = S.define_G(**CONFIG_3D).to(dev)
sample = make_synthetic_mesh(200_000, 3, seed=0).to(dev)
def build():
    p = net.build_plan(sample, inputs_ready=True)
    torch.cuda.current_stream().wait_stream(torch.cuda.current_stream())
    return p
print('shipped plan build of the headline scene (all CSRs, side streams, incl. host): %.1f us' % t(lambda: (build(), torch.cuda.synchronize())), flush=True)
