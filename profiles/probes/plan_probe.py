#!/usr/bin/env python3
"""The headline scene's plan build 20 times (for rocprofv3 --kernel-trace --stats): STIN_PLAN_SORT=0 | 1."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import CONFIG_3D
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
dev = torch.device('cuda:0')
net = S.define_G(**CONFIG_3D).to(dev)
sample = make_synthetic_mesh(200_000, 3, seed=0).to(dev)
for _ in range(20):
    net.build_plan(sample, inputs_ready=True)
    torch.cuda.synchronize()
