#!/usr/bin/env python3
"""The CPU oracle (fwd + loss + bwd, unfused PyG-form restatement) on the HEADLINE mesh itself (200 704 vertices, 3 levels): one
warm-up + median of 3 passes at a fixed thread count (default 16 = the probe-best count of bench.py's cpu_baseline on the
EPYC 9575F boxes).  bench.py times a bounded ~35 k-vertex sample inside the default run; this is the same-mesh figure
SURVEY 8(d) asks for, committed as profiles/rNN_cpu_baseline_200k.json.   python profiles/cpu_baseline_200k.py [threads] > out.json"""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIG_3D, _cpu_model  # noqa: E402
from oracle import stin_oracle  # noqa: E402
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.set_num_threads(threads)
torch.manual_seed(49)
net = stin_oracle.define_G(**CONFIG_3D)
sample = make_synthetic_mesh(200_000, 3, seed=0)


def run():
    net.zero_grad(set_to_none=True)
    t = time.perf_counter()
    stin_oracle.compute_loss(stin_oracle.graph_forward(net, sample), sample.color, sample.mask).backward()
    return time.perf_counter() - t


warm = run()
ts = [run() for _ in range(3)]
nv = int(sample.x.shape[0])
print(json.dumps({'value': nv / statistics.median(ts), 'unit': 'vertices/s', 'cores': threads, 'kind': 'port', 'cpu_model': _cpu_model(),
                  'host_threads': os.cpu_count(), 'sample_vertices': nv, 'edges': int(sample.edge_index.shape[1]), 'warmup_s': round(warm, 3),
                  'passes_s': [round(t, 3) for t in ts],
                  'sample': 'fwd+loss+bwd of the CPU oracle on the headline 200 704-vertex / 1 200 642-edge 3-level synthetic mesh, torch %s CPU '
                            'fp32, 1 warm-up + median of 3 passes, %d threads' % (torch.__version__, threads)}))
