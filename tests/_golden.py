"""Helpers shared by the parity tests: load tests/golden/*.npz fixtures."""
import json
import os

import numpy as np
import torch

from surface_texture_inpainting_net_amd.data import HierarchicalBatch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

MODEL_FIXTURES = ['g1_imagegraph_edgeconv', 'g2_3level_transinv_max', 'g2_3level_transinv_mean',
                  'g3_batch2_unequal', 'g5_sageconv', 'g5_sageconvtransinv', 'g6_graphnorm', 'g7_train_step']
# (g12_batchnorm_step has its own protocol - two forward calls - and its own tests)


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name + '.npz')) as z:
        return {k: z[k] for k in z.files}


class ModelFixture:
    def __init__(self, name):
        z = load_npz(name)
        self.name = name
        self.cfg = json.loads(bytes(z['cfg']).decode())
        self.state_dict = {k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('sd.')}
        self.state_dict_after = {k[4:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('sd1.')}
        self.grads = {k[2:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('g.')}
        self.out = torch.from_numpy(z['out'])
        self.pred = torch.from_numpy(z['pred'])
        self.loss = torch.from_numpy(z['loss'])
        self.gx = torch.from_numpy(z['gx'])
        self._sample = {k[2:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('s.')}

    def sample(self, device='cpu'):
        s = HierarchicalBatch(**{k: v.clone() for k, v in self._sample.items()})
        return s.to(device) if device != 'cpu' else s


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
