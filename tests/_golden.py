"""Helpers shared by the parity tests: load tests/golden/*.npz fixtures."""
import json
import os

import numpy as np
import torch

from surface_texture_inpainting_net_amd.data import HierarchicalBatch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

MODEL_FIXTURES = ['g1_imagegraph_edgeconv', 'g2_3level_transinv_max', 'g2_3level_transinv_mean',
                  'g3_batch2_unequal', 'g5_sageconv', 'g5_sageconvtransinv', 'g6_graphnorm', 'g7_train_step', 'g13_5level']
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


def grad_flip_report(named_params, ref_params, tag, flip_at=1e-3):
    """Weight gradients of the HIP path against the oracle's, as the "decision flip" claim made testable: fp32 re-association
    flips a few near-tie arg-max / ReLU decisions and each flip re-routes ONE gradient contribution, so the error is a handful
    of entries beyond `flip_at` of the gradient scale on top of ~1e-4 noise - not a broad loss of precision.
    -> dict(max_rel = worst entry / scale, beyond = entries beyond flip_at * scale, total, rel_l2); printed, for the bars."""
    refs = list(ref_params)
    scale = max(float(q.grad.abs().max()) for q in refs)
    worst, beyond, total, num, den = 0.0, 0, 0, 0.0, 0.0
    for (k, p), q in zip(named_params, refs):
        d = (p.grad.detach().cpu() - q.grad).abs()
        worst = max(worst, float(d.max()))
        beyond += int((d > flip_at * scale).sum())
        total += d.numel()
        num += float(d.double().pow(2).sum())
        den += float(q.grad.double().pow(2).sum())
    rep = dict(max_rel=worst / scale, beyond=beyond, total=total, rel_l2=(num / max(den, 1e-300)) ** 0.5)
    print('\n[grad] %s: worst entry %.3e of scale, %d of %d entries beyond %.0e of scale, rel-L2 %.3e'
          % (tag, rep['max_rel'], beyond, total, flip_at, rep['rel_l2']))
    return rep


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
