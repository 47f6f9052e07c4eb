"""SingleConvMeshNet (reference models/singleconvmeshnet.py, BatchNorm1d inside the edge MLP).

Fixture g10 holds the reference's OWN class in training mode (output, loss, running statistics after the step) and in
eval mode; its gradients come from the build's restatement, whose training-mode forward the generator checked bit for bit
against the reference (the reference's in-place residual add cannot run backward on torch 2.x).
CPU: restatement vs fixture.  GPU: the HIP-kernel model vs fixture, fp32 tolerance 1e-4 / gradients 1e-3 of their scale."""
import numpy as np
import pytest
import torch

from _golden import load_npz
from oracle import scmn_oracle
from surface_texture_inpainting_net_amd.data import HierarchicalBatch

CFG = dict(feature_number=10, num_propagation_steps=2, filter_sizes=[16, 32, 48], num_classes=3)


def _load(pooling):
    g = load_npz('g10_singleconvmeshnet_%s' % pooling)
    sample = HierarchicalBatch(**{k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('s.')})
    state = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w/')}
    return g, sample, state


@pytest.mark.parametrize('pooling', ['mean', 'max'])
def test_restatement_matches_reference_fixture(pooling):
    g, sample, state = _load(pooling)
    net = scmn_oracle.SingleConvMeshNet(pooling_method=pooling, **CFG)
    net.load_state_dict(state)
    net.train()
    out = net(sample)
    assert np.allclose(out.detach().numpy(), g['out_train'], rtol=0, atol=2e-6)
    loss = ((out - torch.from_numpy(g['target'])) ** 2).mean()
    assert abs(float(loss.detach()) - float(g['loss'])) <= 1e-6
    loss.backward()
    for k, p in net.named_parameters():
        assert np.allclose(p.grad.numpy(), g['g_restatement/' + k], rtol=1e-4, atol=1e-6), k
    for k, v in net.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            assert np.allclose(v.numpy(), g['after/' + k], rtol=1e-6, atol=1e-6), k
    net.eval()
    with torch.no_grad():
        assert np.allclose(net(sample).numpy(), g['out_eval'], rtol=0, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('pooling', ['mean', 'max'])
def test_hip_singleconvmeshnet_matches_reference_fixture(pooling):
    from surface_texture_inpainting_net_amd.singleconvmeshnet import SingleConvMeshNet
    g, sample, state = _load(pooling)
    net = SingleConvMeshNet(pooling_method=pooling, **CFG)
    assert list(net.state_dict().keys()) == list(state.keys()), 'reference checkpoint layout'
    net.load_state_dict(state)
    net = net.to('cuda:0')
    s = sample.to('cuda:0')
    net.train()
    out = net(s)
    assert float((out.detach().cpu() - torch.from_numpy(g['out_train'])).abs().max()) <= 1e-4
    loss = ((out - torch.from_numpy(g['target']).to('cuda:0')) ** 2).mean()
    assert abs(float(loss.detach()) - float(g['loss'])) <= 1e-5
    loss.backward()
    scale = max(float(np.abs(g['g_restatement/' + k]).max()) for k, _ in net.named_parameters())
    for k, p in net.named_parameters():
        assert float((p.grad.cpu() - torch.from_numpy(g['g_restatement/' + k])).abs().max()) <= 2e-3 * scale, k
    for k, v in net.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            assert np.allclose(v.cpu().numpy(), g['after/' + k], rtol=1e-5, atol=1e-5), k
    net.eval()
    with torch.no_grad():
        assert float((net(s).cpu() - torch.from_numpy(g['out_eval'])).abs().max()) <= 1e-4


@pytest.mark.gpu
def test_hip_singleconvmeshnet_full_size_runs_and_is_deterministic():
    from surface_texture_inpainting_net_amd.singleconvmeshnet import SingleConvMeshNet
    from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
    torch.manual_seed(0)
    net = SingleConvMeshNet(10, 2, [32, 64, 128], num_classes=21).to('cuda:0')
    s = make_synthetic_mesh(100_000, 3, seed=4, dilations=()).to('cuda:0')
    a = net(s)
    a.square().mean().backward()
    ga = [p.grad.clone() for p in net.parameters()]
    net.zero_grad()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.reset_running_stats()
    b = net(s)
    b.square().mean().backward()
    assert a.shape == (s.x.shape[0], 21) and torch.equal(a, b)
    assert all(torch.equal(x, p.grad) for x, p in zip(ga, net.parameters())), 'no atomics: bit-reproducible gradients'
