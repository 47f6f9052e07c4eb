"""Parity at BASELINE.json's full size, and the evidence behind the gradient tolerance.

1. The headline workload itself - 200 704 vertices / 1 200 642 directed edges, 3 levels, the shipped 3-D config, seed 49 -
   through the HIP path against the CPU oracle (one fwd + loss + bwd of the oracle: ~25 s on the GPU box's host).
   Bars: forward max-abs <= 1e-4, weight gradients relative L2 <= 1e-3.
2. An A/B on the 12 000-vertex case that separates the two possible sources of gradient error: the split-16-bit MFMA
   GEMMs (fp16x3 forward, bf16x3 backward) versus discrete decision flips (a near-tie arg-max of the max pool or a ReLU
   decision re-routes one gradient entry).  Truth = an fp64 run of the oracle; the fp32 CPU oracle, the shipped GEMM
   precision, exact-fp32 GEMMs and bf16x6 backward GEMMs are each measured against it.
"""
import pytest
import torch

from oracle import stin_oracle
from surface_texture_inpainting_net_amd import functional as SF
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S
from surface_texture_inpainting_net_amd.data import HierarchicalBatch
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CFG = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=9,
           n_levels=2, pooling_type='max', dilations=[1, 1, 1, 2, 4, 8, 16, 1, 1], checkpoint_bottleneck=True)


def _hip_run(net, s):
    net.zero_grad(set_to_none=True)
    sd = s.to(DEV)
    out = net(sd)
    loss = stin_oracle.compute_loss(torch.where((sd.mask > 0).expand_as(sd.color), out, sd.color), sd.color, sd.mask)
    loss.backward()
    torch.cuda.synchronize()
    return out.detach().cpu(), float(loss.detach()), [p.grad.detach().cpu().clone() for p in net.parameters()]


def _grad_errors(grads, truth):
    """-> (global relative L2, worst per-tensor max-abs / global max-abs)."""
    scale = max(float(t.abs().max()) for t in truth)
    num = sum(float((g.double() - t.double()).pow(2).sum()) for g, t in zip(grads, truth))
    den = sum(float(t.double().pow(2).sum()) for t in truth)
    worst = max(float((g.double() - t.double()).abs().max()) for g, t in zip(grads, truth)) / scale
    return (num / den) ** 0.5, worst


@pytest.mark.parametrize('weight_seed,mesh_seed', [(49, 0), (7, 5)])
def test_headline_size_forward_loss_and_gradients_vs_cpu_oracle(weight_seed, mesh_seed):
    """(49, 0) = the bench's weights and scene; (7, 5) = a second draw of both (round-3 review: the backward GEMMs run at a
    16-bit significand and one seed was the only full-size guard)."""
    torch.manual_seed(weight_seed)
    ref = stin_oracle.define_G(**CFG)
    net = S.define_G(**CFG)
    net.load_state_dict(ref.state_dict())
    net = net.to(DEV)
    s = make_synthetic_mesh(200_000, 3, seed=mesh_seed)
    assert s.x.shape[0] == 200_704 and s.edge_index.shape[1] == 1_200_642
    got, loss, grads = _hip_run(net, s)
    torch.set_num_threads(min(32, torch.get_num_threads()))      # torch's CPU scatter / index ops degrade when oversubscribed
    want = ref(s)
    loss_ref = stin_oracle.compute_loss(stin_oracle.graph_forward(ref, s), s.color, s.mask)
    loss_ref.backward()
    err = float((got - want.detach()).abs().max())
    rel, worst = _grad_errors(grads, [p.grad for p in ref.parameters()])
    print('\nfull size (weights %d, mesh %d): ' % (weight_seed, mesh_seed), end='')
    print('forward max-abs %.3e, loss %.7f vs %.7f, weight-gradient rel-L2 %.3e, worst per-tensor max-abs %.3e of scale'
          % (err, loss, float(loss_ref), rel, worst))
    assert err <= 1e-4
    assert abs(loss - float(loss_ref)) <= 1e-6
    assert rel <= 1e-3
    # a flipped arg-max / ReLU decision moves ONE entry by O(|g_i|): bounded against the scale, not against 1e-3
    assert worst <= 5e-3


def test_gradient_error_is_decision_flips_not_gemm_precision(monkeypatch):
    torch.manual_seed(49)
    ref = stin_oracle.define_G(**CFG)
    s = make_synthetic_mesh(12_000, 3, seed=3)
    # fp64 truth: the same oracle in double
    ref64 = stin_oracle.define_G(**CFG).double()
    ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
    s64 = HierarchicalBatch(**{k: (s[k].double() if torch.is_tensor(s[k]) and s[k].is_floating_point() else s[k]) for k in s.keys()})
    out64 = ref64(s64)
    stin_oracle.compute_loss(stin_oracle.graph_forward(ref64, s64), s64.color, s64.mask).backward()
    truth = [p.grad for p in ref64.parameters()]
    out32 = ref(s)
    stin_oracle.compute_loss(stin_oracle.graph_forward(ref, s), s.color, s.mask).backward()
    res = {'cpu oracle fp32': (float((out32.detach().double() - out64.detach()).abs().max()),) +
           _grad_errors([p.grad for p in ref.parameters()], truth)}
    net = S.define_G(**CFG)
    net.load_state_dict(ref.state_dict())
    net = net.to(DEV)
    variants = {'shipped (fp16x3 fwd, bf16x3 bwd)': (SF.GEMM_F16X3, SF.GEMM_BF16X3),
                'bf16x6 bwd': (SF.GEMM_F16X3, SF.GEMM_BF16X6),
                'exact fp32 GEMMs': (SF.GEMM_F32, SF.GEMM_F32)}
    for name, (pf, pb) in variants.items():
        monkeypatch.setattr(SF, 'PREC_FWD', pf)
        monkeypatch.setattr(SF, 'PREC_BWD', pb)
        got, _, grads = _hip_run(net, s)
        res[name] = (float((got.double() - out64.detach()).abs().max()),) + _grad_errors(grads, truth)
    print()
    for name, (fe, rel, worst) in res.items():
        print('%-36s forward max-abs %.2e   grad rel-L2 %.2e   worst entry %.2e of scale' % (name, fe, rel, worst))
    shipped, exact, cpu = res['shipped (fp16x3 fwd, bf16x3 bwd)'], res['exact fp32 GEMMs'], res['cpu oracle fp32']
    for fe, rel, worst in res.values():
        assert fe <= 1e-4 and rel <= 3e-3
    # the split-16-bit GEMMs cost nothing measurable beside exact fp32 GEMMs: the error that remains is the flip noise every
    # fp32 evaluation order has (the fp32 CPU oracle included)
    assert shipped[1] <= max(2.0 * exact[1], 2.0 * cpu[1], 5e-4)
