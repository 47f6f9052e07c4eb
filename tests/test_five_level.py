"""BASELINE config 5's NETWORK (n_levels=4: five graph levels, widths 64 .. 1024, bottleneck EdgeConv 2048 -> 2048 -> 1024,
67 146 563 parameters) against the CPU oracle - forward, loss and every weight gradient - at a size the oracle finishes in
seconds (a ~30 k-vertex five-level mesh with dilated edge sets at the coarsest level), fp32 storage at the fp32 bars and
bf16 storage at stated bars.  The 1 M-vertex run of the same network is a property test (tests/test_hip_bf16.py); the
reference class at this depth is pinned by the golden fixture g13_5level (tests/test_hip_parity.py, tests/test_oracle_golden.py).
Reference: models/surfacetextureinpaintingnet.py:316-338 (encoder / decoder built per level), :398-471 (forward)."""
import json
import os

import pytest
import torch

from oracle import stin_oracle
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh

DEV = 'cuda:0'
CFG5 = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=9, n_levels=4,
            pooling_type='max', dilations=[1, 1, 1, 2, 4, 8, 16, 1, 1], checkpoint_bottleneck=True,
            num_blocks_per_uncheckpointed_block=1)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

_cache = {}


def _oracle_run():
    """The oracle's forward / loss / gradients on the five-level mesh (computed once per session: ~20 s of CPU)."""
    if 'ref' not in _cache:
        torch.manual_seed(49)
        ref = stin_oracle.define_G(**CFG5)
        assert sum(p.numel() for p in ref.parameters()) == 67_146_563
        s = make_synthetic_mesh(30_000, 5, seed=21, dilations=(2, 4, 8, 16))
        want = ref(s)
        loss = stin_oracle.compute_loss(stin_oracle.graph_forward(ref, s), s.color, s.mask)
        loss.backward()
        _cache['ref'] = (ref, s, want.detach(), float(loss.detach()))
    return _cache['ref']


def _truth64():
    """The same oracle in fp64 (the referee between two fp32 evaluation orders): gradients of every parameter."""
    if 'g64' not in _cache:
        from surface_texture_inpainting_net_amd.data import HierarchicalBatch
        ref, s, _, _ = _oracle_run()
        ref64 = stin_oracle.define_G(**CFG5).double()
        ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
        s64 = HierarchicalBatch(**{k: (s[k].double() if torch.is_tensor(s[k]) and s[k].is_floating_point() else s[k]) for k in s.keys()})
        out64 = ref64(s64)
        stin_oracle.compute_loss(stin_oracle.graph_forward(ref64, s64), s64.color, s64.mask).backward()
        _cache['g64'] = (out64.detach(), [p.grad for p in ref64.parameters()])
    return _cache['g64']


def _rel_l2(grads, truth):
    num = sum(float((g.double() - t.double()).pow(2).sum()) for g, t in zip(grads, truth))
    return (num / sum(float(t.double().pow(2).sum()) for t in truth)) ** 0.5


def _hip_run(bf16):
    ref, s, want, loss_ref = _oracle_run()
    net = S.define_G(**CFG5)
    net.load_state_dict(ref.state_dict())
    net = net.to(DEV)
    if bf16:
        net.set_activation_dtype(torch.bfloat16)
    sd = s.to(DEV)
    got = net(sd)
    assert got.dtype == torch.float32 and got.shape == want.shape
    loss = stin_oracle.compute_loss(torch.where((sd.mask > 0).expand_as(sd.color), got, sd.color), sd.color, sd.mask)
    loss.backward()
    d = (got.detach().cpu() - want).abs()
    num = den = 0.0
    worst = (0.0, None)
    for (k, p), q in zip(net.named_parameters(), ref.parameters()):
        assert p.grad is not None and p.grad.dtype == torch.float32, k
        n_k, d_k = float((p.grad.cpu() - q.grad).double().pow(2).sum()), float(q.grad.double().pow(2).sum())
        num, den = num + n_k, den + d_k
        if d_k > 0 and (n_k / d_k) ** 0.5 > worst[0]:
            worst = ((n_k / d_k) ** 0.5, k)
    return dict(fwd_max=float(d.max()), fwd_mean=float(d.mean()), loss=float(loss.detach()), loss_ref=loss_ref,
                grad_rel=(num / den) ** 0.5, worst=worst, out=got.detach().cpu(), grads=[p.grad.detach().cpu() for p in net.parameters()])


@pytest.mark.gpu
def test_five_level_network_vs_oracle_fp32():
    """fp32 storage, 5 levels x 17 blocks incl. the 2048-wide bottleneck products: forward max-abs <= 1e-4 and loss to 1e-6
    against the fp32 oracle (measured 1.3e-5).  Weight gradients over all 67 M parameters: on a 30 k-vertex mesh whose
    coarsest level has 245 vertices a flipped arg-max / ReLU decision weighs ~5x what it does at the headline size, and the
    REFERENCE'S OWN fp32 arithmetic cannot meet 1e-3 here - the fp32 CPU oracle is 9.3e-4 (GPU box) / 7.4e-4 (build container)
    relative L2 away from an fp64 run of itself.  So the fp64 run is the referee: the HIP path must be within 1e-3 of it or within
    twice the fp32 CPU oracle's own distance (measured 1.40e-3 vs 9.3e-4; 1.65e-3 between the two fp32 evaluations).
    Round 6 attribution (profiles/r06_five_level_attribution.md, profiles/probes/five_level_attribution.py - the same run with other
    matrix-core settings): the backward products play no part (exact-fp32 and bf16x6 backward GEMMs give the same 1.398e-3 to four
    digits, K >= 1024 layers 1.420e-3 vs the rest 1.362e-3), and giving the FORWARD products more bits moves the result AWAY from the
    fp64 truth (bf16x6 = 24-bit products 2.61e-3 with 197 entries beyond 1e-3 of scale, exact-fp32 MFMA 2.61e-3 with 1 364; the shipped
    fp16x3 forward: 2 entries) - every fp32 evaluation order lands 1-3e-3 from fp64 on this mesh, which one is closest is chance
    (arg-max / ReLU decisions at a 245-vertex coarsest level), so the bar stays relative to the reference's own fp32 distance."""
    r = _hip_run(False)
    print('\n5-level fp32 vs oracle: fwd max-abs %.3e, loss %.7f vs %.7f, grad rel-L2 %.3e (worst tensor %.3e %s)'
          % (r['fwd_max'], r['loss'], r['loss_ref'], r['grad_rel'], r['worst'][0], r['worst'][1]))
    ref = _oracle_run()[0]
    out64, g64 = _truth64()
    e_hip, e_cpu = _rel_l2(r['grads'], g64), _rel_l2([p.grad for p in ref.parameters()], g64)
    f_hip = float((r['out'].double() - out64).abs().max())
    print('against the fp64 run of the oracle: HIP fwd %.3e grad rel-L2 %.3e | fp32 CPU oracle grad rel-L2 %.3e' % (f_hip, e_hip, e_cpu))
    assert r['fwd_max'] <= 1e-4 and f_hip <= 1e-4
    assert abs(r['loss'] - r['loss_ref']) <= 1e-6
    assert e_hip <= max(1e-3, 2.0 * e_cpu), (e_hip, e_cpu)


@pytest.mark.gpu
def test_five_level_network_vs_oracle_bf16():
    """bf16 activation storage (the mode BASELINE config 5 names) against the fp32 oracle: the build's stated tolerance for this
    depth = 1.5 x what was measured on MI355X in round 5 (fwd max-abs 1.49e-1 / mean-abs 1.78e-2, loss 1.2e-4 relative, gradients
    27.4 % relative L2: two more pool levels and a 245-vertex coarsest level against the 3-level network's 16 %)."""
    r = _hip_run(True)
    print('\n5-level bf16 vs fp32 oracle: fwd max-abs %.3e mean-abs %.3e, loss %.7f vs %.7f, grad rel-L2 %.3e'
          % (r['fwd_max'], r['fwd_mean'], r['loss'], r['loss_ref'], r['grad_rel']))
    assert r['fwd_max'] <= BF16_BARS['fwd_max'] and r['fwd_mean'] <= BF16_BARS['fwd_mean']
    assert abs(r['loss'] - r['loss_ref']) <= BF16_BARS['loss_rel'] * r['loss_ref']
    assert r['grad_rel'] <= BF16_BARS['grad_rel']


# measured x 1.5 (see the docstring above); a 1.5x numerical regression of the bf16 mode fails here
BF16_BARS = dict(fwd_max=0.22, fwd_mean=2.7e-2, loss_rel=5e-4, grad_rel=0.41)


def test_five_level_parameter_count_is_the_survey_constant():
    rec = json.load(open(os.path.join(GOLDEN, 'param_counts.json')))
    net = S.define_G(**CFG5)
    assert sum(p.numel() for p in net.parameters()) == rec['3d_transinv_nl4_nb9'] == 67_146_563
