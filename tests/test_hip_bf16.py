"""GPU tests of the bf16-STORAGE variant of the path (BASELINE.json configs 3 and 5; the reference itself is
fp32-only, so this is the build's extension with its own stated tolerance).

Two layers of checks, all through the C ABI:
  * kernel level, BIT-EXACT: every *_bf16 streaming / gather / norm kernel must equal its *_f32 twin run on the
    widened inputs, rounded once to bf16 (same fp32 arithmetic, same summation order, one final rounding); index
    outputs (ReLU masks, pool arg-max) must be identical.
  * GEMMs against fp64 on the bf16-rounded operands (fp32 accumulation: error bound independent of bf16);
  * whole network (shipped 3-D config, 15 blocks) against the fp32 CPU oracle - the STATED tolerance of the bf16
    mode = 1.5 x what is measured (round 5; NET_BARS): tanh output max-abs <= 0.115 and mean-abs <= 1.5e-2 (measured
    7.6e-2 / 9.9e-3), loss within 5e-4 (measured 2.1e-4), weight gradients within 24 % relative L2 (measured 16.1 %).
    Why the gradient bar is not tighter, and why no "mixed" variant is shipped (tests/tools/bf16_design_probe.py,
    profiles/r02_bf16_sensitivity.md): this network's gradient responds to a forward perturbation of relative size e like
    sqrt(e), not e - ReLU and arg-max (max pool) decisions flip in proportion to e and each flip moves a gradient entry
    by O(1).  fp32 reordering (e = 6e-8) already gives 7e-4 against the reference formulation; ONE bf16 rounding per block
    (block input only, fp32 everything else, fp32 residual stream) gives 9.5 %, the pre-activations Y alone 11 %, all
    tensors 16 %; every backward rounding together < 0.5 %.  A 5 % bar is therefore out of reach of ANY 16-bit storage
    of a forward tensor (fp16's 2^-11 would halve it), and keeping the residual stream in fp32 buys 16 % -> 15 %.  What
    certifies the mode is the training curve: test_bf16_training_curve_tracks_fp32.
"""
import pytest
import torch

from oracle import stin_oracle
from surface_texture_inpainting_net_amd import functional as SF
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S
from surface_texture_inpainting_net_amd.modules import _as_groups
from surface_texture_inpainting_net_amd.plan import EdgeSet, PoolMap
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not SF.USE_EDGE_MASK, reason='bf16 storage needs the saved ReLU mask (STIN_EDGE_MASK=0 set)')]
DEV = 'cuda:0'
BF = torch.bfloat16


def _bad():
    return torch.zeros(1, dtype=torch.int32, device=DEV)


def _graph(n, e, seed, isolated=5):
    g = torch.Generator().manual_seed(seed)
    return torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(isolated, n, (e,), generator=g)]).to(DEV)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(BF)


@pytest.mark.parametrize('H', [128, 256, 512, 1024])
def test_edge_stage_bf16_equals_rounded_fp32_kernels(H):
    n, e = 3000, 17000
    es = EdgeSet(_graph(n, e, H), n, _bad())
    A, B, G = _rand((n, H), 1), _rand((n, H), 2), _rand((n, H), 3)
    words = H // 32
    out16 = torch.empty(n, H + 8, dtype=BF, device=DEV)
    out32 = torch.empty(n, H + 4, dtype=torch.float32, device=DEV)
    m16 = torch.zeros(e * words, dtype=torch.int32, device=DEV)
    m32 = torch.zeros(e * words, dtype=torch.int32, device=DEV)
    SF.edge_relu_mean_fwd(A, B, es.by_dst, out16, indicator=True, mask=m16)
    SF.edge_relu_mean_fwd(A.float(), B.float(), es.by_dst, out32, indicator=True, mask=m32)
    # same ReLU decisions (taken on the same fp32 sums); the bf16 kernels give a lane 8 channels instead of 4, so the BIT
    # ORDER inside an edge slot differs from the fp32 kernels' - compare the per-slot population counts, the backward
    # results below compare the full content
    def popcount(m):
        bits = (m.view(e, words, 1) >> torch.arange(32, device=DEV, dtype=torch.int32)) & 1
        return bits.sum(dim=(1, 2))
    assert torch.equal(popcount(m16), popcount(m32))
    assert torch.equal(out16[:, :H + 1], out32[:, :H + 1].to(BF))
    dA16, dB16 = torch.empty(n, H, dtype=BF, device=DEV), torch.empty(n, H, dtype=BF, device=DEV)
    dA32, dB32 = torch.empty(n, H, device=DEV), torch.empty(n, H, device=DEV)
    SF.edge_relu_mean_bwd_dst_mask(G, m16, es.by_dst, dA16)
    SF.edge_relu_mean_bwd_src_mask(G, m16, es, dB16)
    SF.edge_relu_mean_bwd_dst_mask(G.float(), m32, es.by_dst, dA32)
    SF.edge_relu_mean_bwd_src_mask(G.float(), m32, es, dB32)
    assert torch.equal(dA16, dA32.to(BF))
    assert torch.equal(dB16, dB32.to(BF))
    # both halves in one launch (what the block backward calls), into column slices of one wider matrix
    dY = torch.full((n, 2 * H + 8), 7.0, dtype=BF, device=DEV)
    SF.edge_relu_mean_bwd_mask(G, m16, es, dY[:, :H], dY[:, H:2 * H])
    assert torch.equal(dY[:, :H], dA16) and torch.equal(dY[:, H:2 * H], dB16)
    assert float((dY[:, 2 * H:].float() - 7.0).abs().max()) == 0.0
    dZ = torch.full((n, 2 * H + H // 2 + 8), 7.0, dtype=BF, device=DEV)
    src = _rand((n, H), 9)[:, :H // 2]
    SF.edge_relu_mean_bwd_mask(G, m16, es, dZ[:, :H], dZ[:, H:2 * H], copy_src=src, copy_dst=dZ[:, 2 * H:2 * H + H // 2])
    assert torch.equal(dZ[:, :H], dA16) and torch.equal(dZ[:, H:2 * H], dB16) and torch.equal(dZ[:, 2 * H:2 * H + H // 2], src)
    assert float((dZ[:, 2 * H + H // 2:].float() - 7.0).abs().max()) == 0.0


def test_edge_stage_bf16_on_column_slices_of_a_wider_matrix():
    n, e, H = 1000, 6000, 128
    es = EdgeSet(_graph(n, e, 9), n, _bad())
    Y = _rand((n, 2 * H + 64), 4)
    out = torch.empty(n, H + 8, dtype=BF, device=DEV)
    ref = torch.empty(n, H + 4, device=DEV)
    SF.edge_relu_mean_fwd(Y[:, :H], Y[:, H:2 * H], es.by_dst, out, indicator=True)
    SF.edge_relu_mean_fwd(Y[:, :H].float().contiguous(), Y[:, H:2 * H].float().contiguous(), es.by_dst, ref, indicator=True)
    assert torch.equal(out[:, :H + 1], ref[:, :H + 1].to(BF))


def test_bf16_kernels_reject_unvectorisable_rows():
    n = 100
    es = EdgeSet(_graph(n, 300, 1), n, _bad())
    A = _rand((n, 6), 1)
    with pytest.raises(Exception):
        SF.edge_relu_mean_fwd(A, A, es.by_dst, torch.empty(n, 6, dtype=BF, device=DEV))
    with pytest.raises(TypeError):
        SF.edge_relu_mean_fwd(A, A.float(), es.by_dst, torch.empty(n, 8, dtype=BF, device=DEV))


@pytest.mark.parametrize('C', [8, 64, 128, 320])
def test_segment_sum_pool_unpool_bf16_equal_rounded_fp32(C):
    n, e = 5000, 21000
    es = EdgeSet(_graph(n, e, C), n, _bad())
    x = _rand((n, C), 5)
    for mean in (False, True):
        a = SF.segment_sum(x, es.by_dst.rowptr, es.by_dst.col, n, mean=mean)
        b = SF.segment_sum(x.float(), es.by_dst.rowptr, es.by_dst.col, n, mean=mean)
        assert a.dtype == BF and torch.equal(a, b.to(BF))
    g = torch.Generator().manual_seed(C)
    nc = 1500
    trace = torch.randint(0, nc, (n,), generator=g)
    trace[:nc] = torch.arange(nc)
    pool = PoolMap(trace.to(DEV), n, nc, _bad())
    xq = (x.float() * 4).round().div(4).to(BF)                      # engineered ties
    p16 = SF.PoolMaxFn.apply(xq, pool)
    p32 = SF.PoolMaxFn.apply(xq.float(), pool)
    assert torch.equal(p16, p32.to(BF))
    xr = xq.clone().requires_grad_(True)
    xf = xq.float().requires_grad_(True)
    go = _rand((nc, C), 6)
    SF.PoolMaxFn.apply(xr, pool).backward(go)
    SF.PoolMaxFn.apply(xf, pool).backward(go.float())
    assert torch.equal(xr.grad, xf.grad.to(BF)), 'same arg-max routing (first maximum wins)'
    u16 = SF.UnpoolFn.apply(go, pool)
    assert torch.equal(u16, go[pool.trace.long()])
    m16 = SF.PoolMeanFn.apply(x, pool)
    assert torch.equal(m16, SF.PoolMeanFn.apply(x.float(), pool).to(BF))


@pytest.mark.parametrize('quirk', [False, True])
def test_instance_norm_bf16_equals_rounded_fp32(quirk):
    n, C = 7001, 128
    batch = torch.cat([torch.zeros(3000), torch.ones(1500), torch.full((2501,), 2.)]).long().to(DEV)
    groups = _as_groups(batch, n, torch.device(DEV), quirk)
    x, res, go = _rand((n, C), 7, 2.0), _rand((n, C), 8), _rand((n, C), 9)
    mean16, rstd16 = SF.instance_stats(x, groups)
    mean32, rstd32 = SF.instance_stats(x.float(), groups)
    assert torch.equal(mean16, mean32) and torch.equal(rstd16, rstd32), 'fp64 accumulation of the same values'
    y16 = SF.norm_act_res_fwd(x, mean16, rstd16, groups, res=res, act=True)
    y32 = SF.norm_act_res_fwd(x.float(), mean32, rstd32, groups, res=res.float(), act=True)
    assert y16.dtype == BF and torch.equal(y16, y32.to(BF))
    d16 = SF.instance_norm_act_bwd(x, go, mean16, rstd16, groups, act=True)
    d32 = SF.instance_norm_act_bwd(x.float(), go.float(), mean32, rstd32, groups, act=True)
    assert torch.equal(d16, d32.to(BF))


@pytest.mark.parametrize('M,Nc,K', [(1, 8, 8), (37, 3, 64), (1000, 128, 16), (4097, 320, 64), (3001, 64, 136), (2500, 256, 264),
                                    (513, 256, 1280), (700, 3, 3), (999, 40, 12)])
def test_gemm_bf16_storage_against_fp64(M, Nc, K):
    g = torch.Generator().manual_seed(M + Nc + K)
    A = torch.randn(M, K, generator=g).to(DEV).to(BF)
    W = (torch.randn(Nc, K, generator=g) * 0.1).to(DEV)
    b = torch.randn(Nc, generator=g).to(DEV)
    G = torch.randn(M, Nc, generator=g).to(DEV).to(BF)
    Wr = W.to(BF).double()                                           # the kernel rounds W to bf16 while staging
    want = A.double() @ Wr.t() + b.double()
    scale = float(want.abs().max()) + 1
    got32 = SF.gemm_nt(A, W, b, out_dtype=torch.float32)
    assert got32.dtype == torch.float32
    assert float((got32.double() - want).abs().max()) <= 2e-6 * (K ** 0.5) * scale
    got16 = SF.gemm_nt(A, W, b)
    assert got16.dtype == BF
    assert torch.equal(got16, got32.to(BF)), 'bf16 output = ONE rounding of the fp32 result'
    # masked bias + residual, fused in fp32 before the single rounding
    mask = (torch.rand(M, 1, generator=g) > 0.3).to(DEV).to(BF)
    res = torch.randn(M, Nc, generator=g).to(DEV).to(BF)
    want2 = A.double() @ Wr.t() + mask.double() * b.double() + res.double()
    got2 = SF.gemm_nt(A, W, b, row_mask=mask[:, 0], residual=res)
    assert float((got2.double() - want2).abs().max()) <= 2 ** -8 * (float(want2.abs().max()) + 1)
    assert float((got2.double() - want2).abs().mean()) <= 2 ** -9 * float(want2.abs().mean() + 1e-3)
    # weight gradient (+ weighted bias-gradient column), fp32 result
    w = torch.rand(M, 1, generator=g).to(DEV).to(BF)
    want_tn = torch.cat([G.double().t() @ A.double(), (G.double() * w.double()).sum(0)[:, None]], 1)
    got_tn = SF.gemm_tn(G, A, ones_column=True, row_weight=w[:, 0])
    assert got_tn.dtype == torch.float32
    assert float((got_tn.double() - want_tn).abs().max()) <= 2e-6 * (M ** 0.5) * float(want_tn.abs().max() + 1) / 10 + 1e-5
    assert torch.equal(got_tn, SF.gemm_tn(G, A, ones_column=True, row_weight=w[:, 0])), 'deterministic slab order'
    assert torch.equal(SF.gemm_tn(G, A)[:, :K], got_tn[:, :K])


def test_gemm_bf16_on_strided_views():
    g = torch.Generator().manual_seed(5)
    big = torch.randn(500, 264, generator=g).to(DEV).to(BF)
    A = big[:, 8:136]                                                # ld 264, 16-byte aligned column slice
    W = torch.randn(96, 128, generator=g).to(DEV)
    want = A.double() @ W.to(BF).double().t()
    out = torch.zeros(500, 200, dtype=BF, device=DEV)
    SF.gemm_nt(A, W, out=out[:, 104:200])
    assert float((out[:, 104:200].double() - want).abs().max()) <= 2 ** -8 * float(want.abs().max())
    assert float(out[:, :104].abs().max()) == 0.0
    Gm = big[:, 136:264]
    want_tn = Gm.double().t() @ A.double()
    assert float((SF.gemm_tn(Gm, A).double() - want_tn).abs().max()) <= 1e-4 * float(want_tn.abs().max())


CFG3D = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=9,
             n_levels=2, pooling_type='max', dilations=[1, 1, 1, 2, 4, 8, 16, 1, 1])


def _net_pair(cfg, seed=49):
    torch.manual_seed(seed)
    ref = stin_oracle.define_G(**cfg)
    net = S.define_G(**cfg)
    net.load_state_dict(ref.state_dict())
    return ref, net.to(DEV)


def test_bf16_network_vs_fp32_oracle_stated_tolerance():
    ref, net = _net_pair(CFG3D)
    net.set_activation_dtype(BF)
    s = make_synthetic_mesh(12_000, 3, seed=3)
    want = ref(s)
    loss_ref = stin_oracle.compute_loss(stin_oracle.graph_forward(ref, s), s.color, s.mask)
    loss_ref.backward()
    sd = s.to(DEV)
    got = net(sd)
    assert got.dtype == torch.float32
    loss = stin_oracle.compute_loss(torch.where((sd.mask > 0).expand_as(sd.color), got, sd.color), sd.color, sd.mask)
    loss.backward()
    err = float((got.detach().cpu() - want.detach()).abs().max())
    mean_err = float((got.detach().cpu() - want.detach()).abs().mean())
    num = den = 0.0
    for p, q in zip(net.parameters(), ref.parameters()):
        assert p.grad.dtype == torch.float32
        num += float((p.grad.cpu() - q.grad).double().pow(2).sum())
        den += float(q.grad.double().pow(2).sum())
    rel = (num / den) ** 0.5
    print('bf16 storage vs fp32 oracle: fwd max-abs %.3e mean-abs %.3e, loss %.6f vs %.6f, grad rel-L2 %.3e'
          % (err, mean_err, float(loss.detach()), float(loss_ref.detach()), rel))
    # stated tolerance of the mode = what was measured on MI355X in round 5 x 1.5 (NET_BARS): a 1.5x numerical regression fails
    assert err <= NET_BARS['fwd_max'] and mean_err <= NET_BARS['fwd_mean'], (err, mean_err)
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= NET_BARS['loss_rel'] * float(loss_ref.detach())
    assert rel <= NET_BARS['grad_rel'], rel


# measured on MI355X in round 5: fwd max-abs 7.62e-2, mean-abs 9.87e-3, loss 2.1e-4 relative, gradients 16.1 % relative L2
NET_BARS = dict(fwd_max=0.115, fwd_mean=1.5e-2, loss_rel=5e-4, grad_rel=0.24)


def test_bf16_network_is_deterministic_and_trains():
    from surface_texture_inpainting_net_amd.train_step import TrainStep
    cfg = dict(CFG3D, n_blocks=3, dilations=[1, 2, 4])
    _, net = _net_pair(cfg, seed=1)
    net.set_activation_dtype(BF)
    s = make_synthetic_mesh(6000, 3, seed=5, dilations=(2, 4)).to(DEV)
    a = net(s)
    b = net(s)
    assert torch.equal(a, b), 'no atomics: bit-reproducible'
    step = TrainStep(net, lr=1e-3)
    losses = [float(step(s)) for _ in range(8)]
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses


@pytest.mark.parametrize('seed', [3, 4, 5])
@pytest.mark.parametrize('shape', ['scene', 'crops', 'scene5'])
def test_bf16_training_curve_tracks_fp32(shape, seed):
    """(round 4: three initialisation seeds, and BASELINE config 3's shape - batches of 8 unequal crops, 4 graph levels, per-graph
    instance norm with the linspace slices - beside the single scene.)
    200 Adam steps (lr 5e-5) of the shipped 3-D config on a 20 164-vertex mesh with a smooth (learnable) colour field and
    four hole masks, same initial weights and data order: bf16 storage against fp32 storage, with fp32 storage + exact-fp32
    GEMMs as the CONTROL (two fp32 evaluation orders drift apart too - the network's decisions amplify any perturbation, see
    tests/tools/bf16_design_probe.py).  Measured on MI355X: control +1.0 % / +1.9 %, bf16 +2.1 % / +0.1 % (seeds 3 / 4) in the
    mean loss of the last 50 steps.  Stated bar: bf16 within 4 % of fp32 there, and within 1 % over the first 50 steps (where
    the trajectories have not separated yet)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('bf16_training_curve', os.path.join(os.path.dirname(__file__), 'tools',
                                                                                    'bf16_training_curve.py'))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    if shape == 'scene':
        samples, cfg = [s.to(DEV) for s in tool.learnable_samples(20_000)], None
    elif shape == 'scene5':
        # (round 6) BASELINE config 5's network: five graph levels, 67 M parameters, the 2048-wide bottleneck - the depth at which one
        # bf16 forward/backward is 27 % (relative L2) from the fp32 gradients (tests/test_five_level.py); what certifies the mode is this curve
        samples, cfg = [s.to(DEV) for s in tool.learnable_samples(30_000, levels=5)], dict(tool.CFG, n_levels=4)
    else:
        samples, cfg = [s.to(DEV) for s in tool.learnable_crop_batches()], dict(tool.CFG, n_levels=3)
    curves = {m: tool.curve(samples, 200, m, seed=seed, lr=5e-5, cfg=cfg) for m in ('f32', 'f32-exact-gemm', 'bf16')}
    head = {m: float(c[:50].mean()) for m, c in curves.items()}
    tail = {m: float(c[-50:].mean()) for m, c in curves.items()}
    print('\n%s seed %d: mean loss steps 0-49 %s\nmean loss steps 150-199 %s' % (shape, seed, head, tail))
    assert all(bool(torch.isfinite(c).all()) for c in curves.values())
    # The trajectories are chaotic: two fp32 evaluation orders (shipped split GEMMs vs exact-fp32 GEMMs - the CONTROL) end 200 steps
    # up to 8 % apart, and ANY change of fp32 summation order moves each of them by that much (round 4: making ONE weight-gradient
    # product of the first block exact moved the fp32 tails by -4 .. +7 %).  So the statement that survives such changes is: the bf16
    # run ends within `tail_bar` of the NEARER of the two fp32 runs, which themselves stay within `ctrl_bar` of each other.
    # Measured on MI355X, round 4, two code states (tail = mean loss of steps 150-199; bf16 vs nearer fp32 run | control vs fp32):
    #   scene  seeds 3 / 4 / 5: +2.0 .. +2.9 % / +1.1 .. +2.5 % (inside the band once) / +3.3 % | -4.3 .. +2.6 %
    #   crops  seeds 3 / 4 / 5: -6.4 .. -2.2 % / -2.6 .. -0.6 % / +2.1 .. +2.7 %              | -8.1 .. +0.7 %
    # (the crop batches change every step and hold 8 small graphs: a noisier loss, and it falls to 0.32 of its start in 200 steps
    #  where the single scene reaches 0.17).  Stated bars = bench.py's `dtype_tolerance`.
    trains, head_bar, tail_bar, ctrl_bar = {'scene': (0.25, 1e-2, 5e-2, 8e-2), 'crops': (0.40, 2e-2, 8e-2, 12e-2),
                                            'scene5': SCENE5_BARS}[shape]
    assert tail['f32'] < trains * head['f32'], 'the task must actually train'
    assert abs(head['bf16'] - head['f32']) <= head_bar * head['f32']
    near = min(abs(tail['bf16'] - tail['f32']), abs(tail['bf16'] - tail['f32-exact-gemm']))
    assert near <= tail_bar * tail['f32'], (tail, near / tail['f32'])
    assert abs(tail['f32-exact-gemm'] - tail['f32']) <= ctrl_bar * tail['f32'], 'the two fp32 evaluation orders drifted further apart than ever measured'


# five-level scene (round 6), measured on MI355X, seeds 3 / 4 / 5: loss falls to 0.27 / 0.25 / 0.28 of its start; bf16 vs fp32 over steps
# 0-49: -1.8 / +0.1 / +0.8 %; tail (steps 150-199) bf16 vs the nearer fp32 run: 2.4 / 9.7 / 2.1 %; the two fp32 orders: 6.8 / 0.9 / 6.2 %
# apart.  Bars = 1.5 x the largest measured figure (trains, first-50, last-50, control).
SCENE5_BARS = (0.42, 3e-2, 15e-2, 11e-2)


def test_bf16_mode_is_refused_for_unsupported_variants():
    net = S.define_G(input_nc=10, output_nc=3, ngf=16, filter_type='sageconv', norm='instance', n_blocks=1, n_levels=1,
                     pooling_type='max')
    with pytest.raises(NotImplementedError):
        net.set_activation_dtype(BF)
    with pytest.raises(ValueError):
        net.set_activation_dtype(torch.float16)


def test_bf16_batch_of_unequal_crops_four_levels():
    """BASELINE config 3 in miniature: a batch of 8 unequal crops, 4 graph levels, bf16 storage, full channel widths
    (ngf 64 -> 512 at the coarsest level).  Batched instance norm (per-graph statistics incl. the linspace-slice quirk)
    and the bit-exact integer batch propagation are shared with the fp32 path; the values carry the bf16 tolerance."""
    from surface_texture_inpainting_net_amd.data import collate
    graphs = [make_synthetic_mesh(n, 4, seed=60 + i, dilations=(2,)) for i, n in enumerate((1200, 2800, 1900, 1500, 2400, 2000, 1300, 2600))]
    batch = collate(graphs)
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=3, n_levels=3,
               pooling_type='max', dilations=[1, 2, 1])
    ref, net = _net_pair(cfg, seed=7)
    net.set_activation_dtype(BF)
    want = ref(batch)
    loss_ref = stin_oracle.compute_loss(stin_oracle.graph_forward(ref, batch), batch.color, batch.mask)
    bd = batch.to(DEV)
    got = net(bd)
    loss = stin_oracle.compute_loss(torch.where((bd.mask > 0).expand_as(bd.color), got, bd.color), bd.color, bd.mask)
    d = (got.detach().cpu() - want.detach()).abs()
    print('bf16 crops: fwd max-abs %.3e mean-abs %.3e, loss %.6f vs %.6f' % (float(d.max()), float(d.mean()), float(loss.detach()), float(loss_ref.detach())))
    assert float(d.max()) <= CROPS_BARS['fwd_max'] and float(d.mean()) <= CROPS_BARS['fwd_mean'], (float(d.max()), float(d.mean()))
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= CROPS_BARS['loss_rel'] * float(loss_ref.detach())
    loss.backward()
    assert all(bool(torch.isfinite(p.grad).all()) for p in net.parameters())


# measured on MI355X in round 5 (bars = x 1.5): crops fwd max-abs 1.14e-1, mean-abs 1.37e-2, loss 3.8e-4 relative;
# config 5 (1 M vertices, bf16 vs fp32 storage) max-abs 2.04e-1, mean-abs 1.71e-2
CROPS_BARS = dict(fwd_max=0.17, fwd_mean=2.1e-2, loss_rel=8e-4)
C5_BARS = dict(max=0.31, mean=2.6e-2)


def test_config5_shape_one_million_vertices_five_levels():
    """BASELINE config 5 (roofline stress): 1M-vertex / 6M-edge synthetic mesh, 5 graph levels (67 M parameters,
    EdgeConv hidden width up to 2048).  Too large for the CPU oracle inside a test, so size-independent properties:
    finite tanh-range output, bit-reproducible, fp32 and bf16-storage runs agree within the bf16 tolerance, finite
    gradients for every parameter."""
    cfg = dict(CFG3D, n_levels=4)
    torch.manual_seed(49)
    net = S.define_G(**cfg).to(DEV)
    assert sum(p.numel() for p in net.parameters()) == 67_146_563          # SURVEY §8c probe constant
    s = make_synthetic_mesh(1_000_000, 5, seed=0).to(DEV)
    with torch.no_grad():
        a = net(s)
        b = net(s)
    assert a.shape == (s.x.shape[0], 3) and torch.equal(a, b)
    assert bool(torch.isfinite(a).all()) and float(a.abs().max()) <= 1.0
    net.set_activation_dtype(BF)
    out = net(s)
    d = (out.detach() - a).abs()
    print('config 5 (1 M vertices): bf16 vs fp32 storage max-abs %.3e mean-abs %.3e' % (float(d.max()), float(d.mean())))
    assert float(d.max()) <= C5_BARS['max'] and float(d.mean()) <= C5_BARS['mean'], (float(d.max()), float(d.mean()))
    loss = stin_oracle.compute_loss(torch.where((s.mask > 0).expand_as(s.color), out, s.color), s.color, s.mask)
    loss.backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())
