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
    # tolerances leave room for a different host CPU's BLAS summation order (bit-exact on the generating host)
    assert np.allclose(out.detach().numpy(), g['out_train'], rtol=0, atol=2e-5)
    loss = ((out - torch.from_numpy(g['target'])) ** 2).mean()
    assert abs(float(loss.detach()) - float(g['loss'])) <= 1e-5
    loss.backward()
    scale = max(float(np.abs(g['g_restatement/' + k]).max()) for k, _ in net.named_parameters())
    for k, p in net.named_parameters():
        assert float(np.abs(p.grad.numpy() - g['g_restatement/' + k]).max()) <= 2e-4 * scale, k
    for k, v in net.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            assert np.allclose(v.numpy(), g['after/' + k], rtol=1e-5, atol=1e-5), k
    net.eval()
    with torch.no_grad():
        assert np.allclose(net(sample).numpy(), g['out_eval'], rtol=0, atol=2e-5)


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
    # the fixture's running statistics are those of the reference after ONE forward (its backward cannot run on torch 2.x)
    once = {k: v.cpu().clone() for k, v in net.state_dict().items() if 'running' in k or 'num_batches' in k}
    for k, v in once.items():
        assert np.allclose(v.numpy(), g['after/' + k], rtol=1e-5, atol=1e-5), k
    loss.backward()
    scale = max(float(np.abs(g['g_restatement/' + k]).max()) for k, _ in net.named_parameters())
    for k, p in net.named_parameters():
        assert float((p.grad.cpu() - torch.from_numpy(g['g_restatement/' + k])).abs().max()) <= 2e-3 * scale, k
    # backward = where the reference recomputes its checkpointed blocks (left levels >= 1, right blocks but the last one,
    # models/singleconvmeshnet.py:124-126, :139-144): their BatchNorm statistics take the batch a second time
    twice = {k: v.cpu() for k, v in net.state_dict().items() if k in once}
    recomputed = [k for k in once if k.startswith(('left_geo_cnns.1.', 'left_geo_cnns.2.', 'right_geo_cnns.1.'))]
    assert recomputed and len(recomputed) < len(once)
    m = 0.1
    for k in once:
        if k not in recomputed:
            assert torch.equal(twice[k], once[k]), k
        elif k.endswith('num_batches_tracked'):
            assert int(twice[k]) == 2 and int(once[k]) == 1, k
        else:      # x1 = (1 - m) x0 + m b and x2 = (1 - m) x1 + m b with the same batch value b  =>  x2 = (2 - m) x1 - (1 - m) x0
            assert torch.allclose(twice[k], (2 - m) * once[k] - (1 - m) * state[k], rtol=1e-5, atol=1e-6), k
    net.load_state_dict({**net.state_dict(), **{k: v.to('cuda:0') for k, v in once.items()}})    # the fixture's eval state
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


@pytest.mark.gpu
@pytest.mark.parametrize('C', [64, 30])
@pytest.mark.parametrize('relu', [False, True])
def test_hip_fused_batchnorm_act_matches_fp64_autograd(C, relu):
    """stin_bn_act_{fwd,bwd}_f32 + the STIN_RED_DOT_BN[_RELU] column sums against nn.BatchNorm1d (+ReLU) in fp64."""
    from surface_texture_inpainting_net_amd.singleconvmeshnet import batch_norm_rows
    torch.manual_seed(3)
    n = 5003
    x = (torch.randn(n, C) * 2 + 0.5).to('cuda:0').requires_grad_()
    bn = torch.nn.BatchNorm1d(C).to('cuda:0')
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    w = torch.randn(n, C, device='cuda:0')
    y = batch_norm_rows(x, bn, relu=relu)
    (y * w).sum().backward()
    ref = torch.nn.BatchNorm1d(C).double()
    ref.load_state_dict({k: v.detach().cpu().double() if v.is_floating_point() else v.cpu()
                         for k, v in torch.nn.BatchNorm1d(C).state_dict().items()})
    with torch.no_grad():
        ref.weight.copy_(bn.weight.detach().cpu().double())
        ref.bias.copy_(bn.bias.detach().cpu().double())
    xr = x.detach().cpu().double().requires_grad_()
    yr = ref(xr)
    yr = torch.relu(yr) if relu else yr
    (yr * w.cpu().double()).sum().backward()
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) <= 1e-5
    assert float((x.grad.cpu().double() - xr.grad).abs().max()) <= 1e-4 * float(xr.grad.abs().max())
    for p, q in ((bn.weight, ref.weight), (bn.bias, ref.bias)):
        assert float((p.grad.cpu().double() - q.grad).abs().max()) <= 1e-4 * float(q.grad.abs().max())
    assert torch.allclose(bn.running_mean.cpu().double(), ref.running_mean, atol=1e-6)
    assert torch.allclose(bn.running_var.cpu().double(), ref.running_var, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('H,e', [(128, 4000), (6, 4000), (128, 70_001), (64, 33_333), (512, 9000)])    # (>= 4096 rows: the 4-rows-per-thread kernel)
def test_hip_gather_add_rows_bit_exact(H, e):
    """stin_gather_add_rows_f32: out[e] = y[dst[e], :H] + y[src[e], H:] equals the two-gather form bit for bit, and its
    backward equals index_add in fp64 to rounding."""
    from surface_texture_inpainting_net_amd.singleconvmeshnet import _GatherAddFn, _as_edge_index
    torch.manual_seed(5)
    n = 700
    ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(0, n, (e,))]).to('cuda:0')
    y = torch.randn(n, 2 * H, device='cuda:0', requires_grad=True)
    idx = _as_edge_index(ei, n)
    out = _GatherAddFn.apply(y, idx)
    assert torch.equal(out, y[ei[1], :H] + y[ei[0], H:])
    g = torch.randn(e, H, device='cuda:0')
    out.backward(g)
    ref = torch.zeros(n, 2 * H, dtype=torch.float64)
    ref[:, :H].index_add_(0, ei[1].cpu(), g.cpu().double())
    ref[:, H:].index_add_(0, ei[0].cpu(), g.cpu().double())
    assert float((y.grad.cpu().double() - ref).abs().max()) <= 1e-5 * max(1.0, e / 4000)


@pytest.mark.gpu
@pytest.mark.parametrize('C', [64, 6])
def test_hip_fused_batchnorm_then_mean_matches_fp64_autograd(C):
    """batch_norm_mean = scatter_mean(BatchNorm1d(m), dst) with statistics over the E edge rows, vertices without
    in-edges exactly 0 - forward, all three gradients and the running statistics against fp64 torch."""
    from surface_texture_inpainting_net_amd.singleconvmeshnet import _as_edge_index, batch_norm_mean
    torch.manual_seed(8)
    n, e = 900, 5000
    ei = torch.stack([torch.randint(0, n, (e,)), torch.randint(7, n, (e,))])      # vertices 0..6: no in-edge
    m = (torch.randn(e, C) * 1.5 + 0.3).to('cuda:0').requires_grad_()
    bn = torch.nn.BatchNorm1d(C).to('cuda:0')
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    w = torch.randn(n, C, device='cuda:0')
    out = batch_norm_mean(m, bn, _as_edge_index(ei.to('cuda:0'), n))
    assert out is not None and float(out[:7].abs().max()) == 0.0
    (out * w).sum().backward()
    ref = torch.nn.BatchNorm1d(C).double()
    with torch.no_grad():
        ref.weight.copy_(bn.weight.detach().cpu().double())
        ref.bias.copy_(bn.bias.detach().cpu().double())
    mr = m.detach().cpu().double().requires_grad_()
    y = ref(mr)
    cnt = torch.zeros(n, dtype=torch.float64).index_add_(0, ei[1], torch.ones(e, dtype=torch.float64)).clamp(min=1)
    outr = torch.zeros(n, C, dtype=torch.float64).index_add_(0, ei[1], y) / cnt[:, None]
    (outr * w.cpu().double()).sum().backward()
    assert float((out.detach().cpu().double() - outr.detach()).abs().max()) <= 1e-5
    assert float((m.grad.cpu().double() - mr.grad).abs().max()) <= 1e-4 * float(mr.grad.abs().max())
    for p, q in ((bn.weight, ref.weight), (bn.bias, ref.bias)):
        assert float((p.grad.cpu().double() - q.grad).abs().max()) <= 1e-4 * float(q.grad.abs().max())
    assert torch.allclose(bn.running_mean.cpu().double(), ref.running_mean, atol=1e-6)
    assert torch.allclose(bn.running_var.cpu().double(), ref.running_var, atol=1e-5)


GRAD_BAR = 1.8e-3    # 1.5 x the round-5 measurement on MI355X (max pooling 1.16e-3, mean pooling 2.4e-4: arg-max flips on a 6 000-vertex mesh)


@pytest.mark.gpu
@pytest.mark.parametrize('pooling', ['mean', 'max'])
def test_hip_fused_layer_equals_the_per_op_path(pooling, monkeypatch):
    """The fused EdgeConv(BN) layer node (round 5: one autograd node per layer, the framework's elementwise / copy / stack /
    accumulate launches folded into HIP kernels) runs the same kernels with the same arithmetic as the per-op path: outputs,
    every parameter gradient and every BatchNorm running statistic equal it bit for bit (incl. the second running-statistics
    update of the blocks the reference recomputes under torch.utils.checkpoint)."""
    from surface_texture_inpainting_net_amd import singleconvmeshnet as M
    from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
    s = make_synthetic_mesh(6000, 3, seed=9, dilations=()).to('cuda:0')
    tgt = torch.randn(s.x.shape[0], 5, generator=torch.Generator().manual_seed(1)).to('cuda:0')

    def run(fused, in_gemm=True):
        monkeypatch.setattr(M, 'USE_FUSED_LAYER', fused)
        monkeypatch.setattr(M, 'BN_IN_GEMM', in_gemm)
        monkeypatch.setattr(M, 'STATS_IN_GATHER', in_gemm)    # (fp64 sums in another order: mean / rstd equal to fp32 rounding)
        monkeypatch.setattr(M, 'BN_BWD_IN_GEMM', in_gemm)     # (likewise: the two column sums of the BatchNorm backward)
        torch.manual_seed(3)
        net = M.SingleConvMeshNet(10, 2, [16, 32, 64], num_classes=5, pooling_method=pooling).to('cuda:0')
        outs = []
        for _ in range(2):                                   # two steps: running statistics and counters accumulate
            net.zero_grad(set_to_none=True)
            out = net(s)
            ((out - tgt) ** 2).mean().backward()
            outs.append(out.detach().clone())
        return outs, [p.grad.clone() for p in net.parameters()], {k: v.clone() for k, v in net.state_dict().items()}

    o0, g0, s0 = run(False)
    names = [k for k, _ in M.SingleConvMeshNet(10, 2, [16, 32, 64], num_classes=5).named_parameters()]
    # the fused node with the normalised edge rows materialised: the same kernels on the same data -> bit for bit
    o1, g1, s1 = run(True, False)
    assert all(torch.equal(a, b) for a, b in zip(o0, o1))
    for k, a, b in zip(names, g0, g1):
        assert torch.equal(a, b), k
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
    # BatchNorm + ReLU applied inside the per-edge GEMMs' operand staging (stin_gemm_nt_bn_f32 / stin_gemm_tn_bn_f32: the
    # affine form relu(v s + t), equal to the two-pass expression to fp32 rounding) and the moments of the gather-add output
    # accumulated by the gather-add pass (stin_gather_add_rows_stats_f32): measured 1.2e-5 / 1.2e-3 of scale
    o2, g2, s2 = run(True, True)
    scale = max(float(a.abs().max()) for a in g0)
    eo = max(float((a - b).abs().max()) for a, b in zip(o0, o2))
    eg = max(float((a - b).abs().max()) for a, b in zip(g0, g2)) / scale
    print('\nBN-in-GEMM vs two-pass (%s pooling): outputs %.2e, worst gradient entry %.2e of scale' % (pooling, eo, eg))
    assert eo <= 2e-5
    for k, a, b in zip(names, g0, g2):
        assert float((a - b).abs().max()) <= GRAD_BAR * scale, k
    for k in s0:
        assert torch.allclose(s0[k].float(), s2[k].float(), rtol=1e-5, atol=1e-6), k


@pytest.mark.gpu
@pytest.mark.parametrize('H,e', [(128, 70_001), (64, 33_333), (512, 9000), (256, 1_000_000)])
def test_hip_gather_add_with_moments_equals_gather_add_plus_moments(H, e):
    """stin_gather_add_rows_stats_f32: the gather-add output is bit-identical to stin_gather_add_rows_f32's, and the folded
    (mean, rstd) of its per-block fp64 partials equal stin_colreduce_f32(MOMENTS) over the written matrix to fp32 rounding."""
    from surface_texture_inpainting_net_amd import _lib
    from surface_texture_inpainting_net_amd import functional as SF
    from surface_texture_inpainting_net_amd.singleconvmeshnet import _all_rows, _as_edge_index
    lib = _lib.load()
    n = 5000
    g = torch.Generator().manual_seed(H + e)
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)]).to('cuda:0')
    y = (torch.randn(n, 2 * H, generator=g) * 1.3 + 0.4).to('cuda:0')
    idx = _as_edge_index(ei, n)
    ref = y[ei[1], :H] + y[ei[0], H:]
    groups = int(lib.stin_gather_add_rows_stats_groups(e, H))
    assert 0 < groups <= 1024
    out = torch.empty(e, H, device='cuda:0')
    partial = torch.empty(groups, 2, H, dtype=torch.float64, device='cuda:0')
    SF._call('stin_gather_add_rows_stats_f32', SF._ptr(y), 2 * H, SF._ptr(idx.dst32), y.data_ptr() + 4 * H, 2 * H, SF._ptr(idx.src32), e, H,
             SF._ptr(out), H, SF._ptr(partial), partial.numel() * 8, SF._stream(y))
    assert torch.equal(out, ref)
    ge = _all_rows(e, y.device)
    mean, rstd = SF.moments_final(partial, ge.inv_cnt, eps=1e-5)
    m0, r0 = SF.colreduce(SF.RED_MOMENTS, ref, ge, ge.ptr_sum, eps=1e-5)
    assert float((mean - m0).abs().max()) <= 1e-6 * float(m0.abs().max() + 1) and float((rstd / r0 - 1).abs().max()) <= 1e-6
    p2 = torch.empty_like(partial)
    SF._call('stin_gather_add_rows_stats_f32', SF._ptr(y), 2 * H, SF._ptr(idx.dst32), y.data_ptr() + 4 * H, 2 * H, SF._ptr(idx.src32), e, H,
             SF._ptr(out), H, SF._ptr(p2), p2.numel() * 8, SF._stream(y))
    assert torch.equal(partial, p2), 'deterministic partials'


@pytest.mark.gpu
@pytest.mark.parametrize('C,n,e', [(64, 20_000, 120_000), (128, 6000, 30_011), (256, 1500, 9000), (32, 200_704, 1_200_642)])
def test_hip_segment_mean_with_moments_equals_the_two_passes(C, n, e):
    """stin_segment_mean_stats_f32: the per-target means equal stin_segment_sum_f32(mean) bit for bit (incl. empty rows), and the
    folded moments of the visited edge rows equal stin_colreduce_f32(MOMENTS) over all E rows to fp32 rounding."""
    from surface_texture_inpainting_net_amd import _lib
    from surface_texture_inpainting_net_amd import functional as SF
    from surface_texture_inpainting_net_amd.singleconvmeshnet import _all_rows, _as_edge_index
    lib = _lib.load()
    g = torch.Generator().manual_seed(C + n)
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(7, n, (e,), generator=g)]).to('cuda:0')   # rows 0..6: no in-edge
    m = (torch.randn(e, C, generator=g) * 0.8 - 0.2).to('cuda:0')
    idx = _as_edge_index(ei, n)
    ref = SF.segment_sum(m, idx.by_dst.rowptr, idx.by_dst.col, n, mean=True)
    groups = int(lib.stin_segment_mean_stats_groups(n, C))
    assert 0 < groups <= 2048
    out = torch.empty(n, C, device='cuda:0')
    partial = torch.empty(groups, 2, C, dtype=torch.float64, device='cuda:0')
    SF._call('stin_segment_mean_stats_f32', SF._ptr(m), C, SF._ptr(idx.by_dst.rowptr), SF._ptr(idx.by_dst.col), n, C, SF._ptr(out), C,
             SF._ptr(partial), partial.numel() * 8, SF._stream(m))
    assert torch.equal(out, ref) and float(out[:7].abs().max()) == 0.0
    ge = _all_rows(e, m.device)
    mean, rstd = SF.moments_final(partial, ge.inv_cnt, eps=1e-5)
    m0, r0 = SF.colreduce(SF.RED_MOMENTS, m, ge, ge.ptr_sum, eps=1e-5)
    assert float((mean - m0).abs().max()) <= 1e-6 and float((rstd / r0 - 1).abs().max()) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('e,h2,cout', [(70_001, 128, 64), (33_333, 256, 128), (9000, 512, 128), (5000, 96, 64), (3001, 32, 64), (1_200_642, 128, 64)])
def test_hip_gemm_with_batchnorm_backward_on_its_epilogue_equals_the_three_launches(e, h2, cout, monkeypatch):
    """stin_gemm_nt_bn_bwd_{stats,apply}_f32 (the per-edge input-gradient product run twice, BatchNorm1d + ReLU's backward on its
    epilogue) against stin_gemm_nt_f32 + stin_colreduce_f32(DOT_BN_RELU) + stin_bn_act_bwd_f32: the accumulators are the same
    numbers, so the two column sums agree to fp64-summation-order rounding and the finished gradient to a few fp32 ulps of
    its scale; both tile widths (two or four 32-column tiles per block); deterministic."""
    from surface_texture_inpainting_net_amd import _lib
    from surface_texture_inpainting_net_amd import functional as SF
    from surface_texture_inpainting_net_amd.singleconvmeshnet import _all_rows
    lib = _lib.load()
    g = torch.Generator().manual_seed(e + h2)
    dm = (torch.randn(e, cout, generator=g) * 0.3).to('cuda:0')
    w2T = (torch.randn(h2, cout, generator=g) * 0.1).to('cuda:0')
    pre = (torch.randn(e, h2, generator=g) * 1.1 + 0.2).to('cuda:0')
    gb = torch.stack([torch.rand(h2, generator=g) + 0.5, torch.randn(h2, generator=g) * 0.3]).to('cuda:0')
    ge = _all_rows(e, pre.device)
    mean, rstd = SF.colreduce(SF.RED_MOMENTS, pre, ge, ge.ptr_sum, eps=1e-5)
    st = SF._stream(pre)
    # the three-launch route
    dh = SF.gemm_nt(dm, w2T, precision=SF.PREC_BWD)
    P0, Q0 = SF.colreduce(SF.RED_DOT_BN_RELU, pre, ge, ge.ptr_sum, gout=dh, mean=mean, rstd=rstd, coef=gb)
    ref = torch.empty_like(dh)
    SF._call('stin_bn_act_bwd_f32', SF._ptr(pre), h2, SF._ptr(dh), h2, SF._ptr(mean), SF._ptr(rstd), SF._ptr(gb[0]), SF._ptr(gb[1]),
             SF._ptr(P0), SF._ptr(Q0), 1.0 / e, e, h2, 1, SF._ptr(ref), h2, st)
    for bn in ['2', '4']:
        monkeypatch.setenv('STIN_NT_STREAM_NT', bn)
        groups = int(lib.stin_gemm_nt_bn_bwd_groups(e, h2, cout, int(SF.PREC_BWD)))
        assert 0 < groups <= 2048
        args = (SF._ptr(dm), cout, SF._ptr(w2T), cout, SF._ptr(pre), h2, SF._ptr(mean), SF._ptr(rstd), SF._ptr(gb[0]), SF._ptr(gb[1]))
        res = []
        for _ in range(2):
            partial = torch.full((groups, 2, h2), float('nan'), dtype=torch.float64, device='cuda:0')
            pq = torch.empty(2, h2, device='cuda:0')
            SF._call('stin_gemm_nt_bn_bwd_stats_f32', *args, e, h2, cout, int(SF.PREC_BWD), SF._ptr(partial), partial.numel() * 8, SF._ptr(pq), st)
            out = torch.full((e, h2), float('nan'), device='cuda:0')
            SF._call('stin_gemm_nt_bn_bwd_apply_f32', *args, SF._ptr(pq), 1.0 / e, e, h2, cout, SF._ptr(out), h2, int(SF.PREC_BWD), st)
            res.append((pq, out))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), 'deterministic'
        pq, out = res[0]
        sp, sq = float(P0.abs().max()), float(Q0.abs().max())
        ep, eq = float((pq[0] - P0.view(-1)).abs().max()) / sp, float((pq[1] - Q0.view(-1)).abs().max()) / sq
        eo = float((out - ref).abs().max()) / float(ref.abs().max())
        print('\n[%d x %d x %d, tile width %s] P %.1e, Q %.1e, dx %.1e of scale' % (e, h2, cout, bn, ep, eq, eo))
        assert ep <= 2e-7 and eq <= 2e-7 and eo <= 1e-6
    monkeypatch.setenv('STIN_NT_BNBWD', '0')
    assert int(lib.stin_gemm_nt_bn_bwd_groups(e, h2, cout, int(SF.PREC_BWD))) == 0
    monkeypatch.delenv('STIN_NT_BNBWD')
    assert int(lib.stin_gemm_nt_bn_bwd_groups(e, 512, 256, int(SF.PREC_BWD))) == 0      # (K = 256: the three launches are faster)


@pytest.mark.gpu
@pytest.mark.parametrize('m,nc,k', [(70_001, 128, 64), (70_001, 64, 128), (33_333, 256, 128), (20_000, 128, 256), (5000, 96, 64), (66_000, 64, 128)])
def test_hip_streaming_rows_gemm_is_bit_identical_to_the_tiled_kernels(m, nc, k, monkeypatch):
    """stin_gemm_nt_stream_f32 (persistent blocks, wave-owned 32-row tiles through wave-private LDS, the weight slice split once per
    block) multiplies in the tiled kernels' k and MFMA order: the plain product and the product with BatchNorm1d + ReLU applied to
    the staged operand equal stin_gemm_nt_f32 / stin_gemm_nt_bn_f32 on the 64 x 64 tiling bit for bit, for every precision the
    edge MLP uses; the public entry points route M >= 65 536 rows to it."""
    from surface_texture_inpainting_net_amd import functional as SF
    g = torch.Generator().manual_seed(m + nc)
    A = (torch.randn(m, k, generator=g) * 0.7 + 0.1).to('cuda:0')
    W = (torch.randn(nc, k, generator=g) * 0.1).to('cuda:0')
    mean, rstd = (torch.randn(k, generator=g) * 0.1).to('cuda:0'), (torch.rand(k, generator=g) + 0.5).to('cuda:0')
    gamma, beta = (torch.rand(k, generator=g) + 0.5).to('cuda:0'), (torch.randn(k, generator=g) * 0.2).to('cuda:0')
    st = SF._stream(A)
    monkeypatch.setenv('STIN_NT_STREAM', '0')
    for prec in (SF.PREC_FWD, SF.PREC_BWD, SF.GEMM_BF16X6):                      # (three 16-bit pieces: 6 bytes per element of LDS)
        ref = SF.gemm_nt(A, W, precision=prec)
        ref_bn = torch.empty(m, nc, device='cuda:0')
        SF._call('stin_gemm_nt_bn_f32', SF._ptr(A), k, SF._ptr(W), k, SF._ptr(mean), SF._ptr(rstd), SF._ptr(gamma), SF._ptr(beta), m, nc, k,
                 SF._ptr(ref_bn), nc, int(prec), st)
        for nt in ('2', '4'):
            monkeypatch.setenv('STIN_NT_STREAM_NT', nt)
            out = torch.full((m, nc), float('nan'), device='cuda:0')
            SF._call('stin_gemm_nt_stream_f32', SF._ptr(A), k, SF._ptr(W), k, None, None, None, None, m, nc, k, SF._ptr(out), nc, int(prec), st)
            assert torch.equal(out, ref), (prec, nt)
            out.fill_(float('nan'))
            SF._call('stin_gemm_nt_stream_f32', SF._ptr(A), k, SF._ptr(W), k, SF._ptr(mean), SF._ptr(rstd), SF._ptr(gamma), SF._ptr(beta), m, nc,
                     k, SF._ptr(out), nc, int(prec), st)
            assert torch.equal(out, ref_bn), (prec, nt, 'bn')
    monkeypatch.delenv('STIN_NT_STREAM_NT')
    monkeypatch.setenv('STIN_NT_STREAM', '1')
    assert torch.equal(SF.gemm_nt(A, W, precision=SF.GEMM_BF16X6), ref)            # (routed by size; same numbers either way)
    if m >= 65536:                                                                   # the epilogue forms the public entry point routes as well
        b = torch.randn(nc, generator=g).to('cuda:0')
        mask = (torch.rand(m, generator=g) < 0.7).float().to('cuda:0')
        res = torch.randn(m, nc, generator=g).to('cuda:0')
        for kw in (dict(bias=b, row_mask=mask), dict(bias=b, residual=res), dict(residual=res)):
            monkeypatch.setenv('STIN_NT_STREAM', '0')
            want = SF.gemm_nt(A, W, precision=SF.PREC_BWD, **kw)
            monkeypatch.setenv('STIN_NT_STREAM', '1')
            assert torch.equal(SF.gemm_nt(A, W, precision=SF.PREC_BWD, **kw), want), sorted(kw)


@pytest.mark.gpu
@pytest.mark.parametrize('cs,cu', [(64, 128), (6, 10), (128, 256)])
def test_hip_skip_unpool_concat_equals_cat_of_the_gather(cs, cu):
    """stin_concat_unpool_f32 / _SkipUnpoolConcatFn: torch.cat((skip, coarse[trace]), -1) in one launch - forward bit for bit, and
    both gradients (the left column block as a view; the segment sum of the right one over each coarse vertex's children) equal
    to the cat + UnpoolFn composition bit for bit."""
    from surface_texture_inpainting_net_amd import functional as SF
    from surface_texture_inpainting_net_amd.plan import PoolMap
    from surface_texture_inpainting_net_amd.singleconvmeshnet import _SkipUnpoolConcatFn
    g = torch.Generator().manual_seed(cs + cu)
    n_fine, n_coarse = 20_011, 3000
    trace = torch.randint(0, n_coarse, (n_fine,), generator=g)
    trace[:n_coarse] = torch.arange(n_coarse)                                      # every coarse vertex has a child
    bad = torch.zeros(1, dtype=torch.int32, device='cuda:0')
    pool = PoolMap(trace.to('cuda:0'), n_fine, n_coarse, bad)
    assert int(bad.item()) == 0
    skip = torch.randn(n_fine, cs, generator=g).to('cuda:0').requires_grad_(True)
    coarse = torch.randn(n_coarse, cu, generator=g).to('cuda:0').requires_grad_(True)
    w = torch.randn(n_fine, cs + cu, generator=g).to('cuda:0')
    ref = torch.cat((skip, SF.UnpoolFn.apply(coarse, pool)), -1)
    (ref * w).sum().backward()
    gs, gc = skip.grad.clone(), coarse.grad.clone()
    skip.grad = coarse.grad = None
    out = _SkipUnpoolConcatFn.apply(skip, coarse, pool)
    assert torch.equal(out, ref)
    (out * w).sum().backward()
    assert torch.equal(skip.grad, gs) and torch.equal(coarse.grad, gc)
