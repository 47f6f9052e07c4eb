"""GPU parity tests proper: every case goes through the C ABI of libstin_hip.so (via the autograd
layer) on a real MI355X and is compared with the CPU oracle on the same seeded inputs, with the
committed golden fixtures (generated from the reference's own classes), and - at BASELINE.json's
full 200k-vertex size - through size-independent properties.

Tolerances (BASELINE.json north_star / SURVEY §8d): forward max-abs <= 1e-4 in fp32 versus the CPU
reference; integer / index results (CSR plans, pool arg-max, batch vectors) bit-exact; gradients
within 1e-3 of the fixture-wide gradient scale (summation-order noise over 1e5-1e6 terms).
"""
import numpy as np
import pytest
import torch

from _golden import MODEL_FIXTURES, ModelFixture, grad_flip_report, load_npz
from oracle import scatter_ops, stin_oracle
from surface_texture_inpainting_net_amd import functional as SF
from surface_texture_inpainting_net_amd import modules as M
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S
from surface_texture_inpainting_net_amd.data import HierarchicalBatch
from surface_texture_inpainting_net_amd.plan import EdgeSet, GraphPlan, NormGroups, PoolMap, build_csr, plan_for
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
FWD_TOL = 1e-4


def _bad():
    return torch.zeros(1, dtype=torch.int32, device=DEV)


def _random_graph(n, e, seed, isolated=5):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.randint(isolated, n, (e,), generator=g)      # vertices < isolated have no in-edge
    return torch.stack([src, dst])


# ------------------------------------------------------------------------------- plan
@pytest.mark.parametrize('n,e', [(1, 0), (7, 1), (50, 400), (1000, 0), (4099, 30011), (200_000, 1_200_000)])
def test_csr_plan_bit_exact(n, e):
    g = torch.Generator().manual_seed(n + e)
    key = torch.randint(0, n, (e,), generator=g)
    val = torch.randint(0, n + 3, (e,), generator=g)
    bad = _bad()
    csr = build_csr(key.to(DEV), val.to(DEV), n, n + 3, bad, want_perm=True)
    order = np.argsort(key.numpy(), kind='stable')
    cnt = np.bincount(key.numpy(), minlength=n)
    rowptr = np.concatenate([[0], np.cumsum(cnt)])
    assert np.array_equal(csr.rowptr.cpu().numpy(), rowptr)
    assert np.array_equal(csr.perm.cpu().numpy(), order)
    assert np.array_equal(csr.col.cpu().numpy(), val.numpy()[order])
    assert np.array_equal(csr.inv_deg.cpu().numpy(), (1.0 / np.maximum(cnt, 1)).astype(np.float32))
    assert int(bad.item()) == 0


@pytest.mark.parametrize('n,e', [(5, 0), (64, 500), (4099, 30011), (200_000, 1_200_000)])
def test_edge_csr_pair_bit_exact(n, e):
    g = torch.Generator().manual_seed(7 * n + e)
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)])
    bad = _bad()
    es = EdgeSet(ei.to(DEV), n, bad)
    src, dst = ei[0].numpy(), ei[1].numpy()
    for csr, key, val in ((es.by_dst, dst, src), (es.by_src, src, dst)):
        order = np.argsort(key, kind='stable')
        cnt = np.bincount(key, minlength=n)
        assert np.array_equal(csr.rowptr.cpu().numpy(), np.concatenate([[0], np.cumsum(cnt)]))
        assert np.array_equal(csr.col.cpu().numpy(), val[order])
    assert np.array_equal(es.inv_deg.cpu().numpy(), (1.0 / np.maximum(np.bincount(dst, minlength=n), 1)).astype(np.float32))
    assert int(bad.item()) == 0


def test_plan_build_by_sort_equals_the_counting_sort(monkeypatch):
    """stin_plan_build_many by ONE stable radix sort (round 4, the default) against the counting sort with returning atomics
    (STIN_PLAN_SORT=0): every output of a whole scene's batched build - both CSRs of every edge set, cross map, source weights,
    pool maps with the narrowed trace - identical, including dropped out-of-range pairs, empty rows and an empty edge set."""
    s = make_synthetic_mesh(9000, 3, seed=5, dilations=(2, 4))
    ei0 = s.edge_index.clone()
    ei0[0, 17] = 10 ** 9                      # one out-of-range source, one negative target
    ei0[1, 4000] = -3
    jobs_in = [(ei0, int(s.num_vertices[0, 0])), (s['hierarchy_edge_index_1'], int(s.num_vertices[0, 1])),
               (torch.zeros(2, 0, dtype=torch.long), 50), (torch.randint(0, 7, (2, 300)), 400)]

    def run(flag):
        monkeypatch.setenv('STIN_PLAN_SORT', flag)
        out = []
        bad = _bad()
        from surface_texture_inpainting_net_amd.plan import PlanJobs
        jb = PlanJobs()
        sets = [EdgeSet(ei.to(DEV), n, bad, jobs=jb) for ei, n in jobs_in]
        pm = PoolMap(s['hierarchy_trace_index_1'].to(DEV), int(s.num_vertices[0, 0]), int(s.num_vertices[0, 1]), bad, jb)
        jb.run(bad, torch.device(DEV))
        torch.cuda.synchronize()
        for es in sets:                           # (slots behind the last valid entry - dropped pairs - are never written: compare what is)
            nd, ns = int(es.by_dst.rowptr[-1]), int(es.by_src.rowptr[-1])
            out += [es.by_dst.rowptr, es.by_dst.col[:nd], es.by_src.rowptr, es.by_src.col[:ns], es.xslot[:ns], es.w_src[:ns], es.inv_deg]
        out += [pm.children.rowptr, pm.children.col, pm.trace]
        single = build_csr(torch.tensor([3, 1, 3, 0, 3], device=DEV), torch.tensor([9, 8, 7, 6, 5], device=DEV), 5, 10, bad, want_perm=True)
        out += [single.rowptr, single.col, single.perm, single.inv_deg, bad]
        return [t.clone() for t in out]

    want, got = run('0'), run('1')
    assert len(want) == len(got)
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), i
    assert int(got[-1].item()) != 0           # the out-of-range pairs were flagged by both


def test_csr_flags_out_of_range_indices():
    key = torch.tensor([0, 1, 5, 2], device=DEV)
    bad = _bad()
    build_csr(key, None, 4, 4, bad)
    assert int(bad.item()) != 0
    bad = _bad()
    build_csr(torch.tensor([0, 1], device=DEV), torch.tensor([0, -1], device=DEV), 4, 4, bad)
    assert int(bad.item()) != 0
    s = make_synthetic_mesh(100, 2, seed=0, dilations=()).to(DEV)
    s['hierarchy_trace_index_1'][3] = 10_000
    net = S.define_G(input_nc=10, output_nc=3, ngf=8, filter_type='edgeconv', norm='instance', n_blocks=1, n_levels=1,
                     pooling_type='max').to(DEV)
    with pytest.raises(IndexError):
        net(s)


# ------------------------------------------------------------------------- edge stage
def _edge_oracle(A, B, ei):
    return scatter_ops.scatter_mean(torch.relu(A[ei[1]] + B[ei[0]]), ei[1], dim=0, dim_size=A.shape[0])


@pytest.mark.parametrize('H', [4, 6, 16, 24, 64, 128, 256, 512, 1024])
def test_edge_stage_forward_backward(H):
    n, e = 777, 5000
    ei = _random_graph(n, e, seed=H)
    g = torch.Generator().manual_seed(H)
    A = torch.randn(n, H, generator=g, requires_grad=True)
    B = torch.randn(n, H, generator=g, requires_grad=True)
    W = torch.randn(n, H, generator=g)
    want = _edge_oracle(A, B, ei)
    (want * W).sum().backward()
    es = EdgeSet(ei.to(DEV), n, _bad())
    Ad = A.detach().to(DEV).requires_grad_(True)
    Bd = B.detach().to(DEV).requires_grad_(True)
    got = SF.EdgeReluMeanFn.apply(Ad, Bd, es)
    (got * W.to(DEV)).sum().backward()
    assert float((got.cpu() - want).abs().max()) <= 2e-6
    assert float(got[:5].abs().max()) == 0.0                   # no in-edges -> exactly 0
    assert float((Ad.grad.cpu() - A.grad).abs().max()) <= 1e-5
    assert float((Bd.grad.cpu() - B.grad).abs().max()) <= 1e-5


@pytest.mark.parametrize('H', [128, 256, 512, 1024, 2048])
def test_edge_stage_saved_mask_backward_equals_recompute(H):
    """The forward's ReLU bit-mask (E*H/8 bytes) drives a backward that must equal the recompute form bit for bit."""
    n, e = 1500, 9000
    ei = _random_graph(n, e, seed=H + 1)
    es = EdgeSet(ei.to(DEV), n, _bad())
    g = torch.Generator().manual_seed(H)
    Y = torch.randn(n, 2 * H + 8, generator=g).to(DEV)           # A and B as column slices of a wider matrix
    A, B = Y[:, :H], Y[:, H:2 * H]
    Gr = torch.randn(n, H, generator=g).to(DEV)
    out0 = torch.empty(n, H + 4, device=DEV)
    out1 = torch.empty(n, H + 4, device=DEV)
    mask = torch.zeros(e * (H // 32), dtype=torch.int32, device=DEV)
    SF.edge_relu_mean_fwd(A, B, es.by_dst, out0, indicator=True)
    SF.edge_relu_mean_fwd(A, B, es.by_dst, out1, indicator=True, mask=mask)
    assert torch.equal(out0, out1)
    # every slot carries exactly as many set bits as positive pre-activations (bit order is kernel-private)
    dst = torch.repeat_interleave(torch.arange(n, device=DEV), (es.by_dst.rowptr[1:] - es.by_dst.rowptr[:-1]).long())
    want_pop = (A[dst] + B[es.by_dst.col.long()] > 0).sum(1)
    words = mask.view(e, H // 32).long() & 0xFFFFFFFF
    got_pop = ((words.unsqueeze(-1) >> torch.arange(32, device=DEV)) & 1).sum((1, 2))
    assert torch.equal(got_pop, want_pop)
    dA0, dA1, dB0, dB1 = (torch.empty(n, H, device=DEV) for _ in range(4))
    SF.edge_relu_mean_bwd_dst(A, B, Gr, es.by_dst, dA0)
    SF.edge_relu_mean_bwd_dst_mask(Gr, mask, es.by_dst, dA1)
    SF.edge_relu_mean_bwd_src(A, B, Gr, es.inv_deg, es.by_src, dB0)
    SF.edge_relu_mean_bwd_src_mask(Gr, mask, es, dB1)
    assert torch.equal(dA0, dA1)
    assert torch.equal(dB0, dB1)
    # both halves in one launch, written into column slices of one wider matrix as the block backward does
    dY = torch.full((n, 2 * H + 8), 7.0, device=DEV)
    SF.edge_relu_mean_bwd_mask(Gr, mask, es, dY[:, :H], dY[:, H:2 * H])
    assert torch.equal(dY[:, :H], dA1) and torch.equal(dY[:, H:2 * H], dB1)
    assert float((dY[:, 2 * H:] - 7.0).abs().max()) == 0.0
    # ... with the row-copy rider (the shortcut gradient of the block backward): H / 2 channels from a strided source
    dZ = torch.full((n, 2 * H + H // 2 + 4), 7.0, device=DEV)
    src = torch.randn(n, H, device=DEV)[:, :H // 2]
    SF.edge_relu_mean_bwd_mask(Gr, mask, es, dZ[:, :H], dZ[:, H:2 * H], copy_src=src, copy_dst=dZ[:, 2 * H:2 * H + H // 2])
    assert torch.equal(dZ[:, :H], dA1) and torch.equal(dZ[:, H:2 * H], dB1) and torch.equal(dZ[:, 2 * H:2 * H + H // 2], src)
    assert float((dZ[:, 2 * H + H // 2:] - 7.0).abs().max()) == 0.0


def test_edge_stage_on_column_slices_and_indicator():
    n, H = 300, 32
    ei = _random_graph(n, 2000, seed=1)
    Y = torch.randn(n, 2 * H + 8, device=DEV)
    es = EdgeSet(ei.to(DEV), n, _bad())
    hE = torch.full((n, H + 4), 7.0, device=DEV)
    SF.edge_relu_mean_fwd(Y[:, :H], Y[:, H:2 * H], es.by_dst, hE, indicator=True)
    want = _edge_oracle(Y[:, :H].cpu(), Y[:, H:2 * H].cpu(), ei)
    assert float((hE[:, :H].cpu() - want).abs().max()) <= 2e-6
    deg = torch.bincount(ei[1], minlength=n)
    assert torch.equal(hE[:, H].cpu(), (deg > 0).float())
    assert float(hE[:, H + 1:].abs().max()) == 0.0


@pytest.mark.parametrize('trans_inv', [False, True])
def test_edgeconv_filter_matches_unfused_pyg_form(trans_inv):
    n, cin, cout = 500, 10, 16
    ei = _random_graph(n, 3000, seed=2)
    torch.manual_seed(5)
    f = M.get_gcn_filter(cin, cout, module=M.EdgeConvTransInv if trans_inv else None, double_input=not trans_inv)
    for p in f.parameters():
        if p.dim() == 1:
            torch.nn.init.normal_(p, 0, 0.3)
    x = torch.randn(n, cin)
    want = stin_oracle.edge_conv(x, ei, f.nn[0].weight, f.nn[0].bias, f.nn[2].weight, f.nn[2].bias, trans_inv)
    got = f.to(DEV)(x.to(DEV), ei.to(DEV))
    assert float((got.cpu() - want).abs().max()) <= 1e-5
    assert float(got[:5].abs().max()) == 0.0


# ------------------------------------------------------------------------ MFMA GEMMs
@pytest.mark.parametrize('M,Nc,K', [(1, 3, 10), (37, 3, 64), (1000, 128, 10), (4097, 320, 64), (3001, 64, 132),
                                     (2500, 256, 260), (777, 1024, 256), (513, 256, 1280), (300, 24, 20)])
def test_gemm_nt_and_tn_against_fp64(M, Nc, K):
    g = torch.Generator().manual_seed(M + Nc + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(Nc, K, generator=g)
    b = torch.randn(Nc, generator=g)
    G = torch.randn(M, Nc, generator=g)
    want = (A.double() @ W.double().t() + b.double())
    got = SF.gemm_nt(A.to(DEV), W.to(DEV), b.to(DEV)).cpu().double()
    assert float((got - want).abs().max()) <= 2e-6 * (K ** 0.5) * float(want.abs().max() + 1)
    # exact fp32 fmaf-chain numerics: identical to a sequential fp32 reference to ~1 ulp of the partial sums
    want_tn = torch.cat([G.double().t() @ A.double(), G.double().sum(0)[:, None]], 1)
    got_tn = SF.gemm_tn(G.to(DEV), A.to(DEV), ones_column=True).cpu().double()
    assert float((got_tn - want_tn).abs().max()) <= 2e-6 * (M ** 0.5) * float(want_tn.abs().max() + 1) / 10 + 1e-5
    got_tn2 = SF.gemm_tn(G.to(DEV), A.to(DEV)).cpu().double()
    assert torch.equal(got_tn2, got_tn[:, :-1]), 'deterministic slab order'
    for prec, tol in ((SF.GEMM_BF16X3, 3e-5), (SF.GEMM_BF16X6, 3e-6)):
        got_p = SF.gemm_tn(G.to(DEV), A.to(DEV), ones_column=True, precision=prec).cpu().double()
        assert float((got_p - want_tn).abs().max()) <= tol * float(want_tn.abs().max()) + 1e-6, prec
        assert torch.equal(got_p, SF.gemm_tn(G.to(DEV), A.to(DEV), ones_column=True, precision=prec).cpu().double())


@pytest.mark.parametrize('M', [600_000, 1_300_000, 2_800_000])
def test_skinny_weight_gradient_product_on_long_matrices(M):
    """k_gemm_tn_skinny (K <= 16: the first block's weight gradient) stages a chunk's X rows in dynamic LDS.  The chunk is
    capped at 512 rows, so meshes beyond ~1 M vertices get MORE chunks, not a launch that asks for more than 64 KB (which
    failed the whole backward before the cap).  Exact fp32 products: compared against fp64 sums."""
    lib = _lib_load()
    Nc, K = 64, 12
    assert lib.stin_gemm_tn_workspace_bytes(M, Nc, K, 1) >= ((M + 511) // 512) * Nc * K * 4
    g = torch.Generator(device=DEV).manual_seed(M % 1000)
    G = torch.randn(M, Nc, generator=g, device=DEV)
    X = torch.randn(M, K, generator=g, device=DEV)
    got = SF.gemm_tn(G, X, ones_column=True, precision=SF.PREC_BWD).double()
    want = torch.cat([G.double().t() @ X.double(), G.double().sum(0)[:, None]], 1)
    assert float((got - want).abs().max()) <= 1e-6 * (M ** 0.5) * 8
    assert torch.equal(got, SF.gemm_tn(G, X, ones_column=True, precision=SF.PREC_BWD).double())


@pytest.mark.parametrize('M,Nc,K', [(4097, 320, 64), (2500, 256, 260), (513, 256, 1280), (1000, 128, 10)])
def test_gemm_nt_precision_modes(M, Nc, K):
    """fp32 MFMA chain vs the split-bf16 paths (fp32 accumulate): error against fp64, relative to the
    result scale.  x6 (exact 3-piece split) must be at least as good as fp32; x3 within 2e-5."""
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(Nc, K, generator=g) * 0.1).to(DEV)
    b = torch.randn(Nc, generator=g).to(DEV)
    want = A.double() @ W.double().t() + b.double()
    scale = float(want.abs().max())
    err = {p: float((SF.gemm_nt(A, W, b, precision=p).double() - want).abs().max()) / scale
           for p in (SF.GEMM_F32, SF.GEMM_BF16X3, SF.GEMM_BF16X6, SF.GEMM_F16X3)}
    assert err[SF.GEMM_F32] <= 3e-6
    assert err[SF.GEMM_BF16X6] <= 3e-6
    assert err[SF.GEMM_F16X3] <= 3e-6          # 2 fp16 pieces = 22 bits on unit-scale operands
    assert err[SF.GEMM_BF16X3] <= 2e-5
    small = torch.randint(-3, 4, (M, K), generator=g).float().to(DEV)          # exactly representable: all paths exact
    wi = torch.randint(-3, 4, (Nc, K), generator=g).float().to(DEV)
    for p in (SF.GEMM_F32, SF.GEMM_BF16X3, SF.GEMM_BF16X6, SF.GEMM_F16X3):
        assert torch.equal(SF.gemm_nt(small, wi, precision=p).double(), small.double() @ wi.double().t())
    # pre-split weight operand (split once per step by the pack kernel): the very same arithmetic
    if K % 4 == 0:
        for p in (SF.GEMM_BF16X3, SF.GEMM_F16X3):
            assert torch.equal(SF.gemm_nt(A, SF.split_weights(W, p), b, precision=p | SF.GEMM_W_PRESPLIT),
                               SF.gemm_nt(A, W, b, precision=p)), p
    # fp16 split: out-of-range operands must fail loudly (inf/NaN), tiny ones degrade to absolute precision
    big = A.clone()
    big[0, 0] = 1e4
    assert not bool(torch.isfinite(SF.gemm_nt(big, W, precision=SF.GEMM_F16X3)[0]).all())
    tiny = SF.gemm_nt(A * 1e-3, W, precision=SF.GEMM_F16X3).double() * 1e3
    assert float((tiny - (want - b.double())).abs().max()) / scale <= 2e-5


@pytest.mark.parametrize('M,Nc,K', [(1, 320, 128), (37, 320, 192), (300, 352, 128), (4097, 320, 128), (5000, 1024, 256), (18063, 640, 256),
                                    (2500, 1280, 128), (64, 384, 256), (129, 1024, 128), (777, 1280, 256), (2049, 96, 576),
                                    (130, 64, 36), (1000, 256, 640), (3001, 128, 320), (1, 256, 128), (63, 256, 1024),
                                    (18063, 256, 1024), (18063, 256, 512), (20001, 128, 256), (4100, 128, 1280), (200, 256, 64),
                                    (129, 128, 192), (515, 192, 256)])
def test_gemm_nt_strip_kernel_equals_tiled_kernel(M, Nc, K, monkeypatch):
    """(STIN_NT_PANEL=0: the round-4 column-panel kernel, which now takes some of these shapes by default, has its own test below.)
    The NT kernels that read the weight operand in MFMA fragment order straight from L2 - the resident-strip kernel
    (k_gemm_nt_strip: 64-row strips resident in LDS, Nc >= 320, K <= 256) and the all-columns kernel (k_gemm_nt_wide:
    Nc = 128 / 256, any K, A streamed once through a double-buffered LDS chunk) - against the tiled kernel on the k-group
    layout: same arithmetic in the same order -> bit-identical.  Shapes neither takes (stin_gemm_w_is_frag) keep the
    k-group layout and the tiled kernel under the same flag.  Both against fp64."""
    monkeypatch.setenv('STIN_NT_PANEL', '0')
    g = torch.Generator().manual_seed(M + Nc + K)
    A = torch.randn(M + 3, K + 4, generator=g).to(DEV)[1:M + 1, :K]            # a strided view: lda != K
    W = (torch.randn(Nc, K, generator=g) * 0.1).to(DEV)
    b = torch.randn(Nc, generator=g).to(DEV)
    mask = (torch.rand(M, 3, generator=g) < 0.7).float().to(DEV)[:, 1]
    res = torch.randn(M, Nc, generator=g).to(DEV)
    want = A.double() @ W.double().t()
    scale = float(want.abs().max()) + 1.0
    FR = 0x400
    frag = K % 64 == 0 and ((Nc == 256 and K >= 512) or (Nc == 128 and K >= 256) or (Nc % 32 == 0 and 128 <= K <= 256 and Nc >= 320))
    assert bool(_lib_load().stin_gemm_w_is_frag(Nc, K)) == frag
    for prec in (SF.GEMM_F16X3, SF.GEMM_BF16X3):
        Wk, Wf = SF.split_weights(W, prec), SF.split_weights(W, prec | FR)
        assert torch.equal(Wk, Wf) == (not frag)
        for kw in (dict(), dict(bias=b), dict(bias=b, row_mask=mask), dict(residual=res), dict(bias=b, residual=res)):
            tiled = SF.gemm_nt(A, Wk, precision=prec | SF.GEMM_W_PRESPLIT, **kw)
            for _ in range(2):                                  # twice: the ring / unit logic must not depend on what ran before
                strip = SF.gemm_nt(A, Wf, precision=prec | SF.GEMM_W_PRESPLIT | FR, **kw)
                assert torch.equal(strip, tiled), (prec, sorted(kw))
        wide = torch.full((M + 2, Nc + 8), 3.0, device=DEV)                  # output as a view into a wider matrix
        SF.gemm_nt(A, Wf, b, residual=res, out=wide[1:M + 1, 4:Nc + 4], precision=prec | SF.GEMM_W_PRESPLIT | FR)
        assert torch.equal(wide[1:M + 1, 4:Nc + 4], SF.gemm_nt(A, Wk, b, residual=res, precision=prec | SF.GEMM_W_PRESPLIT))
        wide[1:M + 1, 4:Nc + 4] = 3.0
        assert float((wide - 3.0).abs().max()) == 0.0
        ref = want + b.double() * mask.double()[:, None]
        tol = 3e-6 if prec == SF.GEMM_F16X3 else 2e-5
        assert float((SF.gemm_nt(A, Wf, b, row_mask=mask, precision=prec | SF.GEMM_W_PRESPLIT | FR).double() - ref).abs().max()) <= tol * scale


@pytest.mark.parametrize('M,Nc,K', [(8100, 4096, 1024), (8100, 1024, 2048), (2700, 512, 1024), (1000, 384, 256), (300, 1024, 320), (129, 256, 64),
                                    (1, 128, 64), (127, 1152, 128), (4097, 2048, 512)])
def test_gemm_nt_bf16_lds_dma_kernel_equals_register_staged_kernel(M, Nc, K, monkeypatch):
    """k_gemm_nt_b16_glds (128 x 128 tile, operand tiles staged by LDS-DMA into two buffers, source-side swizzle; the fat
    shapes of the 5-level hierarchy) against the register-staged bf16 kernel: same MFMA sequence per output element ->
    bit-identical, bf16 and fp32 outputs, bias / masked bias / residual, ragged row and column tiles (rows past M / Nc are
    clamped sources), both block -> tile orders (column tiles a multiple of 8 or not).  And against fp64."""
    g = torch.Generator().manual_seed(M + Nc + K)
    A = torch.randn(M + 2, K + 8, generator=g).to(DEV).bfloat16()[1:M + 1, :K]            # lda != K
    W = (torch.randn(Nc, K, generator=g) * 0.1).to(DEV).bfloat16()
    b = torch.randn(Nc, generator=g).to(DEV)
    mask = (torch.rand(M, 3, generator=g) < 0.7).to(DEV).bfloat16()[:, 1]
    res = torch.randn(M, Nc, generator=g).to(DEV).bfloat16()
    for kw in (dict(), dict(bias=b), dict(bias=b, row_mask=mask), dict(residual=res), dict(bias=b, residual=res),
               dict(bias=b, out_dtype=torch.float32)):
        monkeypatch.setenv('STIN_NT_GLDS', '0')
        base = SF.gemm_nt(A, W, **kw)
        monkeypatch.setenv('STIN_NT_GLDS', '1')
        for big in ('0', '1'):                                  # 128 x 128 tiles on 4 waves, 256 x 256 tiles on 8 waves
            monkeypatch.setenv('STIN_NT_BIG', big)
            for _ in range(2):
                got = SF.gemm_nt(A, W, **kw)
                assert got.dtype == base.dtype and torch.equal(got, base), (big, sorted(kw))
        monkeypatch.delenv('STIN_NT_BIG')
    wide = torch.full((M + 2, Nc + 16), 3.0, device=DEV).bfloat16()                     # output as a view into a wider matrix
    SF.gemm_nt(A, W, b, out=wide[1:M + 1, 8:Nc + 8])
    assert torch.equal(wide[1:M + 1, 8:Nc + 8], base.bfloat16() if base.dtype != torch.bfloat16 else SF.gemm_nt(A, W, b))
    wide[1:M + 1, 8:Nc + 8] = 3.0
    assert float((wide.float() - 3.0).abs().max()) == 0.0
    want = A.double() @ W.double().t() + b.double()
    got = SF.gemm_nt(A, W, b, out_dtype=torch.float32).double()
    assert float((got - want).abs().max()) <= 2e-5 * (float(want.abs().max()) + 1.0) * max(1.0, (K / 256) ** 0.5)


@pytest.mark.parametrize('cfg', ['22', '41'])
@pytest.mark.parametrize('M,Nc,K', [(1, 320, 128), (37, 320, 192), (300, 352, 128), (5000, 1024, 256), (18063, 640, 256), (129, 1024, 128),
                                    (777, 1280, 256), (64, 384, 256), (18063, 512, 256), (127, 320, 256), (128, 320, 256), (200, 320, 128)])
def test_gemm_nt_strip_configurations_are_bit_identical(M, Nc, K, cfg, monkeypatch):
    """k_gemm_nt_strip<PT, MT, QM>: 128-row strips on two wave quads (STIN_STRIP_CFG=22) or on one quad with four row tiles
    per wave (41) against the 64-row configuration (21): the same MFMA sequence per output element -> bit-identical, with
    every epilogue (bias, masked bias, residual, strided output), ragged last strips and unit ranges that cross strips."""
    g = torch.Generator().manual_seed(M + Nc + K)
    A = torch.randn(M + 3, K + 4, generator=g).to(DEV)[1:M + 1, :K]
    W = (torch.randn(Nc, K, generator=g) * 0.1).to(DEV)
    b = torch.randn(Nc, generator=g).to(DEV)
    mask = (torch.rand(M, 3, generator=g) < 0.7).float().to(DEV)[:, 1]
    res = torch.randn(M, Nc, generator=g).to(DEV)
    FR = 0x400
    assert _lib_load().stin_gemm_w_is_frag(Nc, K)
    for prec in (SF.GEMM_F16X3, SF.GEMM_BF16X3):
        Wf = SF.split_weights(W, prec | FR)
        for kw in (dict(), dict(bias=b), dict(bias=b, row_mask=mask), dict(residual=res), dict(bias=b, residual=res)):
            monkeypatch.setenv('STIN_STRIP_CFG', '21')
            base = SF.gemm_nt(A, Wf, precision=prec | SF.GEMM_W_PRESPLIT | FR, **kw)
            monkeypatch.setenv('STIN_STRIP_CFG', cfg)
            for _ in range(2):
                got = SF.gemm_nt(A, Wf, precision=prec | SF.GEMM_W_PRESPLIT | FR, **kw)
                assert torch.equal(got, base), (prec, sorted(kw))
        wide = torch.full((M + 2, Nc + 8), 3.0, device=DEV)
        SF.gemm_nt(A, Wf, b, out=wide[1:M + 1, 4:Nc + 4], precision=prec | SF.GEMM_W_PRESPLIT | FR)
        monkeypatch.setenv('STIN_STRIP_CFG', '21')
        assert torch.equal(wide[1:M + 1, 4:Nc + 4], SF.gemm_nt(A, Wf, b, precision=prec | SF.GEMM_W_PRESPLIT | FR))
        wide[1:M + 1, 4:Nc + 4] = 3.0
        assert float((wide - 3.0).abs().max()) == 0.0


@pytest.mark.parametrize('M,Nc,K', [(18063, 256, 512), (18063, 1024, 256), (5000, 320, 128), (777, 128, 260), (65, 132, 128),
                                    (31, 256, 128), (4097, 1280, 128), (60211, 128, 256), (1, 128, 128)])
def test_gemm_tn_ws_kernel_equals_four_wave_kernel(M, Nc, K, monkeypatch):
    """The producer / consumer TN kernel (csrc/stin_wgrad.hip: 8 waves, fixed roles, double-buffered LDS) against the
    four-wave kernel it replaces for 128 x 128-tiled bf16x3 products: same chunking, k order and MFMA order -> the summed
    slabs are bit-identical, with and without the bias column / row weights, ragged row counts and partial edge tiles."""
    g = torch.Generator().manual_seed(M + Nc + K)
    G = torch.randn(M + 2, Nc + 4, generator=g).to(DEV)[1:M + 1, :Nc]              # strided views: ld != width
    X = torch.randn(M, K + 8, generator=g).to(DEV)[:, 4:K + 4]
    w = torch.rand(M, 3, generator=g).to(DEV)[:, 1]
    want64 = torch.cat([G.double().t() @ X.double(), (G.double() * w.double()[:, None]).sum(0)[:, None]], 1)
    res = {}
    for ws in ('0', '1'):
        monkeypatch.setenv('STIN_TN_WS', ws)
        res[ws] = [SF.gemm_tn(G, X, ones_column=True, row_weight=w, precision=SF.GEMM_BF16X3),
                   SF.gemm_tn(G, X, ones_column=True, precision=SF.GEMM_BF16X3),
                   SF.gemm_tn(G, X, precision=SF.GEMM_BF16X3)]
        assert torch.equal(res[ws][0], SF.gemm_tn(G, X, ones_column=True, row_weight=w, precision=SF.GEMM_BF16X3))   # deterministic
    for a, b in zip(res['0'], res['1']):
        assert torch.equal(a, b)
    assert float((res['1'][0].double() - want64).abs().max()) <= 3e-5 * float(want64.abs().max()) + 1e-6


@pytest.mark.parametrize('N,Cin,Cout,shortcut,trans_inv', [(18063, 256, 256, False, False), (5000, 128, 256, True, False),
                                                          (3001, 10, 64, True, True), (2000, 64, 128, True, False),
                                                          (257, 512, 512, False, False), (40, 128, 128, False, True)])
def test_edgeconv_wgrad_equals_two_tn_gemms_plus_unpack(N, Cin, Cout, shortcut, trans_inv):
    """stin_edgeconv_wgrad (both transposed products in one grid where both are 128 x 128-tiled + ONE finalize launch that
    sums the slabs and writes the reference-layout gradients) against what it replaces: stin_gemm_tn_f32 twice +
    stin_edgeconv_unpack_grads_f32 - bit-identical, also where only one / none of the products takes the new kernel."""
    lib = _lib_load()
    g = torch.Generator().manual_seed(N + Cin + Cout)
    H = 2 * Cout
    Cp = (Cin + 3) // 4 * 4
    Yw = 2 * H + (Cout if shortcut else 0)
    dagg = torch.randn(N, Cout, generator=g).to(DEV)
    hE = torch.randn(N, H + 4, generator=g).to(DEV)
    hE[:, H] = (torch.rand(N, generator=g) < 0.8).float().to(DEV)
    dY = torch.randn(N, Yw, generator=g).to(DEV)
    x = torch.zeros(N, Cp, device=DEV)
    x[:, :Cin] = torch.randn(N, Cin, generator=g).to(DEV)
    ld1 = Cin if trans_inv else 2 * Cin
    # reference: the two stand-alone products + the unpack kernel
    dw2b = SF.gemm_tn(dagg, hE[:, :H], ones_column=True, row_weight=hE[:, H], precision=SF.GEMM_BF16X3)
    dwb = SF.gemm_tn(dY, x, ones_column=True, precision=SF.GEMM_BF16X3)
    want = [torch.empty(H, ld1, device=DEV), torch.empty(H, device=DEV), torch.empty(Cout, H, device=DEV), torch.empty(Cout, device=DEV),
            torch.empty(Cout, Cin, device=DEV) if shortcut else None, torch.empty(Cout, device=DEV) if shortcut else None]
    SF._call('stin_edgeconv_unpack_grads_f32', SF._ptr(dwb), SF._ptr(dw2b), Cin, Cp, H, Cout, int(shortcut), int(trans_inv),
             SF._ptr(want[0]), SF._ptr(want[1]), SF._ptr(want[4]), SF._ptr(want[5]), SF._ptr(want[2]), SF._ptr(want[3]), SF._stream(x))
    got = [torch.full_like(t, 7.0) if t is not None else None for t in want]
    ws_bytes = lib.stin_edgeconv_wgrad_workspace_bytes(N, Cp, H, Cout, int(shortcut))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    SF._call('stin_edgeconv_wgrad', 0, SF._ptr(dagg), Cout, SF._ptr(hE), hE.stride(0), SF._ptr(dY), Yw, SF._ptr(x), Cp, N, Cin, Cp, H,
             Cout, int(shortcut), int(trans_inv), SF.GEMM_BF16X3, SF._ptr(got[0]), SF._ptr(got[1]), SF._ptr(got[2]), SF._ptr(got[3]),
             SF._ptr(got[4]), SF._ptr(got[5]), SF._ptr(ws), ws_bytes, SF._stream(x))
    for i, (a, b) in enumerate(zip(got, want)):
        if b is not None:
            assert torch.equal(a, b), i
    # and against fp64
    ref = dagg.double().t() @ hE[:, :H].double()
    assert float((got[2].double() - ref).abs().max()) <= 3e-5 * float(ref.abs().max()) + 1e-6


def _panel_tiles(M, Nc, K, forced, cu=256):
    """Host mirror of panel_tiles() in csrc/stin_gemm.hip: 32-row tiles per block of the balanced column-panel NT kernel (0 = not used)."""
    if forced == 0 or Nc % 128 or K % 64 or M <= 0:
        return 0
    P, rg = Nc // 128, (M + 31) // 32
    for rounds in range(1, (64 if forced == 1 else 1) + 1):
        slots = cu * rounds // P
        if slots < 1:
            continue
        mts = (rg + slots - 1) // slots
        if mts <= 9:
            return (2 if forced == 1 else 0) if mts < 2 else mts
    return 0


@pytest.mark.parametrize('panel', ['0', '1'])
@pytest.mark.parametrize('M,Nc,K', [(18063, 256, 512), (60211, 128, 256), (130, 256, 512), (64, 128, 256), (4097, 256, 1024),
                                    (129, 128, 256), (190, 128, 512)])
def test_gemm_nt_with_fused_column_statistics(M, Nc, K, panel, monkeypatch):
    """stin_gemm_nt_colstats_f32: GEMM2 of a block plus the first stage of the instance-norm statistics of its output in one
    launch.  The output equals the plain call bit for bit; the per-group sums are the fp64 column sums of the stored fp32
    values (exact up to fp64 rounding); mean / rstd equal the two-kernel colreduce route to fp32 rounding.  Both kernels that
    carry the epilogue: the all-columns kernel (STIN_NT_PANEL=0: one group per 64 rows) and the balanced column-panel kernel
    (STIN_NT_PANEL=1: two groups per row block, MT0 and MT1 32-row tiles)."""
    from surface_texture_inpainting_net_amd.plan import NormGroups
    monkeypatch.setenv('STIN_NT_PANEL', panel)
    g = torch.Generator().manual_seed(M + Nc)
    A = torch.rand(M, K + 4, generator=g).to(DEV)
    W = (torch.randn(Nc, K, generator=g) * 0.1).to(DEV)
    b = torch.randn(Nc, generator=g).to(DEV)
    prec = SF.GEMM_F16X3 | SF.GEMM_W_PRESPLIT | 0x400
    Wf = SF.split_weights(W, SF.GEMM_F16X3 | 0x400)
    plain = SF.gemm_nt(A[:, :K], Wf, b, row_mask=A[:, K], precision=prec)
    got = SF.gemm_nt_colstats(A[:, :K], Wf, b, A[:, K], prec, Nc)
    assert got is not None
    C, partial = got
    assert torch.equal(C, plain)
    # no write past the [groups, 2, Nc] buffer (a 128-row block whose second 64-row group lies past M must not store one)
    guard = torch.full((partial.numel() + 4096,), -7.0, dtype=torch.float64, device=DEV)
    SF._call('stin_gemm_nt_colstats_f32', SF._ptr(A), A.stride(0), SF._ptr(Wf), K, SF._ptr(b), SF._ptr(A[:, K]), A.stride(0), None, 0,
             M, Nc, K, SF._ptr(C), Nc, prec, SF._ptr(guard), partial.numel() * 8, SF._stream(A))
    assert torch.equal(guard[:partial.numel()].view_as(partial), partial) and float((guard[partial.numel():] + 7.0).abs().max()) == 0.0
    mts = _panel_tiles(M, Nc, K, int(panel))
    if mts == 0:
        bounds = [(i * 64, min(M, i * 64 + 64)) for i in range((M + 63) // 64)]
    else:                                                   # row block rb: rows [rb BM, rb BM + 32 MT0) and [.., (rb + 1) BM)
        bm, mt0 = 32 * mts, 32 * ((mts + 1) // 2)
        bounds = []
        for rb in range((M + bm - 1) // bm):
            bounds += [(min(M, rb * bm), min(M, rb * bm + mt0)), (min(M, rb * bm + mt0), min(M, rb * bm + bm))]
    assert partial.shape == (len(bounds), 2, Nc)
    Cd = C.double()
    zero = torch.zeros(Nc, dtype=torch.float64, device=DEV)
    want1 = torch.stack([Cd[a:e].sum(0) if e > a else zero for a, e in bounds])
    want2 = torch.stack([(Cd[a:e] ** 2).sum(0) if e > a else zero for a, e in bounds])
    rows = max(e - a for a, e in bounds)
    assert float((partial[:, 0] - want1).abs().max()) <= 1e-11 * rows * float(C.abs().max())
    assert float((partial[:, 1] - want2).abs().max()) <= 1e-11 * rows * float(C.abs().max()) ** 2
    ng = NormGroups(M, torch.device(DEV))
    mean, rstd = SF.moments_final(partial, ng.inv_cnt)
    mean2, rstd2 = SF.instance_stats(C, ng)
    assert float((mean - mean2).abs().max()) <= 1e-6 * float(mean2.abs().max()) + 1e-9
    assert float((rstd / rstd2 - 1).abs().max()) <= 1e-6
    # shapes without an all-columns kernel: no fused form
    assert SF.gemm_nt_colstats(A[:, :K], SF.split_weights(W, SF.GEMM_F16X3), b, A[:, K], SF.GEMM_F16X3 | SF.GEMM_W_PRESPLIT, Nc) is None


@pytest.mark.parametrize('M,Nc,K,res', [(18063, 256, 1024, True), (18063, 256, 1024, False), (4097, 256, 512, True), (130, 128, 256, True),
                                        (8190, 512, 256, False), (33, 256, 512, True)])
def test_gemm_nt_with_fused_norm_backward_statistics(M, Nc, K, res, monkeypatch):
    """stin_gemm_nt_dotelu_f32: the input-gradient product of a block plus the first stage of the NEXT block-in-backward-order's
    instance-norm + ELU backward statistics in one launch (round 4 hand-off, stin_block.hip BwdLink).  The product equals the plain
    call bit for bit; the partial sums are the fp64 column sums of dy xc / dy per row group of the panel kernel; the folded
    coefficients k, m equal the separate stin_colreduce_f32(DOT_ELU, NORM_COEF) route to fp32 rounding."""
    from surface_texture_inpainting_net_amd.plan import NormGroups
    monkeypatch.setenv('STIN_NT_PANEL', '1')
    g = torch.Generator().manual_seed(M + Nc + K)
    A = (torch.randn(M, K, generator=g) * 0.5).to(DEV)
    W = (torch.randn(Nc, K, generator=g) * 0.05).to(DEV)
    R = torch.randn(M, Nc, generator=g).to(DEV) if res else None
    nx = (torch.randn(M, Nc, generator=g) * 1.5 + 0.3).to(DEV)
    prec = SF.GEMM_BF16X3 | SF.GEMM_W_PRESPLIT | 0x400
    Wf = SF.split_weights(W, SF.GEMM_BF16X3 | 0x400)
    ng = NormGroups(M, torch.device(DEV))
    mean, rstd = SF.instance_stats(nx, ng)
    plain = SF.gemm_nt(A, Wf, None, precision=prec, residual=R)
    got = SF.gemm_nt_dotelu(A, Wf, R, prec, Nc, nx, mean, rstd)
    assert got is not None
    C, partial = got
    assert torch.equal(C, plain)
    mts = _panel_tiles(M, Nc, K, 1)
    bm, mt0 = 32 * mts, 32 * ((mts + 1) // 2)
    bounds = []
    for rb in range((M + bm - 1) // bm):
        bounds += [(min(M, rb * bm), min(M, rb * bm + mt0)), (min(M, rb * bm + mt0), min(M, rb * bm + bm))]
    assert partial.shape == (len(bounds), 2, Nc)
    xc = nx - mean
    pre = xc * rstd
    dy = C * torch.where(pre > 0, torch.ones_like(pre), torch.exp(pre))
    t0, t1 = (dy * xc).double(), dy.double()
    zero = torch.zeros(Nc, dtype=torch.float64, device=DEV)
    want0 = torch.stack([t0[a:e].sum(0) if e > a else zero for a, e in bounds])
    want1 = torch.stack([t1[a:e].sum(0) if e > a else zero for a, e in bounds])
    rows = max(e - a for a, e in bounds)
    scale = float(t0.abs().max()) + float(t1.abs().max()) + 1e-30
    assert float((partial[:, 0] - want0).abs().max()) <= 1e-5 * rows * scale          # (__expf vs torch.exp in fp32: ~1e-6 relative)
    assert float((partial[:, 1] - want1).abs().max()) <= 1e-5 * rows * scale
    k, m = SF.norm_coef_from_partials(partial, rstd, ng.inv_cnt)
    k2, m2 = SF.colreduce(SF.RED_DOT_ELU, nx, ng, ng.ptr_true, gout=C, mean=mean, rstd=rstd, post=SF.POST_NORM_COEF)
    assert float((k - k2).abs().max()) <= 2e-6 * float(k2.abs().max()) + 1e-12
    assert float((m - m2).abs().max()) <= 2e-6 * float(m2.abs().max()) + 1e-12


@pytest.mark.parametrize('M,Nc,K', [(1, 256, 128), (8187, 256, 512), (12283, 256, 1024), (16379, 256, 512), (18063, 256, 1024), (24571, 256, 512),
                                    (28667, 256, 128), (32763, 256, 512), (36859, 256, 1024), (18063, 512, 256), (18063, 1024, 256),
                                    (18063, 1280, 128), (18063, 128, 1280), (5000, 1024, 256), (60211, 640, 256), (777, 384, 192), (300, 128, 256)])
def test_gemm_nt_panel_kernel_equals_tiled_kernel(M, Nc, K, monkeypatch):
    """Round 4: the balanced column-panel NT kernel (k_gemm_nt_panel: 128-column panels x row blocks of 32 (MT0 + MT1) rows sized
    so that the grid fills the chip in whole rounds, 8 waves = 2 row groups x 4 column tiles, A streamed once through LDS, W
    fragments from L2) against the tiled kernel on the k-group layout AND against the strip / all-columns kernels it replaces:
    same k order, MFMA order and epilogue expression -> bit-identical.  Every (MT0, MT1) instantiation (2 .. 9 tiles per block),
    ragged last row blocks, single-row input, one and several rounds of blocks, strided A / residual / output views."""
    FR = 0x400
    frag = bool(_lib_load().stin_gemm_w_is_frag(Nc, K))
    g = torch.Generator().manual_seed(M + Nc + K)
    A = torch.randn(M + 3, K + 4, generator=g).to(DEV)[1:M + 1, :K]
    W = (torch.randn(Nc, K, generator=g) * 0.1).to(DEV)
    b = torch.randn(Nc, generator=g).to(DEV)
    mask = (torch.rand(M, 3, generator=g) < 0.7).float().to(DEV)[:, 1]
    res = torch.randn(M, Nc + 4, generator=g).to(DEV)[:, :Nc]
    want = A.double() @ W.double().t()
    scale = float(want.abs().max()) + 1.0
    for prec in (SF.GEMM_F16X3, SF.GEMM_BF16X3):
        Wk, Wf = SF.split_weights(W, prec), SF.split_weights(W, prec | FR)
        for kw in (dict(), dict(bias=b, row_mask=mask), dict(bias=b, residual=res)):
            tiled = SF.gemm_nt(A, Wk, precision=prec | SF.GEMM_W_PRESPLIT, **kw)
            monkeypatch.setenv('STIN_NT_PANEL', '0')
            old = SF.gemm_nt(A, Wf, precision=prec | SF.GEMM_W_PRESPLIT | FR, **kw)
            monkeypatch.setenv('STIN_NT_PANEL', '1')
            for _ in range(2):
                new = SF.gemm_nt(A, Wf, precision=prec | SF.GEMM_W_PRESPLIT | FR, **kw)
                assert torch.equal(new, tiled) and torch.equal(old, tiled), (prec, sorted(kw))
        if frag and Nc % 128 == 0:
            assert _panel_tiles(M, Nc, K, 1) >= 2
            wide = torch.full((M + 2, Nc + 8), 3.0, device=DEV)              # output as a view into a wider matrix
            SF.gemm_nt(A, Wf, b, residual=res, out=wide[1:M + 1, 4:Nc + 4], precision=prec | SF.GEMM_W_PRESPLIT | FR)
            assert torch.equal(wide[1:M + 1, 4:Nc + 4], SF.gemm_nt(A, Wk, b, residual=res, precision=prec | SF.GEMM_W_PRESPLIT))
            wide[1:M + 1, 4:Nc + 4] = 3.0
            assert float((wide - 3.0).abs().max()) == 0.0
        ref = want + b.double() * mask.double()[:, None]
        tol = 3e-6 if prec == SF.GEMM_F16X3 else 2e-5
        assert float((SF.gemm_nt(A, Wf, b, row_mask=mask, precision=prec | SF.GEMM_W_PRESPLIT | FR).double() - ref).abs().max()) <= tol * scale


@pytest.mark.parametrize('M,Nc,K', [(40_001, 64, 128), (40_001, 128, 64), (9000, 64, 256), (20_000, 256, 64), (5003, 64, 320), (3000, 96, 192)])
def test_gemm_nt_streaming_rows_kernel_with_presplit_weights_equals_tiled_kernel(M, Nc, K, monkeypatch):
    """Round 5: the streaming-rows NT kernel (k_gemm_nt_stream: persistent blocks, wave-owned 32-row tiles through wave-private
    LDS, the pre-split weight slice copied once per block) on the k-group layout of the block launches' tall products
    (agg = hE W2^T, dhE = g W2: >= 500 000 rows by default, STIN_NT_STREAM_PRE_ROWS) against the tiled kernel: same k order,
    MFMA order and epilogue expression (bias [* row mask], residual) -> bit-identical; ragged last tile, strided views,
    two and four column tiles per block."""
    g = torch.Generator().manual_seed(M + Nc + K)
    A = torch.randn(M + 3, K + 4, generator=g).to(DEV)[1:M + 1, :K]
    W = (torch.randn(Nc, K, generator=g) * 0.1).to(DEV)
    b = torch.randn(Nc, generator=g).to(DEV)
    mask = (torch.rand(M, 3, generator=g) < 0.7).float().to(DEV)[:, 1]
    res = torch.randn(M, Nc + 4, generator=g).to(DEV)[:, :Nc]
    for prec in (SF.GEMM_F16X3, SF.GEMM_BF16X3):
        Wk = SF.split_weights(W, prec)
        for kw in (dict(), dict(bias=b, row_mask=mask), dict(bias=b, residual=res), dict(residual=res)):
            monkeypatch.setenv('STIN_NT_STREAM_PRE_ROWS', '999999999')
            tiled = SF.gemm_nt(A, Wk, precision=prec | SF.GEMM_W_PRESPLIT, **kw)
            monkeypatch.setenv('STIN_NT_STREAM_PRE_ROWS', '1')
            for nt in ('2', '4'):
                monkeypatch.setenv('STIN_NT_STREAM_NT', nt)
                new = SF.gemm_nt(A, Wk, precision=prec | SF.GEMM_W_PRESPLIT, **kw)
                assert torch.equal(new, tiled), (prec, sorted(kw), nt)
            monkeypatch.delenv('STIN_NT_STREAM_NT')


def _lib_load():
    from surface_texture_inpainting_net_amd import _lib
    return _lib.load()


def test_gemm_nt_on_strided_views_and_linear_autograd():
    g = torch.Generator().manual_seed(3)
    big = torch.randn(500, 200, generator=g).to(DEV)
    A = big[:, 40:104]                                  # ld 200, 16-byte aligned column slice
    W = torch.randn(96, 64, generator=g).to(DEV)
    out = torch.zeros(500, 160, device=DEV)
    SF.gemm_nt(A, W, None, out=out[:, 32:128])
    want = A.double() @ W.double().t()
    assert float((out[:, 32:128].double() - want).abs().max()) <= 1e-4
    assert float(out[:, :32].abs().max()) == 0.0 and float(out[:, 128:].abs().max()) == 0.0
    x = torch.randn(300, 20, generator=g).to(DEV).requires_grad_(True)
    lin = torch.nn.Linear(20, 7).to(DEV)
    y = SF.linear(x, lin.weight, lin.bias)
    y2 = torch.nn.functional.linear(x.detach().requires_grad_(True), lin.weight, lin.bias)
    assert float((y - y2).abs().max()) <= 1e-5
    w = torch.randn(300, 7, device=DEV)
    gx, gw, gb = torch.autograd.grad((y * w).sum(), [x, lin.weight, lin.bias])
    # backward GEMMs run on the 2-piece bf16 split (~4e-6 relative, fp32 accumulate)
    ref_gx, ref_gw = w @ lin.weight, w.t() @ x.detach()
    assert float((gx - ref_gx).abs().max()) <= 3e-5 * float(ref_gx.abs().max())
    assert float((gw - ref_gw).abs().max()) <= 3e-5 * float(ref_gw.abs().max())
    assert float((gb - w.sum(0)).abs().max()) <= 1e-4


@pytest.mark.parametrize('N,K,Nc', [(1, 64, 3), (37, 64, 3), (5000, 64, 3), (200_704, 64, 3), (3000, 128, 4), (777, 256, 1), (1234, 20, 2),
                                    (3025, 4, 3), (40_000, 8, 3), (100_000, 16, 1)])     # (round 5) K <= 16 on grids of > 16 blocks: the fold's LDS parts
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_last_layer_linear_tanh_one_launch_kernels(N, K, Nc, dtype):
    """tanh(x W^T + b) and its backward (stin_linear_tanh_*, csrc/stin_tail.hip) against fp64 on the same inputs:
    forward <= 2e-6, dx / dW / db within summation-order noise of the fp64 result; strided input rows; deterministic."""
    g = torch.Generator().manual_seed(N + K + Nc)
    big = (torch.randn(N, K + 8, generator=g) * 0.7).to(DEV).to(dtype)
    x = big[:, 4:4 + K]                                  # strided rows (ld K + 8), 16-byte (fp32) / 8-byte (bf16) aligned
    x = x.detach().requires_grad_(True)
    lin = torch.nn.Linear(K, Nc).to(DEV)
    assert SF.linear_tanh_eligible(x, lin.weight, lin.bias)
    y = SF.linear_tanh(x, lin.weight, lin.bias)
    assert y.dtype == torch.float32 and y.shape == (N, Nc)
    xd = x.detach().double().requires_grad_(True)
    Wd, bd = lin.weight.detach().double().requires_grad_(True), lin.bias.detach().double().requires_grad_(True)
    yd = torch.tanh(xd @ Wd.t() + bd)
    assert float((y.double() - yd).abs().max()) <= 2e-6
    w = torch.randn(N, Nc, generator=g).to(DEV)
    gx, gw, gb = torch.autograd.grad((y * w).sum(), [x, lin.weight, lin.bias])
    rx, rw, rb = torch.autograd.grad((yd * w.double()).sum(), [xd, Wd, bd])
    tol_x = 1e-5 if dtype == torch.float32 else 1e-2
    assert float((gx.double() - rx).abs().max()) <= tol_x * max(1.0, float(rx.abs().max()))
    scale_w = float((w.double().abs().t() @ x.detach().double().abs()).max()) + 1e-30        # sum of magnitudes: fp32 noise floor
    assert float((gw.double() - rw).abs().max()) <= 2e-6 * scale_w
    assert float((gb.double() - rb).abs().max()) <= 2e-6 * (float(w.abs().sum(0).max()) + 1e-30)
    y2 = SF.linear_tanh(x, lin.weight, lin.bias)
    gx2, gw2, gb2 = torch.autograd.grad((y2 * w).sum(), [x, lin.weight, lin.bias])
    assert torch.equal(y, y2) and torch.equal(gx, gx2) and torch.equal(gw, gw2) and torch.equal(gb, gb2)


def test_linear_backward_separate_weight_and_bias_gradients_and_pretransposed_weight():
    """LinearFn's backward: dW / db through stin_gemm_tn_wb_* (no [Nc, K + 1] intermediate) equal the sliced stin_gemm_tn_f32
    result bit for bit, and a pre-transposed weight operand (PackSet's riding transpose) gives the same input gradient."""
    g = torch.Generator().manual_seed(11)
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.randn(4000, 64, generator=g).to(DEV).to(dtype)
        G = torch.randn(4000, 64, generator=g).to(DEV).to(dtype)
        dwb = SF.gemm_tn(G, x, ones_column=True, precision=SF.PREC_BWD)
        dW, db = torch.empty(64, 64, device=DEV), torch.empty(64, device=DEV)
        SF.gemm_tn_wb(G, x, dW, db, precision=SF.PREC_BWD)
        assert torch.equal(dW, dwb[:, :-1]) and torch.equal(db, dwb[:, -1])
    lin = torch.nn.Linear(64, 64).to(DEV)
    xs = torch.randn(3000, 64, generator=g).to(DEV).requires_grad_(True)
    w = torch.randn(3000, 64, generator=g).to(DEV)
    ps = SF.PackSet([], torch.device(DEV), False, (lin.weight,))
    ps.run()
    assert torch.equal(ps.transposed[0], lin.weight.detach().t().contiguous())
    a = torch.autograd.grad((SF.linear(xs, lin.weight, lin.bias) * w).sum(), [xs, lin.weight, lin.bias])
    b = torch.autograd.grad((SF.linear(xs, lin.weight, lin.bias, wT=ps.transposed[0]) * w).sum(), [xs, lin.weight, lin.bias])
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    # the pack buffer is rewritten by every run and is not version-tracked: a backward that follows a LATER run must not read
    # it (wT_guard) - poisoned here to make a stale read visible
    y = SF.linear(xs, lin.weight, lin.bias, wT=ps.transposed[0], wT_guard=(ps, ps.runs))
    ps.run()
    ps.transposed[0].fill_(float('nan'))
    c = torch.autograd.grad((y * w).sum(), [xs, lin.weight, lin.bias])
    assert all(torch.equal(p, q) for p, q in zip(a, c))


@pytest.mark.parametrize('N,C,groups', [(18_063, 256, 226), (18_063, 512, 126), (1806, 256, 58), (5000, 64, 40), (60_211, 128, 942),
                                        (333, 32, 3), (70, 96, 1)])
def test_norm_with_the_statistics_fold_inside_the_launch_is_bit_identical(N, C, groups):
    """k_norm_fold (round 5): the per-row-group statistics partials of a GEMM epilogue folded by every workgroup of the
    normalisation launch for its own columns - mean / rstd and the outputs must equal the two-launch route (fold launch +
    elementwise launch) BIT FOR BIT, forward and backward; shapes the host rule declines return STIN_E_UNSUPPORTED untouched."""
    from surface_texture_inpainting_net_amd.plan import _ptr, _stream
    lib = _lib_load()
    g = torch.Generator(device=DEV).manual_seed(N + C)
    x = torch.randn(N, C, generator=g, device=DEV) * 1.7 + 0.3
    res = torch.randn(N, C, generator=g, device=DEV)
    go = torch.randn(N, C, generator=g, device=DEV)
    inv = torch.full((1,), 1.0 / N, device=DEV)
    # moment partials of x itself, row groups of unequal size
    cuts = torch.linspace(0, N, groups + 1).long().tolist()
    pm = torch.stack([torch.stack([x[a:b].double().sum(0), x[a:b].double().pow(2).sum(0)]) for a, b in zip(cuts, cuts[1:])]).contiguous()
    mean, rstd = SF.moments_final(pm, inv)
    y_ref = torch.empty(N, C, device=DEV)
    SF._call('stin_norm_act_res_fwd_f32', _ptr(x), C, _ptr(mean), _ptr(rstd), None, _ptr(res), C, N, C, 1, _ptr(y_ref), C, _stream(x))
    rows = lib.stin_norm_fold_rows(N, C, groups)
    m2, r2, y = torch.zeros(1, C, device=DEV), torch.zeros(1, C, device=DEV), torch.zeros(N, C, device=DEV)
    rc = lib.stin_norm_act_res_fwd_fold_f32(_ptr(pm), groups, _ptr(x), C, _ptr(res), C, _ptr(inv), float(SF.EPS), N, C, _ptr(m2), _ptr(r2),
                                            _ptr(y), C, _stream(x))
    if rows <= 0:
        assert rc != 0 and float(y.abs().max()) == 0.0       # declined before anything was enqueued
        return
    assert rc == 0
    assert torch.equal(m2, mean) and torch.equal(r2, rstd) and torch.equal(y, y_ref)
    # backward: partials of (dy xc, dy) as the dx product's epilogue writes them
    xc = (x - mean).double()
    n = (x - mean) * rstd
    dy = (go * torch.where(n > 0, torch.ones_like(n), torch.exp(n))).double()
    pb = torch.stack([torch.stack([(dy[a:b] * xc[a:b]).sum(0), dy[a:b].sum(0)]) for a, b in zip(cuts, cuts[1:])]).contiguous()
    kk, mm = SF.norm_coef_from_partials(pb, rstd, inv)
    d_ref = torch.empty(N, C, device=DEV)
    SF._call('stin_norm_act_bwd_f32', _ptr(x), C, _ptr(go), C, _ptr(mean), _ptr(rstd), _ptr(rstd), _ptr(kk), _ptr(mm), None, None, N, C, 1,
             _ptr(d_ref), C, _stream(x))
    d = torch.zeros(N, C, device=DEV)
    assert lib.stin_norm_act_bwd_fold_f32(_ptr(pb), groups, _ptr(x), C, _ptr(go), C, _ptr(mean), _ptr(rstd), _ptr(inv), N, C, _ptr(d), C,
                                          _stream(x)) == 0
    assert torch.equal(d, d_ref)
    # strided rows (the residual is a column slice of Y in shortcut blocks)
    big = torch.randn(N, C + 8, generator=g, device=DEV)
    y3 = torch.zeros(N, C, device=DEV)
    assert lib.stin_norm_act_res_fwd_fold_f32(_ptr(pm), groups, _ptr(x), C, _ptr(big[:, 4:]), C + 8, _ptr(inv), float(SF.EPS), N, C, _ptr(m2),
                                              _ptr(r2), _ptr(y3), C, _stream(x)) == 0
    SF._call('stin_norm_act_res_fwd_f32', _ptr(x), C, _ptr(mean), _ptr(rstd), None, _ptr(big[:, 4:]), C + 8, N, C, 1, _ptr(y_ref), C, _stream(x))
    assert torch.equal(y3, y_ref)


# --------------------------------------------------------- segment sum / pool / unpool
@pytest.mark.parametrize('C', [1, 3, 8, 64, 100, 256])
def test_segment_sum_matches_scatter_and_is_linear(C):
    n, e = 400, 3000
    g = torch.Generator().manual_seed(C)
    idx = torch.randint(3, n, (e,), generator=g)
    src = torch.randn(e, C, generator=g)
    src2 = torch.randn(e, C, generator=g)
    csr = build_csr(idx.to(DEV), None, n, e, _bad(), want_perm=True)
    got = SF.ScatterAddFn.apply(src.to(DEV), csr)
    want = scatter_ops.scatter_sum(src, idx, dim=0, dim_size=n)
    assert torch.equal(got.cpu(), want), 'same summation order as the sequential CPU scatter_add -> bit-exact'
    got_mean = SF.segment_sum(src.to(DEV), csr.rowptr, csr.perm, n, mean=True)
    assert float((got_mean.cpu() - scatter_ops.scatter_mean(src, idx, dim=0, dim_size=n)).abs().max()) <= 1e-6
    both = SF.ScatterAddFn.apply((src + 2 * src2).to(DEV), csr)
    assert float((both - (got + 2 * SF.ScatterAddFn.apply(src2.to(DEV), csr))).abs().max()) <= 1e-4


def test_pool_unpool_against_reference_fixture():
    z = {k: torch.from_numpy(v) for k, v in load_npz('g4_per_op').items()}
    xv, trace = z['pool.x'], z['pool.trace']
    pm = PoolMap(trace.to(DEV), 7, 5, _bad())
    assert pm.children.col.cpu().tolist() == [0, 1, 2, 3, 6, 4, 5]
    for kind, fn in (('max', SF.PoolMaxFn), ('mean', SF.PoolMeanFn)):
        xx = xv.clone().to(DEV).requires_grad_(True)
        y = fn.apply(xx, pm)
        assert torch.equal(y.cpu(), z['pool.%s.y' % kind]), kind
        (y * torch.arange(1., 16.).view(5, 3).to(DEV)).sum().backward()
        if kind == 'max':
            assert torch.equal(xx.grad.cpu(), z['pool.%s.gx' % kind]), kind
        else:   # g * (1/count) vs the reference's g / count: 1 ulp
            assert float((xx.grad.cpu() - z['pool.%s.gx' % kind]).abs().max()) <= 1e-6
    xx = z['unpool.x'].clone().to(DEV).requires_grad_(True)
    y = SF.UnpoolFn.apply(xx, pm)
    assert torch.equal(y.cpu(), z['unpool.y'])
    (y * torch.arange(1., 22.).view(7, 3).to(DEV)).sum().backward()
    assert torch.equal(xx.grad.cpu(), z['unpool.gx'])
    bvec = torch.tensor([0, 0, 0, 1, 1, 1, 1], device=DEV)
    assert torch.equal(SF.batch_pool(bvec, pm).cpu(), z['batch.pooled'])


@pytest.mark.parametrize('C', [3, 8, 64, 128, 320])
def test_pool_max_random_with_ties_bit_exact(C):
    n_f, n_c = 5000, 1500
    g = torch.Generator().manual_seed(C)
    trace = torch.randint(0, n_c, (n_f,), generator=g)
    x = torch.randint(-3, 4, (n_f, C), generator=g).float()          # many exact ties
    x.requires_grad_(True)
    w = torch.randn(n_c, C, generator=g)
    want, arg = scatter_ops.scatter_max(x, trace, dim=0, dim_size=n_c)
    (want * w).sum().backward()
    pm = PoolMap(trace.to(DEV), n_f, n_c, _bad())
    xd = x.detach().to(DEV).requires_grad_(True)
    got = SF.PoolMaxFn.apply(xd, pm)
    (got * w.to(DEV)).sum().backward()
    assert torch.equal(got.cpu(), want.detach())
    assert torch.equal(xd.grad.cpu(), x.grad), 'gradient goes to the FIRST arg-max only'


def test_batch_vector_propagation_bit_exact():
    fx = ModelFixture('g3_batch2_unequal')
    s = fx.sample(DEV)
    plan = GraphPlan(s)
    b = fx.sample().batch
    for lvl in (1, 2):
        tr = fx.sample()['hierarchy_trace_index_%d' % lvl]
        n = int(fx.sample().num_vertices.sum(0)[lvl])
        b = stin_oracle.pool_batch(b, tr, n)
        got = plan.batch_vector(lvl)
        assert got.dtype == torch.int64 and torch.equal(got.cpu(), b)
    up = SF.batch_unpool(plan.batch_vector(2), plan.pool(2))
    assert torch.equal(up.cpu(), b.index_select(0, fx.sample()['hierarchy_trace_index_2']))


# ------------------------------------------------------------------------------- norms
def test_instance_norm_variants_against_reference_fixture():
    z = {k: torch.from_numpy(v) for k, v in load_npz('g4_per_op').items()}
    x = z['norm.x']
    fi = M.FastInstanceNorm(5)
    for tag, b in (('none', None), ('zeros', torch.zeros(60, dtype=torch.long)), ('eq', z['norm.b_eq']),
                   ('un', z['norm.b_un'])):
        xx = x.clone().to(DEV).requires_grad_(True)
        y = fi(xx, None if b is None else b.to(DEV))
        assert float((y.cpu() - z['fin.%s.y' % tag]).abs().max()) <= 2e-6, tag
        (y * torch.linspace(-1, 1, y.numel()).view_as(y).to(DEV)).sum().backward()
        assert float((xx.grad.cpu() - z['fin.%s.gx' % tag]).abs().max()) <= 5e-5, tag
    gn = M.SingleBatchGraphNorm(5).to(DEV)
    with torch.no_grad():
        gn.weight.copy_(z['gn.weight'])
        gn.bias.copy_(z['gn.bias'])
        gn.mean_scale.copy_(z['gn.mean_scale'])
    for tag, b in (('none', None), ('un', z['norm.b_un'])):
        xx = x.clone().to(DEV).requires_grad_(True)
        y = gn(xx, None if b is None else b.to(DEV))
        assert float((y.cpu() - z['gn.%s.y' % tag]).abs().max()) <= 5e-6, tag
        (y * torch.linspace(-1, 1, y.numel()).view_as(y).to(DEV)).sum().backward()
        assert float((xx.grad.cpu() - z['gn.%s.gx' % tag]).abs().max()) <= 5e-5, tag
        assert float((gn.weight.grad.cpu() - z['gn.%s.gweight' % tag]).abs().max()) <= 5e-5
        assert float((gn.mean_scale.grad.cpu() - z['gn.%s.gmean_scale' % tag]).abs().max()) <= 5e-5
        gn.zero_grad()


@pytest.mark.parametrize('n,c', [(2, 4), (3, 7), (5000, 64), (100_000, 128), (777, 1024)])
def test_instance_norm_large_and_ragged(n, c):
    g = torch.Generator().manual_seed(n)
    x = (torch.randn(n, c, generator=g) * 3 + 50).requires_grad_(True)   # large mean: cancellation stress
    w = torch.randn(n, c, generator=g)
    want = torch.nn.functional.elu(stin_oracle.fast_instance_norm(x))
    (want * w).sum().backward()
    xd = x.detach().to(DEV).requires_grad_(True)
    got = SF.InstanceNormActResFn.apply(xd, None, NormGroups(n, torch.device(DEV)), True)
    (got * w.to(DEV)).sum().backward()
    assert float((got.detach().cpu() - want.detach()).abs().max()) <= 2e-4
    if n > 3:
        assert float((xd.grad.cpu() - x.grad).abs().max()) <= 1e-3 * float(x.grad.abs().max()) + 1e-6


def test_instance_norm_single_vertex_raises_like_reference():
    with pytest.raises(ValueError):
        M.FastInstanceNorm(4)(torch.randn(1, 4, device=DEV))


# ------------------------------------------------------------------------------ blocks
def test_graph_resnet_block_against_reference_fixture():
    z = {k: torch.from_numpy(v) for k, v in load_npz('g4_per_op').items()}
    ei = z['ei'].to(DEV)
    for tag, cin, cout, batch in (('neq', 6, 8, None), ('eq', 8, 8, None), ('eqb', 8, 8, z['batch_uneq'])):
        blk = S.GraphResnetBlock(cin, cout, M.get_gcn_filter, M.FastInstanceNorm, False, True)
        sd = {k[len('blk_%s.sd.' % tag):]: v for k, v in z.items() if k.startswith('blk_%s.sd.' % tag)}
        blk.load_state_dict(sd)
        blk = blk.to(DEV)
        x = z['blk_%s.x' % tag].to(DEV).requires_grad_(True)
        y = blk(x, ei, None if batch is None else batch.to(DEV))          # raw tensors, like an external caller
        assert float((y.cpu() - z['blk_%s.y' % tag]).abs().max()) <= 1e-5, tag
        (y * z['blk_%s.w' % tag].to(DEV)).sum().backward()
        scale = max(float(v.abs().max()) for k, v in z.items() if k.startswith('blk_%s.g.' % tag))
        assert float((x.grad.cpu() - z['blk_%s.gx' % tag]).abs().max()) <= 1e-3 * float(z['blk_%s.gx' % tag].abs().max())
        for k, p in blk.named_parameters():
            assert float((p.grad.cpu() - z['blk_%s.g.%s' % (tag, k)]).abs().max()) <= 1e-3 * scale, (tag, k)


# ------------------------------------------------------------------------------ models
@pytest.mark.parametrize('name', MODEL_FIXTURES)
def test_model_against_reference_fixture(name):
    fx = ModelFixture(name)
    net = S.define_G(**fx.cfg)
    net.load_state_dict(fx.state_dict)
    net = net.to(DEV)
    s = fx.sample(DEV)
    s.x = s.x.clone().requires_grad_(True)
    out = net(s)
    assert out.shape == fx.out.shape
    assert float((out.detach().cpu() - fx.out).abs().max()) <= FWD_TOL
    pred = torch.where((s.mask > 0).expand_as(s.color), out, s.color)
    loss = stin_oracle.compute_loss(pred, s.color, weights=s.mask)
    assert abs(float(loss.detach()) - float(fx.loss)) <= 1e-6
    loss.backward()
    scale = max(float(g.abs().max()) for g in fx.grads.values())
    worst = max(float((p.grad.cpu() - fx.grads[k]).abs().max()) for k, p in net.named_parameters()) / scale
    print('\n[fixture] %s: fwd %.2e, gx %.2e of scale, worst weight-gradient entry %.2e of scale'
          % (name, float((out.detach().cpu() - fx.out).abs().max()), float((s.x.grad.cpu() - fx.gx).abs().max()) / float(fx.gx.abs().max()), worst))
    assert float((s.x.grad.cpu() - fx.gx).abs().max()) <= 1e-3 * float(fx.gx.abs().max())
    for k, p in net.named_parameters():
        assert float((p.grad.cpu() - fx.grads[k]).abs().max()) <= 1e-3 * scale, k


@pytest.mark.parametrize('norm,filter_type', [('batch', 'edgeconv'), ('none', 'edgeconvtransinv'), ('instance', 'sageconvtransinv')])
def test_secondary_norms_and_filters_vs_oracle(norm, filter_type):
    """norm='batch' (PyG BatchNorm wrapper, train mode), no norm, and the SAGE trans-inv filter: not in the shipped
    configs and without a golden fixture, so checked against the oracle directly."""
    cfg = dict(input_nc=10, output_nc=3, ngf=8, filter_type=filter_type, norm=norm, n_blocks=2, n_levels=1,
               pooling_type='mean')
    torch.manual_seed(11)
    ref = stin_oracle.define_G(**cfg)
    for p in ref.parameters():
        if p.dim() == 1:
            torch.nn.init.normal_(p, 0, 0.2)
    net = S.define_G(**cfg)
    net.load_state_dict(ref.state_dict())
    net = net.to(DEV)
    s = make_synthetic_mesh(400, 2, seed=12, dilations=())
    want = ref(s)
    want.square().sum().backward()
    got = net(s.to(DEV))
    got.square().sum().backward()
    assert float((got.detach().cpu() - want.detach()).abs().max()) <= FWD_TOL
    rep = grad_flip_report(net.named_parameters(), ref.parameters(), 'secondary %s/%s' % (norm, filter_type))
    bar = SECONDARY_BARS[(norm, filter_type)]
    assert rep['max_rel'] <= bar[0] and rep['beyond'] <= bar[1] and rep['rel_l2'] <= bar[2], (rep, bar)


# (worst entry / gradient scale, entries beyond 1e-3 of scale, relative L2).  Measured on MI355X in round 5: worst 0.9-2.6e-6,
# none beyond, relative L2 1.8-7.0e-6 - fp32 rounding, no decision flipped on these meshes; the bars leave ~8x for a
# re-associated sum and are 60-100x below the 2e-3 of rounds 1-4
SECONDARY_BARS = {('batch', 'edgeconv'): (2e-5, 0, 3e-5), ('none', 'edgeconvtransinv'): (2e-5, 0, 3e-5),
                  ('instance', 'sageconvtransinv'): (2e-5, 0, 3e-5)}


def test_batch_of_unequal_crops_four_levels_vs_oracle():
    """Config-3 shape in miniature: a batch of 5 unequal crops, 4 graph levels, dilated bottleneck (offsets fixed
    per level), through collate -> HIP model, against the oracle on the same collated batch."""
    from surface_texture_inpainting_net_amd.data import collate
    graphs = [make_synthetic_mesh(n, 4, seed=40 + i, dilations=(2,)) for i, n in enumerate((300, 520, 410, 260, 640))]
    batch = collate(graphs)
    assert batch.num_vertices.shape == (5, 4)
    cfg = dict(input_nc=10, output_nc=3, ngf=8, filter_type='edgeconvtransinv', norm='instance', n_blocks=3, n_levels=3,
               pooling_type='max', dilations=[1, 2, 1])
    torch.manual_seed(21)
    ref = stin_oracle.define_G(**cfg)
    net = S.define_G(**cfg)
    net.load_state_dict(ref.state_dict())
    net = net.to(DEV)
    want = ref(batch)
    stin_oracle.compute_loss(stin_oracle.graph_forward(ref, batch), batch.color, batch.mask).backward()
    bd = batch.to(DEV)
    got = net(bd)
    stin_oracle.compute_loss(torch.where((bd.mask > 0).expand_as(bd.color), got, bd.color), bd.color, bd.mask).backward()
    assert float((got.detach().cpu() - want.detach()).abs().max()) <= FWD_TOL
    rep = grad_flip_report(net.named_parameters(), ref.parameters(), 'crops 4 levels')
    # measured on MI355X in round 5: worst entry 2.6e-5 of scale, none beyond 1e-3, relative L2 1.7e-5
    assert rep['max_rel'] <= CROPS4_BARS[0] and rep['beyond'] <= CROPS4_BARS[1] and rep['rel_l2'] <= CROPS4_BARS[2], rep


CROPS4_BARS = (1e-4, 0, 6e-5)


def test_train_step_against_reference_fixture():
    from surface_texture_inpainting_net_amd.train_step import TrainStep
    fx = ModelFixture('g7_train_step')
    net = S.define_G(**fx.cfg)
    net.load_state_dict(fx.state_dict)
    step = TrainStep(net.to(DEV), lr=7e-5, amsgrad=True)
    loss = step(fx.sample(DEV))
    assert abs(float(loss) - float(fx.loss)) <= 1e-6
    for k, v in step.model.state_dict().items():
        ok = fx.grads[k].abs() > 1e-5
        assert torch.allclose(v.cpu()[ok], fx.state_dict_after[k][ok], rtol=0, atol=5e-7), k


def test_batchnorm_training_step_against_reference_fixture():
    """norm='batch' on the library's BatchNorm kernels through the reference's real training step (fixture g12): output,
    loss, gradients, the Adam update and the running statistics - which the reference's checkpointed blocks update TWICE
    per step (their forward is recomputed inside backward; this build updates them a second time in backward instead)."""
    from surface_texture_inpainting_net_amd.train_step import TrainStep
    fx = ModelFixture('g12_batchnorm_step')
    net = S.define_G(**fx.cfg)
    net.load_state_dict(fx.state_dict)
    net = net.to(DEV).train()
    s = fx.sample(DEV)
    out = net(s)                                              # fixture protocol: one forward-only call in train mode first
    assert float((out.detach().cpu() - fx.out).abs().max()) <= FWD_TOL
    step = TrainStep(net, lr=7e-5, amsgrad=True)
    loss = step(s)
    assert abs(float(loss) - float(fx.loss)) <= 2e-6
    flat = step.bucket.flat.cpu()
    off = 0
    scale = max(float(g.abs().max()) for g in fx.grads.values())
    for k, p in net.named_parameters():
        g = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
        assert float((g - fx.grads[k]).abs().max()) <= 1e-3 * scale, k
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    for k, v in fx.state_dict_after.items():
        if k.endswith('num_batches_tracked'):
            assert int(sd[k]) == int(v), k
        elif 'running_' in k:
            assert float((sd[k] - v).abs().max()) <= 2e-6 + 1e-5 * float(v.abs().max()), k
        else:
            ok = fx.grads[k].abs() > 1e-5 if k in fx.grads else torch.ones_like(v, dtype=torch.bool)
            assert torch.allclose(sd[k][ok], v[ok], rtol=0, atol=5e-7), k
    net.eval()                                                # inference: running statistics through the same kernels
    with torch.no_grad():
        ev = net(s)
    ref = stin_oracle.define_G(**fx.cfg)
    ref.load_state_dict({k: v for k, v in sd.items()})
    ref.eval()
    with torch.no_grad():
        want = ref(fx.sample())
    assert float((ev.cpu() - want).abs().max()) <= FWD_TOL


def test_fused_masked_l1_loss_matches_trainer_formula():
    g = torch.Generator().manual_seed(5)
    n = 5000
    out = (torch.rand(n, 3, generator=g) * 2 - 1).requires_grad_(True)
    color = torch.rand(n, 3, generator=g) * 2 - 1
    mask = torch.where(torch.rand(n, 1, generator=g) < 0.3, torch.randint(1, 17, (n, 1), generator=g), torch.zeros(n, 1, dtype=torch.long))
    out.data[7] = color[7]                                         # exact zeros: |.|' = 0 there, as torch's abs backward
    for use_w in (True, False):
        out.grad = None
        want = stin_oracle.compute_loss(torch.where((mask > 0).expand_as(color), out, color), color, mask if use_w else None)
        want.backward()
        od = out.detach().to(DEV).requires_grad_(True)
        got = SF.masked_l1_loss(od, color.to(DEV), mask.to(DEV), use_w)
        (got * 3.0).backward()
        assert abs(float(got) - float(want)) <= 2e-7
        assert float((od.grad.cpu() / 3.0 - out.grad).abs().max()) <= 1e-9 + 1e-6 * float(out.grad.abs().max())


@pytest.mark.parametrize('amsgrad,wd', [(True, 0.0), (False, 0.0), (True, 0.01)])
def test_flat_adam_matches_torch_adam(amsgrad, wd):
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(10_001, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=7e-5, weight_decay=wd, amsgrad=amsgrad)
    p = p0.clone().to(DEV)
    m, v, vm = (torch.zeros_like(p) for _ in range(3))
    for step in range(1, 6):
        grad = torch.randn(10_001, generator=g) * (0.1 if step != 3 else 10.0)     # a spike: amsgrad's max matters
        ref.grad = grad.clone()
        opt.step()
        SF.adam_step(p, grad.to(DEV), m, v, vm, 7e-5, 0.9, 0.999, 1e-8, wd, step, amsgrad)
        assert float((p.cpu() - ref.detach()).abs().max()) <= 5e-7, step          # 1-2 ulp of |p| ~ 4


def test_model_batched_true_per_graph_mode_equals_separate_graphs():
    """compat_linspace_norm=False: correct segmented statistics; the encoder/bottleneck/decoder
    blocks then treat each graph of the batch independently (io blocks still share statistics, Q1)."""
    fx = ModelFixture('g3_batch2_unequal')
    net = S.define_G(**fx.cfg)
    net.load_state_dict(fx.state_dict)
    net = net.to(DEV)
    net.compat_linspace_norm = False
    out = net(fx.sample(DEV))
    assert torch.isfinite(out).all() and out.shape == fx.out.shape
    # differs from the quirk output (unequal graphs) but stays a valid tanh output
    assert float((out.cpu() - fx.out).abs().max()) > 1e-3 and float(out.abs().max()) < 1.0


def test_graph_metrics_against_reference_fixture():
    from surface_texture_inpainting_net_amd import metrics
    z = {k: torch.from_numpy(v) for k, v in load_npz('g8_metrics').items()}
    pred, ei = z['pred'].to(DEV), z['ei'].to(DEV)
    assert torch.allclose(metrics.graph_laplace_variance(pred, ei).cpu(), z['lap_var'], rtol=1e-5)
    assert torch.allclose(metrics.graph_total_variation(pred, ei).cpu(), z['tv'], rtol=1e-5)
    assert torch.allclose(metrics.psnr(pred, z['color'].to(DEV), data_range=2.0).cpu(), z['psnr'], rtol=1e-5)
    # round 4: the Laplace operator is one HIP pass (stin_graph_laplace_f32) - bit for bit the reference's composition (propagate [1 | x]
    # with aggr = 'add' in edge order, prop[:, 1:] - prop[:, 0:1] * x), which a tensor that needs a gradient still takes; also on a
    # 3-channel strided view and on a graph with empty rows
    g = torch.Generator().manual_seed(3)
    for x, e in ((metrics.grayscale(pred), ei), (torch.randn(5000, 7, generator=g).to(DEV)[:, 2:5], torch.randint(0, 5000, (2, 21000), generator=g).to(DEV))):
        fast = metrics.graph_laplace(x, e)
        slow = metrics.graph_laplace(x.clone().requires_grad_(True), e)
        assert slow.requires_grad and not fast.requires_grad and torch.equal(fast, slow.detach())


# ------------------------------------------- full-size (BASELINE.json) property checks
@pytest.fixture(scope='module')
def big():
    s = make_synthetic_mesh(200_000, 3, seed=0)
    return s, s.to(DEV)


def test_full_size_forward_backward_deterministic_and_bounded(big):
    s_cpu, s = big
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=9,
               n_levels=2, pooling_type='max', dilations=[1, 1, 1, 2, 4, 8, 16, 1, 1], checkpoint_bottleneck=True)
    torch.manual_seed(49)
    net = S.define_G(**cfg).to(DEV)
    outs, grads = [], []
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        out = net(s)
        loss = stin_oracle.compute_loss(torch.where((s.mask > 0).expand_as(s.color), out, s.color), s.color, s.mask)
        loss.backward()
        outs.append(out.detach().clone())
        grads.append(net.input_blocks[0].first_filter.nn[0].weight.grad.clone())
    assert outs[0].shape == (200_704, 3) and torch.isfinite(outs[0]).all() and float(outs[0].abs().max()) < 1.0
    assert torch.equal(outs[0], outs[1]), 'no float atomics: forward is bit-reproducible'
    assert torch.equal(grads[0], grads[1]) or float((grads[0] - grads[1]).abs().max()) <= 1e-7 * float(grads[0].abs().max())


def test_full_size_segment_sum_properties(big):
    _, s = big
    n = s.x.shape[0]
    plan = plan_for(s)
    e = plan.edges('edge_index', 0)
    ones = torch.ones(n, 64, device=DEV)
    deg = (e.by_dst.rowptr[1:] - e.by_dst.rowptr[:-1]).float()
    got = SF.segment_sum(ones, e.by_dst.rowptr, e.by_dst.col, n)
    assert torch.equal(got, deg[:, None].expand(-1, 64).contiguous())            # sum of ones = in-degree
    x = torch.randn(n, 64, device=DEV)
    tot = SF.segment_sum(x, e.by_src.rowptr, e.by_src.col, n).double().sum(0)
    want = (x.double() * deg[:, None].double()).sum(0)                           # symmetric graph: checksum of sums
    assert float((tot - want).abs().max()) <= 1e-6 * float(want.abs().max() + 1)
    pool = plan.pool(1)
    c = torch.randn(pool.n_coarse, 64, device=DEV)
    assert float(SF.PoolMeanFn.apply(SF.UnpoolFn.apply(c, pool), pool).sub(c).abs().max()) <= 1e-6
    assert torch.equal(SF.PoolMaxFn.apply(SF.UnpoolFn.apply(c, pool), pool), c)  # unpool -> max-pool round trip


def test_mid_size_full_width_model_vs_oracle():
    """ngf = 64 (the shipped 3-D config) at a size the CPU oracle finishes in seconds."""
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=9,
               n_levels=2, pooling_type='max', dilations=[1, 1, 1, 2, 4, 8, 16, 1, 1], checkpoint_bottleneck=True)
    torch.manual_seed(49)
    ref = stin_oracle.define_G(**cfg)
    net = S.define_G(**cfg)
    net.load_state_dict(ref.state_dict())
    net = net.to(DEV)
    s = make_synthetic_mesh(12_000, 3, seed=3)
    want = ref(s)
    stin_oracle.compute_loss(stin_oracle.graph_forward(ref, s), s.color, s.mask).backward()
    sd = s.to(DEV)
    got = net(sd)
    stin_oracle.compute_loss(torch.where((sd.mask > 0).expand_as(sd.color), got, sd.color), sd.color, sd.mask).backward()
    assert float((got.detach().cpu() - want.detach()).abs().max()) <= FWD_TOL
    # Gradients at full width: fp32 re-association flips a handful of near-tie arg-max (max pool) and ReLU decisions (out of
    # ~1e7), each of which re-routes one gradient entry - a DISCRETE change.  tests/test_full_size_parity.py measures it
    # against an fp64 run of the oracle on this very case: fp32 CPU oracle 2.4e-4 relative L2 / 8.4e-4 worst entry, this
    # path 6.9e-4 / 7.5e-4 with the shipped split-16-bit GEMMs, 8.1e-4 / 8.2e-4 with exact fp32 GEMMs, identical with
    # bf16x6 backward GEMMs - flip noise, not GEMM precision.  SURVEY 8d's bar (relative 1e-3 on weight gradients) is met
    # at the full 200k size (3.0e-4); at 12k vertices each flip weighs more, hence 1.5e-3 / 3e-3 here.
    rep = grad_flip_report(net.named_parameters(), ref.parameters(), 'mid size, full width')
    assert rep['max_rel'] <= MID_BARS[0] and rep['beyond'] <= MID_BARS[1] and rep['rel_l2'] <= MID_BARS[2], rep


# (worst entry / scale, entries beyond 1e-3 of scale of 4.2 M, relative L2) = 1.5 x the round-5 measurement on MI355X
# (8.2e-4, 0, 7.3e-4; the flip analysis is in the comment above)
MID_BARS = (1.3e-3, 10, 1.1e-3)



def test_plan_prefetch_on_side_streams_equals_lazy_build():
    """GraphPlan.prefetch builds the CSR pieces side by side on side streams; results and the model output must be
    identical to the lazy build on the compute stream, also when the sample's plan is rebuilt while earlier work is
    still queued (record_stream keeps freed plan memory from being reused too early)."""
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=3, n_levels=2,
               pooling_type='max', dilations=[1, 2, 4])
    torch.manual_seed(5)
    net = S.define_G(**cfg).to(DEV)
    s = make_synthetic_mesh(30_000, 3, seed=11, dilations=(2, 4)).to(DEV)
    with torch.no_grad():
        want = net(s)
        lazy = s._plan_cache
        for ready in (False, True, True, True):
            s._plan_cache = None
            plan = net.prefetch_plan(s, inputs_ready=ready)
            got = net(s)
            assert torch.equal(got, want)
        for key, es in plan._edges.items():
            ref = lazy._edges[key]
            assert torch.equal(es.by_dst.rowptr, ref.by_dst.rowptr) and torch.equal(es.by_dst.col, ref.by_dst.col)
            assert torch.equal(es.by_src.col, ref.by_src.col) and torch.equal(es.xslot, ref.xslot)
        for lvl, pm in plan._pools.items():
            assert torch.equal(pm.trace, lazy._pools[lvl].trace) and torch.equal(pm.children.col, lazy._pools[lvl].children.col)


@pytest.mark.parametrize('counts', [[3000, 1, 2500, 0, 1700], [5], [64] * 7, [1000, 1000, 1000, 1000], [7, 100000, 3]])
def test_norm_group_ids_kernel_equals_the_framework_route(counts):
    """plan.NormGroups of a batch: graph id and linspace-slice id per row from ONE launch (stin_norm_group_ids_i64) - the ids the
    framework route (cast, arange, cast, searchsorted(right=True), cast) gives, bit for bit; uneven graphs exercise the slice quirk
    (models/modules/fastinstancenorm.py:53-82), equal ones and B = 1 the no-quirk branch, empty graphs the repeated boundaries."""
    from surface_texture_inpainting_net_amd.plan import NormGroups
    dev = torch.device(DEV)
    cnt = torch.tensor(counts, dtype=torch.int64)
    n = int(cnt.sum())
    batch = torch.repeat_interleave(torch.arange(len(counts)), cnt).to(dev)
    ng = NormGroups(n, dev, batch, cnt, linspace_quirk=True)
    assert ng.gid.dtype == torch.int32 and torch.equal(ng.gid, batch.to(torch.int32))
    sum_ptr = torch.linspace(0, n, len(counts) + 1, dtype=torch.int).to(torch.int64)
    true_ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(cnt, 0)])
    assert ng.quirk == bool((sum_ptr != true_ptr).any())
    want = torch.searchsorted(sum_ptr[1:].to(dev), torch.arange(n, device=dev), right=True).to(torch.int32)
    if ng.quirk:
        assert ng.sid.dtype == torch.int32 and torch.equal(ng.sid, want)
    else:
        assert ng.sid is ng.gid and torch.equal(ng.gid, want)
    assert torch.equal(ng.ptr_sum.cpu().long(), sum_ptr) and torch.equal(ng.ptr_true.cpu().long(), true_ptr)


def test_deferred_plan_validation_reports_bad_indices_one_call_later():
    """plan_validation='deferred' (what TrainStep selects): no host sync in forward; an out-of-range index raises the
    reference's IndexError at the next forward / check instead of inside the call that used it."""
    from surface_texture_inpainting_net_amd.plan import check_deferred
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=1, n_levels=1,
               pooling_type='max', dilations=[1])
    torch.manual_seed(3)
    net = S.define_G(**cfg).to(DEV)
    good = make_synthetic_mesh(3000, 2, seed=1, dilations=()).to(DEV)
    bad = make_synthetic_mesh(3000, 2, seed=2, dilations=()).to(DEV)
    bad.edge_index[1, 5] = bad.x.shape[0] + 7
    with torch.no_grad():
        with pytest.raises(IndexError):
            net(bad)                                          # default 'sync': raised by the call itself
        bad._plan_cache = None
        net.plan_validation = 'deferred'
        out = net(bad)                                        # no raise here, nothing faulted (the pair was left out)
        assert bool(torch.isfinite(out).all())
        torch.cuda.synchronize()
        with pytest.raises(IndexError, match='deferred'):
            net(good)                                         # ... but by the next call
        assert bool(torch.isfinite(net(good)).all())          # the report is delivered once
        bad._plan_cache = None
        net(bad)
        with pytest.raises(IndexError):
            check_deferred(wait=True)
        check_deferred(wait=True)


def test_deferred_validation_of_a_plan_built_ahead_travels_on_the_build_stream():
    """A plan built ahead of its step (net.build_plan = GraphPlan.prefetch(join=False)) under 'deferred' validation copies its flag
    word to the host from the BUILD stream: forward() adds no copy of its own (same pinned word afterwards) unless something was
    built lazily since, and a bad index is still reported - at the latest by the next check."""
    from surface_texture_inpainting_net_amd.plan import check_deferred
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=1, n_levels=1,
               pooling_type='max', dilations=[1])
    torch.manual_seed(3)
    net = S.define_G(**cfg).to(DEV)
    net.plan_validation = 'deferred'
    good = make_synthetic_mesh(3000, 2, seed=1, dilations=()).to(DEV)
    bad = make_synthetic_mesh(3000, 2, seed=2, dilations=()).to(DEV)
    bad.edge_index[1, 5] = bad.x.shape[0] + 7
    check_deferred(wait=True)
    with torch.no_grad():
        plan = net.build_plan(good, inputs_ready=True)
        good._plan_cache = plan
        word, gen = plan._flag_host, plan._flag_gen
        assert word is not None and gen == plan._gen
        net(good)
        assert plan._flag_host is word or plan._validated          # no second copy: the build stream's is the pending one
        check_deferred(wait=True)
        assert plan._validated
        plan = net.build_plan(bad, inputs_ready=True)
        bad._plan_cache = plan
        with pytest.raises(IndexError, match='deferred'):        # by the call that uses the plan when the copy has landed by
            out = net(bad)                                       # then (nothing faulted either way), else by the next check
            assert bool(torch.isfinite(out).all())
            check_deferred(wait=True)
        check_deferred(wait=True)
        # a structure built lazily AFTER the copy gets a copy of its own at validate()
        plan = net.build_plan(good, inputs_ready=True)
        torch.cuda.synchronize()
        check_deferred(wait=True)
        assert plan._validated
        extra = bad.edge_index.clone()
        plan.edges_from_tensor(extra, good.x.shape[0])            # out-of-range pair, built on the compute stream
        assert not plan._validated and plan._flag_gen != plan._gen
        plan.validate()
        assert plan._flag_gen == plan._gen
        with pytest.raises(IndexError, match='deferred'):
            check_deferred(wait=True)
        check_deferred(wait=True)


def test_norm_backward_coefficients_fused_into_the_reduction_equal_the_separate_kernel():
    """STIN_POST_NORM_COEF (what the whole-block backward uses) = DOT_ELU sums followed by stin_norm_bwd_coef_f32, bit for bit."""
    n, C = 9000, 64
    batch = torch.cat([torch.zeros(4000), torch.ones(5000)]).long().to(DEV)
    groups = M._as_groups(batch, n, torch.device(DEV), False)
    g = torch.Generator().manual_seed(5)
    x, go = torch.randn(n, C, generator=g).to(DEV), torch.randn(n, C, generator=g).to(DEV)
    mean, rstd = SF.instance_stats(x, groups)
    T1, S0 = SF.colreduce(SF.RED_DOT_ELU, x, groups, groups.ptr_true, gout=go, mean=mean, rstd=rstd)
    k, m = torch.empty_like(rstd), torch.empty_like(rstd)
    SF._call('stin_norm_bwd_coef_f32', SF._ptr(T1), SF._ptr(S0), SF._ptr(rstd), SF._ptr(groups.inv_cnt), rstd.shape[0], C,
             SF._ptr(k), SF._ptr(m), SF._stream(x))
    k2, m2 = SF.colreduce(SF.RED_DOT_ELU, x, groups, groups.ptr_true, gout=go, mean=mean, rstd=rstd, post=SF.POST_NORM_COEF)
    assert torch.equal(k, k2) and torch.equal(m, m2)


def test_weight_gradient_side_stream_is_bit_identical_in_every_autograd_mode():
    """stin_edgeconv_block_bwd with a wgrad_stream (the default): same bits as the single-stream order for (a) a plain
    backward into empty .grad (join deferred to the end of the pass), (b) accumulation into existing .grad (joined inside
    the call), (c) torch.autograd.grad, repeated so that freed workspaces get recycled while side work is queued."""
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=3, n_levels=2,
               pooling_type='max', dilations=[1, 2, 4])
    torch.manual_seed(7)
    net = S.define_G(**cfg).to(DEV)
    s = make_synthetic_mesh(40_000, 3, seed=12, dilations=(2, 4)).to(DEV)
    params = list(net.parameters())

    def grads(mode):
        net.zero_grad(set_to_none=True)
        loss = net(s).square().mean()
        if mode == 'grad':
            return [g.clone() for g in torch.autograd.grad(loss, params)]
        loss.backward()
        if mode == 'accumulate':
            net(s).square().mean().backward()
        return [p.grad.clone() for p in params]

    old, old_min = SF.USE_WGRAD_STREAM, SF.WGRAD_MIN_WORK
    SF.WGRAD_MIN_WORK = 0.0                      # (a 40k-vertex step would otherwise stay on one stream: launch-bound)
    try:
        for mode in ('backward', 'accumulate', 'grad'):
            SF.USE_WGRAD_STREAM = False
            want = grads(mode)
            SF.USE_WGRAD_STREAM = True
            for _ in range(3):
                got = grads(mode)
                assert all(torch.equal(a, b) for a, b in zip(got, want)), mode
    finally:
        SF.USE_WGRAD_STREAM, SF.WGRAD_MIN_WORK = old, old_min


@pytest.mark.parametrize('Cin,H,Cout,shortcut,trans_inv', [(256, 512, 256, False, False), (128, 512, 256, True, False),
                                                           (64, 128, 64, False, True), (128, 128, 64, True, False),
                                                           (64, 256, 128, True, False), (16, 24, 8, False, False), (8, 8, 8, True, True)])
@pytest.mark.parametrize('modes', ['f16x3+frag / bf16x3+frag', 'f16x3 / bf16x3', 'plain', 'bf16 rows'])
def test_pack8_equals_the_elementwise_pack(Cin, H, Cout, shortcut, trans_inv, modes, monkeypatch):
    """stin_edgeconv_pack_f32 with eight elements of a destination row per thread (round 4, pack_body8) writes the same bytes as
    the one-element-per-thread pack (STIN_PACK8=0) into all five operands, for every storage mode a block uses: fragment-order
    and k-group split layouts, plain fp32, plain bf16 - and the plain-transpose pseudo job of functional.PackSet."""
    F16, BF16, FRAG, WB16 = SF.GEMM_F16X3, SF.GEMM_BF16X3, 0x400, SF.GEMM_W_BF16
    fm, bm = {'f16x3+frag / bf16x3+frag': (F16 | FRAG, BF16 | FRAG), 'f16x3 / bf16x3': (F16, BF16), 'plain': (0, 0),
              'bf16 rows': (WB16, WB16)}[modes]
    g = torch.Generator().manual_seed(Cin + H + Cout)
    Cp = Cin
    W1 = torch.randn(H, Cin if trans_inv else 2 * Cin, generator=g).to(DEV)
    b1 = torch.randn(H, generator=g).to(DEV)
    W2 = torch.randn(Cout, H, generator=g).to(DEV)
    Ws = torch.randn(Cout, Cin, generator=g).to(DEV) if shortcut else None
    bs = torch.randn(Cout, generator=g).to(DEV) if shortcut else None
    Yw = 2 * H + (Cout if shortcut else 0)

    def run(flag):
        monkeypatch.setenv('STIN_PACK8', flag)
        bufs = [torch.full((n,), -3.0, device=DEV) for n in (Yw * Cp, Yw, Yw * Cp, H * Cout, Cout * H)]
        SF._call('stin_edgeconv_pack_f32', SF._ptr(W1), SF._ptr(b1), SF._ptr(Ws), SF._ptr(bs), SF._ptr(W2), Cin, Cp, H, Cout, int(shortcut),
                 int(trans_inv), SF._ptr(bufs[0]), SF._ptr(bufs[1]), SF._ptr(bufs[2]), SF._ptr(bufs[3]), SF._ptr(bufs[4]), fm, bm,
                 SF._stream(W1))
        return [b.view(torch.int32).clone() for b in bufs]

    want, got = run('0'), run('1')
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    ps = SF.PackSet([], torch.device(DEV), False, (W2,))
    ps.run()
    assert torch.equal(ps.transposed[0], W2.t().contiguous())


@pytest.mark.parametrize('batched', [False, True])
def test_pack_many_equals_the_per_block_pack(batched):
    """functional.PackSet: the weight operands of every fused block packed by ONE launch at the start of forward (persistent
    buffers, job table in device memory) - outputs and gradients bit-identical to the per-block pack, over two steps with a
    weight update in between (the table must keep pointing at the live parameters) and after moving to another batch size."""
    from surface_texture_inpainting_net_amd.data import collate
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=3, n_levels=2,
               pooling_type='max', dilations=[1, 2, 1])
    if batched:
        samples = [collate([make_synthetic_mesh(n, 3, seed=70 + i, dilations=(2,)) for i, n in enumerate(ns)]).to(DEV)
                   for ns in ((900, 1500, 700), (1200, 800))]
    else:
        samples = [make_synthetic_mesh(n, 3, seed=70 + n, dilations=(2,)).to(DEV) for n in (5000, 3100)]

    def run(flag):
        old = SF.USE_PACK_MANY
        SF.USE_PACK_MANY = flag
        try:
            torch.manual_seed(9)
            net = S.define_G(**cfg).to(DEV)
            opt = torch.optim.SGD(net.parameters(), lr=0.05)
            outs = []
            for s in samples + samples[:1]:
                opt.zero_grad(set_to_none=True)
                out = net(s)
                out.square().mean().backward()
                outs += [out.detach().clone()] + [p.grad.clone() for p in net.parameters()]
                opt.step()
            assert (net._pack_set is not None) == flag
            return outs
        finally:
            SF.USE_PACK_MANY = old

    want, got = run(False), run(True)
    assert all(torch.equal(a, b) for a, b in zip(got, want))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('batched', [False, True])
def test_block_call_equals_per_kernel_path_bitwise(batched, dtype):
    """stin_edgeconv_block_fwd/bwd only enqueue the individual entry points: outputs and gradients must equal the
    per-kernel host path bit for bit - single graph, and a batch of unequal crops (the reference's linspace-slice
    statistics, fastinstancenorm.py:53-82, inside the block call)."""
    from surface_texture_inpainting_net_amd.data import collate
    if dtype == 'bf16' and not SF.USE_EDGE_MASK:
        pytest.skip('bf16 storage needs the saved ReLU mask (STIN_EDGE_MASK=0 set)')
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=3, n_levels=2,
               pooling_type='max', dilations=[1, 2, 1])
    torch.manual_seed(9)
    net = S.define_G(**cfg).to(DEV)
    if dtype == 'bf16':
        net.set_activation_dtype(torch.bfloat16)
    if batched:
        s = collate([make_synthetic_mesh(n, 3, seed=60 + i, dilations=(2,)) for i, n in enumerate((900, 1500, 700, 1210))]).to(DEV)
    else:
        s = make_synthetic_mesh(5000, 3, seed=60, dilations=(2,)).to(DEV)

    def run():
        net.zero_grad(set_to_none=True)
        out = net(s)
        out.float().square().mean().backward()
        return [out.detach().clone()] + [p.grad.clone() for p in net.parameters()]

    old = SF.USE_BLOCK_CALL
    try:
        SF.USE_BLOCK_CALL = False
        want = run()
        SF.USE_BLOCK_CALL = True
        got = run()
    finally:
        SF.USE_BLOCK_CALL = old
    if batched:
        assert any(g.quirk for g in s._plan_cache._norms.values()), 'the batch must exercise the linspace-slice path'
    assert all(torch.equal(a, b) for a, b in zip(got, want))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('batched', [False, True])
def test_bottleneck_chain_equals_per_block_nodes_bitwise(batched, dtype):
    """functional.EdgeConvChainFn (the bottleneck blocks as ONE autograd node, stin_edgeconv_chain_fwd / _bwd) only loops
    over the whole-block launch sequences: outputs, input gradient and every parameter gradient equal the per-block
    autograd nodes bit for bit - plain autograd and the TrainStep bucket route, single graph and a batch of unequal crops."""
    from surface_texture_inpainting_net_amd.data import collate
    from surface_texture_inpainting_net_amd.train_step import TrainStep
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=4, n_levels=2,
               pooling_type='max', dilations=[1, 2, 4, 1])
    if batched:
        s = collate([make_synthetic_mesh(n, 3, seed=80 + i, dilations=(2, 4)) for i, n in enumerate((900, 1500, 700))]).to(DEV)
    else:
        s = make_synthetic_mesh(5000, 3, seed=80, dilations=(2, 4)).to(DEV)

    def run(chain):
        old, old_net = SF.USE_CHAIN, SF.USE_NET_CALL
        SF.USE_CHAIN, SF.USE_NET_CALL = chain, False          # (the whole-network node would take the bottleneck with it)
        try:
            torch.manual_seed(9)
            net = S.define_G(**cfg).to(DEV)
            if dtype == 'bf16':
                net.set_activation_dtype(torch.bfloat16)
            x = s.x.clone().requires_grad_(True)
            s2 = type(s)(**{k: (x if k == 'x' else s[k]) for k in s.keys()})
            s2._nv_host = s._nv_host
            out = net(s2)
            out.float().square().mean().backward()
            res = [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in net.parameters()]
            # and three optimizer steps through the bucket route
            step = TrainStep(net, lr=1e-3)
            losses = [float(step(s)) for _ in range(3)]
            step.finish()
            return res + [torch.tensor(losses)] + [p.detach().clone() for p in net.parameters()]
        finally:
            SF.USE_CHAIN, SF.USE_NET_CALL = old, old_net

    want = run(False)
    before = SF.EdgeConvChainFn.calls
    got = run(True)
    assert SF.EdgeConvChainFn.calls == before + 4, 'the chain path must have been taken (1 plain + 3 TrainStep forwards)'
    assert len(want) == len(got)
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), i


def test_block_handoff_statistics_from_the_gradient_product_equal_the_separate_reduction(monkeypatch):
    """stin_net_bwd's block-to-block hand-off (stin_block.hip BwdLink): with the panel kernel serving the input-gradient
    products, the instance-norm backward sums of block k - 1 ride on block k's dx product.  Same training run with the
    hand-off on and off (STIN_DOTELU_FUSED): losses, input gradient and every parameter gradient agree to fp32 rounding of the
    two fp64 summation orders (in practice bit for bit), and the fused route was really taken (fewer reduction launches)."""
    from surface_texture_inpainting_net_amd.train_step import TrainStep
    monkeypatch.setenv('STIN_NT_PANEL', '1')
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=4, n_levels=2,
               pooling_type='max', dilations=[1, 2, 4, 1])
    s = make_synthetic_mesh(9000, 3, seed=31, dilations=(2, 4)).to(DEV)

    def run(fused):
        monkeypatch.setenv('STIN_DOTELU_FUSED', '1' if fused else '0')
        torch.manual_seed(5)
        net = S.define_G(**cfg).to(DEV)
        x = s.x.clone().requires_grad_(True)
        s2 = type(s)(**{k: (x if k == 'x' else s[k]) for k in s.keys()})
        s2._nv_host = s._nv_host
        out = net(s2)
        out.float().square().mean().backward()
        res = [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in net.parameters()]
        step = TrainStep(net, lr=1e-3)
        losses = [float(step(s)) for _ in range(3)]
        step.finish()
        return res + [torch.tensor(losses)]

    want, got = run(False), run(True)
    worst = 0.0
    for a, b in zip(got, want):
        scale = float(b.abs().max()) + 1e-30
        worst = max(worst, float((a - b).abs().max()) / scale)
    assert worst <= 1e-6, worst
    lib = _lib_load()
    N2 = int(s.num_vertices.reshape(-1)[-1])
    prec = SF.PREC_BWD | SF.GEMM_W_PRESPLIT | SF.GEMM_W_FRAG
    assert int(lib.stin_gemm_nt_dotelu_groups(N2, 256, 1024, prec)) > 0, 'the bottleneck dx product must be served by the panel kernel'
    monkeypatch.setenv('STIN_DOTELU_FUSED', '0')
    assert int(lib.stin_gemm_nt_dotelu_groups(N2, 256, 1024, prec)) == 0


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('batched', [False, True])
def test_network_graph_part_as_one_node_equals_per_op_nodes_bitwise(batched, dtype):
    """functional.NetFn (every fused block and pool / unpool step of the network as ONE autograd node, stin_net_fwd / _bwd)
    only loops over the per-op entry points: output, input gradient and every parameter gradient equal the per-op autograd
    nodes bit for bit - plain autograd (fresh gradient tensors, weight-gradient side stream with the deferred join) and the
    TrainStep bucket route, single graph and a batch of unequal crops (linspace-slice statistics), 3 levels."""
    from surface_texture_inpainting_net_amd.data import collate
    from surface_texture_inpainting_net_amd.train_step import TrainStep
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=3, n_levels=2,
               pooling_type='max', dilations=[1, 2, 4])
    if batched:
        s = collate([make_synthetic_mesh(n, 3, seed=90 + i, dilations=(2, 4)) for i, n in enumerate((900, 1500, 700))]).to(DEV)
    else:
        s = make_synthetic_mesh(6000, 3, seed=90, dilations=(2, 4)).to(DEV)

    def run(net_call):
        old = SF.USE_NET_CALL
        SF.USE_NET_CALL = net_call
        try:
            torch.manual_seed(11)
            net = S.define_G(**cfg).to(DEV)
            if dtype == 'bf16':
                net.set_activation_dtype(torch.bfloat16)
            x = s.x.clone().requires_grad_(True)
            s2 = type(s)(**{k: (x if k == 'x' else s[k]) for k in s.keys()})
            s2._nv_host = s._nv_host
            out = net(s2)
            out.float().square().mean().backward()
            res = [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in net.parameters()]
            net.zero_grad(set_to_none=True)
            out = net(s)                                       # the network input needs no gradient: op 0 writes none
            out.float().abs().mean().backward()
            res += [p.grad.clone() for p in net.parameters()]
            step = TrainStep(net, lr=1e-3)
            losses = [float(step(s)) for _ in range(3)]
            step.finish()
            return res + [torch.tensor(losses)] + [p.detach().clone() for p in net.parameters()]
        finally:
            SF.USE_NET_CALL = old

    want = run(False)
    before = SF.NetFn.calls
    got = run(True)
    assert SF.NetFn.calls == before + 5, 'the whole-network path must have been taken (2 plain + 3 TrainStep forwards)'
    assert len(want) == len(got)
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), i


def test_batch_of_unequal_crops_full_width_vs_oracle():
    """The linspace-slice quirk through the whole-block calls (ngf = 64: saved-mask widths) against the CPU oracle."""
    from surface_texture_inpainting_net_amd.data import collate
    batch = collate([make_synthetic_mesh(n, 3, seed=70 + i, dilations=(2,)) for i, n in enumerate((800, 1300, 600))])
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=2, n_levels=2,
               pooling_type='max', dilations=[1, 2])
    torch.manual_seed(22)
    ref = stin_oracle.define_G(**cfg)
    net = S.define_G(**cfg)
    net.load_state_dict(ref.state_dict())
    net = net.to(DEV)
    want = ref(batch)
    stin_oracle.compute_loss(stin_oracle.graph_forward(ref, batch), batch.color, batch.mask).backward()
    bd = batch.to(DEV)
    got = net(bd)
    stin_oracle.compute_loss(torch.where((bd.mask > 0).expand_as(bd.color), got, bd.color), bd.color, bd.mask).backward()
    assert float((got.detach().cpu() - want.detach()).abs().max()) <= FWD_TOL
    scale = max(float(p.grad.abs().max()) for p in ref.parameters())
    num = den = 0.0
    for (k, p), q in zip(net.named_parameters(), ref.parameters()):
        d = p.grad.cpu() - q.grad
        assert float(d.abs().max()) <= 1e-2 * scale, k
        num += float(d.double().pow(2).sum())
        den += float(q.grad.double().pow(2).sum())
    assert (num / den) ** 0.5 <= 3e-3


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_train_step_direct_bucket_gradients_equal_autograd_gradients(dtype):
    """TrainStep lets the whole-block backward write weight gradients straight into the flat all-reduce bucket (no
    per-parameter accumulate node, no copy): the bucket must hold exactly the gradients autograd would have produced,
    step after step."""
    from surface_texture_inpainting_net_amd.train_step import TrainStep
    if dtype == 'bf16' and not SF.USE_EDGE_MASK:
        pytest.skip('bf16 storage needs the saved ReLU mask (STIN_EDGE_MASK=0 set)')
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=3, n_levels=2,
               pooling_type='max', dilations=[1, 2, 4])
    torch.manual_seed(13)
    net = S.define_G(**cfg).to(DEV)
    if dtype == 'bf16':
        net.set_activation_dtype(torch.bfloat16)
    s = make_synthetic_mesh(20_000, 3, seed=14, dilations=(2, 4)).to(DEV)
    step = TrainStep(net, lr=0.0)                       # lr 0: parameters stay put, every step must give the same gradients
    names = [k for k, p in net.named_parameters() if p.requires_grad]
    for _ in range(3):
        step(s)
        direct = step.bucket.flat.clone()
        if SF.USE_DIRECT_GRADS and SF.USE_BLOCK_CALL and SF.USE_EDGE_MASK:          # (environment toggles)
            assert sum(step.bucket.written) >= 4 * 7, 'the block weights took the direct path'
    for p in net.parameters():
        p.grad = None
    loss = SF.masked_l1_loss(net(s), s.color, s.mask, True)
    loss.backward()
    off = 0
    for k, p in zip(names, step.bucket.params):
        want = p.grad if p.grad is not None else torch.zeros_like(p)
        assert torch.equal(direct[off:off + p.numel()].view_as(p), want), k
        off += p.numel()


def test_train_step_with_a_frozen_middle_block_trains_the_rest():
    """A partially frozen model (the bottleneck's middle block here, requires_grad=False before the TrainStep is built): the
    whole-network node probes every block BEFORE it touches the bucket's bookkeeping and, finding one block without a slot, hands
    fresh gradient tensors to autograd for all of them (round-3 advisor: it used to raise with `written` half set).  The
    gradients of the trainable parameters equal plain autograd's, the frozen ones get none, repeated steps work."""
    from surface_texture_inpainting_net_amd.train_step import TrainStep
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=3, n_levels=2,
               pooling_type='max', dilations=[1, 2, 4])
    torch.manual_seed(13)
    net = S.define_G(**cfg).to(DEV)
    frozen = [p for k, p in net.named_parameters() if k.startswith('bottleneck_blocks.1.')]
    assert frozen
    for p in frozen:
        p.requires_grad_(False)
    s = make_synthetic_mesh(8_000, 3, seed=14, dilations=(2, 4)).to(DEV)
    step = TrainStep(net, lr=0.0)
    for _ in range(3):
        step(s)
    got = step.bucket.flat.clone()
    assert all(p.grad is None for p in frozen)
    for p in net.parameters():
        p.grad = None
    SF.masked_l1_loss(net(s), s.color, s.mask, True).backward()
    off = 0
    for p in step.bucket.params:
        want = p.grad if p.grad is not None else torch.zeros_like(p)
        assert torch.equal(got[off:off + p.numel()].view_as(p), want)
        off += p.numel()
    assert all(p.grad is None for p in frozen)


def test_training_loop_memory_is_stable_over_changing_scene_sizes():
    """The allocator pool must reach a steady state when scenes of different sizes alternate (a training epoch): the
    weight-gradient side stream keeps its inputs alive by reference until the end-of-backward join instead of
    record_stream, whose deferred frees let the pool grow by ~1 % per step."""
    from surface_texture_inpainting_net_amd.train_step import TrainStep
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=3, n_levels=2,
               pooling_type='max', dilations=[1, 2, 4])
    torch.manual_seed(3)
    net = S.define_G(**cfg).to(DEV)
    step = TrainStep(net, lr=1e-4)
    scenes = [make_synthetic_mesh(n, 3, seed=80 + i, dilations=(2, 4)).to(DEV) for i, n in enumerate((30_000, 44_000, 37_000))]

    def run(n):
        for i in range(n):
            s = scenes[i % 3]
            s._plan_cache = None
            step(s)
        torch.cuda.synchronize()
        return torch.cuda.memory_reserved()

    base = run(24)
    later = run(90)
    step.finish()
    assert later <= base * 1.10 + (64 << 20), (base, later)


@pytest.mark.parametrize('case', ['no_edges', 'two_vertices', 'self_loops_and_duplicates', 'one_hub'])
def test_graph_resnet_block_on_degenerate_graphs_vs_oracle(case):
    """Edge cases through the whole-block calls at a saved-mask width (Cout = 64): an EMPTY edge set (every vertex
    isolated: aggregation 0, masked bias off, output = residual), a 2-vertex graph, repeated / self-loop edges (PyG
    counts each occurrence), and one hub that every other vertex points at (one long CSR row, the rest empty)."""
    torch.manual_seed(31)
    n = {'no_edges': 37, 'two_vertices': 2, 'self_loops_and_duplicates': 50, 'one_hub': 300}[case]
    if case == 'no_edges':
        ei = torch.zeros(2, 0, dtype=torch.int64)
    elif case == 'two_vertices':
        ei = torch.tensor([[0, 1], [1, 0]])
    elif case == 'self_loops_and_duplicates':
        a = torch.randint(0, n, (400,))
        b = torch.randint(0, n, (400,))
        ei = torch.stack([torch.cat([a, a[:100], torch.arange(n)]), torch.cat([b, b[:100], torch.arange(n)])])
    else:
        ei = torch.stack([torch.arange(1, n), torch.zeros(n - 1, dtype=torch.int64)])
    for cin, cout in ((64, 64), (32, 64)):
        ref = stin_oracle.OracleBlock(cin, cout, 'edgeconv', 'instance')
        with torch.no_grad():
            for p in ref.parameters():
                p.uniform_(-0.3, 0.3)                     # non-zero biases: the [deg > 0] bias mask matters
        blk = S.GraphResnetBlock(cin, cout, M.get_gcn_filter, M.FastInstanceNorm, False, True)
        blk.load_state_dict(ref.state_dict())
        blk = blk.to(DEV)
        x = torch.randn(n, cin)
        w = torch.randn(n, cout)
        xr = x.clone().requires_grad_(True)
        yr = ref(xr, ei)
        (yr * w).sum().backward()
        xd = x.to(DEV).requires_grad_(True)
        yd = blk(xd, ei.to(DEV), None)
        (yd * w.to(DEV)).sum().backward()
        # two vertices: the instance norm of two points is +-1 per channel unless they are closer than sqrt(eps) - then a
        # 1e-7 difference in the GEMM is amplified by up to 1/sqrt(eps) = 316: looser bound for that case only
        # (one_hub: the hub normalises to ~sqrt(N) = 17 and leaves the block at ~20: the bound is relative to the output
        # range; the fp32 CPU oracle itself is 1e-4 away from an fp64 run there, this path 2.3e-4)
        tol = 5e-4 if case == 'two_vertices' else 2e-5 * max(1.0, float(yr.detach().abs().max()))
        err = float((yd.detach().cpu() - yr.detach()).abs().max())
        assert err <= tol, (case, cin, err)
        gs = max(1e-6, float(xr.grad.abs().max()))
        gerr = float((xd.grad.cpu() - xr.grad).abs().max())
        assert gerr <= (5e-2 if case == 'two_vertices' else 2e-3) * gs, (case, cin, gerr, gs)
        scale = max(1e-6, max(float(p.grad.abs().max()) for p in ref.parameters()))
        for (k, p), q in zip(blk.named_parameters(), ref.parameters()):
            assert float((p.grad.cpu() - q.grad).abs().max()) <= (5e-2 if case == 'two_vertices' else 2e-3) * scale, (case, cin, k)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_model_with_an_empty_dilated_edge_set(dtype):
    """A dilation whose walk found no edge at all (graph_dilation.py may return an empty list for a level): the
    bottleneck block on that set sees only isolated vertices.  fp32: against the oracle; bf16: runs, finite, and the
    block call equals the per-kernel path."""
    if dtype == 'bf16' and not SF.USE_EDGE_MASK:
        pytest.skip('bf16 storage needs the saved ReLU mask (STIN_EDGE_MASK=0 set)')
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=3, n_levels=2,
               pooling_type='max', dilations=[1, 2, 4])
    torch.manual_seed(17)
    ref = stin_oracle.define_G(**cfg)
    net = S.define_G(**cfg)
    net.load_state_dict(ref.state_dict())
    net = net.to(DEV)
    s = make_synthetic_mesh(3000, 3, seed=33, dilations=(2, 4))
    key = [k for k in s.keys() if k.startswith('hierarchy_dil_4_edge_index')][0]
    s[key] = torch.zeros(2, 0, dtype=torch.int64)
    sd = s.to(DEV)
    if dtype == 'f32':
        want = ref(s)
        want.square().mean().backward()
        got = net(sd)
        got.square().mean().backward()
        assert float((got.detach().cpu() - want.detach()).abs().max()) <= FWD_TOL
        scale = max(float(p.grad.abs().max()) for p in ref.parameters())
        for (k, p), q in zip(net.named_parameters(), ref.parameters()):
            assert float((p.grad.cpu() - q.grad).abs().max()) <= 1e-2 * scale, k
    else:
        net.set_activation_dtype(torch.bfloat16)
        outs = []
        for blockcall in (True, False):
            old = SF.USE_BLOCK_CALL
            SF.USE_BLOCK_CALL = blockcall
            try:
                net.zero_grad(set_to_none=True)
                out = net(sd)
                out.float().square().mean().backward()
                outs.append([out.detach().clone()] + [p.grad.clone() for p in net.parameters()])
            finally:
                SF.USE_BLOCK_CALL = old
        assert all(torch.isfinite(t).all() for t in outs[0])
        assert all(torch.equal(a, b) for a, b in zip(*outs))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_vertex_renumbering_by_locality_is_invisible_at_the_boundary(dtype, monkeypatch):
    """plan.GraphPlan._ensure_order (Morton order of the position channels at level 0, first-child order above) relabels
    index VALUES only: outputs come back in the sample's vertex order and equal the un-renumbered run up to the rounding of
    the instance-norm column sums (other row order); every CSR row keeps its neighbour order and every children list its
    original order - checked on a mesh with exact ties in the max pooling (duplicated feature rows), where the gradient
    must reach the same child."""
    from surface_texture_inpainting_net_amd import plan as P
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=2, n_levels=2,
               pooling_type='max', dilations=[1, 2])
    s = make_synthetic_mesh(4000, 3, seed=31, dilations=(2,))
    m = s['x'][1::3].shape[0]
    s['x'][0:3 * m:3] = s['x'][1::3]                                        # duplicated rows: ties at every level's max pooling
    s = s.to(DEV)

    def run(reorder):
        monkeypatch.setattr(P, 'REORDER', reorder)
        monkeypatch.setattr(P, 'REORDER_MIN', 0)
        torch.manual_seed(5)
        net = S.define_G(**cfg).to(DEV)
        if dtype == 'bf16':
            net.set_activation_dtype(torch.bfloat16)
        s._plan_cache = None
        x = s.x.clone().requires_grad_(True)
        s2 = type(s)(**{k: (x if k == 'x' else s[k]) for k in s.keys()})
        s2._nv_host = s._nv_host
        out = net(s2)
        assert (s2._plan_cache.order0 is not None) == reorder
        (out.float() * torch.linspace(-1, 1, out.numel(), device=DEV).view_as(out)).sum().backward()
        return [out.detach().float(), x.grad.float()] + [p.grad.clone() for p in net.parameters()]

    want, got = run(False), run(True)
    tol = 2e-5 if dtype == 'f32' else 6e-2
    gscale = max(float(b.abs().max()) for b in want[2:])                   # (bias gradients in front of a norm are pure cancellation noise)
    for i, (a, b) in enumerate(zip(got, want)):
        scale = max(float(b.abs().max()), 5e-2 * gscale if i >= 2 else 0.0) + 1e-6
        assert float((a - b).abs().max()) <= tol * scale, (i, float((a - b).abs().max()), scale)


def test_vertex_renumbering_reports_out_of_range_indices():
    """An out-of-range edge / trace id must still raise IndexError when the plan renumbers (the relabel maps it to the
    out-of-range sentinel instead of indexing past the rank table)."""
    from surface_texture_inpainting_net_amd import plan as P
    cfg = dict(input_nc=10, output_nc=3, ngf=8, filter_type='edgeconvtransinv', norm='instance', n_blocks=1, n_levels=1,
               pooling_type='max')
    net = S.define_G(**cfg).to(DEV)
    old = P.REORDER_MIN
    P.REORDER_MIN = 0
    try:
        for key, val in (('edge_index', 10 ** 6), ('hierarchy_trace_index_1', -3), ('hierarchy_edge_index_1', 70000)):
            s = make_synthetic_mesh(600, 2, seed=3, dilations=())
            s[key] = s[key].clone()
            s[key].view(-1)[5] = val
            with pytest.raises(IndexError):
                net(s.to(DEV))
                torch.cuda.synchronize()
    finally:
        P.REORDER_MIN = old


@pytest.mark.parametrize('M,Nc,K', [(8100, 1024, 512), (2700, 512, 1024), (777, 136, 264), (4097, 256, 128), (65, 128, 128), (1, 128, 256),
                                    (20000, 320, 128), (3000, 2048, 256), (1500, 520, 776), (63, 512, 512), (8100, 4096, 1024)])
def test_gemm_tn_bf16_transposed_read_kernel_equals_register_transpose_kernel(M, Nc, K, monkeypatch):
    """k_gemm_tn_b16_tr (row-major LDS tiles, ds_read_b64_tr_b16 fragments, two buffers) against k_gemm_tn_b16 (register
    transposes, k-major LDS image): same products in the same k order -> the weight-gradient block is bit-identical; the bias
    column (sum_m w[m] G[m, i], accumulated from the A fragments) equals it to fp32 rounding; ragged rows / columns, strided
    operands, with and without the ones column and row weights.  And against fp64."""
    g = torch.Generator().manual_seed(M + Nc + K)
    G = torch.randn(M + 2, Nc + 8, generator=g).to(DEV).bfloat16()[1:M + 1, :Nc]
    X = torch.randn(M, K + 16, generator=g).to(DEV).bfloat16()[:, 8:K + 8]
    w = torch.rand(M, 8, generator=g).to(DEV).bfloat16()[:, 3]
    big = Nc >= 512 and K >= 512                  # 256 x 256 tiles: other row chunks -> the slab sums associate differently
    for kw in (dict(), dict(ones_column=True), dict(ones_column=True, row_weight=w)):
        monkeypatch.setenv('STIN_TN_TR', '0')
        monkeypatch.setenv('STIN_TN_BIG', '0')
        base = SF.gemm_tn(G, X, **kw)
        monkeypatch.setenv('STIN_TN_TR', '1')
        monkeypatch.delenv('STIN_TN_BIG')
        for _ in range(2):
            got = SF.gemm_tn(G, X, **kw)
            assert got.shape == base.shape
            if big:
                assert float((got[:, :K] - base[:, :K]).abs().max()) <= 2e-5 * (float(base.abs().max()) + 1.0) * max(1.0, (M / 1000) ** 0.5)
                monkeypatch.setenv('STIN_TN_BIG', '0')
                assert torch.equal(SF.gemm_tn(G, X, **kw)[:, :K], base[:, :K]), sorted(kw)      # the 128 x 128 transposed-read kernel
                monkeypatch.delenv('STIN_TN_BIG')
            else:
                assert torch.equal(got[:, :K], base[:, :K]), sorted(kw)
            if kw:
                assert float((got[:, K] - base[:, K]).abs().max()) <= 2e-5 * (float(base[:, K].abs().max()) + 1.0) * max(1.0, (M / 1000) ** 0.5)
    want = torch.cat([G.double().t() @ X.double(), (G.double() * w.double()[:, None]).sum(0)[:, None]], 1)
    got = SF.gemm_tn(G, X, ones_column=True, row_weight=w).double()
    assert float((got - want).abs().max()) <= 3e-5 * (float(want.abs().max()) + 1.0) * max(1.0, (M / 1000) ** 0.5)


def test_vertex_order_hip_path_equals_the_torch_formulation(monkeypatch):
    """csrc/stin_order.hip (bounding box + Morton keys + radix sorts + batched relabelling, one foreign call each) against the
    framework-op formulation in plan.GraphPlan._ensure_order: same permutation on every level, and the plans built from the two
    are identical array by array (both CSRs of every edge set, cross maps, children lists, traces)."""
    from surface_texture_inpainting_net_amd import plan as P
    s = make_synthetic_mesh(30000, 3, seed=17, dilations=(2, 4)).to(DEV)
    monkeypatch.setattr(P, 'REORDER', True)
    monkeypatch.setattr(P, 'REORDER_MIN', 0)
    edges = [('edge_index', 0), ('hierarchy_edge_index_1', 1), ('hierarchy_edge_index_2', 2), ('hierarchy_dil_2_edge_index_2', 2),
             ('hierarchy_dil_4_edge_index_2', 2)]
    plans = {}
    for impl in ('torch', 'hip'):
        monkeypatch.setattr(P, 'REORDER_IMPL', impl)
        pl = P.GraphPlan(s, positions=(6, 9))
        pl.ensure(edges, [1, 2])
        pl.validate()
        plans[impl] = pl
    a, b = plans['torch'], plans['hip']
    assert b._ranks[0].dtype == torch.int32 and a._ranks[0].dtype == torch.int64
    for ra, rb in zip(a._ranks, b._ranks):
        assert torch.equal(ra, rb.to(torch.int64))
    assert torch.equal(a.order0, b.order0) and torch.equal(a.rank0, b.rank0)
    assert sorted(b.order0.tolist()) == list(range(s.x.shape[0]))
    for key, _ in edges:
        ea, eb = a._edges[key], b._edges[key]
        for name in ('rowptr', 'col'):
            assert torch.equal(getattr(ea.by_dst, name), getattr(eb.by_dst, name)), (key, name)
            assert torch.equal(getattr(ea.by_src, name), getattr(eb.by_src, name)), (key, name)
        assert torch.equal(ea.xslot, eb.xslot) and torch.equal(ea.w_src, eb.w_src) and torch.equal(ea.inv_deg, eb.inv_deg)
    for lvl in (1, 2):
        pa, pb = a._pools[lvl], b._pools[lvl]
        assert torch.equal(pa.trace, pb.trace) and torch.equal(pa.children.rowptr, pb.children.rowptr)
        assert torch.equal(pa.children.col, pb.children.col) and torch.equal(pa.inv_count, pb.inv_count)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_no_grad_forward_is_bit_identical_to_the_training_forward_and_keeps_no_mask(dtype):
    """Round 6: the reference's validation loop runs model(data) under torch.no_grad() (trainers/inpainting3d_trainer.py:204-263).
    There the whole-network call stores no ReLU mask (stin_edgeconv_block_fwd: mask NULL), shares one temporaries region between
    all blocks and ping-pongs the op outputs - same kernels otherwise: the colours equal the differentiable forward's bit for bit,
    no autograd graph is built, and the pass allocates a fraction of the training forward's arena."""
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=3, n_levels=2,
               pooling_type='max', dilations=[1, 2, 4])
    torch.manual_seed(3)
    net = S.define_G(**cfg).to(DEV)
    if dtype == torch.bfloat16:
        net.set_activation_dtype(torch.bfloat16)
    s = make_synthetic_mesh(20000, 3, seed=5, dilations=(2, 4)).to(DEV)
    calls = SF.NetFn.calls
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    want = net(s)
    torch.cuda.synchronize()
    peak_train = torch.cuda.max_memory_allocated() - base
    assert SF.NetFn.calls == calls + 1 and want.requires_grad
    want = want.detach().clone()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with torch.no_grad():
        got = net(s)
    torch.cuda.synchronize()
    peak_eval = torch.cuda.max_memory_allocated() - base
    assert SF.NetFn.calls == calls + 2 and not got.requires_grad and got.grad_fn is None
    assert torch.equal(got, want)
    assert peak_eval < 0.5 * peak_train, (peak_eval, peak_train)
    net.eval()                                                # eval mode changes nothing for instance norm: still the same bits
    with torch.no_grad():
        again = net(s)
    assert torch.equal(again, want)
    for p in net.parameters():                                # frozen parameters + a plain input: no gradient needed either
        p.requires_grad_(False)
    net.train()
    frozen = net(s)
    assert not frozen.requires_grad and torch.equal(frozen, want)


# ----------------------------------------------------------------- translation-invariant blocks, compact layout (round 6)
@pytest.mark.parametrize('H', [128, 256, 512, 1024, 2048])
def test_trans_inv_compact_edge_kernels_equal_the_materialised_form(H):
    """EdgeConvTransInv's message is nn(x_j - x_i) (models/modules/edge_conv_translation_invariance.py:20-22): A_i = b1 - B_i.  The
    compact forward forms A per row instead of reading a GEMM output - same rows and same mask bit for bit - and the compact backward
    writes D = dB - dA (one rounding of the difference of the pair launch's two outputs) plus the column sums of dA (= db1)."""
    n, e = 1500, 9000
    ei = _random_graph(n, e, seed=H + 3)
    es = EdgeSet(ei.to(DEV), n, _bad())
    g = torch.Generator().manual_seed(H)
    Bm = torch.randn(n, H + 8, generator=g).to(DEV)[:, :H]         # a column slice of a wider matrix, as B is of Y
    b1 = torch.randn(H, generator=g).to(DEV)
    for bias in (b1, None):
        A = (-Bm + bias) if bias is not None else (-Bm + 0.0)        # what the [-W1 ; W1] product's A columns hold: fl(b1 - B_i)
        out0, out1 = torch.empty(n, H + 4, device=DEV), torch.full((n, H + 4), 3.0, device=DEV)
        m0 = torch.zeros(e * (H // 32), dtype=torch.int32, device=DEV)
        m1 = torch.zeros_like(m0)
        SF.edge_relu_mean_fwd(A.contiguous(), Bm, es.by_dst, out0, indicator=True, mask=m0)
        SF.edge_relu_mean_fwd_ti(bias, Bm, es.by_dst, out1, indicator=True, mask=m1)
        assert torch.equal(out0, out1) and torch.equal(m0, m1)
        out2 = torch.empty(n, H + 4, device=DEV)
        SF.edge_relu_mean_fwd_ti(bias, Bm, es.by_dst, out2, indicator=True, mask=None)      # the no-grad forward: no mask store
        assert torch.equal(out0, out2)
    Gr = torch.randn(n, H, generator=g).to(DEV)
    dA, dB = torch.empty(n, H, device=DEV), torch.empty(n, H, device=DEV)
    SF.edge_relu_mean_bwd_mask(Gr, m0, es, dA, dB)
    dY = torch.full((n, H + H // 2 + 4), 7.0, device=DEV)
    src = torch.randn(n, H, device=DEV)[:, :H // 2]
    db1 = SF.edge_relu_mean_bwd_mask_ti(Gr, m0, es, dY[:, :H], copy_src=src, copy_dst=dY[:, H:H + H // 2])
    assert torch.equal(dY[:, :H], dB - dA) and torch.equal(dY[:, H:H + H // 2], src)
    assert float((dY[:, H + H // 2:] - 7.0).abs().max()) == 0.0
    want = dA.double().sum(0)
    assert float((db1.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    db1_again = SF.edge_relu_mean_bwd_mask_ti(Gr, m0, es, torch.empty(n, H, device=DEV))
    assert torch.equal(db1, db1_again), 'fixed-order partial sums: bit-reproducible'


@pytest.mark.parametrize('shortcut', [False, True])
def test_trans_inv_compact_block_equals_the_materialised_block(shortcut, monkeypatch):
    """One GraphResnetBlock, EdgeConvTransInv, compact layout (Y = [B | S], H (+ Cout) columns) against both halves materialised
    (STIN_TI_COMPACT=0: Y = [A | B | S]): B has the same bits, A_i = b1 - B_i differs from the [-W1] product's A by <= 1 ulp of the
    accumulator (the MFMA adder is not symmetric under negation), so the forward agrees to ~1e-6 and the backward to fp32 rounding
    plus the handful of ReLU decisions that ulp flips (dx = (dB - dA) W1 with one rounding of the difference instead of a product over
    2 H columns; dW1 = D^T x instead of dB^T x - dA^T x; db1 from the column partials of dA) - on the whole-block C path, the per-op
    path and the whole-network node."""
    from surface_texture_inpainting_net_amd.surfacetextureinpaintingnet import GraphResnetBlock
    s = make_synthetic_mesh(6000, 1, seed=9, dilations=()).to(DEV)
    n = s.x.shape[0]                                           # (the generator rounds to its grid)
    es = plan_for(s).edges('edge_index', 0)
    cin, cout = (64, 128) if shortcut else (128, 128)
    torch.manual_seed(1)
    blk = GraphResnetBlock(cin, cout, M.get_gcn_filter, M.FastInstanceNorm, False, True, module=M.EdgeConvTransInv, double_input=False).to(DEV)
    with torch.no_grad():
        blk.first_filter.nn[0].bias.normal_(0, 0.3)            # (zero at init in the reference: make b1 matter)
    x0 = torch.randn(n, cin, device=DEV)
    w = torch.randn(n, cout, device=DEV)

    def run(compact, per_op):
        monkeypatch.setattr(SF, 'TI_COMPACT', compact)
        monkeypatch.setattr(SF, 'USE_BLOCK_CALL', not per_op)
        blk.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        y = blk(x, es)
        (y * w).sum().backward()
        return [y.detach(), x.grad] + [p.grad.clone() for p in blk.parameters()]

    base = run(False, False)
    names = ['out', 'dx'] + [k for k, _ in blk.named_parameters()]
    for compact, per_op in ((True, False), (True, True)):
        got = run(compact, per_op)
        assert float((got[0] - base[0]).abs().max()) <= 1e-5, 'forward'
        gscale = max(float(b.abs().max()) for b in base[2:])           # (bias gradients in front of a norm are pure cancellation noise)
        for k, a, b in zip(names[1:], got[1:], base[1:]):
            scale = max(float(b.abs().max()), 5e-2 * gscale) + 1e-12
            assert float((a - b).abs().max()) <= 1e-4 * scale, (k, compact, per_op, float((a - b).abs().max()), scale)
    c_fast, c_op = run(True, False), run(True, True)
    for k, a, b in zip(names, c_fast, c_op):                   # same kernels, same fold order on both paths
        assert torch.equal(a, b), k
