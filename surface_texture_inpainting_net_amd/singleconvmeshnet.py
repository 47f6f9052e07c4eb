"""SingleConvMeshNet on MI355X: the reference's U-Net over the mesh hierarchy whose EdgeConv filters
carry BatchNorm1d INSIDE the per-edge MLP (models/singleconvmeshnet.py:10-156,
models/modules/edge_conv_filter.py:34-44 `with_norm=True`) - same constructor, same `forward(sample)`, same state_dict keys.

The BatchNorm statistics run over all E per-edge activations, so the per-vertex restructure of the inpainting net stops
half-way: Lin1 has no bias and is linear, hence  Lin1([x_i ; x_j - x_i]) = A_i + B_j  with per-VERTEX GEMMs
(A = x (Wa - Wb)^T, B = x Wb^T; TransInv: A = -x W1^T, B = x W1^T), but BN1 -> ReLU -> Lin2 -> BN2 need genuine
per-EDGE tensors: h_e = A_dst + B_src (two row gathers), BN over E rows, one per-EDGE MFMA GEMM, BN over E rows, and a
CSR segment-mean back to the vertices.  All of it runs on the library's kernels (row gather, segment sum, fp64-accumulated
column statistics, MFMA GEMMs, pool / unpool); the per-channel affine, ReLU, residual add and concatenation are plain
elementwise framework ops.  No float atomics: backward of a gather is a segment sum over the opposite CSR.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import functional as SF
from .plan import NormGroups, PoolMap, _ptr, _stream, build_csr


class _EdgeIndex:
    """Both CSRs (with the slot -> edge id permutation) and the int32 endpoints of one edge set."""

    def __init__(self, edge_index, n, bad):
        lib = _lib.load()
        src, dst = edge_index[0].contiguous(), edge_index[1].contiguous()
        self.n, self.E = n, int(dst.numel())
        self.by_dst = build_csr(dst, None, n, max(self.E, 1), bad, want_perm=True)
        self.by_src = build_csr(src, None, n, max(self.E, 1), bad, want_perm=True)
        self.src32 = torch.empty(max(self.E, 1), dtype=torch.int32, device=dst.device)[:self.E]
        self.dst32 = torch.empty(max(self.E, 1), dtype=torch.int32, device=dst.device)[:self.E]
        for a, b in ((src, self.src32), (dst, self.dst32)):
            _lib.check(lib.stin_narrow_i64_to_i32(_ptr(a), self.E, n, _ptr(b), _ptr(bad), _stream(a)), 'stin_narrow_i64_to_i32')
        self._has_in = None

    @property
    def has_in(self):
        """[N, 1] float: 1 where the vertex has at least one in-edge."""
        if self._has_in is None:
            rp = self.by_dst.rowptr
            self._has_in = (rp[1:] > rp[:-1]).to(torch.float32).view(-1, 1)
        return self._has_in


def _as_edge_index(ei, n):
    """_EdgeIndex, or a raw [2, E] int64 edge_index as the reference's modules receive it (validated on the spot)."""
    if isinstance(ei, _EdgeIndex):
        return ei
    bad = torch.zeros(1, dtype=torch.int32, device=ei.device)
    out = _EdgeIndex(ei, n, bad)
    if int(bad.item()) != 0:
        raise IndexError('edge index out of range for %d vertices' % n)
    return out


class _GatherRowsFn(torch.autograd.Function):
    """out[e] = x[idx[e]];  backward: segment sum of the edge gradients over the CSR grouped by idx."""

    @staticmethod
    def forward(ctx, x, idx32, csr):
        ctx.csr, ctx.n = csr, x.shape[0]
        return SF.gather_rows(x, idx32)

    @staticmethod
    def backward(ctx, g):
        return SF.segment_sum(g, ctx.csr.rowptr, ctx.csr.col, ctx.n, mean=False), None, None


class _GatherAddFn(torch.autograd.Function):
    """h[e] = y[dst[e], :h] + y[src[e], h:] in one pass over y = [A | B]; backward: the two segment sums of the edge
    gradient over the destination / source CSRs (fixed order, no atomics) written into the halves of dy."""

    @staticmethod
    def forward(ctx, y, ei):
        y, ld = SF._mat(y)
        e, h = ei.dst32.shape[0], y.shape[1] // 2
        out = torch.empty(e, h, dtype=y.dtype, device=y.device)
        SF._call('stin_gather_add_rows_f32', SF._ptr(y), ld, SF._ptr(ei.dst32), y.data_ptr() + 4 * h, ld, SF._ptr(ei.src32),
                 e, h, SF._ptr(out), h, SF._stream(y))
        ctx.ei, ctx.n = ei, y.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        ei, n = ctx.ei, ctx.n
        g, ldg = SF._mat(g)
        h = g.shape[1]
        dy = torch.empty(n, 2 * h, dtype=g.dtype, device=g.device)
        for off, csr in ((0, ei.by_dst), (h, ei.by_src)):
            SF._call('stin_segment_sum_f32', SF._ptr(g), ldg, SF._ptr(csr.rowptr), SF._ptr(csr.col), n, h, 0,
                     dy.data_ptr() + 4 * off, 2 * h, SF._stream(g))
        return dy, None


class _ScatterMeanFn(torch.autograd.Function):
    """out[i] = mean of the edge rows whose target is i (0 for vertices without in-edges): scatter_mean."""

    @staticmethod
    def forward(ctx, m, ei):
        ctx.ei = ei
        return SF.segment_sum(m, ei.by_dst.rowptr, ei.by_dst.col, ei.n, mean=True)

    @staticmethod
    def backward(ctx, g):
        ei = ctx.ei
        return SF.gather_rows(g, ei.dst32, ei.by_dst.inv_deg), None


class _SkipUnpoolConcatFn(torch.autograd.Function):
    """torch.cat((skip, coarse[trace]), -1) in one launch (stin_concat_unpool_f32: the unpool gather writes straight into its half of
    the concatenated rows); backward: the skip gradient is the left column block of the incoming gradient (a view), the coarse
    gradient the segment sum of the right block over each coarse vertex's children - UnpoolFn's backward on a strided view."""

    @staticmethod
    def forward(ctx, skip, coarse, pool):
        skip, lds = SF._mat(skip)
        coarse, ldc = SF._mat(coarse)
        n, cs, cu = skip.shape[0], skip.shape[1], coarse.shape[1]
        out = torch.empty(n, cs + cu, dtype=skip.dtype, device=skip.device)
        SF._call('stin_concat_unpool_f32', SF._ptr(skip), lds, SF._ptr(coarse), ldc, SF._ptr(pool.trace), n, cs, cu, SF._ptr(out), cs + cu,
                 SF._stream(skip))
        ctx.pool, ctx.cs = pool, cs
        return out

    @staticmethod
    def backward(ctx, g):
        p, cs = ctx.pool, ctx.cs
        return g[:, :cs], SF.segment_sum(g[:, cs:], p.children.rowptr, p.children.col, p.n_coarse, mean=False), None


class _RowStatsNormFn(torch.autograd.Function):
    """y = (x - mean) * rstd over ALL rows (biased variance, eps inside the root), plus the batch mean / biased variance
    as non-differentiable outputs for the running statistics: BatchNorm1d's training-mode normalisation."""

    @staticmethod
    def forward(ctx, x, groups, eps, bn2=None):
        x, _ = SF._mat(x)
        mean, rstd = SF.instance_stats(x, groups) if eps == SF.EPS else SF.colreduce(SF.RED_MOMENTS, x, groups, groups.ptr_sum, eps=eps)
        y = SF.norm_act_res_fwd(x, mean, rstd, groups, res=None, act=False)
        ctx.save_for_backward(x, mean, rstd)
        ctx.groups, ctx.bn2 = groups, bn2
        m1, r1 = mean.view(-1), rstd.view(-1)
        ctx.mark_non_differentiable(m1, r1)
        ctx.set_materialize_grads(False)          # (no zero tensors for the statistics outputs' gradients: two fill launches per call)
        return y, m1, r1

    @staticmethod
    def backward(ctx, g, _gm, _gv):
        x, mean, rstd = ctx.saved_tensors
        if ctx.bn2 is not None:
            _update_running(ctx.bn2, mean.view(-1), rstd.view(-1), x.shape[0])
        return SF.instance_norm_act_bwd(x, g, mean, rstd, ctx.groups, act=False), None, None, None


class _BatchNormActFn(torch.autograd.Function):
    """y = act(gamma * (x - mean) * rstd + beta) with batch statistics over ALL rows, in three passes over x forward
    (moments, normalise) and two backward (the dgamma / dbeta column sums, then dx) - the affine map, the ReLU and both
    of their gradients live inside those kernels instead of separate elementwise / reduce launches."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu, bn2=None):
        x, ldx = SF._mat(x)
        n, c = x.shape
        mean, rstd = SF.colreduce(SF.RED_MOMENTS, x, groups, groups.ptr_sum, eps=eps)
        gamma, beta = gamma.detach().contiguous(), beta.detach().contiguous()
        y = torch.empty(n, c, dtype=x.dtype, device=x.device)
        SF._call('stin_bn_act_fwd_f32', SF._ptr(x), ldx, SF._ptr(mean), SF._ptr(rstd), SF._ptr(gamma), SF._ptr(beta), n, c,
                 int(relu), SF._ptr(y), c, SF._stream(x))
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        ctx.groups, ctx.relu, ctx.bn2 = groups, relu, bn2
        m1, r1 = mean.view(-1), rstd.view(-1)
        ctx.mark_non_differentiable(m1, r1)
        ctx.set_materialize_grads(False)          # (no zero tensors for the statistics outputs' gradients: two fill launches per call)
        return y, m1, r1

    @staticmethod
    def backward(ctx, g, _gm, _gv):
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        x, ldx = SF._mat(x)
        g, ldg = SF._mat(g)
        n, c = x.shape
        if ctx.bn2 is not None:      # the reference recomputes this block's forward here (torch.utils.checkpoint): its BatchNorm
            _update_running(ctx.bn2, mean.view(-1), rstd.view(-1), n)      # running statistics take the batch a second time
        gb = torch.stack([gamma, beta])                                     # coef = [gamma ; beta]
        P, Q = SF.colreduce(SF.RED_DOT_BN_RELU if ctx.relu else SF.RED_DOT_BN, x, ctx.groups, ctx.groups.ptr_sum, gout=g,
                            mean=mean, rstd=rstd, coef=gb)
        dx = torch.empty(n, c, dtype=x.dtype, device=x.device)
        SF._call('stin_bn_act_bwd_f32', SF._ptr(x), ldx, SF._ptr(g), ldg, SF._ptr(mean), SF._ptr(rstd), SF._ptr(gamma),
                 SF._ptr(beta), SF._ptr(P), SF._ptr(Q), 1.0 / n, n, c, int(ctx.relu), SF._ptr(dx), c, SF._stream(x))
        return dx, P.view(-1), Q.view(-1), None, None, None, None


class _BatchNormMeanFn(torch.autograd.Function):
    """scatter_mean(BatchNorm1d(m)) over the in-edges of every vertex with the batch statistics of the E edge rows: the
    affine map commutes with the mean, so forward = statistics over m, segment mean of the RAW rows, then one affine
    pass over N rows (vertices without in-edges stay 0); backward = column sums over N rows + one pass that turns the
    vertex gradient into the edge-row gradient (stin_bn_mean_bwd_f32) - instead of a normalise pass, a gather and three
    more passes over [E, C]."""

    @staticmethod
    def forward(ctx, m, gamma, beta, ei, groups_e, groups_n, eps, bn2=None):
        m, ldm = SF._mat(m)
        e, c = m.shape
        mean, rstd = SF.colreduce(SF.RED_MOMENTS, m, groups_e, groups_e.ptr_sum, eps=eps)
        gamma, beta = gamma.detach().contiguous(), beta.detach().contiguous()
        agg = SF.segment_sum(m, ei.by_dst.rowptr, ei.by_dst.col, ei.n, mean=True)
        out = torch.empty(ei.n, c, dtype=m.dtype, device=m.device)
        SF._call('stin_bn_act_fwd_f32', SF._ptr(agg), c, SF._ptr(mean), SF._ptr(rstd), SF._ptr(gamma), SF._ptr(beta), ei.n, c, 0,
                 SF._ptr(out), c, SF._stream(m))
        out.mul_(ei.has_in)
        ctx.save_for_backward(m, agg, mean, rstd, gamma, beta)
        ctx.ei, ctx.groups_n, ctx.bn2 = ei, groups_n, bn2
        m1, r1 = mean.view(-1), rstd.view(-1)
        ctx.mark_non_differentiable(m1, r1)
        ctx.set_materialize_grads(False)          # (no zero tensors for the statistics outputs' gradients: two fill launches per call)
        return out, m1, r1

    @staticmethod
    def backward(ctx, g, _gm, _gv):
        m, agg, mean, rstd, gamma, beta = ctx.saved_tensors
        ei = ctx.ei
        m, ldm = SF._mat(m)
        e, c = m.shape
        if ctx.bn2 is not None:
            _update_running(ctx.bn2, mean.view(-1), rstd.view(-1), e)
        g = g * ei.has_in                                                 # rows without in-edges produced a constant 0
        P, Q = SF.colreduce(SF.RED_DOT_BN, agg, ctx.groups_n, ctx.groups_n.ptr_sum, gout=g, mean=mean, rstd=rstd,
                            coef=torch.stack([gamma, beta]))
        dm = torch.empty(e, c, dtype=m.dtype, device=m.device)
        SF._call('stin_bn_mean_bwd_f32', SF._ptr(m), ldm, SF._ptr(g), c, SF._ptr(ei.dst32), SF._ptr(ei.by_dst.inv_deg), SF._ptr(mean),
                 SF._ptr(rstd), SF._ptr(gamma), SF._ptr(P), SF._ptr(Q), 1.0 / max(e, 1), e, c, SF._ptr(dm), c, SF._stream(m))
        return dm, P.view(-1), Q.view(-1), None, None, None, None, None


_SINGLE_GROUPS = {}


def _all_rows(n, device):
    """The one-range NormGroups of an [n, C] matrix (cached: it is immutable and costs a fill launch to build)."""
    key = (int(n), str(device))
    g = _SINGLE_GROUPS.get(key)
    if g is None:
        if len(_SINGLE_GROUPS) > 256:
            _SINGLE_GROUPS.clear()
        g = _SINGLE_GROUPS[key] = NormGroups(n, device)
    return g


_NBT_BUMPED = [False]        # SingleConvMeshNet.forward bumps every num_batches_tracked in one multi-tensor launch
# (round 5) ... and the SECOND bump of the blocks the reference recomputes (their BatchNorms run a second forward inside backward)
# is one multi-tensor launch too: forward leaves the counters of those layers here, the first recomputed layer whose backward
# runs bumps them all (a full backward visits every one of them; 12 single-element launches per step before)
_NBT_SECOND = {'pending': None}


def _update_running(bn, mean, rstd, n):
    """nn.BatchNorm1d's running statistics from the batch (mean, rstd) of n rows: one HIP launch (+ the batch counter)."""
    if not (bn.training and bn.track_running_stats):
        return
    with torch.no_grad():
        pend = _NBT_SECOND['pending']
        in_bwd = not torch.is_grad_enabled()             # (autograd runs backward functions with gradients off)
        if in_bwd and pend is not None and not _NBT_BUMPED[0] and any(t is bn.num_batches_tracked for t in pend):
            torch._foreach_add_(pend, 1)                 # every recomputed layer's counter at once (backward, second pass)
            _NBT_SECOND['pending'] = ()
        elif in_bwd and pend is not None and len(pend) == 0 and not _NBT_BUMPED[0] and getattr(bn, '_stin_second_pass', False):
            pass                                         # (already bumped with the others)
        elif not _NBT_BUMPED[0]:
            bn.num_batches_tracked += 1
        mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        SF._call('stin_bn_running_stats_f32', SF._ptr(mean), SF._ptr(rstd), mean.numel(), float(bn.eps), n / max(n - 1, 1),
                 float(mom), SF._ptr(bn.running_mean), SF._ptr(bn.running_var), SF._stream(mean))


def _second_pass(bn, recomputed):
    """The module whose running statistics backward updates once more: set when the reference wraps the enclosing block in
    torch.utils.checkpoint (its forward, BatchNorm included, runs a second time inside backward)."""
    return bn if (recomputed and bn.training and bn.track_running_stats and torch.is_grad_enabled()) else None


def batch_norm_mean(m, bn, ei, recomputed=False):
    """scatter_mean(bn(m), edge_index[1]) for an affine BatchNorm1d in training mode (running statistics updated as
    nn.BatchNorm1d does), fused; None when this fast path does not apply."""
    e = m.shape[0]
    if not (bn.training and bn.affine and m.dtype == torch.float32 and e > 1):
        return None
    out, mean, rstd = _BatchNormMeanFn.apply(m, bn.weight, bn.bias, ei, _all_rows(e, m.device), _all_rows(ei.n, m.device),
                                             float(bn.eps), _second_pass(bn, recomputed))
    _update_running(bn, mean, rstd, e)
    return out


def batch_norm_rows(x, bn, relu=False, recomputed=False):
    """nn.BatchNorm1d semantics on [rows, C] (training: batch statistics over all rows + running-stat update with the
    unbiased variance; eval: running statistics), statistics by the fp64-accumulating column-reduction kernels;
    `relu=True` applies the following ReLU in the same kernels.  recomputed: see _second_pass."""
    if bn.training or not bn.track_running_stats:
        n = x.shape[0]
        fused = bn.affine and x.dtype == torch.float32 and n > 0
        if fused:
            y, mean, rstd = _BatchNormActFn.apply(x, bn.weight, bn.bias, _all_rows(n, x.device), float(bn.eps), bool(relu),
                                                  _second_pass(bn, recomputed))
        else:
            y, mean, rstd = _RowStatsNormFn.apply(x, _all_rows(n, x.device), float(bn.eps), _second_pass(bn, recomputed))
        _update_running(bn, mean, rstd, n)
        if fused:
            return y
    elif (bn.affine and x.dtype == torch.float32 and x.shape[0] > 0 and not (
            torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad or bn.bias.requires_grad))):
        # inference: one HIP pass with the running statistics (rstd from running_var: a [C] vector op).  Not taken when
        # ANYTHING here can receive a gradient - frozen-statistics BN with a trainable affine keeps torch's expression
        x, ldx = SF._mat(x)
        n, c = x.shape
        y = torch.empty(n, c, dtype=x.dtype, device=x.device)
        rstd = torch.rsqrt(bn.running_var + bn.eps)
        SF._call('stin_bn_act_fwd_f32', SF._ptr(x), ldx, SF._ptr(bn.running_mean), SF._ptr(rstd), SF._ptr(bn.weight.detach()),
                 SF._ptr(bn.bias.detach()), n, c, int(relu), SF._ptr(y), c, SF._stream(x))
        return y
    else:
        y = (x - bn.running_mean) * torch.rsqrt(bn.running_var + bn.eps)
    if bn.affine:
        y = y * bn.weight + bn.bias
    return F.relu(y) if relu else y


USE_FUSED_LAYER = True      # A/B switch: 0 = the per-op autograd path
# BatchNorm1d + ReLU of the E x 2 cout edge rows applied inside the per-edge GEMMs' operand staging (stin_gemm_nt_bn_f32 /
# stin_gemm_tn_bn_f32): the normalised matrix h is never written or read (1.2 GB per level-0 layer); 0 = materialise it (A/B)
BN_IN_GEMM = True
# the column moments of the gather-add output accumulated by the gather-add pass itself (stin_gather_add_rows_stats_f32)
STATS_IN_GATHER = True
# backward: the per-edge input-gradient product with BatchNorm1d + ReLU's backward on its epilogue, run twice (statistics, then the
# finished gradient: stin_gemm_nt_bn_bwd_{stats,apply}_f32) instead of GEMM + column reduction + elementwise pass
BN_BWD_IN_GEMM = True


def _gemm_nt_bn(pre, W, mean, rstd, gamma, beta, precision):
    e, k = pre.shape
    out = torch.empty(e, W.shape[0], dtype=torch.float32, device=pre.device)
    SF._call('stin_gemm_nt_bn_f32', SF._ptr(pre), k, SF._ptr(W), k, SF._ptr(mean), SF._ptr(rstd), SF._ptr(gamma), SF._ptr(beta), e,
             W.shape[0], k, SF._ptr(out), W.shape[0], int(precision), SF._stream(pre), tag=(e, W.shape[0], k))
    return out


def _gemm_tn_bn(G, pre, mean, rstd, gamma, beta, precision):
    lib = _lib.load()
    e, nc = G.shape
    k = pre.shape[1]
    out = torch.empty(nc, k, dtype=torch.float32, device=G.device)
    ws_bytes = lib.stin_gemm_tn_workspace_bytes(e, nc, k, 0)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=G.device)
    SF._call('stin_gemm_tn_bn_f32', SF._ptr(G), nc, SF._ptr(pre), k, SF._ptr(mean), SF._ptr(rstd), SF._ptr(gamma), SF._ptr(beta), e, nc, k,
             SF._ptr(out), k, int(precision), SF._ptr(ws), ws_bytes, SF._stream(G), tag=(e, nc, k))
    return out


class _EdgeConvBNLayerFn(torch.autograd.Function):
    """One EdgeConv(BN) layer of SingleConvMeshNet in training mode - Lin1 (per vertex) -> gather-add -> BatchNorm1d + ReLU over
    the E edge rows -> Lin2 (per edge) -> BatchNorm1d -> mean over the in-edges -> [+ x] -> [ReLU] - as ONE autograd node
    (round 5; reference models/modules/edge_conv_filter.py:34-44, models/singleconvmeshnet.py:37-66).  The same HIP kernels, in the
    same order and with the same arithmetic as the per-op path below (stin_bn_affine_res_fwd_f32 repeats the framework
    expression it replaces), minus the framework's elementwise / copy / stack / accumulate launches: operand packing is one
    kernel, the residual add and the output ReLU ride on the N-row BatchNorm epilogue, the residual's gradient on the
    input-gradient GEMM's epilogue, and the BatchNorm backward of the E x 2 cout rows overwrites its own input gradient."""

    @staticmethod
    def forward(ctx, x, W1, W2, g1, b1, g2, b2, ei, trans_inv, bn1, bn2, residual, relu, recomputed):
        x, ldx = SF._mat(x)
        n, cin = x.shape
        h2, cout, e = W1.shape[0], W2.shape[0], ei.E
        dev, st = x.device, SF._stream(x)
        f32 = dict(dtype=torch.float32, device=dev)
        wcat, wcatT, w2T = torch.empty(2 * h2, cin, **f32), torch.empty(cin, 2 * h2, **f32), torch.empty(h2, cout, **f32)
        gb1, gb2 = torch.empty(2, h2, **f32), torch.empty(2, cout, **f32)
        SF._call('stin_scmn_pack_f32', SF._ptr(W1), SF._ptr(W2), SF._ptr(g1), SF._ptr(b1), SF._ptr(g2), SF._ptr(b2), cin, h2, cout,
                 int(trans_inv), SF._ptr(wcat), SF._ptr(wcatT), SF._ptr(w2T), SF._ptr(gb1), SF._ptr(gb2), st)
        y = SF.gemm_nt(x, wcat, precision=SF.PREC_FWD)                              # [N, 2 h2] = [A | B]
        pre = torch.empty(e, h2, **f32)
        ge, gn = _all_rows(e, dev), _all_rows(n, dev)
        groups = int(_lib.load().stin_gather_add_rows_stats_groups(e, h2)) if STATS_IN_GATHER else 0
        if groups > 0:       # the moments of `pre` ride on the pass that writes it (no second pass over the [E, 2 cout] matrix)
            partial = torch.empty(groups, 2, h2, dtype=torch.float64, device=dev)
            SF._call('stin_gather_add_rows_stats_f32', SF._ptr(y), 2 * h2, SF._ptr(ei.dst32), y.data_ptr() + 4 * h2, 2 * h2,
                     SF._ptr(ei.src32), e, h2, SF._ptr(pre), h2, SF._ptr(partial), partial.numel() * 8, st, tag=(e, h2))
            mean1, rstd1 = SF.moments_final(partial, ge.inv_cnt, eps=float(bn1.eps))
        else:
            SF._call('stin_gather_add_rows_f32', SF._ptr(y), 2 * h2, SF._ptr(ei.dst32), y.data_ptr() + 4 * h2, 2 * h2, SF._ptr(ei.src32),
                     e, h2, SF._ptr(pre), h2, st, tag=(e, h2))
            mean1, rstd1 = SF.colreduce(SF.RED_MOMENTS, pre, ge, ge.ptr_sum, eps=float(bn1.eps))
        h = None
        if BN_IN_GEMM and h2 > 16:
            m = _gemm_nt_bn(pre, W2, mean1, rstd1, gb1[0], gb1[1], SF.PREC_FWD)     # per-EDGE GEMM on relu(bn(pre)), [E, cout]
        else:
            h = torch.empty(e, h2, **f32)
            SF._call('stin_bn_act_fwd_f32', SF._ptr(pre), h2, SF._ptr(mean1), SF._ptr(rstd1), SF._ptr(gb1[0]), SF._ptr(gb1[1]), e, h2, 1,
                     SF._ptr(h), h2, st)
            m = SF.gemm_nt(h, W2, precision=SF.PREC_FWD)                            # per-EDGE GEMM, [E, cout]
        groups2 = int(_lib.load().stin_segment_mean_stats_groups(n, cout)) if STATS_IN_GATHER else 0
        if groups2 > 0:      # the moments of the E edge rows m ride on the pass that averages them per target vertex
            partial2 = torch.empty(groups2, 2, cout, dtype=torch.float64, device=dev)
            agg = torch.empty(n, cout, **f32)
            SF._call('stin_segment_mean_stats_f32', SF._ptr(m), cout, SF._ptr(ei.by_dst.rowptr), SF._ptr(ei.by_dst.col), n, cout, SF._ptr(agg),
                     cout, SF._ptr(partial2), partial2.numel() * 8, st, tag=(e, n, cout))
            mean2, rstd2 = SF.moments_final(partial2, ge.inv_cnt, eps=float(bn2.eps))
        else:
            mean2, rstd2 = SF.colreduce(SF.RED_MOMENTS, m, ge, ge.ptr_sum, eps=float(bn2.eps))
            agg = SF.segment_sum(m, ei.by_dst.rowptr, ei.by_dst.col, n, mean=True)
        out = torch.empty(n, cout, **f32)
        SF._call('stin_bn_affine_res_fwd_f32', SF._ptr(agg), cout, SF._ptr(mean2), SF._ptr(rstd2), SF._ptr(gb2[0]), SF._ptr(gb2[1]),
                 SF._ptr(ei.by_dst.rowptr), SF._ptr(x) if residual else None, ldx, n, cout, int(relu), SF._ptr(out), cout, st)
        _update_running(bn1, mean1.view(-1), rstd1.view(-1), e)
        _update_running(bn2, mean2.view(-1), rstd2.view(-1), e)
        ctx.save_for_backward(x, W1, out)
        ctx.bufs = (pre, h, m, agg, mean1, rstd1, mean2, rstd2, wcatT, w2T, gb1, gb2)
        ctx.meta = (ei, bool(trans_inv), bn1, bn2, bool(residual), bool(relu), bool(recomputed))
        return out

    @staticmethod
    def backward(ctx, g):
        x, W1, out = ctx.saved_tensors
        pre, h, m, agg, mean1, rstd1, mean2, rstd2, wcatT, w2T, gb1, gb2 = ctx.bufs
        ei, trans_inv, bn1, bn2, residual, relu, recomputed = ctx.meta
        ctx.bufs = None
        x, ldx = SF._mat(x)
        g, ldg = SF._mat(g)
        n, cin = x.shape
        e, h2 = pre.shape
        cout = m.shape[1]
        dev, st = x.device, SF._stream(x)
        f32 = dict(dtype=torch.float32, device=dev)
        if recomputed:       # the reference recomputes this block's forward here (torch.utils.checkpoint): its BatchNorm running
            _update_running(bn2, mean2.view(-1), rstd2.view(-1), e)      # statistics take the batch a second time (_second_pass)
            _update_running(bn1, mean1.view(-1), rstd1.view(-1), e)
        g_eff = torch.empty(n, cout, **f32) if residual else None
        g_in = torch.empty(n, cout, **f32)
        SF._call('stin_relu_mask_bwd_f32', SF._ptr(g), ldg, SF._ptr(out), cout, SF._ptr(ei.by_dst.rowptr), n, cout, int(relu),
                 SF._ptr(g_eff), SF._ptr(g_in), st)
        ge, gn = _all_rows(e, dev), _all_rows(n, dev)
        P2, Q2 = SF.colreduce(SF.RED_DOT_BN, agg, gn, gn.ptr_sum, gout=g_in, mean=mean2, rstd=rstd2, coef=gb2)
        dm = torch.empty(e, cout, **f32)
        SF._call('stin_bn_mean_bwd_f32', SF._ptr(m), cout, SF._ptr(g_in), cout, SF._ptr(ei.dst32), SF._ptr(ei.by_dst.inv_deg),
                 SF._ptr(mean2), SF._ptr(rstd2), SF._ptr(gb2[0]), SF._ptr(P2), SF._ptr(Q2), 1.0 / max(e, 1), e, cout, SF._ptr(dm), cout, st)
        lib = _lib.load()
        two_pass = int(lib.stin_gemm_nt_bn_bwd_groups(e, h2, cout, int(SF.PREC_BWD))) if BN_BWD_IN_GEMM else 0
        if two_pass > 0:
            # dh = dm W2 is only the output gradient of BN1 + ReLU: the product runs twice (column sums on its epilogue, then the
            # finished edge-row gradient) instead of being stored, reduced and rewritten - 3.1 of 4 GB per level-0 layer
            partial = torch.empty(two_pass, 2, h2, dtype=torch.float64, device=dev)
            pq1 = torch.empty(2, h2, **f32)
            bn_args = (SF._ptr(dm), cout, SF._ptr(w2T), cout, SF._ptr(pre), h2, SF._ptr(mean1), SF._ptr(rstd1), SF._ptr(gb1[0]), SF._ptr(gb1[1]))
            SF._call('stin_gemm_nt_bn_bwd_stats_f32', *bn_args, e, h2, cout, int(SF.PREC_BWD), SF._ptr(partial), partial.numel() * 8,
                     SF._ptr(pq1), st, tag=(e, h2, cout))
            P1, Q1 = pq1[0], pq1[1]
            dh = torch.empty(e, h2, **f32)
            SF._call('stin_gemm_nt_bn_bwd_apply_f32', *bn_args, SF._ptr(pq1), 1.0 / e, e, h2, cout, SF._ptr(dh), h2, int(SF.PREC_BWD), st,
                     tag=(e, h2, cout))
        else:
            dh = SF.gemm_nt(dm, w2T, precision=SF.PREC_BWD)                         # [E, h2]
        if h is None:
            dW2 = _gemm_tn_bn(dm, pre, mean1, rstd1, gb1[0], gb1[1], SF.PREC_BWD)   # [cout, h2] from the pre-norm rows
        else:
            dW2 = SF.gemm_tn(dm, h, ones_column=False, precision=SF.PREC_BWD)       # [cout, h2]
        del h, m, agg
        if two_pass <= 0:
            P1, Q1 = SF.colreduce(SF.RED_DOT_BN_RELU, pre, ge, ge.ptr_sum, gout=dh, mean=mean1, rstd=rstd1, coef=gb1)
            SF._call('stin_bn_act_bwd_f32', SF._ptr(pre), h2, SF._ptr(dh), h2, SF._ptr(mean1), SF._ptr(rstd1), SF._ptr(gb1[0]),
                     SF._ptr(gb1[1]), SF._ptr(P1), SF._ptr(Q1), 1.0 / e, e, h2, 1, SF._ptr(dh), h2, st)   # (element-wise: in place over dh)
        dy = torch.empty(n, 2 * h2, **f32)
        for off, csr in ((0, ei.by_dst), (h2, ei.by_src)):
            SF._call('stin_segment_sum_f32', SF._ptr(dh), h2, SF._ptr(csr.rowptr), SF._ptr(csr.col), n, h2, 0, dy.data_ptr() + 4 * off,
                     2 * h2, st)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = SF.gemm_nt(dy, wcatT, precision=SF.PREC_BWD, residual=g_eff)       # (+ the residual branch's gradient)
        dwcat = SF.gemm_tn(dy, x, ones_column=False, precision=SF.PREC_BWD)         # [2 h2, cin]
        dW1 = torch.empty(W1.shape, **f32)
        SF._call('stin_scmn_unpack_f32', SF._ptr(dwcat), cin, h2, int(trans_inv), SF._ptr(dW1), st)
        return dx, dW1, dW2, P1.view(-1), Q1.view(-1), P2.view(-1), Q2.view(-1), None, None, None, None, None, None, None


class EdgeConvBN(nn.Module):
    """EdgeConv(aggr='mean') whose MLP is Lin(no bias) - BatchNorm1d - ReLU - Lin(no bias) - BatchNorm1d
    (edge_conv_filter.py:34-44); `nn` holds the parameters under the reference's names and is never called itself."""

    def __init__(self, cin, cout, trans_inv=False):
        super().__init__()
        self.trans_inv = trans_inv
        self.recomputed = False      # True where the reference checkpoints the enclosing block (running statistics: _second_pass)
        self.nn = nn.Sequential(nn.Linear(cin if trans_inv else 2 * cin, 2 * cout, bias=False), nn.BatchNorm1d(2 * cout),
                                nn.ReLU(), nn.Linear(2 * cout, cout, bias=False), nn.BatchNorm1d(cout))

    def forward(self, x, ei, residual=False, relu=False):
        """residual / relu: the enclosing ResBlock's `x + f(x)` and ReLU, applied here so that the fused layer can carry them."""
        ei = _as_edge_index(ei, x.shape[0])
        lin1, bn1, lin2, bn2 = self.nn[0], self.nn[1], self.nn[3], self.nn[4]
        h2, cin = lin1.weight.shape[0], x.shape[1]
        if (USE_FUSED_LAYER and x.is_cuda and x.dtype == torch.float32 and ei.E > 1 and all(
                b.training and b.affine and b.track_running_stats for b in (bn1, bn2)) and torch.is_grad_enabled() and
                (not residual or lin2.weight.shape[0] == cin)):
            return _EdgeConvBNLayerFn.apply(x, lin1.weight, lin2.weight, bn1.weight, bn1.bias, bn2.weight, bn2.bias, ei, self.trans_inv,
                                            bn1, bn2, residual, relu, self.recomputed)
        out = self._forward_per_op(x, ei, lin1, bn1, lin2, bn2, h2, cin)
        if residual:
            out = x + out
        return F.relu(out) if relu else out

    def _forward_per_op(self, x, ei, lin1, bn1, lin2, bn2, h2, cin):
        if self.trans_inv:
            wcat = torch.cat([-lin1.weight, lin1.weight], dim=0)
        else:
            wa, wb = lin1.weight[:, :cin], lin1.weight[:, cin:]
            wcat = torch.cat([wa - wb, wb], dim=0)
        y = SF.linear(x, wcat)                                             # [N, 2 * h2] = [A | B], per-VERTEX GEMM
        if y.dtype == torch.float32:
            pre = _GatherAddFn.apply(y, ei)               # [E, 2 cout] = A[dst] + B[src]
        else:
            pre = _GatherRowsFn.apply(y[:, :h2], ei.dst32, ei.by_dst) + _GatherRowsFn.apply(y[:, h2:], ei.src32, ei.by_src)
        h = batch_norm_rows(pre, bn1, relu=True, recomputed=self.recomputed)   # [E, 2 cout]
        m = SF.linear(h, lin2.weight)                                      # per-EDGE GEMM, [E, cout]
        out = batch_norm_mean(m, bn2, ei, recomputed=self.recomputed)      # BN2 + mean over the in-edges, fused
        if out is None:
            out = _ScatterMeanFn.apply(batch_norm_rows(m, bn2, recomputed=self.recomputed), ei)
        return out

    def __repr__(self):
        return '{}(nn={}, aggr=mean)'.format('EdgeConvTransInv' if self.trans_inv else 'EdgeConv', self.nn)


class ResBlock(nn.Module):
    def __init__(self, filters):
        super().__init__()
        self.filters = nn.ModuleList(filters)

    def forward(self, x, ei):
        x = self.filters[0](x, ei, relu=True)                         # relu(f0(x))
        for f in list(self.filters)[1:]:
            x = f(x, ei, residual=True, relu=True)                    # relu(x + f(x))
        return x


class SingleConvMeshNet(nn.Module):
    """Same constructor as the reference (models/singleconvmeshnet.py:13-14); forward(sample) -> [N0, num_classes]."""

    def __init__(self, feature_number, num_propagation_steps, filter_sizes, num_classes=3, pooling_method='mean', aggr='mean'):
        super().__init__()
        if aggr != 'mean':
            raise NotImplementedError('aggr=%r: the reference only ever constructs the mean aggregation' % aggr)
        self._pooling_method = pooling_method
        self._graph_levels = len(filter_sizes)
        left, right = [], []
        cur = feature_number
        for level, fs in enumerate(filter_sizes):
            first = EdgeConvBN(cur, fs, trans_inv=(level == 0 and level < len(filter_sizes) - 1))
            left.append(ResBlock([first] + [EdgeConvBN(fs, fs) for _ in range(num_propagation_steps - 1)]))
            if level < len(filter_sizes) - 1:
                right.append(ResBlock([EdgeConvBN(fs + filter_sizes[level + 1], fs)] +
                                      [EdgeConvBN(fs, fs) for _ in range(num_propagation_steps - 1)]))
                cur = fs
        self.left_geo_cnns = nn.ModuleList(left)
        self.right_geo_cnns = nn.ModuleList(right)
        # the reference runs left blocks of levels >= 1 and every right block but the last-executed (level 0) one under
        # torch.utils.checkpoint (models/singleconvmeshnet.py:124-126, :139-144): a second forward inside backward, which
        # numerically only touches the BatchNorm running statistics (two updates per training step)
        L = len(filter_sizes)
        for blk in list(self.left_geo_cnns[1:]) + [self.right_geo_cnns[-level] for level in range(1, L - 1)]:
            for f in blk.filters:
                f.recomputed = True
        f0 = filter_sizes[0]
        self.final_convs = nn.ModuleList([nn.Sequential(nn.Linear(f0, f0 // 2), nn.BatchNorm1d(f0 // 2), nn.ReLU(),
                                                        nn.Linear(f0 // 2, num_classes))])

    # ---- per-sample index structures, cached on the sample ----------------------------------------------------
    def _indices(self, sample):
        cache = getattr(sample, '_scmn_cache', None)
        if cache is None:
            x = sample.x
            assert x.is_cuda, 'the HIP path needs the sample on the GPU (sample.to("cuda"))'
            L = self._graph_levels
            bad = torch.zeros(1, dtype=torch.int32, device=x.device)
            sizes = [x.shape[0]]
            pools = {}
            for level in range(1, L):
                trace = sample['hierarchy_trace_index_%d' % level]
                n_coarse = int(trace.max()) + 1                       # scatter_* without dim_size (:111-113)
                pools[level] = PoolMap(trace, sizes[-1], n_coarse, bad)
                sizes.append(n_coarse)
            edges = {0: _EdgeIndex(sample.edge_index, sizes[0], bad)}
            for level in range(1, L):
                edges[level] = _EdgeIndex(sample['hierarchy_edge_index_%d' % level], sizes[level], bad)
            if int(bad.item()) != 0:
                raise IndexError('edge / trace index out of range')
            cache = (edges, pools)
            try:
                object.__setattr__(sample, '_scmn_cache', cache)
            except Exception:
                pass
        return cache

    def _pooling(self, x, pool):
        if self._pooling_method == 'mean':
            return SF.PoolMeanFn.apply(x, pool)
        if self._pooling_method == 'max':
            return SF.PoolMaxFn.apply(x, pool)
        raise ValueError('Unkown pooling type {}'.format(self._pooling_method))

    def forward(self, sample):
        bumped = False
        if self.training:                                    # every BatchNorm's batch counter in ONE multi-tensor launch
            nbt = [m.num_batches_tracked for m in self.modules()
                   if isinstance(m, nn.BatchNorm1d) and m.track_running_stats and m.training]
            if nbt:
                with torch.no_grad():
                    torch._foreach_add_(nbt, 1)
                bumped = True
        _NBT_BUMPED[0] = bumped
        second = []
        if bumped and torch.is_grad_enabled():
            for f in (f for m in self.modules() if isinstance(m, EdgeConvBN) and m.recomputed for f in (m.nn[1], m.nn[4])):
                f._stin_second_pass = True
                second.append(f.num_batches_tracked)
        _NBT_SECOND['pending'] = second if second else None
        try:
            return self._forward(sample)
        finally:
            _NBT_BUMPED[0] = False

    def _forward(self, sample):
        edges, pools = self._indices(sample)
        L = self._graph_levels
        levels = [self.left_geo_cnns[0](sample.x, edges[0])]
        for level in range(1, L):
            levels.append(self.left_geo_cnns[level](self._pooling(levels[-1], pools[level]), edges[level]))
        current = levels[-1]
        for level in range(1, L):
            skip, pool = levels[-(level + 1)], pools[L - level]
            if USE_FUSED_LAYER and skip.is_cuda and skip.dtype == torch.float32 and current.dtype == torch.float32:
                fused = _SkipUnpoolConcatFn.apply(skip, current, pool)              # cat((skip, current[trace]), -1), one launch
            else:
                fused = torch.cat((skip, SF.UnpoolFn.apply(current, pool)), -1)
            current = self.right_geo_cnns[-level](fused, edges[L - level - 1])
        lin1, bn, lin2 = self.final_convs[0][0], self.final_convs[0][1], self.final_convs[0][3]
        out = batch_norm_rows(SF.linear(current, lin1.weight, lin1.bias), bn, relu=True)
        return SF.linear(out, lin2.weight, lin2.bias)
