"""torch.autograd bindings of the HIP kernels (libstin_hip.so) - the operator layer
the nn.Modules in surfacetextureinpaintingnet.py are written on.

Every function here runs on GPU tensors only and calls through the C ABI
(include/stin_hip.h); there is no eager/CPU fallback.  The dense per-vertex GEMMs are
the hand-written MFMA kernels too (gemm_nt / gemm_tn); the library-GEMM A/B aid of round 1 lives in
profiles/gemm_bench.py, outside the package.
"""
import os

import torch

from . import _lib
from .plan import _ptr, _stream

EPS = 1e-5
# save the edge-stage ReLU decisions as a bit-mask in forward (STIN_EDGE_MASK=0: recompute them in backward)
USE_EDGE_MASK = os.environ.get('STIN_EDGE_MASK', '1') != '0'
RED_SUM, RED_CSQ, RED_DOT_ELU, RED_COEF_XC, RED_MOMENTS, RED_DOT_BN, RED_DOT_BN_RELU = 0, 1, 2, 3, 4, 5, 6
POST_NONE, POST_SCALE, POST_RSTD, POST_NORM_COEF = 0, 1, 2, 3


def _mat(t):
    """2-D fp32 (or bf16-storage) GPU tensor with unit inner stride -> (tensor, ld)."""
    if not (t.dim() == 2 and t.dtype in (torch.float32, torch.bfloat16) and t.is_cuda):
        raise TypeError('the STINet HIP path takes 2-D float32 (or bfloat16-storage) CUDA tensors (no CPU / eager '
                        'fallback exists); got shape %s dtype %s device %s' % (tuple(t.shape), t.dtype, t.device))
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0)))


def _sfx(t):
    """C-ABI suffix of the kernel family for a tensor's storage type."""
    return '_bf16' if t.dtype == torch.bfloat16 else '_f32'


def _same(ref, *ts):
    """The optional operands of a kernel must share the storage type of its main operand."""
    for t in ts:
        if t is not None and t.dtype != ref.dtype:
            raise TypeError('mixed storage types in one kernel call: %s vs %s' % (ref.dtype, t.dtype))


class KernelTimer:
    """Optional HIP-event bracket around chosen C-ABI launches (bench.py's live roofline numbers).
    Events are recorded on the stream the kernel is enqueued on (torch's current stream)."""
    enabled = False
    names = ()
    records = []          # (name, tag, start_event, end_event)
    max_records = 400     # thousands of outstanding timing events slow the HIP runtime down: sample, do not flood
    in_net = False        # round 3: edge-stage brackets INSIDE the whole-network call (stin_net_op_t::ev_edge0 / 1) - the step stays
                          # on the fast path (one C call per direction); other names still need the per-kernel path

    @classmethod
    def start(cls, names, max_records=400, in_net=False):
        cls.enabled, cls.names, cls.records, cls.max_records = True, tuple(names), [], max_records
        cls.in_net = bool(in_net)

    @classmethod
    def per_kernel_path(cls):
        """True while brackets are on that only the per-kernel host path can place."""
        return cls.enabled and not cls.in_net

    @classmethod
    def edge_events(cls, name, tag):
        """-> (ev0, ev1) raw hipEvent_t handles bracketing one edge-stage launch inside stin_net_fwd / _bwd, or (0, 0)."""
        if not (cls.enabled and cls.in_net and name in cls.names) or len(cls.records) >= cls.max_records:
            return 0, 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()                                  # (creates the handles; the call re-records them around the kernel)
        b.record()
        cls.records.append((name, tag, a, b))
        return a.cuda_event, b.cuda_event

    @classmethod
    def stop(cls):
        cls.enabled = False
        cls.in_net = False
        torch.cuda.synchronize()
        out = {}
        for name, tag, a, b in cls.records:
            out.setdefault((name, tag), []).append(a.elapsed_time(b) * 1e-3)
        cls.records = []
        return out


def _call(name, *args, tag=None):
    if KernelTimer.enabled and name in KernelTimer.names:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(getattr(_lib.load(), name)(*args), name)
        b.record()
        KernelTimer.records.append((name, tag, a, b))
        if len(KernelTimer.records) >= KernelTimer.max_records:
            KernelTimer.enabled = False          # the rest of the timed region runs un-instrumented
        return
    _lib.check(getattr(_lib.load(), name)(*args), name)


# ------------------------------------------------------------------ raw kernel wrappers
def edge_relu_mean_fwd(A, B, csr, out, indicator=False, mask=None):
    A, lda = _mat(A)
    B, ldb = _mat(B)
    _same(A, B, out)
    H = A.shape[1]
    _call('stin_edge_relu_mean_fwd' + _sfx(A), _ptr(A), lda, _ptr(B), ldb, _ptr(csr.rowptr), _ptr(csr.col), A.shape[0], H,
          _ptr(out), out.stride(0), int(indicator), _ptr(mask), _stream(A), tag=(A.shape[0], csr.n_entries, H))
    return out


def edge_mask_supported(H):
    """The saved-ReLU-mask path stores wave-ballot words of full rows: H = 128, 256, 512, 1024 or 2048."""
    return H in (128, 256, 512, 1024, 2048)


def edge_relu_mean_bwd_dst_mask(G, mask, csr, dA):
    G, ldg = _mat(G)
    _same(G, dA)
    _call('stin_edge_relu_mean_bwd_dst_mask' + _sfx(G), _ptr(G), ldg, _ptr(mask), _ptr(csr.rowptr), G.shape[0], G.shape[1],
          _ptr(dA), dA.stride(0), _stream(G), tag=(G.shape[0], csr.n_entries, G.shape[1]))
    return dA


def edge_relu_mean_bwd_src_mask(G, mask, edges, dB):
    G, ldg = _mat(G)
    cs = edges.by_src
    _same(G, dB)
    _call('stin_edge_relu_mean_bwd_src_mask' + _sfx(G), _ptr(G), ldg, _ptr(edges.w_src), _ptr(mask), _ptr(cs.rowptr),
          _ptr(cs.col), _ptr(edges.xslot), G.shape[0], G.shape[1], _ptr(dB), dB.stride(0), _stream(G),
          tag=(G.shape[0], cs.n_entries, G.shape[1]))
    return dB


def edge_relu_mean_bwd_mask(G, mask, edges, dA, dB, copy_src=None, copy_dst=None):
    """dA and dB of the mask backward in one launch (bit-identical to the two separate kernels); optionally
    copy_dst[:, :] = copy_src (row views of the same height) in the same launch."""
    G, ldg = _mat(G)
    cs = edges.by_src
    _same(G, dA)
    _same(G, dB)
    cp = (None, 0, None, 0, 0)
    if copy_src is not None:
        _same(G, copy_src, copy_dst)
        cp = (_ptr(copy_src), copy_src.stride(0), _ptr(copy_dst), copy_dst.stride(0), copy_src.shape[1])
    _call('stin_edge_relu_mean_bwd_mask' + _sfx(G), _ptr(G), ldg, _ptr(mask), _ptr(edges.by_dst.rowptr), _ptr(edges.w_src),
          _ptr(cs.rowptr), _ptr(cs.col), _ptr(edges.xslot), G.shape[0], G.shape[1], _ptr(dA), dA.stride(0), _ptr(dB),
          dB.stride(0), *cp, _stream(G), tag=(G.shape[0], cs.n_entries, G.shape[1]))
    return dA, dB


def edge_relu_mean_fwd_ti(b1, B, csr, out, indicator=False, mask=None):
    """h = mean_j ReLU(A_i + B_j) with A_i = fl(b1 - B_i) formed per row (translation-invariant filter, compact layout): the same
    bits as edge_relu_mean_fwd given that A.  b1 [H] or None.  fp32 rows, saved-mask widths."""
    B, ldb = _mat(B)
    _same(B, out)
    H = B.shape[1]
    _call('stin_edge_relu_mean_fwd_ti_f32', _ptr(b1), _ptr(B), ldb, _ptr(csr.rowptr), _ptr(csr.col), B.shape[0], H,
          _ptr(out), out.stride(0), int(indicator), _ptr(mask), _stream(B), tag=(B.shape[0], csr.n_entries, H))
    return out


def edge_relu_mean_bwd_mask_ti(G, mask, edges, D, copy_src=None, copy_dst=None):
    """D = dB - dA of the mask backward in one row of H columns (compact trans-inv layout) -> db1 = sum_i dA_i [H] (the kernel's
    per-workgroup column partials, folded in a fixed order)."""
    G, ldg = _mat(G)
    cs = edges.by_src
    _same(G, D)
    N, H = G.shape
    rows = int(_lib.load().stin_edge_bwd_ti_colsum_rows(N, H))
    colsum = torch.empty(max(rows, 1), H, dtype=torch.float32, device=G.device)
    cp = (None, 0, None, 0, 0)
    if copy_src is not None:
        _same(G, copy_src, copy_dst)
        cp = (_ptr(copy_src), copy_src.stride(0), _ptr(copy_dst), copy_dst.stride(0), copy_src.shape[1])
    _call('stin_edge_relu_mean_bwd_mask_ti_f32', _ptr(G), ldg, _ptr(mask), _ptr(edges.by_dst.rowptr), _ptr(edges.w_src),
          _ptr(cs.rowptr), _ptr(cs.col), _ptr(edges.xslot), N, H, _ptr(D), D.stride(0), *cp, _ptr(colsum), rows, _stream(G),
          tag=(N, cs.n_entries, H))
    db1 = torch.empty(H, dtype=torch.float32, device=G.device)
    _call('stin_edge_bwd_ti_colsum_fold_f32', _ptr(colsum), rows if N > 0 else 0, H, _ptr(db1), _stream(G))
    return db1


def edge_relu_mean_bwd_dst(A, B, G, csr, dA):
    A, lda = _mat(A)
    B, ldb = _mat(B)
    G, ldg = _mat(G)
    _call('stin_edge_relu_mean_bwd_dst_f32', _ptr(A), lda, _ptr(B), ldb, _ptr(G), ldg, _ptr(csr.rowptr), _ptr(csr.col),
          A.shape[0], A.shape[1], _ptr(dA), dA.stride(0), _stream(A), tag=(A.shape[0], csr.n_entries, A.shape[1]))
    return dA


def edge_relu_mean_bwd_src(A, B, G, inv_deg, csr_src, dB):
    A, lda = _mat(A)
    B, ldb = _mat(B)
    G, ldg = _mat(G)
    _call('stin_edge_relu_mean_bwd_src_f32', _ptr(A), lda, _ptr(B), ldb, _ptr(G), ldg, _ptr(inv_deg),
          _ptr(csr_src.rowptr), _ptr(csr_src.col), A.shape[0], A.shape[1], _ptr(dB), dB.stride(0), _stream(A),
          tag=(A.shape[0], csr_src.n_entries, A.shape[1]))
    return dB


def segment_sum(src, rowptr, col, n_rows, mean=False):
    src, ld = _mat(src)
    out = torch.empty(n_rows, src.shape[1], dtype=src.dtype, device=src.device)
    # (STIN_SEG_NONTEMPORAL = 2: a source that cannot be Infinity-Cache resident is read with non-temporal loads)
    flags = int(mean) | (2 if (src.dtype == torch.float32 and src.shape[0] * ld * 4 > (256 << 20)) else 0)
    _call('stin_segment_sum' + _sfx(src), _ptr(src), ld, _ptr(rowptr), _ptr(col), n_rows, src.shape[1], flags, _ptr(out),
          out.stride(0) if n_rows > 1 else src.shape[1], _stream(src), tag=(int(src.shape[0]), int(n_rows), int(src.shape[1])))
    return out


def gather_rows(src, idx, row_scale=None):
    src, ld = _mat(src)
    n = idx.numel()
    out = torch.empty(n, src.shape[1], dtype=src.dtype, device=src.device)
    _call('stin_gather_rows' + _sfx(src), _ptr(src), ld, _ptr(idx), _ptr(row_scale), n, src.shape[1], _ptr(out), src.shape[1],
          _stream(src))
    return out


def batch_pool(batch, pool):
    out = torch.empty(pool.n_coarse, dtype=torch.int64, device=batch.device)
    _call('stin_batch_pool_i64', _ptr(batch), _ptr(pool.children.rowptr), _ptr(pool.children.col), pool.n_coarse, _ptr(out),
          _stream(batch))
    return out


def batch_unpool(batch, pool):
    out = torch.empty(pool.n_fine, dtype=torch.int64, device=batch.device)
    _call('stin_gather_i64', _ptr(batch), _ptr(pool.trace), pool.n_fine, _ptr(out), _stream(batch))
    return out


def colreduce(mode, x, groups, ptr, *, gout=None, mean=None, rstd=None, coef=None, post=POST_NONE, eps=EPS,
              use_sid=False):
    """Column sums of `x` rows over the ranges `ptr` (None: all rows) -> [B, C] (two for DOT_ELU)."""
    lib = _lib.load()
    x, ldx = _mat(x)
    N, C = x.shape
    B = groups.B
    out0 = torch.empty(B, C, dtype=torch.float32, device=x.device)
    out1 = torch.empty(B, C, dtype=torch.float32, device=x.device) if mode in (RED_DOT_ELU, RED_MOMENTS, RED_DOT_BN, RED_DOT_BN_RELU) else None
    ldg = 0
    if gout is not None:
        gout, ldg = _mat(gout)
        _same(x, gout)
    ws_bytes = lib.stin_colreduce_workspace_bytes(C, B)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    _call('stin_colreduce' + _sfx(x), mode, _ptr(x), ldx, _ptr(gout), ldg, N, C, _ptr(ptr), B, _ptr(groups.gid),
          _ptr(groups.sid if use_sid else None), _ptr(mean), _ptr(rstd), _ptr(coef), post, _ptr(groups.inv_cnt),
          float(eps), _ptr(out0), _ptr(out1), _ptr(ws), ws_bytes, _stream(x))
    return (out0, out1) if out1 is not None else out0


def colsum(x):
    """[N, C] -> [C] column sums (bias gradients), fixed summation order."""
    class _One:
        B, gid, sid, inv_cnt = 1, None, None, None
    return colreduce(RED_SUM, x, _One, None).view(-1)


def instance_stats(x, groups, eps=EPS):
    """-> (mean, rstd) [B, C]: biased variance, eps inside the sqrt (F.instance_norm /
    FastInstanceNorm semantics, reference models/modules/fastinstancenorm.py:44-98)."""
    if groups.gid is None and x.shape[0] == 1:   # F.instance_norm's own check on the batch=None branch
        raise ValueError('Expected more than 1 spatial element when training, got input size {}'.format(
            torch.Size([1, x.shape[1], 1])))
    if not groups.quirk:
        # one pass: fp64 sum x and sum x^2 over each graph's rows (slices == graphs here)
        return colreduce(RED_MOMENTS, x, groups, groups.ptr_sum, eps=eps)
    # linspace-slice quirk: sums over slices, centring through the graph id -> two passes as the reference does
    mean = colreduce(RED_SUM, x, groups, groups.ptr_sum, post=POST_SCALE)
    rstd = colreduce(RED_CSQ, x, groups, groups.ptr_sum, mean=mean, post=POST_RSTD, eps=eps)
    return mean, rstd


def norm_act_res_fwd(x, mean, rstd, groups, res=None, act=True):
    x, ldx = _mat(x)
    N, C = x.shape
    ldres = 0
    if res is not None:
        res, ldres = _mat(res)
        _same(x, res)
    y = torch.empty(N, C, dtype=x.dtype, device=x.device)
    _call('stin_norm_act_res_fwd' + _sfx(x), _ptr(x), ldx, _ptr(mean), _ptr(rstd), _ptr(groups.gid), _ptr(res), ldres, N, C,
          int(act), _ptr(y), C, _stream(x))
    return y


def instance_norm_act_bwd(x, gout, mean, rstd, groups, act=True, out=None):
    """d/dx of y = act((x - mean[g]) * rstd[g]) with the statistics taken over `groups`
    (incl. the linspace-slice quirk: sums over slices sigma, centring through g)."""
    x, ldx = _mat(x)
    gout, ldg = _mat(gout)
    _same(x, gout, out)
    N, C = x.shape
    if act:
        T1, S0 = colreduce(RED_DOT_ELU, x, groups, groups.ptr_true, gout=gout, mean=mean, rstd=rstd)
    else:
        # without activation dY = gout: sum dY*xc and sum dY via the same kernel with rstd = 0 -> ELU'(0)=1
        T1, S0 = colreduce(RED_DOT_ELU, x, groups, groups.ptr_true, gout=gout, mean=mean, rstd=torch.zeros_like(rstd))
    if not groups.quirk:                                            # sum_g xc = 0 when slices == graphs
        k = torch.empty_like(rstd)
        m = torch.empty_like(rstd)
        _call('stin_norm_bwd_coef_f32', _ptr(T1), _ptr(S0), _ptr(rstd), _ptr(groups.inv_cnt), rstd.shape[0], C, _ptr(k),
              _ptr(m), _stream(x))
    else:
        # k = -(rstd^3) T1 / n, indexed by the SUM slice (sid): the coefficient kernel's k (its m is overwritten below) - the same
        # float operations as the four framework launches this replaces, as stin_edgeconv_block_bwd's slice-quirk path does
        k = torch.empty_like(rstd)
        m = torch.empty_like(rstd)
        _call('stin_norm_bwd_coef_f32', _ptr(T1), _ptr(S0), _ptr(rstd), _ptr(groups.inv_cnt), rstd.shape[0], C, _ptr(k),
              _ptr(m), _stream(x))
        U = colreduce(RED_COEF_XC, x, groups, groups.ptr_true, mean=mean, coef=k, use_sid=True)
        _call('stin_norm_bwd_coef_m_quirk_f32', _ptr(S0), _ptr(U), _ptr(rstd), _ptr(groups.inv_cnt), rstd.shape[0], C, _ptr(m),
              _stream(x))
    dx = out if out is not None else torch.empty(N, C, dtype=x.dtype, device=x.device)
    _call('stin_norm_act_bwd' + _sfx(x), _ptr(x), ldx, _ptr(gout), ldg, _ptr(mean), _ptr(rstd), _ptr(rstd), _ptr(k.contiguous()),
          _ptr(m.contiguous()), _ptr(groups.gid), _ptr(groups.sid), N, C, int(act), _ptr(dx), dx.stride(0), _stream(x))
    return dx


GEMM_F32, GEMM_BF16X3, GEMM_BF16X6, GEMM_F16X3 = 0, 2, 3, 4
PREC_NAMES = {GEMM_F32: 'fp32', GEMM_BF16X3: 'bf16x3', GEMM_BF16X6: 'bf16x6', GEMM_F16X3: 'fp16x3'}
# matrix-core path per GEMM role (env override for A/B experiments: STIN_GEMM_FWD / STIN_GEMM_BWD = 0 | 2 | 3 | 4)
# Defaults: forward GEMMs on the 2-piece fp16 split (3 MFMAs; rms 1-5e-7 against fp64 = the fp32 MFMA chain's
# accuracy on unit-scale activations, 1.7x faster than it; whole-net forward error 5-6e-6 like fp32 and bf16x6,
# where bf16x3 gives 1.2e-4); backward GEMMs on the 2-piece bf16 split (gradients need bf16's exponent range;
# rms 4e-6, far inside the 1e-3 gradient tolerance).
PREC_FWD = int(os.environ.get('STIN_GEMM_FWD', GEMM_F16X3))
PREC_BWD = int(os.environ.get('STIN_GEMM_BWD', GEMM_BF16X3))
GEMM_W_PRESPLIT = 0x100            # nt: the weight operand already holds its two 16-bit pieces (stin_hip.h)
GEMM_W_BF16 = 0x200                # stin_gemm_nt_bf16: the weight operand holds bf16 [Nc][K] (stin_hip.h)
# pre-split operands in MFMA fragment order where the shape allows (stin_hip.h STIN_GEMM_W_FRAG): what the resident-strip NT
# kernel reads.  STIN_NT_STRIP=0 keeps the k-group layout and with it the tiled kernel (A/B aid).
GEMM_W_FRAG = 0x400                 # (a module constant since round 6; tests flip the attribute)
WEIGHT_PRESPLIT = True
# one C call per GraphResnetBlock and direction (stin_edgeconv_block_fwd/bwd enqueue the same kernels in the same order
# as the per-kernel path below): removes ~25 Python-level foreign calls per block.  STIN_BLOCK_CALL=0 = per-kernel path.
USE_BLOCK_CALL = os.environ.get('STIN_BLOCK_CALL', '1') != '0'


def gemm_nt_colstats(A, W, bias, row_mask, precision, n_cols):
    """hE W2^T + masked bias AND the first stage of the column statistics of the result in one launch, or None when this
    shape / precision has no all-columns kernel.  -> (C, partial [groups, 2, n_cols] float64)."""
    A, lda = _mat(A)
    M, K = A.shape
    lib = _lib.load()
    groups = int(lib.stin_gemm_nt_colstats_groups(M, n_cols, K, int(precision))) if A.dtype == torch.float32 else 0
    if groups <= 0 or groups > 1024:
        return None
    C = torch.empty(M, n_cols, dtype=torch.float32, device=A.device)
    partial = torch.empty(groups, 2, n_cols, dtype=torch.float64, device=A.device)
    mk, ldm = (row_mask, row_mask.stride(0)) if row_mask is not None else (None, 0)
    _call('stin_gemm_nt_colstats_f32', _ptr(A), lda, _ptr(W), K, _ptr(bias), _ptr(mk), ldm, None, 0, M, n_cols, K, _ptr(C), n_cols,
          int(precision), _ptr(partial), partial.numel() * 8, _stream(A), tag=(M, n_cols, K))
    return C, partial


def gemm_nt_dotelu(A, W, residual, precision, n_cols, nx, nmean, nrstd):
    """A W^T (+ residual) AND the first stage of the instance-norm + ELU backward statistics of the layer whose output gradient
    the product is (nx / nmean / nrstd: that layer's pre-norm rows and statistics), or None when the shape is not served by the
    panel kernel.  -> (C, partial [groups, 2, n_cols] float64).  (stin_gemm_nt_dotelu_f32; test / probe helper - the block
    backward calls it inside stin_net_bwd.)"""
    A, lda = _mat(A)
    M, K = A.shape
    lib = _lib.load()
    groups = int(lib.stin_gemm_nt_dotelu_groups(M, n_cols, K, int(precision))) if A.dtype == torch.float32 else 0
    if groups <= 0:
        return None
    C = torch.empty(M, n_cols, dtype=torch.float32, device=A.device)
    partial = torch.empty(groups, 2, n_cols, dtype=torch.float64, device=A.device)
    res, ldr = (residual, residual.stride(0)) if residual is not None else (None, 0)
    _call('stin_gemm_nt_dotelu_f32', _ptr(A), lda, _ptr(W), K, None, _ptr(res), ldr, M, n_cols, K, _ptr(C), n_cols, int(precision),
          _ptr(nx), nx.stride(0), _ptr(nmean), _ptr(nrstd), _ptr(partial), partial.numel() * 8, _stream(A), tag=(M, n_cols, K))
    return C, partial


def norm_coef_from_partials(partial, rstd, inv_cnt):
    """Second stage: the instance-norm backward coefficients k, m [1, C] of one graph from gemm_nt_dotelu's partials."""
    groups, _, C = partial.shape
    out = torch.empty(2, 1, C, dtype=torch.float32, device=partial.device)
    _call('stin_norm_coef_from_partials_f32', _ptr(partial), groups, C, _ptr(rstd), _ptr(inv_cnt), _ptr(out[0]), _ptr(out[1]),
          _stream(partial))
    return out[0], out[1]


def moments_final(partial, inv_cnt, eps=EPS):
    """Second stage: mean, rstd [1, C] of one instance (inv_cnt [1] = 1 / rows)."""
    groups, _, C = partial.shape
    out = torch.empty(2, 1, C, dtype=torch.float32, device=partial.device)
    _call('stin_moments_final_f32', _ptr(partial), groups, C, _ptr(inv_cnt), float(eps), _ptr(out[0]), _ptr(out[1]), _stream(partial))
    return out[0], out[1]


def gemm_nt(A, W, bias=None, out=None, row_mask=None, precision=GEMM_F32, residual=None, out_dtype=None):
    """A[M, K] . W[Nc, K]^T + bias * row_mask -> [M, Nc]  (hand-written MFMA kernels).
    row_mask: optional [M] column view (stride = its row pitch) multiplying the bias per row.
    A in bf16 storage: W and bias stay fp32, one bf16 MFMA per k-step (`precision` is ignored), the result is bf16
    unless out_dtype=torch.float32."""
    if A.dtype == torch.bfloat16:
        return _gemm_nt_bf16(A, W, bias, out, row_mask, residual, out_dtype)
    A, lda = _mat(A)
    W, ldw = _mat(W)
    M, K = A.shape
    Nc = W.shape[0]
    assert W.shape[1] == K
    if out is None:
        out = torch.empty(M, Nc, dtype=torch.float32, device=A.device)
    ld_res = 0
    if residual is not None:
        residual, ld_res = _mat(residual)
    _call('stin_gemm_nt_f32', _ptr(A), lda, _ptr(W), ldw, _ptr(bias), _ptr(row_mask),
          row_mask.stride(0) if row_mask is not None else 0, _ptr(residual), ld_res, M, Nc, K, _ptr(out),
          out.stride(0) if M > 1 else max(Nc, out.stride(0)), int(precision), _stream(A), tag=(M, Nc, K))
    return out


def _gemm_nt_bf16(A, W, bias, out, row_mask, residual, out_dtype):
    A, lda = _mat(A)
    W, ldw = _mat(W)
    if W.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError('gemm_nt on bf16 activations takes the fp32 master weights or their bf16 rounding')
    _same(A, row_mask, residual)
    M, K = A.shape
    Nc = W.shape[0]
    assert W.shape[1] == K
    if out is None:
        out = torch.empty(M, Nc, dtype=out_dtype or torch.bfloat16, device=A.device)
    ld_res = 0
    if residual is not None:
        residual, ld_res = _mat(residual)
    _call('stin_gemm_nt_bf16', _ptr(A), lda, _ptr(W), ldw, _ptr(bias), _ptr(row_mask),
          row_mask.stride(0) if row_mask is not None else 0, _ptr(residual), ld_res, M, Nc, K, _ptr(out),
          out.stride(0) if M > 1 else max(Nc, out.stride(0)),
          int(out.dtype == torch.float32) | (GEMM_W_BF16 if W.dtype == torch.bfloat16 else 0), _stream(A), tag=(M, Nc, K))
    return out


def split_weights(W, precision):
    """fp32 W [Nc, K] -> the STIN_GEMM_W_PRESPLIT form for `precision` (GEMM_BF16X3 or GEMM_F16X3, optionally
    | GEMM_W_FRAG: fragment order where the shape allows - pass the same flag to gemm_nt), same shape."""
    W, ldw = _mat(W)
    out = torch.empty_like(W, memory_format=torch.contiguous_format)
    _call('stin_gemm_split_weights_f32', _ptr(W), ldw, W.shape[0], W.shape[1], int(precision), _ptr(out), W.shape[1],
          _stream(W))
    return out


def gemm_tn(G, X, ones_column=False, row_weight=None, precision=GEMM_F32):
    """G[M, Nc]^T . [X[M, K] | w] -> [Nc, K (+1)]  (weight gradient; last column = bias gradient
    sum_m w[m] G[m, :], w = row_weight (an [M] column view) or 1).  bf16-storage operands -> fp32 result."""
    if G.dtype == torch.bfloat16:
        lib = _lib.load()
        G, ldg = _mat(G)
        X, ldx = _mat(X)
        _same(G, X, row_weight)
        M, Nc = G.shape
        K = X.shape[1]
        assert X.shape[0] == M
        Kp = K + int(ones_column)
        out = torch.empty(Nc, Kp, dtype=torch.float32, device=G.device)
        ws_bytes = lib.stin_gemm_tn_workspace_bytes(M, Nc, K, int(ones_column))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=G.device)
        _call('stin_gemm_tn_bf16', _ptr(G), ldg, _ptr(X), ldx, M, Nc, K, int(ones_column), _ptr(row_weight),
              row_weight.stride(0) if row_weight is not None else 0, _ptr(out), Kp, _ptr(ws), ws_bytes, _stream(G),
              tag=(M, Nc, K))
        return out
    lib = _lib.load()
    G, ldg = _mat(G)
    X, ldx = _mat(X)
    M, Nc = G.shape
    K = X.shape[1]
    assert X.shape[0] == M
    Kp = K + int(ones_column)
    out = torch.empty(Nc, Kp, dtype=torch.float32, device=G.device)
    ws_bytes = lib.stin_gemm_tn_workspace_bytes(M, Nc, K, int(ones_column))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=G.device)
    _call('stin_gemm_tn_f32', _ptr(G), ldg, _ptr(X), ldx, M, Nc, K, int(ones_column), _ptr(row_weight),
          row_weight.stride(0) if row_weight is not None else 0, _ptr(out), Kp, int(precision), _ptr(ws), ws_bytes,
          _stream(G),
          tag=(M, Nc, K))
    return out


def gemm_tn_wb(G, X, dW, db, precision=GEMM_F32):
    """dW[Nc, K] = G^T X and db[Nc] = column sums of G, written into the two given fp32 tensors (contiguous; e.g. the views of
    an nn.Linear's gradients in a TrainStep bucket) - no [Nc, K + 1] intermediate and no slicing copies (stin_gemm_tn_wb_*)."""
    lib = _lib.load()
    G, ldg = _mat(G)
    X, ldx = _mat(X)
    _same(G, X)
    M, Nc = G.shape
    K = X.shape[1]
    assert X.shape[0] == M and tuple(dW.shape) == (Nc, K) and db.numel() == Nc and dW.is_contiguous() and db.is_contiguous()
    ws_bytes = lib.stin_gemm_tn_workspace_bytes(M, Nc, K, 1)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=G.device)
    if G.dtype == torch.bfloat16:
        _call('stin_gemm_tn_wb_bf16', _ptr(G), ldg, _ptr(X), ldx, M, Nc, K, None, 0, _ptr(dW), K, _ptr(db), _ptr(ws), ws_bytes,
              _stream(G), tag=(M, Nc, K))
    else:
        _call('stin_gemm_tn_wb_f32', _ptr(G), ldg, _ptr(X), ldx, M, Nc, K, None, 0, _ptr(dW), K, _ptr(db), int(precision), _ptr(ws),
              ws_bytes, _stream(G), tag=(M, Nc, K))


class LinearFn(torch.autograd.Function):
    """y = x W^T + b on the MFMA kernels (the tail Linears and the generic filter paths).  wT: the weight already
    transposed ([K, Nc] fp32, written by the network's PackSet beside the block operands) - without it the backward
    transposes the weight itself.  wT points into a buffer that every PackSet.run() rewrites and is NOT under autograd's version
    tracking, so wT_guard = (pack set, its run count at this forward) is compared again in backward: after another pack run
    (a second forward following an in-place weight change) the backward transposes the SAVED, version-checked weight instead.
    Gradients of weight / bias go straight into an accepting TrainStep bucket."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_fp32=False, precision=None, wT=None, wT_guard=None):
        x, _ = _mat(x)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        ctx.wT = wT
        ctx.wT_guard = wT_guard
        return gemm_nt(x, weight, bias, precision=PREC_FWD if precision is None else precision,
                       out_dtype=torch.float32 if out_fp32 else None)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        if g.dtype != x.dtype:                     # fp32 network output on bf16-storage activations
            g = g.to(x.dtype)
        g, _ = _mat(g)
        fresh = ctx.wT is not None and (ctx.wT_guard is None or getattr(ctx.wT_guard[0], 'runs', None) == ctx.wT_guard[1])
        wT = ctx.wT if fresh else weight.t().contiguous()
        dx = gemm_nt(g, wT, precision=PREC_BWD) if ctx.needs_input_grad[0] else None
        if not ctx.has_bias:
            return dx, gemm_tn(g, x, ones_column=False, precision=PREC_BWD), None, None, None, None, None
        direct = _direct_grad_views(ctx.params) if g.is_cuda else None
        if direct is not None:                     # written where the optimizer reads them: nothing for autograd to copy
            gemm_tn_wb(g, x, direct[0], direct[1], precision=PREC_BWD)
            return dx, None, None, None, None, None, None
        if g.is_cuda:
            dW = torch.empty(weight.shape, dtype=torch.float32, device=g.device)
            db = torch.empty(weight.shape[0], dtype=torch.float32, device=g.device)
            gemm_tn_wb(g, x, dW, db, precision=PREC_BWD)
            return dx, dW, db, None, None, None, None
        dwb = gemm_tn(g, x, ones_column=True, precision=PREC_BWD)
        return dx, dwb[:, :-1].contiguous(), dwb[:, -1].contiguous(), None, None, None, None


def linear_tanh_eligible(x, weight, bias):
    """The one-launch last layer (stin_linear_tanh_*): at most 4 output channels from K <= 256 channels, 16-byte rows."""
    Nc, K = weight.shape
    return (USE_TAIL_KERNEL and x.is_cuda and x.dim() == 2 and x.dtype in (torch.float32, torch.bfloat16) and x.stride(1) == 1 and
            x.stride(0) % 4 == 0 and K % 4 == 0 and K <= 256 and 1 <= Nc <= 4 and Nc * K + Nc <= 1024 and
            weight.dtype == torch.float32 and weight.is_contiguous() and (bias is None or bias.is_contiguous()) and
            x.data_ptr() % (16 if x.dtype == torch.float32 else 8) == 0)


USE_TAIL_KERNEL = True


class LinearTanhFn(torch.autograd.Function):
    """tanh(x W^T + b) -> fp32 [N, Nc], the network's last layer (reference models/surfacetextureinpaintingnet.py:470-471), one
    launch per direction (csrc/stin_tail.hip).  Gradients of W / b go straight into an accepting TrainStep bucket."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        N, K = x.shape
        Nc = weight.shape[0]
        y = torch.empty(N, Nc, dtype=torch.float32, device=x.device)
        sfx = '_bf16' if x.dtype == torch.bfloat16 else '_f32'
        _call('stin_linear_tanh_fwd' + sfx, _ptr(x), x.stride(0), _ptr(weight), _ptr(bias), N, K, Nc, _ptr(y), _stream(x), tag=(N, Nc, K))
        ctx.save_for_backward(x, weight, y)
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        lib = _lib.load()
        N, K = x.shape
        Nc = weight.shape[0]
        g = g.contiguous()
        if g.dtype != torch.float32:
            g = g.float()
        dev = x.device
        dx = torch.empty(N, K, dtype=x.dtype, device=dev) if ctx.needs_input_grad[0] else None
        direct = _direct_grad_views(ctx.params)
        if direct is not None:
            dW, db = direct
        else:
            dW = torch.empty(Nc, K, dtype=torch.float32, device=dev)
            db = torch.empty(Nc, dtype=torch.float32, device=dev) if ctx.params[1] is not None else None
        ws_bytes = lib.stin_linear_tanh_bwd_workspace_bytes(N, K, Nc)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        sfx = '_bf16' if x.dtype == torch.bfloat16 else '_f32'
        _call('stin_linear_tanh_bwd' + sfx, _ptr(g), _ptr(y), _ptr(x), x.stride(0), _ptr(weight), N, K, Nc, _ptr(dx), K, _ptr(dW),
              _ptr(db), _ptr(ws), ws_bytes, _stream(x), tag=(N, Nc, K))
        if direct is not None:
            return dx, None, None
        return dx, dW, db


def linear_tanh(x, weight, bias=None, precision=None):
    """tanh(x W^T + b) as fp32: the one-launch kernel where the shape allows it (exact fp32 products), else the GEMM
    (`precision` as in linear()) + torch.tanh."""
    if linear_tanh_eligible(x, weight, bias):
        return LinearTanhFn.apply(x, weight, bias)
    return torch.tanh(linear(x, weight, bias, out_fp32=True, precision=precision))


def linear(x, weight, bias=None, out_fp32=False, precision=None, wT=None, wT_guard=None):
    """x W^T + b.  out_fp32: fp32 result from bf16-storage activations (the network's final output).
    precision: forward matrix-core path (None = PREC_FWD); see forward_precision().  wT / wT_guard: see LinearFn."""
    return LinearFn.apply(x, weight, bias, out_fp32, precision, wT, wT_guard)


def forward_precision(unbounded_input):
    """Forward GEMM path for a layer.  fp16x3 (the default PREC_FWD) carries fp32-grade products for operands inside
    fp16's range (|activation| < 8188, |weight| < 1023; out-of-range operands give inf/NaN, never a silently wrong
    value) - true for everything downstream of an instance / batch / graph norm.  A layer fed by RAW data (the first
    block: un-normalised vertex features) or living in a network built without norms gets bf16x6 instead: bf16 pieces
    have fp32's exponent range, the 3-piece split is exact, and the first block's GEMMs are tiny (K = 12)."""
    if unbounded_input and PREC_FWD == GEMM_F16X3:
        return GEMM_BF16X6
    return PREC_FWD


# ----------------------------------------------------------------------- autograd ops
# ---- weight-gradient side stream of the whole-block backward ----------------------------------------------------------
USE_WGRAD_STREAM = os.environ.get('STIN_WGRAD_STREAM', '1') == '1'
WGRAD_DEFER_JOIN = True
# the overlap pays where kernels are too short to fill the GPU; a weight-gradient GEMM of N * Yw * Cp above this keeps the
# whole chip busy for hundreds of microseconds and only slows the critical-path kernels it runs beside (measured: 500 k
# vertices / 4 levels +4.7 %, every block <= 1.5e10; 1 M vertices / 3 levels -13 %, every block >= 2e10)
WGRAD_MAX_WORK = float(os.environ.get('STIN_WGRAD_MAX_WORK', 1.6e10))
# ... and below this a step is launch-bound on the host, where the second stream only adds host work (3 event records / waits
# per block): 40 k vertices 4.2-4.8 ms per step without vs 5.4-5.5 with (every block <= 1.6e9), 60 k equal, 100 k 6.09 without
# vs 5.60 with (bottleneck blocks 2.4e9), batch of 8 crops / 4 levels 9.8 vs 9.3
WGRAD_MIN_WORK = float(os.environ.get('STIN_WGRAD_MIN_WORK', 2.0e9))
_WGRAD_SIDE = {}


class _WgradSide:
    """Per device: the HIP stream the weight-gradient GEMMs of stin_edgeconv_block_bwd run on and a ring of event
    triples that order it against the compute stream (a fresh triple per block call, so no event is re-recorded while
    a wait on its previous record may still be queued)."""
    RING = 64

    def __init__(self, dev):
        self.stream = torch.cuda.Stream(device=dev)
        self.ring = [[torch.cuda.Event() for _ in range(3)] for _ in range(self.RING)]
        for tri in self.ring:
            for e in tri:
                e.record(self.stream)               # forces creation of the hipEvent_t handles
        self.at = 0
        self.last_done = self.ring[0][2]
        self.hold = []                              # tensors the side stream may still be reading (dropped at the join)

    def next_events(self, track=True):
        """track=False: an event triple for a block that stays on the compute stream (its ev_done is recorded there): it must not
        become `last_done`, the event the deferred join of the SIDE stream waits for."""
        self.at = (self.at + 1) % self.RING
        tri = self.ring[self.at]
        if track:
            self.last_done = tri[2]
        return tri


def _wgrad_side(dev):
    side = _WGRAD_SIDE.get(dev.index)
    if side is None:
        side = _WGRAD_SIDE[dev.index] = _WgradSide(dev)
    return side


USE_DIRECT_GRADS = True


def _direct_grad_views(params, dry_run=False):
    """(dW1, db1, dW2, db2, dWs, dbs) as the parameters' views in an accepting train_step.FlatGradBucket, or None.
    The backward then OVERWRITES those views (one backward per step, every weight used by one block) and returns no
    gradient to autograd for them: no per-parameter accumulate node, no copy into the bucket afterwards, and nothing of
    autograd can read a gradient before the end-of-backward join of the weight-gradient stream."""
    if not USE_DIRECT_GRADS:
        return None
    bucket, out = None, []
    for p in params:
        if p is None:
            out.append(None)
            continue
        slot = getattr(p, '_stin_slot', None)
        if slot is None or not slot[0].accepting or (bucket is not None and slot[0] is not bucket) or slot[0].written[slot[1]]:
            return None
        bucket = slot[0]
        out.append(bucket.views[slot[1]])
    if bucket is None:
        return None
    if dry_run:                  # eligibility probe only (a node that covers several blocks decides for ALL of them first)
        return out
    for p in params:
        if p is not None:
            bucket.written[p._stin_slot[1]] = True
    return out


def _plain_autograd_may_defer():
    """The deferred join is safe under plain autograd only while nothing outside this module can read a gradient before
    the end of the backward pass.  DistributedDataParallel registers its reducer as C++ post-hooks on the AccumulateGrad
    nodes - invisible to `p._backward_hooks` / `_post_accumulate_grad_hooks` - and copies each gradient into its bucket on
    the compute stream as soon as autograd produces it.  DDP cannot exist without an initialised multi-rank process
    group, so in that situation only the TrainStep bucket route (which bypasses autograd for these gradients) uses the
    side stream; every other caller keeps the block on one stream."""
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)


def _wgrad_side_args(dev, keep_alive, params, direct=False, work=0):
    """-> (wgrad_stream, ev_dagg, ev_dy, ev_done, join) for stin_edgeconv_block_bwd.  The join with the compute stream is
    deferred to the end of the backward pass when nothing can read the gradients earlier: the TrainStep bucket route
    (`direct`), or plain single-process autograd with every parameter a leaf whose .grad is None (autograd then adopts
    the returned tensor without a kernel) and without hooks; otherwise the block stays on one stream."""
    if not USE_WGRAD_STREAM or work > WGRAD_MAX_WORK or work < WGRAD_MIN_WORK:
        return 0, 0, 0, 0, 0
    deferred = WGRAD_DEFER_JOIN and (direct or (_plain_autograd_may_defer() and all(
        p is None or (p.is_leaf and p.grad is None and not p._backward_hooks and
                      not getattr(p, '_post_accumulate_grad_hooks', None)) for p in params)))
    if not deferred:
        # a join inside every call measured SLOWER than one stream (9.70 vs 9.45 ms per step): gradient accumulation,
        # parameter hooks, DDP and non-leaf weights simply keep the whole block on the compute stream
        return 0, 0, 0, 0, 0
    side = _wgrad_side(dev)
    # the side stream reads these after this call has returned: keep them referenced until the join (then they are freed in
    # compute-stream order AFTER the join - no record_stream: its deferred frees made the caching allocator's pool grow by
    # ~80 MB per step over hundreds of steps with changing scene sizes)
    side.hold.append(keep_alive)
    ev = side.next_events()
    return side.stream.cuda_stream, ev[0].cuda_event, ev[1].cuda_event, ev[2].cuda_event, 0


def wgrad_side_settle(dev):
    """Defensive join for callers that own the step (TrainStep.forward_backward's finally): when a backward pass raised
    after a block had put work on the weight-gradient stream, the autograd engine dropped the end-of-backward callbacks
    and the join never ran - wait for the side stream now and release what it was reading / writing.  A no-op after a
    normal pass (the join emptied `hold`)."""
    side = _WGRAD_SIDE.get(dev.index if dev.index is not None else torch.cuda.current_device())
    if side is not None and side.hold:
        torch.cuda.current_stream(dev).wait_event(side.last_done)
        side.hold.clear()


def _wgrad_deferred_join(dev, params, grads):
    """Queue the end-of-backward join for one block whose weight gradients were left on the side stream: the compute
    stream waits for ev_done (recorded after every side-stream kernel enqueued so far), and the gradient tensors must
    have been adopted by autograd as they are - a copy would have read them before this join.  One callback per block
    call and no state between passes, so an aborted backward cannot leave a join behind."""
    side = _wgrad_side(dev)
    adopted = [(p, t.data_ptr()) for p, t in zip(params, grads) if p is not None and t is not None]

    def join():
        torch.cuda.current_stream(dev).wait_event(side.last_done)    # the newest ev_done: after all side work so far
        side.hold.clear()                                             # freed from here on = ordered after the join
        for p, ptr in adopted:
            if p.grad is not None and p.grad.data_ptr() != ptr:
                raise RuntimeError('weight-gradient stream: autograd copied a gradient before the end-of-backward join; '
                                   'set STIN_WGRAD_STREAM=0 for this autograd configuration')

    torch.autograd.Variable._execution_engine.queue_callback(join)


def block_split_modes(prec_fwd, b16, Cout):
    """(fwd_split, bwd_split) of one fused block: the storage form of its forward / backward weight operands."""
    fsp = (prec_fwd | GEMM_W_FRAG) if (not b16 and prec_fwd in (GEMM_BF16X3, GEMM_F16X3) and WEIGHT_PRESPLIT) else 0
    bsp = (PREC_BWD | GEMM_W_FRAG) if (not b16 and PREC_BWD in (GEMM_BF16X3, GEMM_F16X3) and WEIGHT_PRESPLIT and Cout % 4 == 0) else 0
    return fsp, bsp


# Translation-invariant blocks in the COMPACT layout (round 6; include/stin_hip.h STIN_TI_COMPACT): the reference's message is
# nn(x_j - x_i) (models/modules/edge_conv_translation_invariance.py:20-22), so W1 (x_j - x_i) + b1 = A_i + B_j with B = x W1^T and
# A_i = b1 - B_i: only B is a GEMM output (Yw = H (+ Cout) instead of 2 H (+ Cout)), the edge stage forms A_i per row - what the
# [-W1 ; W1] product wrote up to one ulp of its accumulator - and the backward pass carries D = dB - dA in H columns.  Half the first Linear's GEMM work in
# every direction.  fp32 storage with a saved-mask hidden width; STIN_TI_COMPACT=0 keeps both halves materialised (A/B switch).
TI_COMPACT = os.environ.get('STIN_TI_COMPACT', '1') != '0'
TI_MODE_COMPACT = 2


def trans_inv_mode(trans_inv, b16, H):
    """The `trans_inv` value the C entry points take: 0 = EdgeConv, 1 = translation-invariant (A and B materialised), 2 = compact."""
    if not trans_inv:
        return 0
    return TI_MODE_COMPACT if (TI_COMPACT and not b16 and USE_EDGE_MASK and edge_mask_supported(H)) else 1


def block_yw(H, Cout, has_shortcut, ti_mode):
    """Width of Y / dY / the packed first-Linear operand (csrc/stin_common.h: stin_yw)."""
    return (H if ti_mode == TI_MODE_COMPACT else 2 * H) + (Cout if has_shortcut else 0)


BLOCK_PACKED = 0x800
USE_PACK_MANY = True


class PackSet:
    """The weight packs of ALL fused blocks of a network in one launch per step (stin_edgeconv_pack_many_f32) instead of
    one tiny launch at the head of every block: persistent per-block operand buffers (the block call's forward workspace
    and wcatT | w2T), a job table in device memory written once.  specs: [(W1, b1, W2, b2, Ws, bs, trans_inv, prec_fwd, B)]
    in block order, fp32 storage.  Valid while the parameters stay where they are (`matches`)."""

    def __init__(self, specs, dev, b16=False, transposes=()):
        import ctypes
        import struct
        lib = _lib.load()
        self.key = self.key_of(specs) + (bool(b16),) + tuple((_ptr(W), tuple(W.shape)) for W in transposes)
        self.b16 = bool(b16)
        self.buffers = []                      # per block: (ws, wts, fwd_split, bwd_split, wcatT view, w2T view, b16)
        blob, self.max_elems = b'', 0
        pad = 8 if b16 else 4
        for (W1, b1, W2, b2, Ws, bs, trans_inv, prec_fwd, B) in specs:
            H, Cout = W1.shape[0], W2.shape[0]
            Cin = W1.shape[1] if trans_inv else W1.shape[1] // 2
            Cp = (Cin + pad - 1) // pad * pad
            has_sc = Ws is not None
            ti = trans_inv_mode(trans_inv, b16, H)
            Yw = block_yw(H, Cout, has_sc, ti)
            fsp, bsp = block_split_modes(prec_fwd, b16, Cout)       # what the block call is handed (bf16 rows: 0, 0)
            # what the pack writes: bf16 rows -> plain bf16 operands where every reduction length is a multiple of 8
            # (the rule of stin_edgeconv_block_fwd), else the fp32 / split form of the fp32-storage path
            jf, jb = ((GEMM_W_BF16, GEMM_W_BF16) if (Cp % 8 == 0 and Cout % 8 == 0) else (0, 0)) if b16 else (fsp, bsp)
            ws = torch.empty(lib.stin_edgeconv_block_fwd_workspace_bytes(Cin, Cp, H, Cout, int(has_sc), B), dtype=torch.uint8, device=dev)
            wts = torch.empty(Yw * Cp + H * Cout, dtype=torch.float32, device=dev)
            off = [ctypes.c_size_t(0) for _ in range(3)]
            _lib.check(lib.stin_edgeconv_block_fwd_pack_offsets(Cp, H, Cout, int(has_sc), *[ctypes.byref(o) for o in off]),
                       'stin_edgeconv_block_fwd_pack_offsets')
            base = (ws.data_ptr() + 255) & ~255
            wcat, w2s, bcat = (base + o.value for o in off)
            W1c, W2c = W1.contiguous(), W2.contiguous()
            assert W1c.data_ptr() == W1.data_ptr() and W2c.data_ptr() == W2.data_ptr(), 'pack_many needs contiguous weights'
            blob += struct.pack('<10Q8i', _ptr(W1), _ptr(b1), _ptr(Ws), _ptr(bs), _ptr(W2), wcat, bcat, _ptr(wts),
                                _ptr(wts) + 4 * Yw * Cp, w2s if jf else 0, Cin, Cp, H, Cout, int(has_sc), ti, jf, jb)
            self.max_elems = max(self.max_elems, Yw * Cp + H * Cout)
            # (the two backward operands as ready-made views: no tensor views are created inside autograd.Function.forward)
            self.buffers.append((ws, wts, fsp, bsp, wts[:Yw * Cp].view(Cp, Yw), wts[Yw * Cp:].view(H, Cout), bool(b16)))
        # plain transposes riding on the same launch (the tail Linear's backward operand W^T): a job whose first operand is
        # empty (Cin = Cp = 0) and whose "second Linear" is the weight - the pack writes w2T [K, Nc] = W^T in plain fp32
        self.transposed = []
        for W in transposes:
            Nc, K = W.shape
            assert W.is_contiguous() and W.dtype == torch.float32
            wT = torch.empty(K, Nc, dtype=torch.float32, device=dev)
            blob += struct.pack('<10Q8i', 0, 0, 0, 0, _ptr(W), 0, 0, 0, _ptr(wT), 0, 0, 0, K, Nc, 0, 0, 0, 0)
            self.max_elems = max(self.max_elems, K * Nc)
            self.transposed.append(wT)
        self.jobs = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
        self.n = len(specs) + len(transposes)

    @staticmethod
    def key_of(specs):
        return tuple((_ptr(W1), _ptr(b1), _ptr(W2), _ptr(Ws), _ptr(bs), tuple(W1.shape), tuple(W2.shape), bool(t), int(pf), int(B),
                      PREC_BWD, WEIGHT_PRESPLIT, GEMM_W_FRAG, TI_COMPACT, USE_EDGE_MASK) for (W1, b1, W2, b2, Ws, bs, t, pf, B) in specs)

    def matches(self, specs, b16=False, transposes=()):
        return self.key == self.key_of(specs) + (bool(b16),) + tuple((_ptr(W), tuple(W.shape)) for W in transposes)

    def run(self):
        self.runs = getattr(self, 'runs', 0) + 1       # what a consumer of `transposed` compares (LinearFn's wT guard)
        _call('stin_edgeconv_pack_many_f32', _ptr(self.jobs), self.n, self.max_elems, _stream(self.jobs))


class EdgeConvBlockFn(torch.autograd.Function):
    """One GraphResnetBlock with an EdgeConv(mean) filter and instance norm, fused at the
    autograd level (reference models/surfacetextureinpaintingnet.py:507-521), taking the
    reference-layout parameters directly:

        Y   = x Wcat^T + bcat          Wcat = [Wa-Wb ; Wb ; Ws]   (per-VERTEX MFMA GEMM)
        h   = mean_j ReLU(A_i + B_j)   A = Y[:, :H], B = Y[:, H:2H]   (HIP edge stage)
        agg = h W2^T + b2 [deg > 0]                                (per-VERTEX MFMA GEMM, masked bias)
        out = (Y[:, 2H:] or x) + ELU(InstanceNorm(agg))            (HIP epilogue)

    Saved for backward: per-vertex tensors plus the E*H-bit ReLU mask (recompute kernels when the hidden width does not
    support the mask).  Fast path: one C call per direction (stin_edgeconv_block_fwd / _bwd, same kernels, same order)."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, Ws, bs, edges, groups, trans_inv, eps=EPS, prec_fwd=None, prepacked=None):
        prec_fwd = PREC_FWD if prec_fwd is None else int(prec_fwd)
        x, _ = _mat(x)
        N, Cin = x.shape
        H, Cout = W1.shape[0], W2.shape[0]
        has_shortcut = Ws is not None
        dev = x.device
        b16 = x.dtype == torch.bfloat16
        ti = trans_inv_mode(trans_inv, b16, H)                    # 2 = compact: Y = [B | S], A_i = b1 - B_i formed by the edge stage
        Yw = block_yw(H, Cout, has_shortcut, ti)
        oB, oS = (0 if ti == TI_MODE_COMPACT else H), Yw - (Cout if has_shortcut else 0)   # columns of B and of the shortcut in Y / dY
        pad = 8 if b16 else 4
        Cp = (Cin + pad - 1) // pad * pad                         # inner dimension padded for the 16-byte GEMM paths
        if Cp != Cin:                                             # (the 10-channel network input -> 12; bf16: 16)
            xp = torch.empty(x.shape[0], Cp, dtype=x.dtype, device=dev)
            _call('stin_pad_rows' + ('_bf16' if b16 else '_f32'), _ptr(x), x.stride(0), x.shape[0], Cin, Cp, _ptr(xp), _stream(x))
        else:
            xp = x
        # forward / backward weight operands, pre-split once here into the two 16-bit pieces the split GEMMs use
        # (instead of once per GEMM block); plain fp32 for the other precisions and for bf16-storage activations
        fsp, bsp = block_split_modes(prec_fwd, b16, Cout)
        fast = (USE_BLOCK_CALL and USE_EDGE_MASK and edge_mask_supported(H) and N > 1
                and not KernelTimer.per_kernel_path())   # (the bench's per-kernel HIP-event brackets need the per-kernel path)
        ctx.fast = fast
        if fast:
            lib = _lib.load()
            B = groups.B
            packed = 0
            if prepacked is not None and len(prepacked) > 6 and prepacked[6] == b16 and prepacked[2] == fsp and prepacked[3] == bsp:
                ws, packed = prepacked[0], BLOCK_PACKED                           # operands already packed (PackSet.run)
                wcatT, w2T = prepacked[4], prepacked[5]
            else:                                                                 # backward weight operands (no views in here)
                wcatT = torch.empty(Cp, Yw, dtype=torch.float32, device=dev)
                w2T = torch.empty(H, Cout, dtype=torch.float32, device=dev)
            Y = torch.empty(N, Yw, dtype=x.dtype, device=dev)
            hE = torch.empty(N, H + pad, dtype=x.dtype, device=dev)
            mask = torch.empty(max(edges.n_edges, 1) * (H // 32), dtype=torch.int32, device=dev)
            agg = torch.empty(N, Cout, dtype=x.dtype, device=dev)
            mean = torch.empty(B, Cout, dtype=torch.float32, device=dev)
            rstd = torch.empty(B, Cout, dtype=torch.float32, device=dev)
            out = torch.empty(N, Cout, dtype=x.dtype, device=dev)
            ws_bytes = lib.stin_edgeconv_block_fwd_workspace_bytes(Cin, Cp, H, Cout, int(has_shortcut), B)
            if not packed:
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            elif ws.numel() < ws_bytes:
                raise RuntimeError('prepacked workspace too small for this batch (PackSet built for another batch size)')
            W1c, W2c = W1.contiguous(), W2.contiguous()
            cd = edges.by_dst
            _call('stin_edgeconv_block_fwd', int(b16), _ptr(xp), xp.stride(0), N, Cin, Cp, H, Cout, int(has_shortcut),
                  ti, _ptr(W1c), _ptr(b1), _ptr(W2c), _ptr(b2), _ptr(Ws), _ptr(bs), _ptr(cd.rowptr), _ptr(cd.col),
                  _ptr(groups.ptr_sum), B, _ptr(groups.gid), _ptr(groups.inv_cnt), int(groups.quirk), float(eps), prec_fwd,
                  fsp | packed, bsp,
                  _ptr(wcatT), _ptr(w2T), _ptr(Y), Yw, _ptr(hE), H + pad, _ptr(mask), _ptr(agg), _ptr(mean), _ptr(rstd),
                  _ptr(out), Cout, _ptr(ws), ws_bytes, _stream(x))
            ctx.save_for_backward(xp, Y, hE, agg, mean, rstd, wcatT, w2T)
            ctx.mask = mask
            ctx.cin = Cin
            ctx.edges, ctx.groups, ctx.H, ctx.has_shortcut, ctx.trans_inv = edges, groups, H, has_shortcut, ti
            ctx.has_b1, ctx.has_b2, ctx.has_bs = b1 is not None, b2 is not None, bs is not None
            ctx.w1_shape = tuple(W1.shape)
            ctx.bsp = bsp
            ctx.params = (W1, b1, W2, b2, Ws, bs)
            return out
        pack = torch.empty(Yw * Cp * 2 + 2 * H * Cout + Yw, dtype=torch.float32, device=dev)
        wcat = pack[:Yw * Cp].view(Yw, Cp)
        wcatT = pack[Yw * Cp:2 * Yw * Cp].view(Cp, Yw)
        w2T = pack[2 * Yw * Cp:2 * Yw * Cp + H * Cout].view(H, Cout)
        w2s = pack[2 * Yw * Cp + H * Cout:2 * Yw * Cp + 2 * H * Cout].view(Cout, H)
        bcat = pack[2 * Yw * Cp + 2 * H * Cout:]
        W1c, W2c = W1.contiguous(), W2.contiguous()
        _call('stin_edgeconv_pack_f32', _ptr(W1c), _ptr(b1), _ptr(Ws), _ptr(bs), _ptr(W2c), Cin, Cp, H, Cout,
              int(has_shortcut), ti, _ptr(wcat), _ptr(bcat), _ptr(wcatT), _ptr(w2T), _ptr(w2s) if fsp else None,
              fsp, bsp, _stream(x))
        pf = (prec_fwd | GEMM_W_PRESPLIT | GEMM_W_FRAG) if fsp else prec_fwd
        Y = gemm_nt(xp, wcat, bcat, precision=pf)
        hE = torch.empty(N, H + pad, dtype=x.dtype, device=dev)     # [h | (deg > 0) | pad]: rows stay 16-byte multiples
        # ReLU decisions as bits (E*H/8 bytes): backward then needs no recompute gathers
        use_mask = USE_EDGE_MASK and edge_mask_supported(H) and Y.stride(0) % 4 == 0
        if b16 and not use_mask:
            raise NotImplementedError('bf16 storage needs the saved ReLU mask: hidden width %d not in '
                                      '{128, 256, 512, 1024, 2048} (or STIN_EDGE_MASK=0)' % H)
        mask = torch.empty(max(edges.n_edges, 1) * (H // 32), dtype=torch.int32, device=dev) if use_mask else None
        if ti == TI_MODE_COMPACT:
            if not use_mask:
                raise RuntimeError('compact trans-inv layout needs the saved ReLU mask (16-byte aligned rows)')
            edge_relu_mean_fwd_ti(b1, Y[:, :H], edges.by_dst, hE, indicator=True, mask=mask)
        else:
            edge_relu_mean_fwd(Y[:, :H], Y[:, H:2 * H], edges.by_dst, hE, indicator=True, mask=mask)
        fused = None
        if fsp and groups.B == 1 and groups.gid is None and not groups.quirk and N > 1:
            fused = gemm_nt_colstats(hE[:, :H], w2s, b2, hE[:, H], pf, Cout)       # GEMM2 + column sums in one launch
        if fused is not None:
            agg = fused[0]
            mean, rstd = moments_final(fused[1], groups.inv_cnt, eps)
        else:
            agg = gemm_nt(hE[:, :H], w2s if fsp else W2c, b2, row_mask=hE[:, H], precision=pf)
            mean, rstd = instance_stats(agg, groups, eps)
        res = Y[:, oS:] if has_shortcut else x
        out = norm_act_res_fwd(agg, mean, rstd, groups, res=res, act=True)
        ctx.save_for_backward(xp, Y, hE, agg, mean, rstd, wcatT, w2T)
        ctx.mask = mask
        ctx.cin = Cin
        ctx.edges, ctx.groups, ctx.H, ctx.has_shortcut, ctx.trans_inv = edges, groups, H, has_shortcut, ti
        ctx.has_b1, ctx.has_b2, ctx.has_bs = b1 is not None, b2 is not None, bs is not None
        ctx.w1_shape = tuple(W1.shape)
        ctx.prec_bwd_nt = (PREC_BWD | GEMM_W_PRESPLIT | GEMM_W_FRAG) if bsp else PREC_BWD
        ctx.bsp = bsp
        return out

    @staticmethod
    def backward(ctx, g):
        x, Y, hE, agg, mean, rstd, wcatT, w2T = ctx.saved_tensors          # x: the (possibly channel-padded) block input
        edges, groups, H = ctx.edges, ctx.groups, ctx.H
        Cin, Cp, Cout = ctx.cin, x.shape[1], agg.shape[1]
        g, ldg = _mat(g)
        if ctx.fast:
            lib = _lib.load()
            dev, N, b16 = x.device, x.shape[0], x.dtype == torch.bfloat16
            _same(x, g)
            dx = torch.empty(N, Cp, dtype=x.dtype, device=dev) if ctx.needs_input_grad[0] else None
            direct = _direct_grad_views(ctx.params)
            if direct is not None:            # a TrainStep bucket is accepting: write the gradients where the optimizer reads them
                dW1, db1, dW2, db2, dWs, dbs = direct
            else:
                dW1 = torch.empty(ctx.w1_shape, dtype=torch.float32, device=dev)
                db1 = torch.empty(H, dtype=torch.float32, device=dev) if ctx.has_b1 else None
                dWs = torch.empty(Cout, Cin, dtype=torch.float32, device=dev) if ctx.has_shortcut else None
                dbs = torch.empty(Cout, dtype=torch.float32, device=dev) if ctx.has_bs else None
                dW2 = torch.empty(Cout, H, dtype=torch.float32, device=dev)
                db2 = torch.empty(Cout, dtype=torch.float32, device=dev) if ctx.has_b2 else None
            ws_bytes = lib.stin_edgeconv_block_bwd_workspace_bytes(N, Cp, H, Cout, int(ctx.has_shortcut), groups.B, int(b16))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            cs = edges.by_src
            side = _wgrad_side_args(dev, (ws, x, hE, g), ctx.params, direct is not None, work=float(N) * Y.shape[1] * Cp)
            if side[0] and direct is None:
                # the side stream WRITES these fresh tensors after this call has returned.  They must not be referenced from
                # here (autograd adopts a returned gradient only when it holds the sole reference - otherwise it copies it on
                # the compute stream, before the join); record_stream makes the allocator wait for the side stream should
                # an aborted backward free them early (a few MB of weights: no pool growth, unlike the activations)
                ss = _wgrad_side(dev).stream
                for t in (dW1, db1, dW2, db2, dWs, dbs):
                    if t is not None:
                        t.record_stream(ss)
            _call('stin_edgeconv_block_bwd', int(b16), _ptr(g), ldg, _ptr(x), x.stride(0), N, Cin, Cp, H, Cout,
                  int(ctx.has_shortcut), int(ctx.trans_inv), _ptr(Y), Y.stride(0), _ptr(hE), hE.stride(0), _ptr(ctx.mask),
                  _ptr(agg), _ptr(mean), _ptr(rstd), _ptr(wcatT), _ptr(w2T), _ptr(edges.by_dst.rowptr), _ptr(cs.rowptr),
                  _ptr(cs.col), _ptr(edges.xslot), _ptr(edges.w_src), _ptr(groups.ptr_true), groups.B, _ptr(groups.gid),
                  _ptr(groups.sid if groups.quirk else None), _ptr(groups.inv_cnt), int(PREC_BWD), ctx.bsp, _ptr(dx), Cp, _ptr(dW1), _ptr(db1), _ptr(dW2), _ptr(db2),
                  _ptr(dWs), _ptr(dbs), _ptr(ws), ws_bytes, _stream(x), *side)
            ctx.mask = None
            if side[0] and not side[4]:
                _wgrad_deferred_join(dev, ctx.params, () if direct is not None else (dW1, db1, dW2, db2, dWs, dbs))
            if direct is not None:
                # a bucket that reduces in segments hands every completed one to RCCL now (train_step.FlatGradBucket)
                # behind the side stream's NEWEST event whenever any block of THIS backward pass has put weight-gradient work
                # there (`hold` is emptied by the end-of-backward join) - not only when this block did: a segment spans
                # several blocks, and an earlier one may have used the side stream while the block completing it did not
                sd = _WGRAD_SIDE.get(dev.index if dev.index is not None else torch.cuda.current_device())
                ctx.params[0]._stin_slot[0].block_done(sd.last_done if (sd is not None and sd.hold) else None)
                if dx is not None and Cp != Cin:
                    dx = dx[:, :Cin]
                return (dx,) + (None,) * 12
            if dx is not None and Cp != Cin:
                dx = dx[:, :Cin]
            return dx, dW1, db1, dW2, db2, dWs, dbs, None, None, None, None, None, None
        dagg = instance_norm_act_bwd(agg, g, mean, rstd, groups, act=True)
        dw2b = gemm_tn(dagg, hE[:, :H], ones_column=True, row_weight=hE[:, H], precision=PREC_BWD)   # [Cout, H + 1] = dW2 | db2
        dhE = gemm_nt(dagg, w2T, precision=ctx.prec_bwd_nt)                                             # [N, H] = dagg W2
        dY = torch.empty_like(Y)
        compact = ctx.trans_inv == TI_MODE_COMPACT
        oS = Y.shape[1] - (Cout if ctx.has_shortcut else 0)
        db1_ti = None
        if compact:                                     # D = dB - dA in one row of H columns (+ the column sums of dA = db1)
            db1_ti = edge_relu_mean_bwd_mask_ti(dhE, ctx.mask, edges, dY[:, :H])
            ctx.mask = None
        elif ctx.mask is not None:
            edge_relu_mean_bwd_mask(dhE, ctx.mask, edges, dY[:, :H], dY[:, H:2 * H])
            ctx.mask = None
        else:
            A, B = Y[:, :H], Y[:, H:2 * H]
            edge_relu_mean_bwd_dst(A, B, dhE, edges.by_dst, dY[:, :H])
            edge_relu_mean_bwd_src(A, B, dhE, edges.inv_deg, edges.by_src, dY[:, H:2 * H])
        if ctx.has_shortcut:
            dY[:, oS:].copy_(g)                     # (the whole-block C call lets this ride on the edge launch)
        dwb = gemm_tn(dY, x, ones_column=True, precision=PREC_BWD)           # [Yw, Cin + 1]: packed weight grad | bias grad
        # dx = dY Wcat (+ g: the identity-residual path, added in the GEMM epilogue); skipped when the block input
        # needs no gradient (the network input of the first block)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = gemm_nt(dY, wcatT, precision=ctx.prec_bwd_nt, residual=None if ctx.has_shortcut else g)
            if Cp != Cin:
                dx = dx[:, :Cin]
        dev = x.device
        dW1 = torch.empty(ctx.w1_shape, dtype=torch.float32, device=dev)
        db1 = torch.empty(H, dtype=torch.float32, device=dev) if ctx.has_b1 else None
        dWs = torch.empty(Cout, Cin, dtype=torch.float32, device=dev) if ctx.has_shortcut else None
        dbs = torch.empty(Cout, dtype=torch.float32, device=dev) if ctx.has_bs else None
        dW2 = torch.empty(Cout, H, dtype=torch.float32, device=dev)
        db2 = torch.empty(Cout, dtype=torch.float32, device=dev) if ctx.has_b2 else None
        _call('stin_edgeconv_unpack_grads_f32', _ptr(dwb), _ptr(dw2b), Cin, Cp, H, Cout, int(ctx.has_shortcut),
              int(ctx.trans_inv), _ptr(dW1), _ptr(db1), _ptr(dWs), _ptr(dbs), _ptr(dW2), _ptr(db2), _stream(x))
        if compact and db1 is not None:
            db1 = db1_ti                            # (the packed product's ones column holds sum_i D_i ~ 0, not db1)
        return dx, dW1, db1, dW2, db2, dWs, dbs, None, None, None, None, None, None


# ---- a chain of fused blocks of one level in ONE autograd node ------------------------------------------------------------
USE_CHAIN = True
_CHAIN_JOB = None


def _chain_struct():
    global _CHAIN_JOB
    if _CHAIN_JOB is None:
        import struct
        _CHAIN_JOB = struct.Struct('<27Q4i')             # stin_chain_job_t (include/stin_hip.h): 232 bytes
    return _CHAIN_JOB


def chain_eligible(blocks, x, edges_list, groups):
    """The n blocks can run as one EdgeConvChainFn: all fused EdgeConv + instance-norm blocks of the same width without
    shortcut (the bottleneck of the network), operands packed by the network's PackSet, saved-mask path available."""
    if not (USE_CHAIN and USE_BLOCK_CALL and USE_EDGE_MASK and not KernelTimer.per_kernel_path() and len(blocks) >= 2 and x.is_cuda):
        return False
    if x.dim() != 2 or x.shape[0] <= 1 or x.dtype not in (torch.float32, torch.bfloat16):
        return False
    C = x.shape[1]
    pad = 8 if x.dtype == torch.bfloat16 else 4
    if C % pad != 0:
        return False
    if not edge_mask_supported(2 * C):
        return False
    b16 = x.dtype == torch.bfloat16
    fsp, bsp = block_split_modes(PREC_FWD, b16, C)
    packed = [b._prepacked is not None for b in blocks]
    if any(packed) != all(packed):                                # all operands packed by the network's PackSet, or none
        return False
    for b in blocks:
        pp = b._prepacked
        if pp is not None and (len(pp) < 7 or pp[6] != b16 or pp[2] != fsp or pp[3] != bsp):
            return False
        if (b.dim_in != C or b.dim_out != C or b.unbounded_input or hasattr(b, 'shortcut')
                or b.first_norm.eps != blocks[0].first_norm.eps):
            return False
    return True


class EdgeConvChainFn(torch.autograd.Function):
    """n consecutive GraphResnetBlocks (EdgeConv(mean) + instance norm + ELU + identity residual) of one level as ONE
    autograd node and one C call per direction (stin_edgeconv_chain_fwd / _bwd: a loop over the whole-block launch
    sequences, same kernels in the same order -> bit-identical to n EdgeConvBlockFn nodes).  What it removes is host time:
    n - 1 autograd nodes, 2 (n - 1) foreign calls and ~12 (n - 1) tensor allocations per step (the 9 bottleneck blocks of
    the 3-D config).  args: x, meta = (edges_list, groups, eps, prec_fwd, prepacked_list, trans_inv_list), then the flat
    parameters W1, b1, W2, b2 of every block."""

    calls = 0                    # (tests check that the chain path was actually taken)

    @staticmethod
    def forward(ctx, x, meta, *params):
        EdgeConvChainFn.calls += 1
        edges_list, groups, eps, prec_fwd, prepacked, trans_inv = meta
        n = len(edges_list)
        lib = _lib.load()
        x, _ = _mat(x)
        N, C = x.shape
        H = 2 * C
        dev, dt = x.device, x.dtype
        b16 = dt == torch.bfloat16
        pad = 8 if b16 else 4
        B = groups.B
        Yw = 2 * H
        fsp, bsp = block_split_modes(prec_fwd, b16, C)
        # one arena per tensor kind (no views of them are created in here: block i's slices are addressed by pointer)
        Y = torch.empty(n, N, Yw, dtype=dt, device=dev)
        hE = torch.empty(n, N, H + pad, dtype=dt, device=dev)
        agg = torch.empty(n, N, C, dtype=dt, device=dev)
        outs = torch.empty(max(n - 1, 1), N, C, dtype=dt, device=dev)         # outputs of blocks 0 .. n - 2
        out = torch.empty(N, C, dtype=dt, device=dev)                         # the chain's output (block n - 1)
        stats = torch.empty(n, 2, B, C, dtype=torch.float32, device=dev)
        words = [max(e.n_edges, 1) * (H // 32) for e in edges_list]
        mask = torch.empty(sum(words), dtype=torch.int32, device=dev)
        es = x.element_size()
        ws_bytes = lib.stin_edgeconv_block_fwd_workspace_bytes(C, C, H, C, 0, B)
        packed = prepacked[0] is not None
        wts_n = Yw * C + H * C                                                # backward weight operands wcatT | w2T per block
        wts = ws = None
        if not packed:                                                        # the block calls run their own pack: own buffers
            wts = torch.empty(n, wts_n, dtype=torch.float32, device=dev)
            ws = torch.empty(n, ws_bytes, dtype=torch.uint8, device=dev)
        st = _chain_struct()
        blob, moff = [], 0
        pY, pH, pA, pO, pS, pM = _ptr(Y), _ptr(hE), _ptr(agg), _ptr(outs), _ptr(stats), _ptr(mask)
        for i in range(n):
            W1, b1, W2, b2 = params[4 * i:4 * i + 4]
            pp = prepacked[i]
            if packed:
                if pp[2] != fsp or pp[3] != bsp or pp[0].numel() < ws_bytes:
                    raise RuntimeError('EdgeConvChainFn: prepacked operands do not match this call (stale PackSet)')
                p_wcatT, p_w2T, p_ws, flag = _ptr(pp[4]), _ptr(pp[5]), _ptr(pp[0]), fsp | BLOCK_PACKED
            else:
                p_wcatT = _ptr(wts) + i * wts_n * 4
                p_w2T, p_ws, flag = p_wcatT + Yw * C * 4, _ptr(ws) + i * ws_bytes, fsp
            cd = edges_list[i].by_dst
            o_i = _ptr(out) if i == n - 1 else pO + i * N * C * es
            blob.append(st.pack(_ptr(W1.contiguous()), _ptr(b1), _ptr(W2.contiguous()), _ptr(b2), p_wcatT, p_w2T, p_ws,
                                _ptr(cd.rowptr), _ptr(cd.col), 0, 0, 0, 0,
                                pY + i * N * Yw * es, pH + i * N * (H + pad) * es, pM + moff * 4, pA + i * N * C * es,
                                pS + (2 * i) * B * C * 4, pS + (2 * i + 1) * B * C * 4, o_i, 0, 0, 0, 0, 0, 0, 0,
                                int(trans_inv[i]), flag, bsp, int(prec_fwd)))
            moff += words[i]
        import ctypes
        buf = ctypes.create_string_buffer(b''.join(blob), n * st.size)
        _call('stin_edgeconv_chain_fwd', int(b16), buf, n, _ptr(x), x.stride(0), N, C, C, H, _ptr(groups.ptr_sum), B, _ptr(groups.gid),
              _ptr(groups.inv_cnt), int(groups.quirk), float(eps), ws_bytes, _stream(x))
        ctx.save_for_backward(x, Y, hE, agg, outs, stats, mask)
        ctx.meta = (edges_list, groups, prepacked, trans_inv, words, bsp, wts)
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, g):
        x, Y, hE, agg, outs, stats, mask = ctx.saved_tensors
        edges_list, groups, prepacked, trans_inv, words, bsp, wts = ctx.meta
        params = ctx.params
        n = len(edges_list)
        lib = _lib.load()
        N, C = x.shape
        b16 = x.dtype == torch.bfloat16
        H, Yw, pad, B = 2 * C, 4 * C, (8 if b16 else 4), groups.B
        wts_n = Yw * C + H * C
        dev, dt, es = x.device, x.dtype, x.element_size()
        g, ldg = _mat(g)
        _same(x, g)
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty(N, C, dtype=dt, device=dev) if need_dx else None
        scratch = torch.empty(2, N, C, dtype=dt, device=dev)
        # gradients: straight into an accepting TrainStep bucket (all blocks or none), else fresh tensors handed to autograd
        direct = []                                        # (all blocks probed before the bucket's bookkeeping changes, as in NetFn)
        if all(_direct_grad_views(tuple(params[4 * i:4 * i + 4]) + (None, None), dry_run=True) is not None for i in range(n)):
            direct = [_direct_grad_views(tuple(params[4 * i:4 * i + 4]) + (None, None)) for i in range(n)]
        if len(direct) != n:
            direct = []
            grads = []
            for i in range(n):
                W1, b1, W2, b2 = params[4 * i:4 * i + 4]
                grads += [torch.empty(W1.shape, dtype=torch.float32, device=dev),
                          torch.empty(H, dtype=torch.float32, device=dev) if b1 is not None else None,
                          torch.empty(C, H, dtype=torch.float32, device=dev),
                          torch.empty(C, dtype=torch.float32, device=dev) if b2 is not None else None]
        ws_bytes = lib.stin_edgeconv_block_bwd_workspace_bytes(N, C, H, C, 0, B, int(b16))
        ws = torch.empty(n, ws_bytes, dtype=torch.uint8, device=dev)          # every block its own: the side stream reads it later
        work = float(N) * Yw * C
        all_params = [p for p in params]
        use_side = (USE_WGRAD_STREAM and WGRAD_MIN_WORK <= work <= WGRAD_MAX_WORK and WGRAD_DEFER_JOIN and
                    (bool(direct) or (_plain_autograd_may_defer() and all(
                        p is None or (p.is_leaf and p.grad is None and not p._backward_hooks and
                                      not getattr(p, '_post_accumulate_grad_hooks', None)) for p in all_params))))
        side_stream, evs = 0, [(0, 0)] * n
        if use_side:
            side = _wgrad_side(dev)
            side.hold.append((ws, x, hE, outs, Y, agg, mask, scratch, wts))
            side_stream = side.stream.cuda_stream
            evs = [None] * n
            for i in reversed(range(n)):          # in BACKWARD order: side.last_done must be the event recorded last (block 0's)
                tri = side.next_events()
                evs[i] = (tri[1].cuda_event, tri[2].cuda_event)
            if not direct:
                for t in grads:
                    if t is not None:
                        t.record_stream(side.stream)
        st = _chain_struct()
        blob, moff = [], 0
        pY, pH, pA, pO, pS, pM, pW = _ptr(Y), _ptr(hE), _ptr(agg), _ptr(outs), _ptr(stats), _ptr(mask), _ptr(ws)
        for i in range(n):
            pp = prepacked[i]
            e = edges_list[i]
            cs = e.by_src
            if direct:
                dW1, db1, dW2, db2 = direct[i][:4]
            else:
                dW1, db1, dW2, db2 = grads[4 * i:4 * i + 4]
            if pp is not None:
                p_wcatT, p_w2T = _ptr(pp[4]), _ptr(pp[5])
            else:
                p_wcatT = _ptr(wts) + i * wts_n * 4
                p_w2T = p_wcatT + Yw * C * 4
            blob.append(st.pack(0, 0, 0, 0, p_wcatT, p_w2T, 0,
                                _ptr(e.by_dst.rowptr), 0, _ptr(cs.rowptr), _ptr(cs.col), _ptr(e.xslot), _ptr(e.w_src),
                                pY + i * N * Yw * es, pH + i * N * (H + pad) * es, pM + moff * 4, pA + i * N * C * es,
                                pS + (2 * i) * B * C * 4, pS + (2 * i + 1) * B * C * 4, (pO + i * N * C * es) if i < n - 1 else 0,
                                _ptr(dW1), _ptr(db1), _ptr(dW2), _ptr(db2), pW + i * ws_bytes, evs[i][0], evs[i][1],
                                int(trans_inv[i]), 0, bsp, 0))
            moff += words[i]
        import ctypes
        buf = ctypes.create_string_buffer(b''.join(blob), n * st.size)
        _call('stin_edgeconv_chain_bwd', int(b16), buf, n, _ptr(g), ldg, _ptr(x), x.stride(0), N, C, C, H, _ptr(groups.ptr_true), B,
              _ptr(groups.gid), _ptr(groups.sid if groups.quirk else None), _ptr(groups.inv_cnt), int(PREC_BWD), _ptr(dx), C,
              _ptr(scratch[0]), _ptr(scratch[1]), ws_bytes, _stream(x), side_stream)
        if use_side:
            _wgrad_deferred_join(dev, all_params if not direct else (), () if direct else grads)
        if direct:
            sd = _WGRAD_SIDE.get(dev.index if dev.index is not None else torch.cuda.current_device())
            params[0]._stin_slot[0].block_done(sd.last_done if (sd is not None and sd.hold) else None)
            return (dx, None) + (None,) * len(params)
        return (dx, None) + tuple(grads)


def edgeconv_chain(x, blocks, edges_list, groups, eps, prec_fwd):
    params = []
    for b in blocks:
        lin1, lin2 = b.first_filter.nn[0], b.first_filter.nn[2]
        params += [lin1.weight, lin1.bias, lin2.weight, lin2.bias]
    b16 = x.dtype == torch.bfloat16
    meta = (list(edges_list), groups, float(eps), int(prec_fwd), [b._prepacked for b in blocks],
            [trans_inv_mode(b.first_filter.trans_inv, b16, b.first_filter.nn[0].weight.shape[0]) for b in blocks])
    return EdgeConvChainFn.apply(x, meta, *params)


# ---- the graph part of the network in ONE autograd node ----------------------------------------------------------------------
USE_NET_CALL = os.environ.get('STIN_NET_CALL', '1') != '0'
_NET_OP = None
OP_BLOCK, OP_POOL_MAX, OP_UNPOOL = 0, 1, 2


def _net_struct():
    global _NET_OP
    if _NET_OP is None:
        import struct
        _NET_OP = struct.Struct('<16ifi7q2Q42Q')             # stin_net_op_t (include/stin_hip.h): 480 bytes
        assert _NET_OP.size == 480
    return _NET_OP


def _align256(n):
    return (n + 255) & ~255


def net_eligible(steps, x):
    """steps = [('block', GraphResnetBlock, EdgeSet, NormGroups) | ('pool', PoolMap) | ('unpool', PoolMap)] can run as one
    NetFn: fused EdgeConv + instance-norm blocks on the whole-block path (saved ReLU mask available), max pooling, fp32 or
    bf16 storage; operands of every block packed by the network's PackSet or of none."""
    if not (USE_NET_CALL and USE_BLOCK_CALL and USE_EDGE_MASK and not KernelTimer.per_kernel_path() and x.is_cuda and x.dim() == 2):
        return False
    if x.dtype not in (torch.float32, torch.bfloat16) or x.shape[0] <= 1:
        return False
    b16 = x.dtype == torch.bfloat16
    packed = []
    for st in steps:
        if st[0] != 'block':
            continue
        b = st[1]
        H = b.first_filter.nn[0].weight.shape[0]
        if not edge_mask_supported(H) or st[2].n <= 1:
            return False
        pp = b._prepacked
        if pp is not None:
            fsp, bsp = block_split_modes(forward_precision(b.unbounded_input), b16, b.dim_out)
            if len(pp) < 7 or pp[6] != b16 or pp[2] != fsp or pp[3] != bsp:
                return False
        packed.append(pp is not None)
    return bool(packed) and any(packed) == all(packed)


class NetFn(torch.autograd.Function):
    """Every fused block and pool / unpool step of the network's graph part as ONE autograd node and one C call per direction
    (stin_net_fwd / _bwd: loops over the per-op entry points, same kernels in the same order -> bit-identical to the per-op
    nodes).  All tensors backward needs live in one arena allocation, the op table is packed on the host (480 bytes per op).
    args: x, meta = (steps, prec list), then the flat parameters (W1, b1, W2, b2, Ws, bs) of every block in step order."""

    calls = 0

    @staticmethod
    def forward(ctx, x, meta, *params):
        NetFn.calls += 1
        steps, need_grad = meta
        lib = _lib.load()
        x, _ = _mat(x)
        dev, dt = x.device, x.dtype
        b16 = dt == torch.bfloat16
        sfx = '_bf16' if b16 else '_f32'
        es = x.element_size()
        pad = 8 if b16 else 4
        N0, Cin0 = x.shape
        Cp0 = (Cin0 + pad - 1) // pad * pad
        if Cp0 != Cin0:                                          # (the 10-channel network input -> 12; bf16: 16)
            xp = torch.empty(N0, Cp0, dtype=x.dtype, device=x.device)
            _call('stin_pad_rows' + sfx, _ptr(x), x.stride(0), N0, Cin0, Cp0, _ptr(xp), _stream(x))   # one launch (F.pad: fill + copy)
        else:
            xp = x
        # ---- pass 1: shapes and arena layout
        plan, off, pi = [], 0, 0
        n_rows, width = N0, Cin0

        # (round 6) no-grad forward: nothing is kept for a backward pass, so every block's temporaries (Y, hE, agg, statistics, unpacked
        # weights) share ONE region and the op outputs ping-pong between two - the arena of an evaluation pass is the largest block's
        # working set instead of the sum over blocks (200 704 vertices: 0.5 GB instead of 3 GB), and it stays cache-warm
        toff, tmax, omax = 0, 0, 0

        def take(nbytes, kind='keep'):
            nonlocal off, toff, tmax, omax
            if not need_grad and kind == 'tmp':
                o = toff
                toff = _align256(toff + nbytes)
                tmax = max(tmax, toff)
                return ('tmp', o)
            if not need_grad and kind == 'out':
                omax = max(omax, _align256(nbytes))
                return ('out', 0)
            o = off
            off = _align256(off + nbytes)
            return o
        for si, stp in enumerate(steps):
            toff = 0
            last = si == len(steps) - 1
            if stp[0] == 'block':
                blk, edges, groups = stp[1], stp[2], stp[3]
                W1, b1, W2, b2, Ws, bs = params[pi:pi + 6]
                pi += 6
                H, Cout = W1.shape[0], W2.shape[0]
                Cin = width
                Cp = (Cin + pad - 1) // pad * pad
                sc = Ws is not None
                ti = trans_inv_mode(blk.first_filter.trans_inv, b16, H)
                Yw = block_yw(H, Cout, sc, ti)
                B = groups.B
                prec = forward_precision(blk.unbounded_input)
                fsp, bsp = block_split_modes(prec, b16, Cout)
                pp = blk._prepacked
                ws_bytes = lib.stin_edgeconv_block_fwd_workspace_bytes(Cin, Cp, H, Cout, int(sc), B)
                d = dict(kind=OP_BLOCK, N=n_rows, Cin=Cin, Cp=Cp, H=H, Cout=Cout, sc=sc, Yw=Yw, B=B, prec=prec, fsp=fsp, bsp=bsp, pp=pp,
                         ws_bytes=ws_bytes, edges=edges, groups=groups, ti=ti, eps=float(blk.first_norm.eps),
                         params=(W1, b1, W2, b2, Ws, bs))
                d['oY'] = take(n_rows * Yw * es, 'tmp')
                d['oH'] = take(n_rows * (H + pad) * es, 'tmp')
                # (round 6) a forward nobody differentiates - torch.no_grad(), the reference's validation loop - keeps no ReLU mask
                d['oM'] = take(max(edges.n_edges, 1) * (H // 32) * 4) if need_grad else None
                d['oA'] = take(n_rows * Cout * es, 'tmp')
                d['oS'] = take(2 * B * Cout * 4, 'tmp')
                if pp is None:
                    d['oW'] = take((Yw * Cp + H * Cout) * 4, 'tmp')
                    d['oWS'] = take(ws_bytes, 'tmp')
                elif pp[0].numel() < ws_bytes:
                    raise RuntimeError('NetFn: prepacked workspace too small for this batch (PackSet built for another batch size)')
                d['oO'] = None if last else take(n_rows * Cout * es, 'out')
                width = Cout
            else:
                pool = stp[1]
                if stp[0] == 'pool':
                    d = dict(kind=OP_POOL_MAX, pool=pool, n_in=pool.n_fine, n_out=pool.n_coarse, C=width)
                    d['oArg'] = take(pool.n_coarse * width * 4, 'tmp')
                else:
                    d = dict(kind=OP_UNPOOL, pool=pool, n_in=pool.n_coarse, n_out=pool.n_fine, C=width)
                n_rows = d['n_out']
                d['oO'] = None if last else take(n_rows * width * es, 'out')
            plan.append(d)
        if not need_grad:                                        # resolve the shared regions: [kept | out 0 | out 1 | temporaries]
            o_base, t_base, flip = off, off + 2 * omax, 0
            for d in plan:
                for k, v in d.items():
                    if isinstance(v, tuple) and len(v) == 2 and v[0] == 'tmp':
                        d[k] = t_base + v[1]
                    elif isinstance(v, tuple) and len(v) == 2 and v[0] == 'out':
                        d[k] = o_base + flip * omax
                        flip ^= 1
            off = t_base + tmax
        arena = torch.empty(off, dtype=torch.uint8, device=dev)
        out = torch.empty(n_rows, width, dtype=dt, device=dev)
        base, p_out = _ptr(arena), _ptr(out)
        assert base % 256 == 0
        # ---- pass 2: the op table
        stc = _net_struct()
        blob = []
        xin, ldx = _ptr(xp), xp.stride(0)
        for d in plan:
            o = p_out if d['oO'] is None else base + d['oO']
            if d['kind'] == OP_BLOCK:
                W1, b1, W2, b2, Ws, bs = d['params']
                pp = d['pp']
                if pp is not None:
                    p_wcatT, p_w2T, p_ws, flag = _ptr(pp[4]), _ptr(pp[5]), _ptr(pp[0]), d['fsp'] | BLOCK_PACKED
                else:
                    p_wcatT = base + d['oW']
                    p_w2T, p_ws, flag = p_wcatT + d['Yw'] * d['Cp'] * 4, base + d['oWS'], d['fsp']
                d['p_wcatT'], d['p_w2T'] = p_wcatT, p_w2T
                g, cd = d['groups'], d['edges'].by_dst
                H, Cout, B = d['H'], d['Cout'], d['B']
                d['x'], d['ldx'], d['out'] = xin, ldx, o
                blob.append(stc.pack(OP_BLOCK, d['Cin'], d['Cp'], H, Cout, int(d['sc']), int(d['ti']), int(d['prec']), flag, d['bsp'], B,
                                     int(g.quirk), 0, 0, 0, 0, d['eps'], 0,
                                     d['N'], d['N'], ldx, Cout, 0, d['Yw'], H + pad, d['ws_bytes'], 0,
                                     xin, o, 0,
                                     _ptr(W1.contiguous()), _ptr(b1), _ptr(W2.contiguous()), _ptr(b2), _ptr(Ws), _ptr(bs), p_wcatT, p_w2T, p_ws,
                                     _ptr(cd.rowptr), _ptr(cd.col), 0, 0, 0, 0,
                                     _ptr(g.ptr_sum), 0, _ptr(g.gid), 0, _ptr(g.inv_cnt),
                                     base + d['oY'], base + d['oH'], (base + d['oM']) if need_grad else 0, base + d['oA'], base + d['oS'],
                                     base + d['oS'] + B * Cout * 4, 0, 0,
                                     0, 0, 0, 0, 0, 0, 0, 0, 0,
                                     *KernelTimer.edge_events('stin_edge_relu_mean_fwd' + sfx, (d['N'], d['edges'].n_edges, H))))
                xin, ldx = o, Cout
            else:
                pool, C = d['pool'], d['C']
                d['x'], d['ldx'], d['out'] = xin, ldx, o
                ch = pool.children
                blob.append(stc.pack(d['kind'], C, C, 0, C, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0,
                                     d['n_out'], d['n_in'], ldx, C, 0, 0, 0, 0, 0,
                                     xin, o, 0,
                                     0, 0, 0, 0, 0, 0, 0, 0, 0,
                                     _ptr(ch.rowptr), _ptr(ch.col), 0, 0, 0, 0,
                                     0, 0, 0, 0, 0,
                                     0, 0, 0, 0, 0, 0,
                                     (base + d['oArg']) if d['kind'] == OP_POOL_MAX else 0, _ptr(pool.trace),
                                     0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0))
                xin, ldx = o, C
        import ctypes
        buf = ctypes.create_string_buffer(b''.join(blob), len(plan) * stc.size)
        _call('stin_net_fwd', int(b16), buf, len(plan), _stream(x))
        if not need_grad:
            return out
        ctx.save_for_backward(xp, arena)
        ctx.plan = plan
        ctx.cin0 = Cin0
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, g):
        xp, arena = ctx.saved_tensors
        plan, params = ctx.plan, ctx.params
        lib = _lib.load()
        dev, dt = xp.device, xp.dtype
        b16 = dt == torch.bfloat16
        sfx = '_bf16' if b16 else '_f32'
        es = xp.element_size()
        pad = 8 if b16 else 4
        g, ldg = _mat(g)
        _same(xp, g)
        base = _ptr(arena)
        need_dx = ctx.needs_input_grad[0]
        blocks = [d for d in plan if d['kind'] == OP_BLOCK]
        # gradients: straight into an accepting TrainStep bucket (all blocks or none), else fresh tensors handed to autograd
        # (probe every block BEFORE any bucket bookkeeping changes: with a frozen or unslotted parameter in one block - a frozen
        # decoder, say - the whole node hands fresh tensors to autograd instead of leaving `written` half set and raising)
        direct = []
        if all(_direct_grad_views(d['params'], dry_run=True) is not None for d in blocks):
            direct = [_direct_grad_views(d['params']) for d in blocks]
        grads = None
        if len(direct) != len(blocks):
            direct = []
            grads = []
            for d in blocks:
                W1, b1, W2, b2, Ws, bs = d['params']
                H, Cout = d['H'], d['Cout']
                grads += [torch.empty(W1.shape, dtype=torch.float32, device=dev),
                          torch.empty(H, dtype=torch.float32, device=dev) if b1 is not None else None,
                          torch.empty(Cout, H, dtype=torch.float32, device=dev),
                          torch.empty(Cout, dtype=torch.float32, device=dev) if b2 is not None else None,
                          torch.empty(Ws.shape, dtype=torch.float32, device=dev) if Ws is not None else None,
                          torch.empty(Cout, dtype=torch.float32, device=dev) if bs is not None else None]
        # scratch: input gradients ping-pong between two buffers of the largest size; every block its own backward workspace
        # (the side stream reads it after this call has returned)
        dx_bytes, ws_off, off = 0, [], 0
        for i, d in enumerate(plan):
            if d['kind'] == OP_BLOCK:
                if i > 0 or need_dx:
                    dx_bytes = max(dx_bytes, d['N'] * d['Cp'] * es)
                d['bwd_ws_bytes'] = lib.stin_edgeconv_block_bwd_workspace_bytes(d['N'], d['Cp'], d['H'], d['Cout'], int(d['sc']), d['B'], int(b16))
                ws_off.append(off)
                off = _align256(off + d['bwd_ws_bytes'])
            else:
                dx_bytes = max(dx_bytes, d['n_in'] * d['C'] * es)
        dx_bytes = _align256(dx_bytes)
        scratch = torch.empty(2 * dx_bytes + off, dtype=torch.uint8, device=dev)
        p_scr = _ptr(scratch)
        p_ws = p_scr + 2 * dx_bytes
        dx0 = None
        d0 = plan[0]
        if need_dx:
            dx0 = torch.empty(d0['N'] if d0['kind'] == OP_BLOCK else d0['n_in'], d0['Cp'] if d0['kind'] == OP_BLOCK else d0['C'],
                              dtype=dt, device=dev)
        all_params = [p for d in blocks for p in d['params']]
        side_ok = (USE_WGRAD_STREAM and WGRAD_DEFER_JOIN and
                   (bool(direct) or (_plain_autograd_may_defer() and all(
                       p is None or (p.is_leaf and p.grad is None and not p._backward_hooks and
                                     not getattr(p, '_post_accumulate_grad_hooks', None)) for p in all_params))))
        use = [side_ok and WGRAD_MIN_WORK <= float(d['N']) * d['Yw'] * d['Cp'] <= WGRAD_MAX_WORK for d in blocks]
        side_stream, any_side = 0, any(use)
        evs = [(0, 0)] * len(blocks)
        # Overlapped gradient all-reduce (train_step.FlatGradBucket.enable_overlap, round 4): every block records its ev_done when
        # its parameter gradients are written - on the weight-gradient stream, or on the compute stream for a block that does
        # not use it - INSIDE the C call's kernel sequence, and the bucket hands each completed segment to RCCL behind exactly
        # those events: the reduction of the decoder's gradients runs while the encoder's backward kernels are still queued.
        bucket = None
        if direct:
            for p in all_params:
                if p is not None:
                    bucket = p._stin_slot[0]
                    break
        seg_events = bucket is not None and bucket.wants_block_events()
        done_ev = [None] * len(blocks)
        if any_side or seg_events:
            side = _wgrad_side(dev)
        if any_side:
            side.hold.append((scratch, xp, arena, g))
            side_stream = side.stream.cuda_stream
        if any_side or seg_events:
            for bi in reversed(range(len(blocks))):      # in BACKWARD order: side.last_done = the event recorded last
                if use[bi]:
                    tri = side.next_events()
                    evs[bi] = (tri[1].cuda_event, tri[2].cuda_event)
                    done_ev[bi] = tri[2]
                elif seg_events:
                    tri = side.next_events(track=False)
                    evs[bi] = (0, tri[2].cuda_event)
                    done_ev[bi] = tri[2]
            if grads is not None:
                for t in grads:
                    if t is not None:
                        t.record_stream(side.stream)
        stc = _net_struct()
        blob, bi = [], 0
        for i, d in enumerate(plan):
            p_dx = (_ptr(dx0) if i == 0 else p_scr + (i & 1) * dx_bytes)
            if i == 0 and not need_dx:
                p_dx = 0
            if d['kind'] == OP_BLOCK:
                e, gr = d['edges'], d['groups']
                cs = e.by_src
                H, Cout, B = d['H'], d['Cout'], d['B']
                gs = direct[bi] if direct else grads[6 * bi:6 * bi + 6]
                ev_dy, ev_done = evs[bi]
                blob.append(stc.pack(OP_BLOCK, d['Cin'], d['Cp'], H, Cout, int(d['sc']), int(d['ti']), 0, 0, d['bsp'], B,
                                     int(gr.quirk), int(use[bi]), 0, 0, 0, d['eps'], 0,
                                     d['N'], d['N'], d['ldx'], Cout, d['Cp'], d['Yw'], H + pad, 0, d['bwd_ws_bytes'],
                                     d['x'], d['out'], p_dx,
                                     0, 0, 0, 0, 0, 0, d['p_wcatT'], d['p_w2T'], 0,
                                     _ptr(e.by_dst.rowptr), 0, _ptr(cs.rowptr), _ptr(cs.col), _ptr(e.xslot), _ptr(e.w_src),
                                     0, _ptr(gr.ptr_true), _ptr(gr.gid), _ptr(gr.sid if gr.quirk else None), _ptr(gr.inv_cnt),
                                     base + d['oY'], base + d['oH'], base + d['oM'], base + d['oA'], base + d['oS'],
                                     base + d['oS'] + B * Cout * 4, 0, 0,
                                     _ptr(gs[0]), _ptr(gs[1]), _ptr(gs[2]), _ptr(gs[3]), _ptr(gs[4]), _ptr(gs[5]),
                                     p_ws + ws_off[bi], ev_dy, ev_done,
                                     *KernelTimer.edge_events('stin_edge_relu_mean_bwd_mask' + ('_ti' if d['ti'] == TI_MODE_COMPACT else '') + sfx,
                                                              (d['N'], e.n_edges, H))))
                bi += 1
            else:
                pool, C = d['pool'], d['C']
                ch = pool.children
                blob.append(stc.pack(d['kind'], C, C, 0, C, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0,
                                     d['n_out'], d['n_in'], d['ldx'], C, C, 0, 0, 0, 0,
                                     d['x'], d['out'], p_dx,
                                     0, 0, 0, 0, 0, 0, 0, 0, 0,
                                     _ptr(ch.rowptr), _ptr(ch.col), 0, 0, 0, 0,
                                     0, 0, 0, 0, 0,
                                     0, 0, 0, 0, 0, 0,
                                     (base + d['oArg']) if d['kind'] == OP_POOL_MAX else 0, _ptr(pool.trace),
                                     0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0))
        import ctypes
        buf = ctypes.create_string_buffer(b''.join(blob), len(plan) * stc.size)
        _call('stin_net_bwd', int(b16), buf, len(plan), _ptr(g), ldg, int(PREC_BWD), _stream(xp), side_stream)
        if any_side:
            _wgrad_deferred_join(dev, all_params if not direct else (), () if direct else grads)
        if dx0 is not None and dx0.shape[1] != ctx.cin0:
            dx0 = dx0[:, :ctx.cin0]
        if direct:
            sd = _WGRAD_SIDE.get(dev.index if dev.index is not None else torch.cuda.current_device())
            slot = None
            for p in all_params:
                if p is not None:
                    slot = p._stin_slot[0]
                    break
            if seg_events:                   # completed segments go to RCCL now, each behind its own blocks' events
                slot.blocks_done([([p._stin_slot[1] for p in d['params'] if p is not None], done_ev[bi]) for bi, d in enumerate(blocks)])
            else:
                slot.block_done(sd.last_done if (sd is not None and sd.hold) else None)
            return (dx0, None) + (None,) * len(params)
        return (dx0, None) + tuple(grads)


def run_net(x, steps):
    """x through `steps` (see net_eligible) as one NetFn node."""
    params = []
    for st in steps:
        if st[0] == 'block':
            b = st[1]
            lin1, lin2 = b.first_filter.nn[0], b.first_filter.nn[2]
            sc = b.shortcut if b.dim_in != b.dim_out else None
            params += [lin1.weight, lin1.bias, lin2.weight, lin2.bias, None if sc is None else sc.weight, None if sc is None else sc.bias]
    # need_grad is decided HERE: inside an autograd.Function's forward the grad mode is always off
    need_grad = torch.is_grad_enabled() and (x.requires_grad or any(p is not None and p.requires_grad for p in params))
    return NetFn.apply(x, (steps, need_grad), *params)


class EdgeReluMeanFn(torch.autograd.Function):
    """h = mean_j ReLU(A_i + B_j) as a standalone differentiable op."""

    @staticmethod
    def forward(ctx, A, B, edges):
        A, _ = _mat(A)
        B, _ = _mat(B)
        out = torch.empty(A.shape[0], A.shape[1], dtype=torch.float32, device=A.device)
        edge_relu_mean_fwd(A, B, edges.by_dst, out)
        ctx.save_for_backward(A, B)
        ctx.edges = edges
        return out

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        g, _ = _mat(g)
        dA = torch.empty_like(A)
        dB = torch.empty_like(B)
        edge_relu_mean_bwd_dst(A, B, g, ctx.edges.by_dst, dA)
        edge_relu_mean_bwd_src(A, B, g, ctx.edges.inv_deg, ctx.edges.by_src, dB)
        return dA, dB, None


def _cols_axpy_rowmask(dst, src, rowptr, c0, c1, alpha):
    """dst[:, c0:c1] += alpha * src[:, c0:c1] on the rows that have an in-edge, in place (stin_cols_axpy_rowmask_*)."""
    _same(dst, src)
    _call('stin_cols_axpy_rowmask' + _sfx(dst), _ptr(dst), dst.stride(0), _ptr(src), src.stride(0), _ptr(rowptr), dst.shape[0], int(c0),
          int(c1), float(alpha), _stream(dst))


class NeighborMeanFn(torch.autograd.Function):
    """agg_i = mean_{j in N(i)} x_j (SAGEConv aggregation; sum for mean=False).  sub_cols = (c0, c1): the translation-invariant
    message x_j[:, c0:c1] - x_i[:, c0:c1] on those columns (models/modules/sage_conv_filter.py:87-90), i.e.
    agg[:, c0:c1] -= x_i[:, c0:c1] on rows with an in-edge - one in-place HIP pass per direction."""

    @staticmethod
    def forward(ctx, x, edges, mean, sub_cols=None):
        ctx.edges, ctx.mean, ctx.sub_cols = edges, mean, sub_cols
        agg = segment_sum(x, edges.by_dst.rowptr, edges.by_dst.col, edges.n, mean=mean)
        if sub_cols is not None:
            xm = x if (x.dim() == 2 and x.stride(1) == 1) else x.contiguous()
            _cols_axpy_rowmask(agg, xm, edges.by_dst.rowptr, sub_cols[0], sub_cols[1], -1.0)
        return agg

    @staticmethod
    def backward(ctx, g):
        e = ctx.edges
        gs = g * e.inv_deg.view(-1, 1) if ctx.mean else g
        dx = segment_sum(gs, e.by_src.rowptr, e.by_src.col, e.n, mean=False)
        if ctx.sub_cols is not None:
            gm = g if (g.dim() == 2 and g.stride(1) == 1) else g.contiguous()
            _cols_axpy_rowmask(dx, gm, e.by_dst.rowptr, ctx.sub_cols[0], ctx.sub_cols[1], -1.0)
        return dx, None, None, None


class ScatterAddFn(torch.autograd.Function):
    """out[n] = sum_{e: index[e]=n} src[e]  (torch_scatter.scatter_sum with a prebuilt CSR)."""

    @staticmethod
    def forward(ctx, src, csr):
        ctx.csr = csr
        return segment_sum(src, csr.rowptr, csr.perm, csr.n_rows, mean=False)

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError('ScatterAddFn is forward-only (benchmark / metrics op)')


class PoolMaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pool):
        x, ldx = _mat(x)
        C = x.shape[1]
        out = torch.empty(pool.n_coarse, C, dtype=x.dtype, device=x.device)
        arg = torch.empty(pool.n_coarse, C, dtype=torch.int32, device=x.device)
        _call('stin_pool_max_fwd' + _sfx(x), _ptr(x), ldx, _ptr(pool.children.rowptr), _ptr(pool.children.col), pool.n_coarse, C,
              _ptr(out), C, _ptr(arg), _stream(x))
        ctx.save_for_backward(arg)
        ctx.pool = pool
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        pool = ctx.pool
        g, ldg = _mat(g)
        C = g.shape[1]
        gx = torch.empty(pool.n_fine, C, dtype=g.dtype, device=g.device)
        _call('stin_pool_max_bwd' + _sfx(g), _ptr(g), ldg, _ptr(arg), _ptr(pool.trace), pool.n_fine, C, _ptr(gx), C, _stream(g))
        return gx, None


class PoolMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pool):
        ctx.pool = pool
        return segment_sum(x, pool.children.rowptr, pool.children.col, pool.n_coarse, mean=True)

    @staticmethod
    def backward(ctx, g):
        return gather_rows(g, ctx.pool.trace, ctx.pool.inv_count), None


class UnpoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pool):
        ctx.pool = pool
        return gather_rows(x, pool.trace)

    @staticmethod
    def backward(ctx, g):
        p = ctx.pool
        return segment_sum(g, p.children.rowptr, p.children.col, p.n_coarse, mean=False), None


class PermuteRowsFn(torch.autograd.Function):
    """y = x[idx] for a PERMUTATION idx (inv = its inverse): the two gathers at the model's boundary when the plan has
    renumbered the vertices (plan.GraphPlan._ensure_order); backward = the gather with the inverse."""

    @staticmethod
    def forward(ctx, x, idx, inv):
        ctx.inv = inv
        return gather_rows(x, idx)

    @staticmethod
    def backward(ctx, g):
        return gather_rows(g.contiguous(), ctx.inv), None, None


class InstanceNormActResFn(torch.autograd.Function):
    """y = res + act(InstanceNorm(x)) (res optional, act = ELU or identity)."""

    @staticmethod
    def forward(ctx, x, res, groups, act, eps=EPS):
        x, _ = _mat(x)
        mean, rstd = instance_stats(x, groups, eps)
        y = norm_act_res_fwd(x, mean, rstd, groups, res=res, act=act)
        ctx.save_for_backward(x, mean, rstd)
        ctx.groups, ctx.act, ctx.has_res = groups, act, res is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, mean, rstd = ctx.saved_tensors
        dx = instance_norm_act_bwd(x, g, mean, rstd, ctx.groups, act=ctx.act)
        return dx, (g if ctx.has_res else None), None, None, None


class MaskedL1LossFn(torch.autograd.Function):
    """The inpainting trainer's loss in one kernel: pred = where(mask > 0, out, color);
    loss = mean(|pred - color| * 0.99^mask) (reference trainers/inpainting3d_trainer.py:127-137).  The gradient
    w.r.t. `out` is produced by the same pass and only scaled in backward."""

    @staticmethod
    def forward(ctx, out, color, mask, use_weight):
        lib = _lib.load()
        out, _ = _mat(out)
        color, _ = _mat(color)
        out, color = out.contiguous(), color.contiguous()
        N, C = out.shape
        m = mask.reshape(-1).contiguous()
        if m.dtype != torch.int64:
            m = m.long()
        loss = torch.empty((), dtype=torch.float32, device=out.device)
        grad = torch.empty_like(out)
        ws_bytes = lib.stin_masked_l1_workspace_bytes(N, C)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=out.device)
        _call('stin_masked_l1_loss_f32', _ptr(out), _ptr(color), _ptr(m), N, C, int(use_weight), _ptr(loss), _ptr(grad),
              _ptr(ws), ws_bytes, _stream(out))
        ctx.save_for_backward(grad)
        return loss

    # a caller that differentiates the loss itself with weight 1 can pass THIS tensor as the seed (`loss.backward(unit_seed(dev))`):
    # backward then returns the stored gradient as it is - no ones-fill for the seed and no [N, C] multiply (two launches)
    _UNIT = {}

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        u = MaskedL1LossFn._UNIT.get(g.device.index)
        if u is not None and g.data_ptr() == u.data_ptr():
            return grad, None, None, None
        return grad * g, None, None, None


def unit_seed(device):
    """The cached scalar 1.0 of `device` that MaskedL1LossFn.backward recognises (see there)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    u = MaskedL1LossFn._UNIT.get(idx)
    if u is None:
        u = MaskedL1LossFn._UNIT[idx] = torch.ones((), dtype=torch.float32, device=device)
    return u


def masked_l1_loss(out, color, mask, use_weight=True):
    return MaskedL1LossFn.apply(out, color, mask, use_weight)


def adam_step(p, g, m, v, vmax, lr, beta1, beta2, eps, weight_decay, step, amsgrad=True):
    """torch.optim.Adam(amsgrad) on flat fp32 buffers, one launch."""
    _call('stin_adam_f32', _ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(vmax), p.numel(), float(lr), float(beta1), float(beta2),
          float(eps), float(weight_decay), int(step), int(amsgrad), _stream(p))


class SliceSumFn(torch.autograd.Function):
    """[N, C] -> [B, C] sums over contiguous row ranges (SingleBatchGraphNorm statistics)."""

    @staticmethod
    def forward(ctx, x, groups):
        ctx.groups = groups
        return colreduce(RED_SUM, x, groups, groups.ptr_sum)

    @staticmethod
    def backward(ctx, g):
        gr = ctx.groups
        if gr.sid is None:
            return g.expand(gr.n_rows, -1).contiguous(), None
        return g.index_select(0, gr.sid.long()), None
