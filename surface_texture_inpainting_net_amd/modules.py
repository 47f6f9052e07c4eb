"""Graph-conv filters and norms of the hot path, on the HIP operator layer.

Host-side mirror of the reference's models/modules/*: same factory names, argument
meaning, parameter names (``nn.0``/``nn.2``, ``sage1.lin_l``/``sage1.lin_r``,
``weight``/``bias``/``mean_scale``, ``module.*``) and error behaviour, so that
reference checkpoints load unchanged.  The arithmetic runs in libstin_hip.so.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as SF
from .plan import EdgeSet, NormGroups


def _as_edges(edges, n):
    """EdgeSet, or a raw int64 [2, E] edge_index (external callers) -> EdgeSet."""
    if isinstance(edges, EdgeSet):
        return edges
    bad = torch.zeros(1, dtype=torch.int32, device=edges.device)
    es = EdgeSet(edges, n, bad)
    if int(bad.item()) != 0:
        raise IndexError('edge_index out of range for %d vertices' % n)
    return es


def _as_groups(batch, n, device, linspace_quirk=True):
    """NormGroups, None (single instance), or a raw int64 batch vector -> NormGroups."""
    if isinstance(batch, NormGroups):
        return batch
    if batch is None:
        return NormGroups(n, device)
    nb = int(batch.max()) + 1                      # same host sync the reference does (fastinstancenorm.py:51)
    counts = torch.bincount(batch, minlength=nb).cpu()
    return NormGroups(n, device, batch, counts, linspace_quirk)


class EdgeConv(nn.Module):
    """EdgeConv(nn=Seq(Lin, ReLU, Lin), aggr='mean') - PyG's EdgeConv as the reference
    configures it (models/modules/edge_conv_filter.py:46-57), evaluated in the exact
    per-vertex form  W2 mean_j ReLU(A_i + B_j) + b2 [deg_i > 0]  (DESIGN.md §2)."""
    trans_inv = False

    def __init__(self, nn_seq, aggr='mean'):
        super().__init__()
        if aggr != 'mean':
            raise NotImplementedError("only aggr='mean' is on the STINet path (the linear restructure needs it)")
        self.nn = nn_seq
        self.aggr = aggr

    def hidden(self):
        return self.nn[0].out_features

    def fused_weights(self, shortcut=None):
        """-> (Wcat [2H(+Cout), Cin], bcat, W2e [Cout, H+4]) from the reference-layout parameters."""
        lin1, lin2 = self.nn[0], self.nn[2]
        W1, H = lin1.weight, lin1.out_features
        b1 = lin1.bias if lin1.bias is not None else W1.new_zeros(H)
        if self.trans_inv:                      # nn(x_j - x_i): A = -x W1^T + b1, B = x W1^T
            rows = [-W1, W1]
        else:                                   # nn([x_i, x_j - x_i]): A = x (Wa - Wb)^T + b1, B = x Wb^T
            cin = W1.shape[1] // 2
            rows = [W1[:, :cin] - W1[:, cin:], W1[:, cin:]]
        biases = [b1, b1.new_zeros(H)]
        if shortcut is not None:
            rows.append(shortcut.weight)
            biases.append(shortcut.bias if shortcut.bias is not None else b1.new_zeros(shortcut.out_features))
        b2 = lin2.bias if lin2.bias is not None else W1.new_zeros(lin2.out_features)
        w2e = torch.cat([lin2.weight, b2[:, None], b2.new_zeros(b2.shape[0], 3)], dim=1)
        return torch.cat(rows, 0), torch.cat(biases, 0), w2e

    def forward(self, x, edge_index):
        edges = _as_edges(edge_index, x.shape[0])
        wcat, bcat, w2e = self.fused_weights()
        H = self.hidden()
        prec = getattr(self, 'fwd_precision', None)
        Y = SF.linear(x, wcat, bcat, precision=prec)
        h = SF.EdgeReluMeanFn.apply(Y[:, :H], Y[:, H:], edges)
        has_in = (edges.by_dst.rowptr[1:] > edges.by_dst.rowptr[:-1]).to(h.dtype).unsqueeze(1)
        return SF.linear(h, self.nn[2].weight, precision=prec) + has_in * w2e[:, H]

    def __repr__(self):
        return '{}(nn={}, aggr={})'.format(self.__class__.__name__, self.nn, self.aggr)


class EdgeConvTransInv(EdgeConv):
    """First-layer EdgeConv whose message is nn(x_j - x_i)
    (models/modules/edge_conv_translation_invariance.py:9-24)."""
    trans_inv = True


def get_gcn_filter(input_size, output_size, activation=nn.ReLU, inplace=False, aggregation='mean', bias=True,
                   module=None, double_input=True, with_norm=False):
    """Same arguments as the reference's edge_conv_filter.get_gcn_filter (:10-57)."""
    assert input_size >= 0
    assert output_size >= 0
    if with_norm:
        # BatchNorm1d inside the edge MLP (edge_conv_filter.py:34-44): per-EDGE tensors, see singleconvmeshnet.EdgeConvBN
        from .singleconvmeshnet import EdgeConvBN
        assert aggregation == 'mean' and issubclass(activation, torch.nn.ReLU), 'the reference only builds mean / ReLU'
        if module not in (None, EdgeConv, EdgeConvTransInv):
            raise NotImplementedError('with_norm filters exist for EdgeConv / EdgeConvTransInv')
        return EdgeConvBN(input_size, output_size, trans_inv=(module is EdgeConvTransInv))
    if activation is not nn.ReLU:
        raise NotImplementedError('the fused edge stage implements ReLU only')
    cin = 2 * input_size if double_input else input_size
    if module is None:
        module = EdgeConv
    inner = nn.Sequential(nn.Linear(cin, 2 * output_size, bias=bias), nn.ReLU(inplace=inplace),
                          nn.Linear(2 * output_size, output_size, bias=bias))
    return module(inner, aggr=aggregation)


class SAGEConv(nn.Module):
    """lin_l(mean_j x_j) + lin_r(x_i) (PyG SAGEConv, root_weight=True, normalize=False)."""
    trans_inv = False

    def __init__(self, in_channels, out_channels, normalize=False, root_weight=True, bias=True):
        super().__init__()
        assert not normalize and root_weight
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_l = nn.Linear(in_channels, out_channels, bias=bias)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, x, edge_index, size=None):
        edges = _as_edges(edge_index, x.shape[0])
        # trans_inv message: x_j[:, 3:9] -= x_i[:, 3:9] (models/modules/sage_conv_filter.py:87-90); under the mean this is
        # agg[:, 3:9] - x_i[:, 3:9] * [deg_i > 0] - one in-place HIP pass inside the aggregation node
        agg = SF.NeighborMeanFn.apply(x, edges, True, (3, 9) if self.trans_inv else None)
        prec = getattr(self, 'fwd_precision', None)
        return SF.linear(agg, self.lin_l.weight, self.lin_l.bias, precision=prec) + SF.linear(x, self.lin_r.weight, precision=prec)


class SAGEConvTransInv(SAGEConv):
    trans_inv = True


class SumSAGEConv(nn.Module):
    def __init__(self, module, *args, **kwargs):
        super().__init__()
        self.sage1 = module(*args, **kwargs)

    def forward(self, x, edge_index, size=None):
        self.sage1.fwd_precision = getattr(self, 'fwd_precision', None)
        return self.sage1(x, edge_index, size)


def get_sage_filter(input_size, output_size, activation=None, inplace=False, aggregation='mean', bias=True,
                    module=None, double_input=False):
    """Same arguments as the reference's sage_conv_filter.get_gcn_filter (:102-138)."""
    assert input_size >= 0
    assert output_size >= 0
    if module is None:
        module = SAGEConv
    return SumSAGEConv(module, input_size, output_size, normalize=False, root_weight=True, bias=bias)


class FastInstanceNorm(nn.Module):
    """Per-graph instance norm on [N, C] node features, no affine, no running stats
    (models/modules/fastinstancenorm.py:42-107; the shipped default `norm: instance`)."""

    def __init__(self, in_channels, eps=1e-5, momentum=0.1, affine=False, track_running_stats=False):
        super().__init__()
        if affine or track_running_stats:
            raise NotImplementedError('FastInstanceNorm is used with affine=False, track_running_stats=False')
        self.num_features, self.eps = in_channels, eps
        self.linspace_quirk = True

    def forward(self, x, batch=None):
        groups = _as_groups(batch, x.shape[0], x.device, self.linspace_quirk)
        return SF.InstanceNormActResFn.apply(x, None, groups, False, self.eps)

    def __repr__(self):
        return '{}({})'.format(self.__class__.__name__, self.num_features)


class _GraphNormFn(torch.autograd.Function):
    """y = w (x - a mu[g]) r[g] + b with mu = mean(x), r = 1/sqrt(mean(x^2) + eps) over the linspace row slices of the
    reference (singlebatchgroupnorm.py:46-71) - forward and backward on the library's kernels: slice sums by
    stin_colreduce (sum x; sum x^2 as CSQ around a zero mean), the affine normalisation as stin_norm_act_res_fwd with
    per-group rows [a mu, w r] (+ the bias row), the backward reductions sum G (x - a mu), sum G by the DOT mode and
    dx = (w r)[g] G + k[s] x + m[s] by stin_norm_act_bwd.  No [N, C] framework kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, mean_scale, groups, eps):
        x, _ = SF._mat(x)
        n_rows, C = x.shape
        B = groups.B
        if B == 1:
            inv_n = torch.full((1, 1), 1.0 / float(n_rows), device=x.device)
        else:
            inv_n = 1.0 / (groups.ptr_sum[1:] - groups.ptr_sum[:-1]).to(torch.float32).unsqueeze(1)
        zero = torch.zeros(B, C, dtype=torch.float32, device=x.device)
        mu = SF.colreduce(SF.RED_SUM, x, groups, groups.ptr_sum) * inv_n                    # [B, C] per slice
        q = SF.colreduce(SF.RED_CSQ, x, groups, groups.ptr_sum, mean=zero) * inv_n          # sum x^2: CSQ around a zero mean
        r = torch.rsqrt(q + eps)
        m = (mu * mean_scale).contiguous()
        wr = (r * weight).contiguous()
        u = SF.norm_act_res_fwd(x, m, wr, groups, res=None, act=False)                      # (x - m[g]) * (w r)[g]
        y = u.add_(bias)                                                                    # in place: one row broadcast
        ctx.save_for_backward(x, weight, mean_scale, mu, r, m, wr, inv_n)
        ctx.groups = groups
        return y

    @staticmethod
    def backward(ctx, G):
        x, weight, mean_scale, mu, r, m, wr, inv_n = ctx.saved_tensors
        groups = ctx.groups
        G, _ = SF._mat(G)
        zero = torch.zeros_like(r)
        # per TRUE group (rows with gid == g): T1 = sum G (x - m[g]), S0 = sum G
        T1, S0 = SF.colreduce(SF.RED_DOT_ELU, x, groups, groups.ptr_true, gout=G, mean=m, rstd=zero)
        dbias = S0.sum(0)
        dweight = (r * T1).sum(0)
        dm = -(wr * S0)                                   # d/d(a mu[g])
        dr = weight * T1
        dscale = (dm * mu).sum(0)
        dmu = dm * mean_scale
        dq = -0.5 * r * r * r * dr
        k = (2.0 * dq * inv_n).contiguous()               # applied through the SLICE id: the statistics ran over slices
        mm = (dmu * inv_n).contiguous()
        dx = torch.empty_like(x)
        SF._call('stin_norm_act_bwd' + SF._sfx(x), SF._ptr(x), x.stride(0), SF._ptr(G), G.stride(0), SF._ptr(zero), SF._ptr(zero),
                 SF._ptr(wr), SF._ptr(k), SF._ptr(mm), SF._ptr(groups.gid), SF._ptr(groups.sid), x.shape[0], x.shape[1], 0,
                 SF._ptr(dx), dx.stride(0), SF._stream(x))
        return dx, dweight, dbias, dscale, None, None


class SingleBatchGraphNorm(nn.Module):
    """weight * (x - mean_scale * mean) / sqrt(mean(x^2) + eps) + bias over linspace row
    slices - the reference's GraphNorm variant incl. its raw-x^2 variance
    (models/modules/singlebatchgroupnorm.py:46-71)."""

    def __init__(self, in_channels, eps=1e-5):
        super().__init__()
        self.in_channels, self.eps = in_channels, eps
        self.weight = nn.Parameter(torch.ones(in_channels))
        self.bias = nn.Parameter(torch.zeros(in_channels))
        self.mean_scale = nn.Parameter(torch.ones(in_channels))

    def forward(self, x, batch=None):
        groups = _as_groups(batch, x.shape[0], x.device, True)
        return _GraphNormFn.apply(x, self.weight, self.bias, self.mean_scale, groups, float(self.eps))

    def __repr__(self):
        return '{}({})'.format(self.__class__.__name__, self.in_channels)


class BatchNorm2Param(nn.Module):
    """PyG BatchNorm (BatchNorm1d held as ``.module``) that ignores ``batch``
    (models/surfacetextureinpaintingnet.py:236-241), on the library's kernels: fp64-accumulated column moments, the
    affine normalisation and its backward as one pass each (stin_bn_act_fwd / _bwd), running statistics in one launch.
    recomputed: the reference checkpoints the enclosing block, so backward updates the running statistics once more."""

    def __init__(self, in_channels, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.module = nn.BatchNorm1d(in_channels, eps, momentum, affine, track_running_stats)

    def forward(self, x, batch=None, recomputed=False):
        from .singleconvmeshnet import batch_norm_rows
        return batch_norm_rows(x, self.module, relu=False, recomputed=recomputed)


class Identity(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, x, batch=None):
        return x
