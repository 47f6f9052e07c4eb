"""CPU oracle for the STINet graph-convolution hot path (fp32, torch CPU).

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this file, and only as the
checker / reported CPU baseline - never as a product code path.

It is an op-for-op, UNFUSED restatement of what the reference executes through
PyG / torch_scatter on CPU (gather x_j, x_i -> sub -> cat -> Linear -> ReLU ->
Linear -> scatter_add / count / divide; arg-first scatter_max pool; gather
unpool; F.instance_norm; ELU; residual), in the reference's op ORDER so that the
fp32 summation order matches the PyG CPU path.  Each function cites the
reference file:line it follows (paths relative to /root/reference).

Pinning: oracle/make_golden.py imports the reference's own classes in the build
container (under oracle/pyg_shim, a restatement of the absent, unpinned
third-party torch_geometric 2.0.x / torch_scatter 2.0.9 ops) and writes
tests/golden/*.npz; tests/test_oracle_golden.py checks this file against them.
The reference repo holds no tests or golden vectors of its own for this path
(SURVEY.md §4), and the third-party ops cannot be run here, so parity is pinned
to "reference composition x restated PyG op semantics" - see DESIGN.md §Oracle.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .scatter_ops import scatter_max, scatter_mean, scatter_sum


# --------------------------------------------------------------------------- ops
def edge_conv(x, edge_index, w1, b1, w2, b2, trans_inv=False):
    """PyG EdgeConv(nn=Seq(Lin, ReLU, Lin), aggr='mean').

    models/modules/edge_conv_filter.py:46-57 (MLP + aggr='mean');
    PyG EdgeConv.message = nn(cat[x_i, x_j - x_i]);
    models/modules/edge_conv_translation_invariance.py:19-21 (nn(x_j - x_i)).
    flow source_to_target: edge_index[0] = source j, edge_index[1] = target i.
    """
    src, dst = edge_index[0], edge_index[1]
    x_j = x.index_select(0, src)
    x_i = x.index_select(0, dst)
    feat = (x_j - x_i) if trans_inv else torch.cat([x_i, x_j - x_i], dim=-1)
    msg = F.linear(F.relu(F.linear(feat, w1, b1)), w2, b2)
    return scatter_mean(msg, dst, dim=0, dim_size=x.shape[0])


def sage_conv(x, edge_index, wl, bl, wr, trans_inv=False):
    """PyG SAGEConv (mean) as wrapped by models/modules/sage_conv_filter.py:122-138;
    SAGEConvTransInv.message (:87-90) subtracts x_i[:, 3:9] from x_j[:, 3:9]."""
    src, dst = edge_index[0], edge_index[1]
    x_j = x.index_select(0, src)
    if trans_inv:
        x_i = x.index_select(0, dst)
        x_j = torch.cat([x_j[:, :3], x_j[:, 3:9] - x_i[:, 3:9], x_j[:, 9:]], dim=1)
    agg = scatter_mean(x_j, dst, dim=0, dim_size=x.shape[0])
    return F.linear(agg, wl, bl) + F.linear(x, wr)


def fast_instance_norm(x, batch=None, eps=1e-5):
    """models/modules/fastinstancenorm.py:42-107 (affine=False, no running stats).

    batch None -> F.instance_norm over all rows (:44-49).  batch given -> the
    reference's quirk: per-graph SUMS are taken over equal-length
    ``linspace`` row slices while the divisors are the true per-graph counts and
    the centring/scaling is applied through ``batch`` (:53-82, :98)."""
    if batch is None:
        return F.instance_norm(x.t().unsqueeze(0), None, None, None, None, True, 0.1, eps).squeeze(0).t()
    nb = int(batch.max()) + 1
    ptr = torch.linspace(0, x.shape[0], nb + 1, dtype=torch.int)
    cnt = torch.zeros(nb, dtype=x.dtype).scatter_add_(0, batch, torch.ones(batch.shape[0], dtype=x.dtype))
    cnt = cnt.clamp_(min=1).view(-1, 1)
    mean = torch.stack([x[ptr[i]:ptr[i + 1]].sum(dim=0) for i in range(nb)]) / cnt
    xc = x - mean.index_select(0, batch)
    var = torch.stack([xc[ptr[i]:ptr[i + 1]].pow(2).sum(dim=0) for i in range(nb)]) / cnt
    return xc / (var + eps).sqrt().index_select(0, batch)


def single_batch_graph_norm(x, weight, bias, mean_scale, batch=None, eps=1e-5):
    """models/modules/singlebatchgroupnorm.py:46-71 - note the variance is the mean
    of the RAW x**2 over the linspace slice (:68), not of the centred value."""
    if batch is None:
        batch = x.new_zeros(x.shape[0], dtype=torch.long)
    nb = int(batch.max()) + 1
    ptr = torch.linspace(0, x.shape[0], nb + 1, dtype=torch.int)
    mean = torch.stack([x[ptr[i]:ptr[i + 1]].mean(dim=0) for i in range(nb)]).index_select(0, batch)
    out = x - mean * mean_scale
    var = torch.stack([x[ptr[i]:ptr[i + 1]].pow(2).mean(dim=0) for i in range(nb)])
    std = (var + eps).sqrt().index_select(0, batch)
    return weight * out / std + bias


def pool(x, trace, n_out, kind):
    """models/surfacetextureinpaintingnet.py:382-388."""
    if kind == 'mean':
        return scatter_mean(x, trace, dim=0, dim_size=n_out)
    if kind == 'max':
        return scatter_max(x, trace, dim=0, dim_size=n_out)[0]
    raise ValueError('Unknown pooling type {}'.format(kind))  # reference raises AttributeError here (Q5)


def unpool(x, trace):
    """models/surfacetextureinpaintingnet.py:390-391."""
    return x[trace]


def pool_batch(batch, trace, n_out):
    """models/surfacetextureinpaintingnet.py:421-422 (int64 scatter_max)."""
    return scatter_max(batch, trace, dim=0, dim_size=n_out)[0]


def graph_laplace_variance(x, edge_index):
    """utils/metrics/graph_metrics.py:6-35: aggr='add' of [1, gray] over in-edges."""
    gray = 0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3]
    xi = torch.cat([gray.new_ones(gray.shape[0], 1), gray], dim=1)
    prop = scatter_sum(xi.index_select(0, edge_index[0]), edge_index[1], dim=0, dim_size=x.shape[0])
    lap = prop[:, 1:] - prop[:, 0:1] * gray
    return torch.var(lap, dim=0, unbiased=False)


def graph_total_variation(x, edge_index):
    """utils/metrics/graph_metrics.py:38-42."""
    return torch.abs(x[edge_index[0]] - x[edge_index[1]]).sum() / (x.shape[0] * x.shape[1])


# ------------------------------------------------------------------------ modules
class _Filter(nn.Module):
    """Parameter holder with the reference's key names (``nn.0``, ``nn.2`` /
    ``sage1.lin_l``, ``sage1.lin_r``)."""

    def __init__(self, cin, cout, filter_type, first):
        super().__init__()
        self.kind = filter_type
        self.first = first
        if filter_type.startswith('edgeconv'):
            tinv = first and filter_type == 'edgeconvtransinv'
            self.trans_inv = tinv
            self.nn = nn.Sequential(nn.Linear(cin if tinv else 2 * cin, 2 * cout), nn.ReLU(),
                                    nn.Linear(2 * cout, cout))
        elif filter_type.startswith('sageconv'):
            self.trans_inv = first and filter_type == 'sageconvtransinv'
            self.sage1 = nn.Module()
            self.sage1.lin_l = nn.Linear(cin, cout, bias=True)
            self.sage1.lin_r = nn.Linear(cin, cout, bias=False)
        else:
            raise NotImplementedError('No filter implemented for gcn filter type {}'.format(filter_type))

    def forward(self, x, edge_index):
        if self.kind.startswith('edgeconv'):
            return edge_conv(x, edge_index, self.nn[0].weight, self.nn[0].bias,
                             self.nn[2].weight, self.nn[2].bias, self.trans_inv)
        return sage_conv(x, edge_index, self.sage1.lin_l.weight, self.sage1.lin_l.bias,
                         self.sage1.lin_r.weight, self.trans_inv)


class _Norm(nn.Module):
    def __init__(self, kind, c, momentum=0.1):
        super().__init__()
        self.kind = kind
        if kind == 'graph':
            self.weight = nn.Parameter(torch.ones(c))
            self.bias = nn.Parameter(torch.zeros(c))
            self.mean_scale = nn.Parameter(torch.ones(c))
        elif kind == 'batch':
            self.module = nn.BatchNorm1d(c, momentum=momentum)

    def forward(self, x, batch=None):
        if self.kind == 'instance':
            return fast_instance_norm(x, batch)
        if self.kind == 'graph':
            return single_batch_graph_norm(x, self.weight, self.bias, self.mean_scale, batch)
        if self.kind == 'batch':
            return self.module(x)
        return x


class OracleBlock(nn.Module):
    """GraphResnetBlock, models/surfacetextureinpaintingnet.py:474-521: one conv ->
    norm -> ELU, residual add with a Linear shortcut when Cin != Cout."""

    def __init__(self, cin, cout, filter_type, norm, first=False, is_checkpointed=False, recomputed=False):
        super().__init__()
        # recomputed: the reference wraps this block in torch.utils.checkpoint (:429, :438, :451, :454), i.e. its forward
        # runs a SECOND time inside backward.  Numerically that is a no-op except for BatchNorm's running statistics
        # (norm='batch'): they are updated twice per training step (and num_batches_tracked += 2).
        self.recomputed = recomputed
        self.first_filter = _Filter(cin, cout, filter_type, first)
        mom = math.sqrt(0.1) if (is_checkpointed and norm == 'batch') else 0.1
        self.first_norm = _Norm(norm, cout, mom)
        if cin != cout:
            self.shortcut = nn.Linear(cin, cout)
        self.cin, self.cout = cin, cout

    def forward(self, x, edge_index, batch=None):
        conv = self.first_filter(x, edge_index)
        if self.recomputed and self.first_norm.kind == 'batch' and self.training and conv.requires_grad:
            detached = conv.detach()

            def _recompute(grad, norm=self.first_norm, t=detached, b=batch):
                with torch.no_grad():                               # the recompute pass of the reference's checkpoint, run
                    norm(t, b)                                      # when backward reaches the block: statistics once more
                return grad
            conv.register_hook(_recompute)
        out = F.elu(self.first_norm(conv, batch))
        if self.cin != self.cout:
            x = self.shortcut(x)
        return x + out


class OracleSTINet(nn.Module):
    """SurfaceTextureInpaintingNet, models/surfacetextureinpaintingnet.py:208-471,
    same constructor meaning and state_dict keys; activation checkpointing (:429,
    :438, :451, :454) is a memory device whose only numeric effect - BatchNorm running
    statistics updated by the recompute pass too - is restated in OracleBlock."""

    def __init__(self, input_nc, output_nc, filter_type, ngf=64, norm_type='instance', n_blocks=6,
                 n_levels=2, n_repeated_io_convs=1, pooling_type='mean', checkpoint_bottleneck=False,
                 num_blocks_per_uncheckpointed_block=1, use_label_embedding=False, num_classes=None,
                 num_embedding=None, dilations=None):
        super().__init__()
        assert n_blocks >= 0
        if filter_type not in ('edgeconv', 'edgeconvtransinv', 'sageconv', 'sageconvtransinv'):
            raise NotImplementedError('No filter implemented for gcn filter type {}'.format(filter_type))
        norm = norm_type if norm_type in ('batch', 'instance', 'graph') else 'none'
        self.pooling_type = pooling_type
        self.dilations = list(dilations) if dilations is not None else [1] * n_blocks
        if use_label_embedding:  # created, never used by forward (:277-278, :409-410)
            self.label_embedding = nn.Embedding(num_classes, num_embedding, padding_idx=0)
        mk = lambda ci, co, **k: OracleBlock(ci, co, filter_type, norm, **k)  # noqa: E731
        self.input_blocks = nn.ModuleList(
            [mk(input_nc, ngf if i == n_repeated_io_convs - 1 else input_nc, first=(i == 0))
             for i in range(n_repeated_io_convs)])                                           # :281-313
        enc = []
        for i in range(n_levels):                                                            # :316-325
            cin = ngf * 2 ** i + (num_embedding if (i == 0 and use_label_embedding) else 0)
            enc.append(mk(cin, ngf * 2 ** (i + 1), recomputed=True))
        self.encoder_blocks = nn.ModuleList(enc)
        w = ngf * 2 ** n_levels
        self.bottleneck_blocks = nn.ModuleList(
            [mk(w, w, is_checkpointed=checkpoint_bottleneck,
                recomputed=checkpoint_bottleneck and (i + 1) % num_blocks_per_uncheckpointed_block == 0)
             for i in range(n_blocks)])                                                      # :327-331, :433-440
        self.decoder_blocks = nn.ModuleList(
            [mk(ngf * 2 ** (n_levels - i), ngf * 2 ** (n_levels - i) // 2, recomputed=True) for i in range(n_levels)])  # :333-338
        self.output_blocks = nn.ModuleList([mk(ngf, ngf) for _ in range(n_repeated_io_convs)])  # :342-352
        self.final_linear1 = nn.Linear(ngf, ngf)
        self.final_norm1 = _Norm(norm, ngf)
        self.final_linear2 = nn.Linear(ngf, output_nc)
        for m in self.modules():                                                             # :360-374
            if isinstance(m, nn.Linear) and m.bias is not None:
                nn.init.zeros_(m.bias)

    def forward(self, sample):
        """:398-471."""
        levels = len(self.decoder_blocks) + 1
        out = sample.x
        for blk in self.input_blocks:
            out = blk(out, sample.edge_index)                       # no batch: whole-batch norm (Q1)
        n_per_level = sample.num_vertices.sum(dim=0)
        batch = sample.batch if sample.batch.max() > 0 else None
        for i, blk in enumerate(self.encoder_blocks):
            lvl = i + 1
            trace = sample['hierarchy_trace_index_%d' % lvl]
            if batch is not None:
                batch = pool_batch(batch, trace, int(n_per_level[lvl]))
            out = pool(out, trace, int(n_per_level[lvl]), self.pooling_type)
            out = blk(out, sample['hierarchy_edge_index_%d' % lvl], batch)
        for i, blk in enumerate(self.bottleneck_blocks):
            d = self.dilations[i]
            key = ('hierarchy_dil_%s_edge_index_%d' % (d, levels - 1)) if d > 1 else \
                ('hierarchy_edge_index_%d' % (levels - 1))
            out = blk(out, sample[key], batch)
        for i, blk in enumerate(self.decoder_blocks):
            lvl = i + 1
            trace = sample['hierarchy_trace_index_%d' % (levels - lvl)]
            out = unpool(out, trace)
            if batch is not None:
                batch = batch.index_select(0, trace)
            edges = sample.edge_index if lvl == levels - 1 else \
                sample['hierarchy_edge_index_%d' % (levels - lvl - 1)]
            out = blk(out, edges, batch)
        for blk in self.output_blocks:
            out = blk(out, sample.edge_index)
        out = self.final_linear1(out)
        out = self.final_norm1(out, batch=sample.batch)             # always the batched branch (Q3)
        out = F.elu(out)
        return torch.tanh(self.final_linear2(out))


def define_G(input_nc, output_nc, ngf, filter_type, norm='batch', dilation_order=0, use_dropout=False,
             n_blocks=6, n_levels=2, n_repeated_io_convs=1, init_type='normal', pooling_type='stride',
             io_receptive_field_type='large', checkpoint_bottleneck=False,
             num_blocks_per_uncheckpointed_block=1, use_label_embedding=False, num_classes=None,
             num_embedding=None, dilations=None, init_gain=0.02, gpu_ids=[]):
    """models/surfacetextureinpaintingnet.py:157-199 (graph branch only)."""
    return OracleSTINet(input_nc, output_nc, filter_type, ngf, norm_type=norm, n_blocks=n_blocks,
                        n_levels=n_levels, n_repeated_io_convs=n_repeated_io_convs,
                        pooling_type=pooling_type, checkpoint_bottleneck=checkpoint_bottleneck,
                        num_blocks_per_uncheckpointed_block=num_blocks_per_uncheckpointed_block,
                        use_label_embedding=use_label_embedding, num_classes=num_classes,
                        num_embedding=num_embedding, dilations=dilations)


# ------------------------------------------------------------- train-step harness
def graph_forward(model, data):
    """trainers/inpainting3d_trainer.py:127-129."""
    out = model(data)
    return torch.where((data.mask > 0).expand_as(data.color), out, data.color)


def compute_loss(output, target, weights=None):
    """trainers/inpainting3d_trainer.py:132-137 with criterion = L1Loss(reduction='none')."""
    loss = (output - target).abs()
    if weights is not None:
        loss = loss * torch.pow(0.99, weights.squeeze()).unsqueeze(1)
    return loss.mean()
