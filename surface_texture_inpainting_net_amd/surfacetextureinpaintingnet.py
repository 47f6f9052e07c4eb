"""SurfaceTextureInpaintingNet on MI355X: the reference's nn.Module surface
(models/surfacetextureinpaintingnet.py: define_G :157-199, SurfaceTextureInpaintingNet
:202-471, GraphResnetBlock :474-521) - same constructor arguments, ``forward(sample)``
contract and state_dict keys - with every graph operation executed by the hand-written
HIP kernels of libstin_hip.so through functional.py.  Drop-in for the graph branch of
``define_G``; the 2-D Conv2d baselines (filter_type conv2d / cfconv2d) are out of scope.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as SF
from . import modules as M
from .plan import EdgeSet, NormGroups, check_deferred, plan_for


class GraphResnetBlock(nn.Module):
    """conv -> norm -> ELU, plus residual (Linear shortcut when Cin != Cout)."""

    def __init__(self, dim_in, dim_out, get_gcn_filter, norm_layer, inplace, use_bias, is_checkpointed=False,
                 module=None, double_input=None):
        super().__init__()
        self.dim_in, self.dim_out = dim_in, dim_out
        self.act = nn.ELU()
        # True where the reference wraps the block in torch.utils.checkpoint (:429, :438, :451, :454): its forward runs a
        # second time inside backward, which numerically only matters for BatchNorm running statistics (norm='batch')
        self.recomputed = False
        self._prepacked = None        # (workspace, wcatT | w2T, fwd_split, bwd_split) of functional.PackSet, set by the network
        # True for a block fed by un-normalised data (the network's first block, or any block of a norm-free network):
        # its forward GEMMs take the range-safe matrix-core path (functional.forward_precision)
        self.unbounded_input = False
        if module is not None:
            self.first_filter = get_gcn_filter(dim_in, dim_out, inplace=inplace, bias=use_bias, module=module,
                                               double_input=double_input)
        else:
            self.first_filter = get_gcn_filter(dim_in, dim_out, inplace=inplace, bias=use_bias)
        if is_checkpointed and issubclass(norm_layer, M.BatchNorm2Param):
            self.first_norm = norm_layer(dim_out, momentum=math.sqrt(0.1))
        else:
            self.first_norm = norm_layer(dim_out)
        if dim_in != dim_out:
            self.shortcut = nn.Linear(dim_in, dim_out)

    def pack_spec(self, B):
        """The arguments of this block's weight pack (functional.PackSet), or None when the block is not the fused kind."""
        if not (isinstance(self.first_filter, M.EdgeConv) and isinstance(self.first_norm, M.FastInstanceNorm)):
            return None
        shortcut = self.shortcut if self.dim_in != self.dim_out else None
        lin1, lin2 = self.first_filter.nn[0], self.first_filter.nn[2]
        return (lin1.weight, lin1.bias, lin2.weight, lin2.bias, None if shortcut is None else shortcut.weight,
                None if shortcut is None else shortcut.bias, self.first_filter.trans_inv,
                SF.forward_precision(self.unbounded_input), int(B))

    def forward(self, x, edges, batch=None):
        n = x.shape[0]
        edges = M._as_edges(edges, n)
        fused = isinstance(self.first_filter, M.EdgeConv) and isinstance(self.first_norm, M.FastInstanceNorm)
        if fused:
            groups = M._as_groups(batch, n, x.device, self.first_norm.linspace_quirk)
            shortcut = self.shortcut if self.dim_in != self.dim_out else None
            lin1, lin2 = self.first_filter.nn[0], self.first_filter.nn[2]
            return SF.EdgeConvBlockFn.apply(x, lin1.weight, lin1.bias, lin2.weight, lin2.bias,
                                            None if shortcut is None else shortcut.weight,
                                            None if shortcut is None else shortcut.bias, edges, groups,
                                            self.first_filter.trans_inv, self.first_norm.eps,
                                            SF.forward_precision(self.unbounded_input), self._prepacked)
        self.first_filter.fwd_precision = SF.forward_precision(self.unbounded_input)
        out = self.first_filter(x, edges)
        res = (SF.linear(x, self.shortcut.weight, self.shortcut.bias, precision=SF.forward_precision(self.unbounded_input))
               if self.dim_in != self.dim_out else x)
        if isinstance(self.first_norm, M.FastInstanceNorm):
            groups = M._as_groups(batch, n, x.device, self.first_norm.linspace_quirk)
            return SF.InstanceNormActResFn.apply(out, res, groups, True, self.first_norm.eps)
        if isinstance(self.first_norm, M.BatchNorm2Param):
            return res + self.act(self.first_norm(out, batch, recomputed=self.recomputed))
        return res + self.act(self.first_norm(out, batch))


class SurfaceTextureInpaintingNet(nn.Module):
    """ResNet-style U-shaped GNN over a mesh hierarchy (see the module docstring)."""

    def __init__(self, input_nc, output_nc, filter_type, ngf=64, norm_type='instance', n_blocks=6, n_levels=2,
                 n_repeated_io_convs=1, pooling_type='mean', checkpoint_bottleneck=False,
                 num_blocks_per_uncheckpointed_block=1, use_label_embedding=False, num_classes=None,
                 num_embedding=None, dilations=None):
        assert (n_blocks >= 0)
        super().__init__()
        if filter_type in ('edgeconv', 'edgeconvtransinv'):
            get_gcn_filter = M.get_gcn_filter
        elif filter_type in ('sageconv', 'sageconvtransinv'):
            get_gcn_filter = M.get_sage_filter
        else:
            raise NotImplementedError('No filter implemented for gcn filter type {}'.format(filter_type))
        if norm_type == 'batch':
            self.norm, self.using_norm = M.BatchNorm2Param, True
        elif norm_type == 'instance':
            self.norm, self.using_norm = M.FastInstanceNorm, True
        elif norm_type == 'graph':
            self.norm, self.using_norm = M.SingleBatchGraphNorm, True
        else:
            self.norm, self.using_norm = M.Identity, False
        self._pack_set, self._pack_key, self._pack_probe, self._tail_wT = None, None, None, None
        self._pooling_type = pooling_type
        self.checkpoint_bottleneck = checkpoint_bottleneck          # accepted for config compatibility: the HIP
        self.num_blocks_per_uncheckpointed_block = num_blocks_per_uncheckpointed_block  # blocks save only per-vertex
        self._use_embedding = use_label_embedding                   # tensors, so no recompute is needed
        self.dilations = list(dilations) if dilations is not None else [1] * n_blocks
        # batched-norm compatibility switch: True reproduces the reference's linspace slices.  The
        # plan's NormGroups carry the choice; blocks called directly with a raw batch tensor use their own
        # FastInstanceNorm.linspace_quirk (default True).
        self.compat_linspace_norm = True
        # storage type of the per-vertex activations and their gradients (parameters, statistics and weight
        # gradients are always fp32): torch.float32 = the reference's numerics (1e-4 bar); torch.bfloat16 = the
        # build's mixed-precision extension for BASELINE configs 3/5 (see set_activation_dtype)
        self.activation_dtype = torch.float32
        # 'sync': an out-of-range index raises IndexError in the forward call that used it (one host sync per new plan);
        # 'deferred': it raises at the next forward / TrainStep call instead and the host never stalls (plan.validate)
        self.plan_validation = 'sync'
        # columns of sample.x that hold the vertex positions: the key of the plan's optional vertex renumbering by locality
        # (plan.GraphPlan._ensure_order; inverted at the boundary of forward()).  The reference's 3-D inpainting features are
        # x = [rgb * known, normal, pos, known] (datasets/scannetcolorgraph_dataloader.py:113-121): columns 6..8 when
        # input_nc == 10.  None = no renumbering.
        self.position_channels = (6, 9) if input_nc == 10 else None
        self._filter_type, self._norm_type = filter_type, norm_type
        inplace, use_bias = False, True
        if self._use_embedding:  # created but never used by forward, as in the reference (:277-278, :409-410)
            self.label_embedding = nn.Embedding(num_classes, num_embedding, padding_idx=0)

        blocks = []
        for i in range(n_repeated_io_convs):
            cout = ngf if i == n_repeated_io_convs - 1 else input_nc
            if i == 0:
                first, double_input = {
                    'edgeconvtransinv': (M.EdgeConvTransInv, False), 'edgeconv': (None, True),
                    'sageconvtransinv': (M.SAGEConvTransInv, False), 'sageconv': (None, False)}[filter_type]
                blocks.append(GraphResnetBlock(input_nc, cout, get_gcn_filter, self.norm, inplace, use_bias,
                                               module=first, double_input=double_input))
            else:
                blocks.append(GraphResnetBlock(input_nc, cout, get_gcn_filter, self.norm, inplace, use_bias))
        self.input_blocks = nn.ModuleList(blocks)

        blocks = []
        for i in range(n_levels):
            cin = ngf * 2 ** i
            if i == 0 and self._use_embedding:
                cin += num_embedding
            blocks.append(GraphResnetBlock(cin, ngf * 2 ** (i + 1), get_gcn_filter, self.norm, inplace, use_bias))
        self.encoder_blocks = nn.ModuleList(blocks)

        width = ngf * 2 ** n_levels
        self.bottleneck_blocks = nn.ModuleList(
            [GraphResnetBlock(width, width, get_gcn_filter, self.norm, inplace, use_bias,
                              is_checkpointed=self.checkpoint_bottleneck) for _ in range(n_blocks)])
        self.decoder_blocks = nn.ModuleList(
            [GraphResnetBlock(ngf * 2 ** (n_levels - i), int(ngf * 2 ** (n_levels - i) / 2), get_gcn_filter, self.norm,
                              inplace, use_bias) for i in range(n_levels)])
        self.output_blocks = nn.ModuleList(
            [GraphResnetBlock(ngf, ngf, get_gcn_filter, self.norm, inplace, use_bias) for _ in range(n_repeated_io_convs)])
        self.final_linear1 = nn.Linear(ngf, ngf, bias=use_bias)
        self.final_norm1 = self.norm(ngf)
        self.final_linear2 = nn.Linear(ngf, output_nc)
        for m in self.modules():
            if isinstance(m, GraphResnetBlock):
                m.unbounded_input = not self.using_norm
        self.input_blocks[0].unbounded_input = True                 # raw vertex features (fp16's range is not guaranteed)
        for blk in list(self.encoder_blocks) + list(self.decoder_blocks):
            blk.recomputed = True
        for i, blk in enumerate(self.bottleneck_blocks):
            blk.recomputed = bool(self.checkpoint_bottleneck) and (i + 1) % self.num_blocks_per_uncheckpointed_block == 0
        for m in self.modules():                                    # reference zeroes every Linear bias (:360-374)
            if isinstance(m, nn.Linear) and m.bias is not None:
                nn.init.zeros_(m.bias)

    def set_activation_dtype(self, dtype):
        """torch.float32 (default) or torch.bfloat16: bf16 STORAGE of activations / activation gradients with fp32
        accumulation, fp32 statistics, fp32 master weights and weight gradients.  The reference has no such mode
        (fp32 only, no autocast): results then agree with the fp32 path to ~1e-2 on the tanh output, not 1e-4."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError('activation dtype must be torch.float32 or torch.bfloat16')
        if dtype == torch.bfloat16 and not (self._filter_type in ('edgeconv', 'edgeconvtransinv')
                                            and self._norm_type == 'instance'):
            raise NotImplementedError('bf16 storage is implemented for the EdgeConv + instance-norm network '
                                      '(the shipped configs), not for filter %s / norm %s'
                                      % (self._filter_type, self._norm_type))
        self.activation_dtype = dtype
        return self

    # -- pooling ------------------------------------------------------------------------
    def _pooling(self, vertex_features, pool_map):
        if self._pooling_type == 'mean':
            return SF.PoolMeanFn.apply(vertex_features, pool_map)
        if self._pooling_type == 'max':
            return SF.PoolMaxFn.apply(vertex_features, pool_map)
        raise ValueError('Unknown pooling type {}'.format(self._pooling_type))

    def _unpooling(self, vertex_features, pool_map):
        return SF.UnpoolFn.apply(vertex_features, pool_map)

    def _pack_weights(self, x, num_graphs):
        """All fused blocks' weight operands in ONE launch (functional.PackSet) instead of one tiny launch at the head of
        every block: 15 launches of ~8 us on the critical path become one.  fp32 and (round 3) bf16 storage, whole-block path only."""
        use = (SF.USE_PACK_MANY and SF.USE_BLOCK_CALL and SF.USE_EDGE_MASK and not SF.KernelTimer.per_kernel_path() and x.is_cuda and
               x.dtype in (torch.float32, torch.bfloat16) and self.norm is M.FastInstanceNorm)
        b16 = x.dtype == torch.bfloat16
        # per-step validity check of the cached set: the data pointers of EVERY packed tensor, read from the modules'
        # parameter tables each step (never from cached Parameter objects) - load_state_dict(assign=True), module surgery
        # (`lin.weight = nn.Parameter(...)`, a replaced filter / shortcut module) and `p.data = ...` on any one of them all
        # change the key.  ~90 dict lookups + data_ptr() calls: ~25 us of host time per step.
        probe = self._pack_probe
        if probe is None:
            probe = self._pack_probe = [b for grp in (self.input_blocks, self.encoder_blocks, self.bottleneck_blocks,
                                                      self.decoder_blocks, self.output_blocks) for b in grp]
        ptrs = []
        for b in probe:
            mods = b._modules
            flt = mods['first_filter']
            seq = flt._modules.get('nn')
            lins = (seq._modules.get('0'), seq._modules.get('2'), mods.get('shortcut')) if seq is not None else (mods.get('shortcut'),)
            for lin in lins:
                if lin is not None:
                    for t in lin._parameters.values():
                        if t is not None:
                            ptrs.append(t.data_ptr())
        tail_w = self.final_linear1._parameters['weight']             # its transpose (backward operand) rides on the pack launch
        ptrs.append(tail_w.data_ptr())
        key = (use, tuple(ptrs), int(num_graphs), SF.PREC_FWD, SF.PREC_BWD, SF.GEMM_W_FRAG, SF.WEIGHT_PRESPLIT, b16)
        if key == self._pack_key:
            if use:
                self._pack_set.run()
            return
        self._pack_key = key
        self._tail_wT = None
        blocks = [b for grp in (self.input_blocks, self.encoder_blocks, self.bottleneck_blocks, self.decoder_blocks,
                                self.output_blocks) for b in grp]
        specs = None
        if use:
            whole = set(id(b) for b in list(self.input_blocks) + list(self.output_blocks))       # norm over the whole batch: B = 1
            specs = [b.pack_spec(1 if id(b) in whole else num_graphs) for b in blocks]
            use = all(sp is not None for sp in specs)
        if not use:
            # (the set itself stays: a captured HIP graph may hold pointers into its buffers, and the per-kernel path of a
            # bracketed bench step only bypasses it)
            self._pack_key = (False,) + key[1:]
            for b in blocks:
                b._prepacked = None
            return
        ps = self._pack_set
        tr = (tail_w,) if (tail_w.is_contiguous() and tail_w.dtype == torch.float32) else ()
        if ps is None or not ps.matches(specs, b16, tr):
            ps = self._pack_set = SF.PackSet(specs, x.device, b16, tr)
        ps.run()
        for b, buf in zip(blocks, ps.buffers):
            b._prepacked = buf
        self._tail_wT = ps.transposed[0] if tr else None

    def _norm_arg(self, plan, level, whole_batch=False):
        """What a block's norm receives: NormGroups for the instance norm; for the other norms the
        reference's `batch` tensor / None."""
        if self.norm is M.FastInstanceNorm:
            g = plan.norm_groups(level, whole_batch)
            return g
        return None if whole_batch else plan.batch_vector(level)

    def _plan_items(self):
        """The CSR structures forward() will ask the plan for: ([(edge key, level), ...], [pool level, ...])."""
        num_levels = len(self.decoder_blocks) + 1
        last = num_levels - 1
        items = [('edge_index', 0)] + [('hierarchy_edge_index_%d' % l, l) for l in range(1, num_levels)]
        for d in self.dilations[:len(self.bottleneck_blocks)]:
            if d > 1:
                items.append(('hierarchy_dil_{}_edge_index_{}'.format(d, last), last))
        seen, uniq = set(), []
        for it in items:
            if it[0] not in seen:
                seen.add(it[0])
                uniq.append(it)
        return uniq, list(range(1, num_levels))

    def prefetch_plan(self, sample, inputs_ready=False, reorder=None):
        """Build the sample's CSR plan now, its independent pieces side by side on side streams (plan.GraphPlan.prefetch).
        Optional: forward() builds whatever is missing at first use on the compute stream.  A data pipeline that hands
        over GPU-resident index tensors can call this with inputs_ready=True as soon as the sample exists, so that the
        build overlaps with the step still running (measured on the 200k-vertex step: no net gain while the step is
        launch-bound on the host)."""
        plan = plan_for(sample, linspace_quirk=self.compat_linspace_norm, validation=self.plan_validation,
                        positions=self.position_channels, reorder=reorder)
        edges, pools = self._plan_items()
        plan.prefetch(edges, pools, inputs_ready=inputs_ready)
        return plan

    def build_plan(self, sample, inputs_ready=True, after=None, reorder=None):
        """A NEW, complete GraphPlan of `sample`, built on the side streams WITHOUT making the compute stream wait (it
        waits when the plan is first used) and without touching the sample's cached plan: the data pipeline's way to
        prepare step k+1 while step k runs.  Hand it over with `sample._plan_cache = plan` (TrainStep.prefetch and
        loader.SceneLoader do).  after = event of the stream that uploads the sample's index tensors, if one does."""
        from .plan import GraphPlan
        plan = GraphPlan(sample, linspace_quirk=self.compat_linspace_norm, validation=self.plan_validation,
                         positions=self.position_channels, reorder=reorder)
        edges, pools = self._plan_items()
        return plan.prefetch(edges, pools, inputs_ready=inputs_ready, join=False, after=after)

    def _net_steps(self, plan, e0, bn_edges, num_levels):
        """The graph part as the op list of functional.NetFn (reference order, :404-455), or None when a block is not the
        fused kind (EdgeConv + instance norm) or the pooling is not max."""
        if self.norm is not M.FastInstanceNorm or self._pooling_type != 'max':
            return None
        groups = (list(self.input_blocks), list(self.encoder_blocks), list(self.bottleneck_blocks), list(self.decoder_blocks),
                  list(self.output_blocks))
        if not all(isinstance(b.first_filter, M.EdgeConv) and isinstance(b.first_norm, M.FastInstanceNorm) for grp in groups for b in grp):
            return None
        last = num_levels - 1
        steps = [('block', blk, e0, self._norm_arg(plan, 0, whole_batch=True)) for blk in groups[0]]
        for i, blk in enumerate(groups[1]):
            level = i + 1
            steps.append(('pool', plan.pool(level)))
            steps.append(('block', blk, plan.edges('hierarchy_edge_index_%d' % level, level), self._norm_arg(plan, level)))
        steps += [('block', blk, edges, self._norm_arg(plan, last)) for blk, edges in zip(groups[2], bn_edges)]
        for i, blk in enumerate(groups[3]):
            level = i + 1
            tgt = num_levels - level - 1
            steps.append(('unpool', plan.pool(num_levels - level)))
            steps.append(('block', blk, e0 if tgt == 0 else plan.edges('hierarchy_edge_index_%d' % tgt, tgt), self._norm_arg(plan, tgt)))
        steps += [('block', blk, e0, self._norm_arg(plan, 0, whole_batch=True)) for blk in groups[4]]
        return steps

    def _forward_per_block(self, out, plan, e0, bn_edges, num_levels):
        for blk in self.input_blocks:                               # norm over the WHOLE batch (reference :406-407)
            out = blk(out, e0, self._norm_arg(plan, 0, whole_batch=True))
        for i, blk in enumerate(self.encoder_blocks):
            level = i + 1
            out = self._pooling(out, plan.pool(level))
            out = blk(out, plan.edges('hierarchy_edge_index_%d' % level, level), self._norm_arg(plan, level))
        last = num_levels - 1
        bn = list(self.bottleneck_blocks)
        if (self.norm is M.FastInstanceNorm and bn and all(isinstance(b.first_filter, M.EdgeConv) for b in bn)
                and SF.chain_eligible(bn, out, bn_edges, None)):
            # the whole bottleneck as ONE autograd node / one foreign call per direction (functional.EdgeConvChainFn)
            out = SF.edgeconv_chain(out, bn, bn_edges, self._norm_arg(plan, last), bn[0].first_norm.eps, SF.PREC_FWD)
        else:
            for blk, edges in zip(bn, bn_edges):
                out = blk(out, edges, self._norm_arg(plan, last))
        for i, blk in enumerate(self.decoder_blocks):
            level = i + 1
            out = self._unpooling(out, plan.pool(num_levels - level))
            tgt = num_levels - level - 1
            edges = e0 if tgt == 0 else plan.edges('hierarchy_edge_index_%d' % tgt, tgt)
            out = blk(out, edges, self._norm_arg(plan, tgt))
        for blk in self.output_blocks:
            out = blk(out, e0, self._norm_arg(plan, 0, whole_batch=True))
        return out

    def forward(self, sample):
        check_deferred()                                                      # deferred index checks of earlier calls
        plan = plan_for(sample, linspace_quirk=self.compat_linspace_norm,      # pieces not prefetched are built at first use
                        validation=self.plan_validation, positions=self.position_channels)
        plan.ensure(*self._plan_items())                                      # ONE batched build of whatever is missing
        num_levels = len(self.decoder_blocks) + 1
        out = sample.x
        if plan.order0 is not None:                                           # the plan renumbered the vertices: enter its order
            out = SF.PermuteRowsFn.apply(out, plan.order0, plan.rank0)
        if self.activation_dtype != out.dtype:
            out = out.to(self.activation_dtype)
        e0 = plan.edges('edge_index', 0)
        self._pack_weights(out, plan.num_graphs)
        last = num_levels - 1
        bn_edges = []
        for i, blk in enumerate(self.bottleneck_blocks):
            if self.dilations[i] > 1:
                key = 'hierarchy_dil_{}_edge_index_{}'.format(self.dilations[i], last)
            else:
                key = 'hierarchy_edge_index_{}'.format(last)
            bn_edges.append(e0 if last == 0 and self.dilations[i] <= 1 else plan.edges(key, last))
        steps = self._net_steps(plan, e0, bn_edges, num_levels)
        if steps is not None and SF.net_eligible(steps, out):
            # the whole graph part as ONE autograd node / one foreign call per direction (functional.NetFn)
            out = SF.run_net(out, steps)
        else:
            out = self._forward_per_block(out, plan, e0, bn_edges, num_levels)
        tail_prec = SF.forward_precision(not self.using_norm)
        packed_tail = self._pack_key is not None and self._pack_key[0] and self._tail_wT is not None
        out = SF.linear(out, self.final_linear1.weight, self.final_linear1.bias, precision=tail_prec,
                        wT=self._tail_wT if packed_tail else None,
                        wT_guard=(self._pack_set, self._pack_set.runs) if packed_tail else None)
        if self.norm is M.FastInstanceNorm:                         # per-graph branch even for B = 1 (:465, Q3)
            out = SF.InstanceNormActResFn.apply(out, None, plan.norm_groups(0), True, self.final_norm1.eps)
        else:
            out = F.elu(self.final_norm1(out, batch=sample.batch))
        out = SF.linear_tanh(out, self.final_linear2.weight, self.final_linear2.bias, precision=tail_prec)
        if plan.order0 is not None:                                           # ... and leave it: outputs in the sample's vertex order
            out = SF.PermuteRowsFn.apply(out, plan.rank0, plan.order0)
        plan.validate()
        return out


def init_net(net, init_type='normal', init_gain=0.02, gpu_ids=[]):
    """Reference :138-154: only moves the net to gpu_ids[0]; weights keep torch's default init."""
    if len(gpu_ids) > 0:
        net.to(gpu_ids[0])
    return net


def define_G(input_nc, output_nc, ngf, filter_type, norm='batch', dilation_order=0, use_dropout=False, n_blocks=6,
             n_levels=2, n_repeated_io_convs=1, init_type='normal', pooling_type='stride',
             io_receptive_field_type='large', checkpoint_bottleneck=False, num_blocks_per_uncheckpointed_block=1,
             use_label_embedding=False, num_classes=None, num_embedding=None, dilations=None, init_gain=0.02,
             gpu_ids=[]):
    """Create the generator - same signature as the reference's define_G (:157-161)."""
    if filter_type in ('conv2d', 'cfconv2d'):
        raise NotImplementedError('the dense Conv2d baselines (Resnet2D) are outside the STINet graph hot path')
    net = SurfaceTextureInpaintingNet(
        input_nc, output_nc, filter_type, ngf, norm_type=norm, n_blocks=n_blocks, n_levels=n_levels,
        n_repeated_io_convs=n_repeated_io_convs, pooling_type=pooling_type,
        checkpoint_bottleneck=checkpoint_bottleneck,
        num_blocks_per_uncheckpointed_block=num_blocks_per_uncheckpointed_block,
        use_label_embedding=use_label_embedding, num_classes=num_classes, num_embedding=num_embedding,
        dilations=dilations)
    return init_net(net, init_type, init_gain, gpu_ids)
