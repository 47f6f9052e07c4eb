"""CPU restatement of the reference's SingleConvMeshNet - TEST INFRASTRUCTURE ONLY.

models/singleconvmeshnet.py:10-156: a U-Net over the mesh hierarchy whose EdgeConv filters carry BatchNorm1d INSIDE
the per-edge MLP (models/modules/edge_conv_filter.py:34-44, `with_norm=True`):

    message_e = BN2( Lin2( ReLU( BN1( Lin1( [x_i ; x_j - x_i] ) ) ) ) )      Lin1, Lin2 without bias,
    out_i     = mean_{e -> i} message_e                                       BN statistics over ALL E EDGES

(first level: EdgeConvTransInv, message input x_j - x_i only).  ResBlock (:91-107): x = ReLU(f0(x)); then
x = ReLU(x + f_k(x)).  Encoder: pool (scatter_mean / scatter_max[0], :109-115) then ResBlock; decoder: unpool by the trace,
concatenate the skip level, ResBlock; head: Lin -> BatchNorm1d -> ReLU -> Lin (:79-86).  Module tree and state_dict keys
are those of the reference (left_geo_cnns.{l}.filters.{k}.nn.{0,1,3,4}.*, right_geo_cnns.*, final_convs.0.{0,1,3}.*).
Unfused, op for op, with the torch-scatter semantics of oracle/scatter_ops.py.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import scatter_ops


class EdgeConvBN(nn.Module):
    def __init__(self, cin, cout, trans_inv=False):
        super().__init__()
        self.trans_inv = trans_inv
        self.nn = nn.Sequential(nn.Linear(cin if trans_inv else 2 * cin, 2 * cout, bias=False), nn.BatchNorm1d(2 * cout),
                                nn.ReLU(), nn.Linear(2 * cout, cout, bias=False), nn.BatchNorm1d(cout))

    def forward(self, x, edge_index):
        x_j, x_i = x[edge_index[0]], x[edge_index[1]]                     # source_to_target flow
        m = x_j - x_i if self.trans_inv else torch.cat([x_i, x_j - x_i], dim=-1)
        return scatter_ops.scatter_mean(self.nn(m), edge_index[1], dim=0, dim_size=x.shape[0])


class ResBlock(nn.Module):
    def __init__(self, filters):
        super().__init__()
        self.filters = nn.ModuleList(filters)

    def forward(self, x, edge_index):
        x = F.relu(self.filters[0](x, edge_index))
        for f in list(self.filters)[1:]:
            x = F.relu(x + f(x, edge_index))
        return x


class SingleConvMeshNet(nn.Module):
    def __init__(self, feature_number, num_propagation_steps, filter_sizes, num_classes=3, pooling_method='mean', aggr='mean'):
        super().__init__()
        assert aggr == 'mean'
        self._pooling_method = pooling_method
        self._graph_levels = len(filter_sizes)
        left, right = [], []
        cur = feature_number
        for level, fs in enumerate(filter_sizes):
            first = EdgeConvBN(cur, fs, trans_inv=(level == 0 and level < len(filter_sizes) - 1))
            left.append(ResBlock([first] + [EdgeConvBN(fs, fs) for _ in range(num_propagation_steps - 1)]))
            if level < len(filter_sizes) - 1:
                cat = fs + filter_sizes[level + 1]
                right.append(ResBlock([EdgeConvBN(cat, fs)] + [EdgeConvBN(fs, fs) for _ in range(num_propagation_steps - 1)]))
                cur = fs
        self.left_geo_cnns = nn.ModuleList(left)
        self.right_geo_cnns = nn.ModuleList(right)
        f0 = filter_sizes[0]
        self.final_convs = nn.ModuleList([nn.Sequential(nn.Linear(f0, f0 // 2), nn.BatchNorm1d(f0 // 2), nn.ReLU(),
                                                        nn.Linear(f0 // 2, num_classes))])

    def _pooling(self, x, trace):
        n = int(trace.max()) + 1
        if self._pooling_method == 'mean':
            return scatter_ops.scatter_mean(x, trace, dim=0, dim_size=n)
        if self._pooling_method == 'max':
            return scatter_ops.scatter_max(x, trace, dim=0, dim_size=n)[0]
        raise ValueError('Unkown pooling type {}'.format(self._pooling_method))

    def forward(self, sample):
        L = self._graph_levels
        levels = [self.left_geo_cnns[0](sample.x, sample.edge_index)]
        for level in range(1, L):
            cur = self._pooling(levels[-1], sample['hierarchy_trace_index_%d' % level])
            levels.append(self.left_geo_cnns[level](cur, sample['hierarchy_edge_index_%d' % level]))
        current = levels[-1]
        for level in range(1, L):
            back = current[sample['hierarchy_trace_index_%d' % (L - level)]]
            fused = torch.cat((levels[-(level + 1)], back), -1)
            ei = sample.edge_index if level == L - 1 else sample['hierarchy_edge_index_%d' % (L - level - 1)]
            current = self.right_geo_cnns[-level](fused, ei)
        out = current
        for conv in self.final_convs:
            out = conv(out)
        return out
