#!/usr/bin/env python3
"""Design probe (CPU, test infrastructure): which bf16 roundings of a GraphResnetBlock cost how much accuracy.

The oracle's network is re-evaluated with the block in the RESTRUCTURED form the HIP path computes
(Y = x [Wa-Wb; Wb; Ws]^T, hE = mean_j ReLU(A_i + B_j), agg = hE W2^T + b2 [deg>0], out = res + ELU(IN(agg))) and
straight-through bf16 roundings switched on per tensor; forward / weight-gradient error against the plain fp32 oracle.

    python tests/tools/bf16_design_probe.py [--vertices 12000]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import stin_oracle  # noqa: E402
from oracle.scatter_ops import scatter_mean  # noqa: E402
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh  # noqa: E402

CFG = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=9,
           n_levels=2, pooling_type='max', dilations=[1, 1, 1, 2, 4, 8, 16, 1, 1], checkpoint_bottleneck=True)
FLAGS = set()


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def qf(t, flag):
    return _RoundFwd.apply(t) if flag in FLAGS else t


def qb(t, flag):
    return _RoundBwd.apply(t) if flag in FLAGS else t


def block_forward(self, x, edge_index, batch=None):
    f = self.first_filter
    w1, b1, w2, b2 = f.nn[0].weight, f.nn[0].bias, f.nn[2].weight, f.nn[2].bias
    cin = x.shape[1]
    if f.trans_inv:
        wa, wb = -w1, w1
    else:
        wa, wb = w1[:, :cin] - w1[:, cin:], w1[:, cin:]
    src, dst = edge_index[0], edge_index[1]
    x = qb(x, 'g')                                  # gradient of the residual stream rounded at the block boundary
    xin = qf(x, 'x')
    A = qb(qf(F.linear(xin, qf(wa, 'w'), b1), 'Y'), 'dY')
    B = qb(qf(F.linear(xin, qf(wb, 'w')), 'Y'), 'dY')
    h = F.relu(A.index_select(0, dst) + B.index_select(0, src))
    hE = qb(qf(scatter_mean(h, dst, dim=0, dim_size=x.shape[0]), 'hE'), 'dhE')
    deg = torch.zeros(x.shape[0], dtype=x.dtype).index_add_(0, dst, torch.ones(dst.shape[0], dtype=x.dtype))
    agg = F.linear(hE, qf(w2, 'w')) + b2 * (deg > 0).to(x.dtype)[:, None]
    agg = qb(qf(agg, 'agg'), 'dagg')
    out = F.elu(self.first_norm(agg, batch))
    res = x
    if self.cin != self.cout:
        res = qb(qf(F.linear(xin if 'sc_x16' in FLAGS else x, qf(self.shortcut.weight, 'wsc'), self.shortcut.bias), 'sc'), 'dsc')
    return qf(res + out, 'out')


def run(net, s):
    net.zero_grad(set_to_none=True)
    out = net(s)
    loss = stin_oracle.compute_loss(torch.where((s.mask > 0).expand_as(s.color), out, s.color), s.color, s.mask)
    loss.backward()
    return out.detach(), float(loss), [p.grad.clone() for p in net.parameters()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--vertices', type=int, default=12000)
    args = ap.parse_args()
    torch.manual_seed(49)
    net = stin_oracle.define_G(**CFG)
    s = make_synthetic_mesh(args.vertices, 3, seed=3)
    out0, loss0, g0 = run(net, s)
    stin_oracle.OracleBlock.forward = block_forward
    designs = {
        'restructured, no rounding': '',
        'x only': 'x', 'w only': 'w wsc', 'Y only': 'Y', 'hE only': 'hE', 'agg only': 'agg', 'out (stream) only': 'out',
        'g (stream grad) only': 'g', 'dagg only': 'dagg', 'dhE only': 'dhE', 'dY only': 'dY',
        'all-bf16 storage (round 1 mode)': 'x w wsc Y hE agg out g dagg dhE dY sc dsc sc_x16',
        'mixed A: fp32 stream; bf16 x,w,Y,hE,agg + all grads': 'x w wsc Y hE agg dagg dhE dY sc dsc sc_x16',
        'mixed B: A with agg/dagg fp32': 'x w wsc Y hE dhE dY sc dsc sc_x16',
        'mixed C: B with fp32 shortcut (x, Ws fp32)': 'x w Y hE dhE dY',
        'mixed D: C with fp32 weights everywhere': 'x Y hE dhE dY',
        'mixed E: only Y, hE bf16 (fwd) and dhE, dY (bwd)': 'Y hE dhE dY',
        'mixed F: only Y, dY': 'Y dY',
    }
    print('%-58s %10s %10s %10s %10s' % ('design', 'fwd max', 'fwd relL2', 'loss rel', 'grad relL2'))
    for name, flags in designs.items():
        FLAGS.clear()
        FLAGS.update(flags.split())
        out, loss, g = run(net, s)
        num = sum(float((a.double() - b.double()).pow(2).sum()) for a, b in zip(g, g0))
        den = sum(float(b.double().pow(2).sum()) for b in g0)
        print('%-58s %10.2e %10.2e %10.2e %10.2e' % (name, float((out - out0).abs().max()),
                                                      float((out - out0).norm() / out0.norm()), abs(loss - loss0) / loss0,
                                                      (num / den) ** 0.5))


if __name__ == '__main__':
    main()
