import torch
import torch.nn.functional as F
from torch.nn import Linear
from .message_passing import MessagePassing


class SAGEConv(MessagePassing):
    def __init__(self, in_channels, out_channels, normalize=False, root_weight=True, bias=True, **kwargs):
        kwargs.setdefault('aggr', 'mean')
        super().__init__(**kwargs)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.normalize, self.root_weight = normalize, root_weight
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        self.lin_l = Linear(in_channels[0], out_channels, bias=bias)
        if self.root_weight:
            self.lin_r = Linear(in_channels[1], out_channels, bias=False)

    def forward(self, x, edge_index, size=None):
        if torch.is_tensor(x):
            x = (x, x)
        out = self.propagate(edge_index, x=x, size=size)
        out = self.lin_l(out)
        if self.root_weight and x[1] is not None:
            out = out + self.lin_r(x[1])
        if self.normalize:
            out = F.normalize(out, p=2., dim=-1)
        return out

    def message(self, x_j):
        return x_j
