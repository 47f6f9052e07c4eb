import torch
from .message_passing import MessagePassing


class EdgeConv(MessagePassing):
    def __init__(self, nn, aggr='max', **kwargs):
        super().__init__(aggr=aggr, **kwargs)
        self.nn = nn

    def forward(self, x, edge_index):
        if torch.is_tensor(x):
            x = (x, x)
        return self.propagate(edge_index, x=x, size=None)

    def message(self, x_i, x_j):
        return self.nn(torch.cat([x_i, x_j - x_i], dim=-1))

    def __repr__(self):
        return '{}(nn={})'.format(self.__class__.__name__, self.nn)
