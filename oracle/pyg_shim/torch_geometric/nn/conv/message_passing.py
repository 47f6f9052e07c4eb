import inspect
import torch
from oracle.scatter_ops import scatter


class MessagePassing(torch.nn.Module):
    """propagate() = gather (x_j by edge_index[0], x_i by edge_index[1]) ->
    message() -> scatter(reduce=aggr) at edge_index[1] -> update()."""

    def __init__(self, aggr='add', flow='source_to_target', node_dim=-2, **kwargs):
        super().__init__()
        assert flow == 'source_to_target' and node_dim == -2
        self.aggr = aggr
        self.flow = flow
        self.node_dim = node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        assert torch.is_tensor(edge_index) and edge_index.dim() == 2 and edge_index.shape[0] == 2
        j, i = edge_index[0], edge_index[1]
        params = [p for p in inspect.signature(self.message).parameters]
        args = {}
        dim_size = None
        for name in params:
            if name.endswith('_i') or name.endswith('_j'):
                data = kwargs[name[:-2]]
                which = 1 if name.endswith('_i') else 0
                if isinstance(data, (tuple, list)):
                    if data[1] is not None:
                        dim_size = data[1].size(0)
                    data = data[which]
                elif dim_size is None:
                    dim_size = data.size(0)
                args[name] = data.index_select(0, i if which == 1 else j)
            else:
                args[name] = kwargs[name]
        if size is not None and size[1] is not None:
            dim_size = size[1]
        msg = self.message(**args)
        out = self.aggregate(msg, i, dim_size=dim_size)
        return self.update(out)

    def message(self, x_j):
        return x_j

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        return scatter(inputs, index, dim=0, dim_size=dim_size, reduce=self.aggr)

    def update(self, inputs):
        return inputs
