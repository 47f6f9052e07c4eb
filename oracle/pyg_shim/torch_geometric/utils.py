import torch


def degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else int(num_nodes)
    out = torch.zeros(n, dtype=dtype if dtype is not None else torch.get_default_dtype(),
                      device=index.device)
    one = torch.ones(index.shape[0], dtype=out.dtype, device=index.device)
    return out.scatter_add_(0, index, one)


def coalesce(edge_index, edge_attr=None, num_nodes=None, reduce='add'):
    assert edge_attr is None
    n = int(edge_index.max()) + 1 if num_nodes is None else int(num_nodes)
    key = edge_index[0] * n + edge_index[1]
    key = torch.unique(key, sorted=True)
    return torch.stack([key // n, key % n], dim=0)
