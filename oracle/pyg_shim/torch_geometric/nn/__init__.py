import torch
from . import conv, inits  # noqa: F401
from .conv import MessagePassing, EdgeConv, SAGEConv  # noqa: F401


class BatchNorm(torch.nn.Module):
    """PyG BatchNorm: BatchNorm1d over node features, held as ``.module``."""
    def __init__(self, in_channels, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.module = torch.nn.BatchNorm1d(in_channels, eps, momentum, affine, track_running_stats)

    def reset_parameters(self):
        self.module.reset_parameters()

    def forward(self, x):
        return self.module(x)


class InstanceNorm(torch.nn.Module):  # imported by the reference, never constructed on the path
    def __init__(self, *a, **k):
        raise NotImplementedError


class GraphNorm(torch.nn.Module):  # imported by the reference, never constructed on the path
    def __init__(self, *a, **k):
        raise NotImplementedError


class DataParallel(torch.nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError
