"""Per-step graph metrics of the reference trainer on the HIP segment-sum kernel
(reference utils/metrics/graph_metrics.py:6-74, called every step at
trainers/inpainting3d_trainer.py:254-263).  GraphLaplaceOperator there is a PyG MessagePassing with
aggr='add' - a literal scatter-add of [E, 2] rows - which is exactly stin_segment_sum_f32 over the
destination CSR of the plan that the forward pass has already built (no second sort)."""
import torch

from . import functional as SF
from .plan import EdgeSet


def _edges(edge_index, n):
    if isinstance(edge_index, EdgeSet):
        return edge_index
    bad = torch.zeros(1, dtype=torch.int32, device=edge_index.device)
    return EdgeSet(edge_index, n, bad)


def grayscale(x):
    return 0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3]


def graph_laplace(x, edge_index):
    """sum_{j in N(i)} x_j - deg_i x_i   (graph_metrics.py:6-16): one HIP pass over the destination CSR (stin_graph_laplace_f32; the
    metric is evaluated under no_grad by the trainer); a tensor that needs a gradient takes the differentiable segment-sum form."""
    e = _edges(edge_index, x.shape[0])
    if x.is_cuda and x.dtype == torch.float32 and not (torch.is_grad_enabled() and x.requires_grad):
        from .plan import _ptr, _stream
        xm, ldx = SF._mat(x.detach())
        out = torch.empty(xm.shape[0], xm.shape[1], dtype=torch.float32, device=x.device)
        SF._call('stin_graph_laplace_f32', _ptr(xm), ldx, _ptr(e.by_dst.rowptr), _ptr(e.by_dst.col), xm.shape[0], xm.shape[1], _ptr(out),
                 out.stride(0), _stream(xm))
        return out
    xi = torch.cat([x.new_ones(x.shape[0], 1), x], dim=1).contiguous()
    prop = SF.segment_sum(xi, e.by_dst.rowptr, e.by_dst.col, e.n, mean=False)
    return prop[:, 1:] - prop[:, 0:1] * x


def graph_laplace_variance(x, edge_index):
    """graph_metrics.py:19-31."""
    with torch.no_grad():
        return torch.var(graph_laplace(grayscale(x), edge_index), dim=0, unbiased=False)


def graph_total_variation(x, edge_index):
    """sum_e |x_src - x_dst| / (N * C)   (graph_metrics.py:34-38): one HIP pass over the destination CSR (each vertex walks
    its in-edges), fp64 partial sums in a fixed order - no [E, C] temporaries, no host sync."""
    from . import _lib
    from .plan import _ptr, _stream
    e = _edges(edge_index, x.shape[0])
    x, ldx = SF._mat(x.detach().float())
    n, c = x.shape
    lib = _lib.load()
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws_bytes = lib.stin_total_variation_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    SF._call('stin_total_variation_f32', _ptr(x), ldx, _ptr(e.by_dst.rowptr), _ptr(e.by_dst.col), n, c, _ptr(out), _ptr(ws),
             ws_bytes, _stream(x))
    return out[0]


def psnr(x, y, data_range=1.0, convert_to_greyscale=False):
    """graph_metrics.py:41-74."""
    x = x / data_range
    y = y / data_range
    if x.size(1) == 3 and convert_to_greyscale:
        x, y = grayscale(x), grayscale(y)
    mse = torch.mean((x - y) ** 2, dim=[0, 1])
    return -10 * torch.log10(mse + 1e-8)
