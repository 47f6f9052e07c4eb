"""CPU restatement of the torch-scatter 2.0.9 ops the STINet hot path calls.

TEST INFRASTRUCTURE ONLY (oracle).  Nothing under ``oracle/`` may be imported by
the product package; only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg use it, and only as the checker.

torch_scatter is a third-party dependency of the reference that is NOT vendored
under /root/reference and is unpinned there (README.md:38-41 of the reference:
``conda install pyg -c pyg``; era torch-scatter 2.0.9).  The published
semantics restated here (not copied):

* ``scatter_sum(src, index, dim=0, dim_size=N)`` = ``zeros(N, C).scatter_add_``;
  on CPU the adds happen sequentially in ``index`` order.
* ``scatter_mean`` = ``scatter_sum / scatter_sum(ones).clamp(min=1)``; float:
  true divide, integer: floor divide.  Empty segments give 0.
* ``scatter_max(src, index, dim=0, dim_size=N) -> (out, arg)``: ``out`` starts
  at the lowest value, update on strict ``>`` walking ``src`` in order, hence
  **the first occurrence wins ties**; empty segments get value 0 and
  ``arg = src.size(0)`` (sentinel).  Backward routes the gradient to ``arg``
  only (one element per (segment, channel)).

Reference call sites: models/surfacetextureinpaintingnet.py:384 (scatter_mean
pool), :386 (scatter_max pool), :422 (scatter_max on the int64 batch vector).
"""
import torch


def _expand_index(index, src):
    if index.dim() == 1 and src.dim() > 1:
        shape = [1] * src.dim()
        shape[0] = -1
        index = index.view(shape).expand_as(src)
    return index


def scatter_sum(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    size = list(src.shape)
    size[0] = int(dim_size)
    res = src.new_zeros(size)
    return res.scatter_add_(0, _expand_index(index, src), src)


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0
    total = scatter_sum(src, index, 0, None, dim_size)
    ones = torch.ones(index.shape[0], dtype=src.dtype, device=src.device)
    count = scatter_sum(ones, index, 0, None, total.shape[0]).clamp_(min=1)
    shape = [1] * total.dim()
    shape[0] = -1
    count = count.view(shape)
    if total.is_floating_point():
        return total / count
    return torch.div(total, count, rounding_mode='floor')


class _ScatterMaxArgFirst(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, dim_size):
        n = src.shape[0]
        idx = _expand_index(index, src)
        size = list(src.shape)
        size[0] = int(dim_size)
        if src.is_floating_point():
            lowest = torch.finfo(src.dtype).min
        else:
            lowest = torch.iinfo(src.dtype).min
        out = src.new_full(size, lowest)
        out.scatter_reduce_(0, idx, src, reduce='amax', include_self=True)
        # arg = FIRST (lowest row id) element equal to the segment max
        rows = torch.arange(n, device=src.device).view([-1] + [1] * (src.dim() - 1)).expand_as(src)
        is_max = src == out.gather(0, idx)
        cand = torch.where(is_max, rows, torch.full_like(rows, n))
        arg = torch.full(size, n, dtype=torch.long, device=src.device)
        arg.scatter_reduce_(0, idx, cand, reduce='amin', include_self=True)
        empty = arg == n
        out = torch.where(empty, torch.zeros_like(out), out)
        ctx.save_for_backward(arg)
        ctx.n = n
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, grad_out, _grad_arg):
        (arg,) = ctx.saved_tensors
        n = ctx.n
        size = list(grad_out.shape)
        size[0] = n + 1  # row n swallows the sentinel of empty segments
        grad_src = grad_out.new_zeros(size)
        grad_src.scatter_(0, arg, grad_out)
        return grad_src[:n], None, None


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    return _ScatterMaxArgFirst.apply(src, index, int(dim_size))


def scatter(src, index, dim=0, out=None, dim_size=None, reduce='sum'):
    if reduce in ('sum', 'add'):
        return scatter_sum(src, index, dim, out, dim_size)
    if reduce == 'mean':
        return scatter_mean(src, index, dim, out, dim_size)
    if reduce == 'max':
        return scatter_max(src, index, dim, out, dim_size)[0]
    raise ValueError(reduce)
