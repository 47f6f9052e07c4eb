"""Graph preprocessing on the GPU for the data either side of the hot path.

* ``dilated_edges``     - the reference's ``preprocessing/graph_dilation.compute_all_node_dilated_edges`` (:50-75), which
  produces the ``hierarchy_dil_{d}_edge_index_{L}`` sets the bottleneck blocks run on.  The per-vertex python loops of
  the reference (~30 min per ScanNet scene, README.md:89) become ONE launch of ``stin_dilated_walk_*`` (one thread per
  directed edge) plus a sort/unique per dilation.
* ``vertex_clustering`` - ``preprocessing/graph_level_generation.vertex_clustering`` (:193-244), the Rossignac voxel
  clustering alternative to QEM for building the hierarchy (trace + coarse edges + coarse coordinates); index
  arithmetic only, expressed with device-side sort/unique.

Both take and return tensors in the reference's own formats.  QEM decimation itself stays out of scope (it shells out
to vcglib's ``tridecimator``).
"""
import ctypes

import torch

from . import _lib
from .plan import _ptr, _stream


def coalesce(edge_index, num_nodes):
    """pyg.utils.coalesce: sort by (row 0, row 1) and drop duplicates.  [2, E] int64 -> [2, E'] int64."""
    if edge_index.numel() == 0:
        return edge_index.reshape(2, 0)
    key = torch.unique(edge_index[0] * num_nodes + edge_index[1], sorted=True)
    return torch.stack([torch.div(key, num_nodes, rounding_mode='floor'), key % num_nodes])


def dilated_edges(edge_index, pos, normals, dilations):
    """edge_index: [2, E] int64 CUDA tensor (row 0 -> row 1; any order, duplicates allowed);
    pos, normals: [N, 3] float32 or float64 (the arithmetic type of the walk; the reference pipeline uses float64);
    dilations: ascending ints in [2, 63].
    -> one entry per dilation: an [E_d, 2] int64 tensor of rows [far vertex, centre] sorted by (far, centre) without
    duplicates - exactly what the reference stores in ``dilated_edges[level][i]`` - or ``[]`` when no walker got that far
    (the reference leaves an empty python list there)."""
    lib = _lib.load()
    if not (edge_index.is_cuda and pos.is_cuda and normals.is_cuda):
        raise TypeError('dilated_edges runs on the GPU only (no CPU fallback exists)')
    if pos.dtype not in (torch.float32, torch.float64) or normals.dtype != pos.dtype:
        raise TypeError('pos and normals must both be float32 or both float64')
    dil = [int(d) for d in dilations]
    if any(d < 2 or d > 63 for d in dil) or any(b <= a for a, b in zip(dil, dil[1:])):
        raise ValueError('dilations must be ascending ints in [2, 63]')
    n = pos.shape[0]
    if edge_index.numel() and (int(edge_index.min()) < 0 or int(edge_index.max()) >= n):
        raise IndexError('edge_index refers to a vertex outside [0, %d)' % n)
    ei = coalesce(edge_index.long(), n)
    e = ei.shape[1]
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=pos.device)
    if e:
        rowptr[1:] = torch.cumsum(torch.bincount(ei[0], minlength=n), 0)
    rowptr, row_of, col = rowptr.int(), ei[0].int().contiguous(), ei[1].int().contiguous()
    pos, normals = pos.contiguous(), normals.contiguous()
    out = torch.empty(len(dil), max(e, 1), dtype=torch.int32, device=pos.device)[:, :e]
    harr = (ctypes.c_int32 * len(dil))(*dil)
    fn = lib.stin_dilated_walk_f64 if pos.dtype == torch.float64 else lib.stin_dilated_walk_f32
    _lib.check(fn(_ptr(rowptr), _ptr(col), _ptr(row_of), _ptr(pos), _ptr(normals), n, e, harr, len(dil),
                  _ptr(out) if e else None, _stream(pos)), 'stin_dilated_walk')
    res = []
    for i in range(len(dil)):
        far = out[i] if e else out.new_zeros(0)
        ok = far >= 0
        if e == 0 or not bool(ok.any()):
            res.append([])
            continue
        pairs = coalesce(torch.stack([far[ok].long(), row_of[ok].long()]), n)
        res.append(pairs.t().contiguous())
    return res


def vertex_clustering(coords, edge_index, voxel_size):
    """coords [N, 3] float (CUDA), edge_index [2, E] int64, voxel_size float ->
    (new_coords float32 [Nc, 3], trace int64 [N], coarse_edges int64 [Ec, 2] sorted by (row 0, row 1)).
    Coarse ids follow the lexicographic order of the voxel bins (``np.unique(bins, axis=0)``)."""
    if not coords.is_cuda:
        raise TypeError('vertex_clustering runs on the GPU only')
    c64 = coords.double()
    bins = torch.div(c64, float(voxel_size), rounding_mode='floor')
    _, trace = torch.unique(bins, dim=0, return_inverse=True)
    trace = trace.reshape(-1)
    nc = int(trace.max()) + 1 if trace.numel() else 0
    ce = torch.stack([trace[edge_index[0]], trace[edge_index[1]]]) if edge_index.numel() else edge_index.reshape(2, 0)
    ce = coalesce(ce[:, ce[0] != ce[1]], max(nc, 1))
    sums = torch.zeros(nc, 3, dtype=torch.float64, device=coords.device).index_add_(0, trace, c64)
    cnt = torch.bincount(trace, minlength=nc).clamp(min=1).double().unsqueeze(1)
    return (sums / cnt).float(), trace, ce.t().contiguous()
