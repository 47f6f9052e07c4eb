"""Graph preprocessing on the GPU for the data either side of the hot path.

* ``dilated_edges``     - the reference's ``preprocessing/graph_dilation.compute_all_node_dilated_edges`` (:50-75), which
  produces the ``hierarchy_dil_{d}_edge_index_{L}`` sets the bottleneck blocks run on.  The per-vertex python loops of
  the reference (~30 min per ScanNet scene, README.md:89) become ONE launch of ``stin_dilated_walk_*`` (one thread per
  directed edge) plus a sort/unique per dilation.
* ``vertex_clustering`` - ``preprocessing/graph_level_generation.vertex_clustering`` (:193-244), the Rossignac voxel
  clustering alternative to QEM for building the hierarchy (trace + coarse edges + coarse coordinates): one 63-bit voxel key
  per vertex, a stable radix sort and a scan (``stin_voxel_cluster_f64``), the coarse edges through ``stin_coalesce_pairs_i64``.

Both take and return tensors in the reference's own formats.  QEM decimation itself stays out of scope (it shells out
to vcglib's ``tridecimator``).
"""
import ctypes

import torch

from . import _lib
from .plan import _ptr, _stream


def coalesce(edge_index, num_nodes, vertex_map=None, drop_loops=False):
    """pyg.utils.coalesce: sort by (row 0, row 1) and drop duplicates.  [2, E] int64 CUDA -> [2, E'] int64.
    vertex_map: both endpoints go through this int64 map first (a trace: the coarse edges of a clustering);
    drop_loops: pairs with equal endpoints are left out.  One radix sort + scan on the GPU (stin_coalesce_pairs_i64)."""
    if edge_index.numel() == 0:
        return edge_index.reshape(2, 0)
    if not edge_index.is_cuda:
        raise TypeError('coalesce runs on the GPU only')
    lib = _lib.load()
    ei = edge_index.long().contiguous()
    E, dev = ei.shape[1], ei.device
    out = torch.empty(2, E, dtype=torch.int64, device=dev)
    state = torch.empty(5, dtype=torch.int64, device=dev)
    ws_bytes = lib.stin_coalesce_workspace_bytes(E)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    vm = vertex_map.contiguous() if vertex_map is not None else None
    _lib.check(lib.stin_coalesce_pairs_i64(_ptr(ei[0]), _ptr(ei[1]), _ptr(vm), (int(vm.numel()) if vm is not None else 0), E, int(num_nodes), int(bool(drop_loops)), _ptr(out[0]),
                                           _ptr(out[1]), _ptr(state), _ptr(ws), ws_bytes, _stream(ei)), 'stin_coalesce_pairs_i64')
    st = state.cpu()
    if int(st[3]) != 0:
        raise IndexError('coalesce: an endpoint lies outside [0, %d)%s' % (
            num_nodes, '' if vm is None else ' or a raw endpoint outside the vertex map [0, %d)' % vm.numel()))
    return out[:, :int(st[4])]


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
    lib = _lib.load()
    c64 = coords.double().contiguous()
    n, dev = c64.shape[0], c64.device
    trace = torch.empty(n, dtype=torch.int64, device=dev)
    new_coords = torch.empty(max(n, 1), 3, dtype=torch.float32, device=dev)
    state = torch.empty(5, dtype=torch.int64, device=dev)
    ws_bytes = lib.stin_voxel_cluster_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.stin_voxel_cluster_f64(_ptr(c64), n, float(voxel_size), _ptr(trace), _ptr(new_coords), _ptr(state), _ptr(ws),
                                          ws_bytes, _stream(c64)), 'stin_voxel_cluster_f64')
    st = state.cpu()
    if int(st[3]) != 0:
        raise ValueError('vertex_clustering: non-finite coordinates or more than 2^21 voxels along an axis')
    nc = int(st[4])
    if edge_index.numel():
        ce = coalesce(edge_index, max(nc, 1), vertex_map=trace, drop_loops=True)
    else:
        ce = edge_index.reshape(2, 0)
    return new_coords[:nc], trace, ce.t().contiguous()
