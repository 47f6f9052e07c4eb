"""Synthetic ScanNet-sized hierarchical meshes (no dataset on the GPU box).

Follows SURVEY.md §8(d): a jittered triangulated grid (mean directed degree ~6,
symmetric edges, no self loops, ``edge_index`` grouped by source like the
reference's preprocessing/graph_level_generation.py:395-398), optional random
vertex permutation so memory order is not grid-like, 30 % contraction per level
(graph-Voronoi clusters, every coarse vertex non-empty) mimicking QEM with
``--level_params 100 30 30 30``, coarse edges = image of the fine edges under the
trace, and directed "dilated" edge sets at the last level with row 0 = far node,
row 1 = centre (reference preprocessing/graph_dilation.py:83-137).  Features
follow datasets/scannetcolorgraph_dataloader.py:92-121:
x = [rgb * known, normal, pos / 1.5, known] (10 channels).
"""
import numpy as np
import torch

from .data import HierarchicalBatch


def _grid_mesh(rows, cols, rng):
    r, c = np.meshgrid(np.arange(rows), np.arange(cols), indexing='ij')
    vid = (r * cols + c)
    pairs = []
    pairs.append((vid[:, :-1].ravel(), vid[:, 1:].ravel()))        # right
    pairs.append((vid[:-1, :].ravel(), vid[1:, :].ravel()))        # down
    pairs.append((vid[:-1, :-1].ravel(), vid[1:, 1:].ravel()))     # diagonal
    a = np.concatenate([p[0] for p in pairs])
    b = np.concatenate([p[1] for p in pairs])
    src = np.concatenate([a, b])
    dst = np.concatenate([b, a])
    n = rows * cols
    pos = np.stack([r.ravel() + 0.0, c.ravel() + 0.0, np.zeros(n)], 1)
    pos[:, :2] += rng.uniform(-0.3, 0.3, size=(n, 2))
    pos[:, 2] = 0.5 * np.sin(pos[:, 0] / 17.0) * np.cos(pos[:, 1] / 23.0) + rng.normal(0, 0.05, n)
    nrm = np.stack([rng.normal(0, 0.1, n), rng.normal(0, 0.1, n), np.ones(n)], 1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return n, src, dst, pos, nrm


def _delaunay_mesh(n, rng):
    """Irregular 2-manifold-like triangulation: the Delaunay triangulation of n uniformly random points of a square
    (vertex degrees 3 ... ~14, mean ~6, standard deviation ~1.3 - the valence spread of a QEM-decimated scan, where the
    jittered grid is 6-regular).  Same height field / normals as the grid mesh."""
    from scipy.spatial import Delaunay
    side = float(np.sqrt(n))
    xy = rng.uniform(0.0, side, size=(n, 2))
    tri = Delaunay(xy).simplices.astype(np.int64)
    a = np.concatenate([tri[:, 0], tri[:, 1], tri[:, 2]])
    b = np.concatenate([tri[:, 1], tri[:, 2], tri[:, 0]])
    key = np.unique(np.concatenate([a * n + b, b * n + a]))          # symmetric, duplicates (shared triangle edges) removed
    src, dst = key // n, key % n
    pos = np.zeros((n, 3))
    pos[:, :2] = xy
    pos[:, 2] = 0.5 * np.sin(pos[:, 0] / 17.0) * np.cos(pos[:, 1] / 23.0) + rng.normal(0, 0.05, n)
    nrm = np.stack([rng.normal(0, 0.1, n), rng.normal(0, 0.1, n), np.ones(n)], 1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return n, src, dst, pos, nrm


def _group_by_source(src, dst):
    order = np.lexsort((dst, src))
    return src[order], dst[order]


def _contract(n, src, dst, keep, rng):
    """Graph-Voronoi contraction to ``keep`` clusters -> trace [n] in [0, keep)."""
    seeds = rng.choice(n, size=keep, replace=False)
    label = np.full(n, -1, dtype=np.int64)
    label[seeds] = np.arange(keep)
    for _ in range(64):
        m = (label[src] >= 0) & (label[dst] < 0)
        if not m.any():
            break
        label[dst[m]] = label[src[m]]
    left = np.flatnonzero(label < 0)      # isolated leftovers join random clusters
    if left.size:
        label[left] = rng.integers(0, keep, size=left.size)
    return label


def _coarse_edges(trace, src, dst, n_coarse):
    cs, cd = trace[src], trace[dst]
    m = cs != cd
    key = np.unique(cs[m] * n_coarse + cd[m])
    return key // n_coarse, key % n_coarse


def _mean_by(trace, vals, n_coarse):
    out = np.zeros((n_coarse, vals.shape[1]))
    np.add.at(out, trace, vals)
    cnt = np.bincount(trace, minlength=n_coarse).astype(np.float64)
    return out / np.maximum(cnt, 1)[:, None]


def _neighbor_matrix(n, src, dst):
    order = np.argsort(src, kind='stable')
    s, d = src[order], dst[order]
    deg = np.bincount(s, minlength=n)
    start = np.concatenate([[0], np.cumsum(deg)[:-1]])
    within = np.arange(s.size) - start[s]
    nbr = np.full((n, int(deg.max()) if n else 0), -1, dtype=np.int64)
    nbr[s, within] = d
    return nbr


def _dilated_edges(n, src, dst, pos, nrm, dilations):
    """Greedy straight walks: for every (centre, 1-hop) pair continue along the
    current direction (projected into the local tangent plane), never stepping back
    or into the centre's 1-hop ring; emit (far, centre) at each requested hop count."""
    nbr = _neighbor_matrix(n, src, dst)
    centre, cur = src.copy(), dst.copy()
    last = centre.copy()
    direction = pos[cur] - pos[last]
    alive = np.ones(centre.size, dtype=bool)
    ring = nbr[centre]                            # [W, D]
    out = {}
    for hop in range(2, max(dilations) + 1):
        cand = nbr[cur]                           # [W, D]
        ok = (cand >= 0) & (cand != last[:, None])
        ok &= ~(cand[:, :, None] == ring[:, None, :]).any(-1)
        nvec = nrm[cur]
        u = direction - nvec * (direction * nvec).sum(1, keepdims=True)
        v = pos[np.maximum(cand, 0)] - pos[cur][:, None, :]
        v = v - nvec[:, None, :] * (v * nvec[:, None, :]).sum(-1, keepdims=True)
        un = np.linalg.norm(u, axis=1, keepdims=True) + 1e-12
        vn = np.linalg.norm(v, axis=2) + 1e-12
        sim = (v * u[:, None, :]).sum(-1) / (vn * un)
        sim = np.where(ok, sim, -2.0)
        best = sim.argmax(1)
        bsim = sim[np.arange(sim.shape[0]), best]
        alive &= bsim >= 0.0
        nxt = cand[np.arange(cand.shape[0]), best]
        if hop in dilations:
            far, ctr = nxt[alive], centre[alive]
            key = np.unique(far * n + ctr)
            out[hop] = (key // n, key % n)
        last = np.where(alive, cur, last)
        cur = np.where(alive, nxt, cur)
        direction = np.where(alive[:, None], u / un, direction)
    return out


def make_synthetic_mesh(n0=200_000, levels=3, seed=0, dilations=(2, 4, 8, 16), permute=True,
                        keep_ratio=0.3, masked_fraction=0.25, dtype=torch.float32, irregular=False):
    """-> HierarchicalBatch (CPU tensors) for ONE graph with ``levels`` graph levels.

    ``n0`` is rounded UP to a square grid (200 000 -> 448 x 448 = 200 704
    vertices, 1 200 642 directed edges).  irregular=True: the same vertex count on a Delaunay
    triangulation of random points instead of the 6-regular jittered grid (degrees 3 ... ~14)."""
    rng = np.random.default_rng(seed)
    side = max(2, int(np.ceil(np.sqrt(n0))))
    if irregular:
        n, src, dst, pos, nrm = _delaunay_mesh(side * side, rng)
    else:
        n, src, dst, pos, nrm = _grid_mesh(side, side, rng)
    if permute:
        perm = rng.permutation(n)                 # new id of old vertex v = perm[v]
        inv = np.empty(n, dtype=np.int64)
        inv[perm] = np.arange(n)
        src, dst = perm[src], perm[dst]
        pos, nrm = pos[inv], nrm[inv]
    src, dst = _group_by_source(src, dst)

    sample = HierarchicalBatch()
    nv = [n]
    lv_src, lv_dst, lv_pos, lv_nrm, lv_n = src, dst, pos, nrm, n
    sample['edge_index'] = torch.from_numpy(np.stack([src, dst]))
    for lvl in range(1, levels):
        keep = max(1, int(np.floor(keep_ratio * lv_n)))
        trace = _contract(lv_n, lv_src, lv_dst, keep, rng)
        c_src, c_dst = _coarse_edges(trace, lv_src, lv_dst, keep)
        c_pos, c_nrm = _mean_by(trace, lv_pos, keep), _mean_by(trace, lv_nrm, keep)
        c_nrm /= (np.linalg.norm(c_nrm, axis=1, keepdims=True) + 1e-12)
        sample['hierarchy_trace_index_%d' % lvl] = torch.from_numpy(trace)
        sample['hierarchy_edge_index_%d' % lvl] = torch.from_numpy(np.stack([c_src, c_dst]))
        lv_src, lv_dst, lv_pos, lv_nrm, lv_n = c_src, c_dst, c_pos, c_nrm, keep
        nv.append(keep)
    dil = [d for d in (dilations or ()) if d > 1]
    if dil and lv_src.size:
        sets = _dilated_edges(lv_n, lv_src, lv_dst, lv_pos, lv_nrm, dil)
        for d in dil:
            far, ctr = sets.get(d, (np.zeros(0, np.int64), np.zeros(0, np.int64)))
            sample['hierarchy_dil_%d_edge_index_%d' % (d, levels - 1)] = torch.from_numpy(np.stack([far, ctr]))

    rgb = rng.uniform(-1, 1, size=(n, 3))
    mask = np.where(rng.uniform(size=n) < masked_fraction, rng.integers(1, 17, size=n), 0)
    known = (mask == 0).astype(np.float64)[:, None]
    p = pos - pos.min(0, keepdims=True)
    p = p / max(p.max(), 1e-9)
    x = np.concatenate([rgb * known, nrm, p / 1.5, known], 1)
    sample['x'] = torch.from_numpy(x).to(dtype)
    sample['color'] = torch.from_numpy(rgb).to(dtype)
    sample['mask'] = torch.from_numpy(mask.astype(np.int64))[:, None]
    sample['num_vertices'] = torch.tensor([nv], dtype=torch.int32)
    sample['batch'] = torch.zeros(n, dtype=torch.long)
    return sample


def _morton3(p, bits=10):
    """30-bit Morton code of positions p [n, 3] (quantised on the bounding box)."""
    lo, hi = p.min(0, keepdims=True), p.max(0, keepdims=True)
    q = np.minimum(((p - lo) / np.maximum(hi - lo, 1e-12) * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    code = np.zeros(p.shape[0], dtype=np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return code


def renumber_by_locality(sample, position_channels=(6, 9)):
    """HOST-side experiment / reader-side preprocessing: renumber the vertices of every level of a single-graph sample so that
    memory order follows space - level 0 by the Morton code of its positions (x[:, 6:9] in the reference's feature layout),
    every coarser level by the smallest new id among its children.  Edge lists keep their order (so every CSR row keeps its
    neighbour order), index values are relabelled.  Returns (new sample, perm0) with new_x = x[perm0]."""
    x = sample.x.numpy()
    n_levels = int(sample.num_vertices.shape[-1])
    order = np.argsort(_morton3(x[:, position_channels[0]:position_channels[1]].astype(np.float64)), kind='stable')
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    out = HierarchicalBatch()
    out['x'] = sample.x[torch.from_numpy(order)]
    out['color'] = sample.color[torch.from_numpy(order)]
    out['mask'] = sample.mask[torch.from_numpy(order)]
    out['batch'] = sample.batch.clone() if 'batch' in sample else torch.zeros(order.size, dtype=torch.long)
    out['num_vertices'] = sample.num_vertices.clone()
    out['edge_index'] = torch.from_numpy(rank[sample.edge_index.numpy()])
    ranks = [rank]
    for lvl in range(1, n_levels):
        tr = sample['hierarchy_trace_index_%d' % lvl].numpy()
        nc = int(tr.max()) + 1
        first = np.full(nc, np.iinfo(np.int64).max, dtype=np.int64)
        np.minimum.at(first, tr, ranks[-1])
        c_order = np.argsort(first, kind='stable')
        c_rank = np.empty_like(c_order)
        c_rank[c_order] = np.arange(nc)
        new_tr = np.empty_like(tr)
        new_tr[ranks[-1]] = c_rank[tr]
        out['hierarchy_trace_index_%d' % lvl] = torch.from_numpy(new_tr)
        out['hierarchy_edge_index_%d' % lvl] = torch.from_numpy(c_rank[sample['hierarchy_edge_index_%d' % lvl].numpy()])
        ranks.append(c_rank)
    from .data import sample_keys
    for k in sample_keys(sample):
        if k.startswith('hierarchy_dil_'):
            lvl = int(k.rsplit('_', 1)[1])
            out[k] = torch.from_numpy(ranks[lvl][sample[k].numpy()])
        elif k not in out and k not in ('x', 'color', 'mask', 'batch', 'num_vertices', 'edge_index'):
            out[k] = sample[k]                        # (name, labels of other levels, ...: carried over unchanged)
    return out, torch.from_numpy(order)
