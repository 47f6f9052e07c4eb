"""CPU restatement of the reference's offline graph preprocessing that feeds the hot path.

TEST INFRASTRUCTURE: only tests/, oracle/make_golden.py and bench-side checkers may import this module.

* `dilated_edges`   - preprocessing/graph_dilation.py:50-137 (`compute_all_node_dilated_edges` /
  `compute_dilated_edges`): from every centre c and each of its one-hop neighbours h a walker steps away from c,
  at each step taking the neighbour of the current vertex whose direction - projected into the current vertex's
  tangent plane by the reference's own `plane_projection` formula (:27-28, NOT an orthogonal projection unless
  its arguments are unit vectors) - has the largest cosine with the running direction (`>=`: the LAST maximum in
  adjacency order wins, candidates start at similarity 0, :108-118); vertices adjacent to c and the previous
  vertex are excluded (:111); for each requested dilation d the vertex reached after d hops yields the edge
  [far, c] (:125-128); the result per dilation is coalesced (sorted by (row 0, row 1), duplicates dropped, :66-70)
  and returned row-wise [E_d, 2].  The arithmetic type is the one of `poses` (the pipeline passes float64,
  preprocessing/graph_level_generation.py:463-465; `dil_test` float32).
  Every floating-point operation is written out in a fixed order (dot = (a0 b0 + a1 b1) + a2 b2, no FMA) so that
  the HIP kernel can reproduce it bit for bit; torch's own dot/norm may associate differently, which can only
  matter for similarities that tie to the last bit.
* `vertex_clustering` - preprocessing/graph_level_generation.py:193-244: voxel bins `coords // voxel_size`, unique
  bins in lexicographic order = coarse ids, trace = bin of every vertex, coarse edges = image of the fine edges
  without self loops and duplicates (returned sorted, the reference's order inside a bin is a Python set order),
  coarse coordinates = mean of the members (float32 as the reference stores them).
"""
import numpy as np


def coalesce(edge_index, n=None):
    """torch_geometric.utils.coalesce on a [2, E] int array: sort by (row 0, row 1), drop duplicates."""
    edge_index = np.asarray(edge_index, dtype=np.int64)
    if edge_index.size == 0:
        return edge_index.reshape(2, 0)
    n = int(edge_index.max()) + 1 if n is None else int(n)
    key = np.unique(edge_index[0] * n + edge_index[1])
    return np.stack([key // n, key % n])


def _dot(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def _norm(a):
    return np.sqrt(_dot(a, a))


def _plane_projection(n, u):
    d = _dot(u, n)
    den = _norm(n) * _norm(u)
    return np.array([u[0] - (n[0] * d) / den, u[1] - (n[1] * d) / den, u[2] - (n[2] * d) / den], dtype=u.dtype)


def _cos(a, b):
    return _dot(a, b) / (_norm(a) * _norm(b))


def dilated_edges(edge_index, poses, norms, dilations):
    """edge_index [2, E] (any order, duplicates allowed), poses/norms [N, 3] float32 or float64,
    dilations: ints >= 2 -> list of [E_d, 2] int64 arrays (row = [far vertex, centre]), one per dilation
    (an empty list entry [] when a dilation produced no edge, as the reference leaves it)."""
    dilations = [int(d) for d in dilations]
    poses = np.asarray(poses)
    norms = np.asarray(norms, dtype=poses.dtype)
    ei = coalesce(edge_index)
    n_adj = int(ei.max()) + 1 if ei.size else 0
    adj = [[] for _ in range(max(n_adj, 0))]
    for s, d in zip(ei[0], ei[1]):
        adj[s].append(int(d))
    out = [[] for _ in dilations]
    max_d = max(dilations)
    with np.errstate(all='ignore'):
        for c in range(poses.shape[0]):          # vertices beyond the largest edge endpoint have no adjacency list in the
            if c >= n_adj:                       # reference (it would raise IndexError there); treated as isolated here
                continue
            hood = adj[c]
            for h in hood:
                if h == c:
                    continue
                last, cur = c, h
                cur_norm = norms[cur]
                direction = poses[cur] - poses[last]
                di = 0
                for d in range(2, max_d + 1):
                    best, best_sim = -1, poses.dtype.type(0.0)
                    for nb in adj[cur]:
                        if nb in hood or nb == last:
                            continue
                        nd = poses[nb] - poses[cur]
                        sim = _cos(_plane_projection(cur_norm, direction), _plane_projection(cur_norm, nd))
                        if sim >= best_sim:
                            best, best_sim = nb, sim
                    if best == -1:
                        break
                    if d in dilations:
                        out[di].append((best, c))
                        di += 1
                    last, cur = cur, best
                    cur_norm = norms[cur]
                    direction = _plane_projection(cur_norm, direction)
                    nn = _norm(direction)
                    direction = np.array([direction[0] / nn, direction[1] / nn, direction[2] / nn], dtype=poses.dtype)
    res = []
    for lst in out:
        if not lst:
            res.append([])
        else:
            res.append(coalesce(np.array(lst, dtype=np.int64).T).T.copy())
    return res


def vertex_clustering(coords, edge_index, voxel_size):
    """-> (new_coords float32 [Nc, 3], trace int64 [N], coarse_edges int64 [Ec, 2] sorted by (row 0, row 1))."""
    coords = np.asarray(coords)
    bins = coords // voxel_size
    _, trace = np.unique(bins, axis=0, return_inverse=True)
    trace = trace.reshape(-1).astype(np.int64)
    nc = int(trace.max()) + 1 if trace.size else 0
    ei = np.asarray(edge_index, dtype=np.int64)
    ce = np.stack([trace[ei[0]], trace[ei[1]]]) if ei.size else np.zeros((2, 0), np.int64)
    ce = ce[:, ce[0] != ce[1]]
    ce = coalesce(ce, nc)
    new_coords = np.empty((nc, 3), dtype=np.float32)
    for b in range(nc):
        new_coords[b] = coords[trace == b].mean(axis=0)
    return new_coords, trace, ce.T.copy()
