"""Graph preprocessing either side of the hot path: dilated-edge generation
(preprocessing/graph_dilation.py) and voxel vertex clustering (preprocessing/graph_level_generation.py:193-244).

CPU part: the restatement oracle/dilation_oracle.py against fixture g9, which holds the outputs of the reference's
OWN functions (its dil_test toy graph in float32 - the only known-answer candidate the reference has - and a
jittered mesh in float64 as the pipeline runs it).  GPU part: the HIP walk kernel through the C ABI against the
oracle and the fixture, BIT-EXACT (index work), plus structural properties at 200k vertices.
"""
import numpy as np
import pytest
import torch

from _golden import load_npz
from oracle import dilation_oracle as D

DEV = 'cuda:0'


def _same(a, b):
    if isinstance(a, list) or isinstance(b, list):
        return isinstance(a, list) and isinstance(b, list) and len(a) == 0 and len(b) == 0
    return np.array_equal(np.asarray(a), np.asarray(b))


def _jittered_mesh(n_side, seed, dtype=np.float64):
    rng = np.random.default_rng(seed)
    idx = np.arange(n_side * n_side).reshape(n_side, n_side)
    e = list(zip(idx[:, :-1].ravel(), idx[:, 1:].ravel())) + list(zip(idx[:-1, :].ravel(), idx[1:, :].ravel())) + \
        list(zip(idx[:-1, :-1].ravel(), idx[1:, 1:].ravel()))
    e = np.array(e).T
    e = np.concatenate([e, e[::-1]], axis=1)
    perm = rng.permutation(n_side * n_side)
    e = perm[e]
    gx, gy = np.meshgrid(np.arange(n_side), np.arange(n_side), indexing='ij')
    pos = np.stack([gx.ravel(), gy.ravel(), np.zeros(n_side * n_side)], 1).astype(np.float64) + rng.normal(0, 0.2, (n_side * n_side, 3))
    p = np.empty_like(pos)
    p[perm] = pos
    nrm = rng.normal(0, 0.3, pos.shape)
    nrm[:, 2] += 1.0
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return e[:, rng.permutation(e.shape[1])].astype(np.int64), p.astype(dtype), nrm.astype(dtype)


# ------------------------------------------------------------------------------ CPU: oracle vs the reference's outputs
def test_oracle_dilated_edges_match_reference_fixture():
    g = load_npz('g9_preprocessing')
    out = D.dilated_edges(g['toy_edge_index'], g['toy_pos'], g['toy_nrm'], g['toy_dilations'])
    for d, o in zip(g['toy_dilations'], out):
        assert _same(o, g['toy_d%d' % d]), d
    out = D.dilated_edges(g['mesh_edge_index'], g['mesh_pos'], g['mesh_nrm'], g['mesh_dilations'])
    for d, o in zip(g['mesh_dilations'], out):
        assert _same(o, g['mesh_d%d' % d]), d
    assert g['mesh_pos'].dtype == np.float64 and g['toy_pos'].dtype == np.float32


def test_oracle_vertex_clustering_matches_reference_fixture():
    g = load_npz('g9_preprocessing')
    coords, trace, edges = D.vertex_clustering(g['mesh_pos'], g['mesh_edge_index'], float(g['vc_voxel']))
    assert np.array_equal(trace, g['vc_trace'])
    assert np.array_equal(edges, g['vc_edges'])
    assert coords.dtype == np.float32 and np.array_equal(coords, g['vc_coords'])


def test_oracle_dilation_edge_cases():
    pos = np.zeros((4, 3), np.float64)
    nrm = np.tile(np.array([0., 0., 1.]), (4, 1))
    assert D.dilated_edges(np.zeros((2, 0), np.int64), pos, nrm, [2, 4]) == [[], []]
    # a path 0-1-2-3 along x: from centre 0 via 1 the walk reaches 2 (d=2) and 3 (d=3); coincident points give NaN -> no edge
    pos = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0]], np.float64)
    ei = np.array([[0, 1, 1, 2, 2, 3], [1, 0, 2, 1, 3, 2]])
    d2, d3 = D.dilated_edges(ei, pos, nrm, [2, 3])
    assert d2.tolist() == [[0, 2], [1, 3], [2, 0], [3, 1]] and d3.tolist() == [[0, 3], [3, 0]]
    assert D.dilated_edges(ei, np.zeros((4, 3)), nrm, [2]) == [[]]


# ------------------------------------------------------------------------------ GPU: HIP kernel vs oracle / fixture
@pytest.mark.gpu
def test_hip_dilated_edges_match_reference_fixture_bit_exact():
    from surface_texture_inpainting_net_amd import preprocessing as P
    g = load_npz('g9_preprocessing')
    for tag in ('toy', 'mesh'):
        out = P.dilated_edges(torch.from_numpy(g[tag + '_edge_index']).to(DEV), torch.from_numpy(g[tag + '_pos']).to(DEV),
                              torch.from_numpy(g[tag + '_nrm']).to(DEV), g[tag + '_dilations'].tolist())
        for d, o in zip(g[tag + '_dilations'], out):
            assert _same(o if isinstance(o, list) else o.cpu().numpy(), g['%s_d%d' % (tag, d)]), (tag, d)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('n_side,seed', [(9, 0), (24, 3)])
def test_hip_dilated_edges_equal_oracle(dtype, n_side, seed):
    from surface_texture_inpainting_net_amd import preprocessing as P
    e, p, nrm = _jittered_mesh(n_side, seed, dtype)
    e = np.concatenate([e, e[:, :40]], axis=1)                         # duplicates must not matter
    dil = [2, 3, 4, 8, 16]
    want = D.dilated_edges(e, p, nrm, dil)
    got = P.dilated_edges(torch.from_numpy(e).to(DEV), torch.from_numpy(p).to(DEV), torch.from_numpy(nrm).to(DEV), dil)
    for d, w, o in zip(dil, want, got):
        assert _same(o if isinstance(o, list) else o.cpu().numpy(), w), d


@pytest.mark.gpu
def test_hip_dilated_edges_empty_ragged_and_errors():
    from surface_texture_inpainting_net_amd import preprocessing as P
    pos = torch.zeros(5, 3, dtype=torch.float64, device=DEV)
    nrm = torch.zeros(5, 3, dtype=torch.float64, device=DEV)
    nrm[:, 2] = 1
    assert P.dilated_edges(torch.zeros(2, 0, dtype=torch.long, device=DEV), pos, nrm, [2, 4]) == [[], []]
    ei = torch.tensor([[0, 1, 1, 2, 2, 3, 4], [1, 0, 2, 1, 3, 2, 4]], device=DEV)      # vertex 4: only a self loop
    pos = torch.tensor([[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0], [9, 9, 9]], dtype=torch.float64, device=DEV)
    d2, d3, d5 = P.dilated_edges(ei, pos, nrm, [2, 3, 5])
    assert d2.tolist() == [[0, 2], [1, 3], [2, 0], [3, 1]] and d3.tolist() == [[0, 3], [3, 0]] and d5 == []
    with pytest.raises(IndexError):
        P.dilated_edges(torch.tensor([[0], [7]], device=DEV), pos, nrm, [2])
    with pytest.raises(ValueError):
        P.dilated_edges(ei, pos, nrm, [4, 2])
    with pytest.raises(TypeError):
        P.dilated_edges(ei, pos, nrm.float(), [2])


@pytest.mark.gpu
def test_hip_dilated_edges_full_size_properties():
    """200k-vertex / 1.2M-edge mesh (the reference needs ~30 min per scene for this step)."""
    from surface_texture_inpainting_net_amd import preprocessing as P
    from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
    s = make_synthetic_mesh(200_000, 1, seed=0, dilations=())
    n = s.x.shape[0]
    ei = s.edge_index.to(DEV)
    pos = s.x[:, 6:9].double().to(DEV)
    nrm = torch.nn.functional.normalize(s.x[:, 3:6].double(), dim=1).to(DEV)
    dil = [2, 4, 8, 16]
    a = P.dilated_edges(ei, pos, nrm, dil)
    b = P.dilated_edges(ei[:, torch.randperm(ei.shape[1], device=DEV)], pos, nrm, dil)
    adj = set(map(tuple, ei.t().cpu().numpy().tolist()))
    for d, x, y in zip(dil, a, b):
        assert torch.equal(x, y), 'independent of the input edge order'
        key = x[:, 0] * n + x[:, 1]
        assert bool((key[1:] > key[:-1]).all()), 'sorted by (far, centre), no duplicates'
        assert bool((x[:, 0] != x[:, 1]).all())
        assert 0 < x.shape[0] <= ei.shape[1], 'at most one edge per walker (fewer after duplicates are merged)'
    far, c = a[0][:5000, 0].cpu().numpy(), a[0][:5000, 1].cpu().numpy()
    assert not any((int(f), int(k)) in adj for f, k in zip(far, c)), 'a 2-dilated vertex is never a one-hop neighbour'


@pytest.mark.gpu
def test_vertex_clustering_matches_fixture_and_oracle():
    from surface_texture_inpainting_net_amd import preprocessing as P
    g = load_npz('g9_preprocessing')
    coords, trace, edges = P.vertex_clustering(torch.from_numpy(g['mesh_pos']).to(DEV), torch.from_numpy(g['mesh_edge_index']).to(DEV),
                                               float(g['vc_voxel']))
    assert np.array_equal(trace.cpu().numpy(), g['vc_trace']) and np.array_equal(edges.cpu().numpy(), g['vc_edges'])
    assert np.allclose(coords.cpu().numpy(), g['vc_coords'], rtol=1e-6, atol=1e-6)
    rng = np.random.default_rng(1)
    n = 5000
    c = rng.uniform(-2, 3, (n, 3))
    ei = rng.integers(0, n, (2, 30000))
    wc, wt, we = D.vertex_clustering(c, ei, 0.37)
    gc, gt, ge = P.vertex_clustering(torch.from_numpy(c).to(DEV), torch.from_numpy(ei).to(DEV), 0.37)
    assert np.array_equal(gt.cpu().numpy(), wt) and np.array_equal(ge.cpu().numpy(), we)
    assert np.allclose(gc.cpu().numpy(), wc, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_coalesce_and_vertex_clustering_kernels_edge_cases():
    """stin_coalesce_pairs_i64 / stin_voxel_cluster_f64 against numpy on ragged inputs: duplicates, self loops, a single
    vertex, negative and exactly-on-the-boundary coordinates, every vertex in one voxel, every vertex its own voxel."""
    from surface_texture_inpainting_net_amd import preprocessing as P
    rng = np.random.default_rng(5)
    for n, e in ((1, 1), (5, 40), (1000, 5000), (70000, 300000)):
        ei = rng.integers(0, n, (2, e))
        want = np.unique(ei[0] * n + ei[1])
        got = P.coalesce(torch.from_numpy(ei).to(DEV), n).cpu().numpy()
        assert np.array_equal(got[0] * n + got[1], want)
        keep = ei[:, ei[0] != ei[1]]
        want = np.unique(keep[0] * n + keep[1])
        got = P.coalesce(torch.from_numpy(ei).to(DEV), n, drop_loops=True).cpu().numpy()
        assert np.array_equal(got[0] * n + got[1], want) and got.shape[1] == want.size
    with pytest.raises(IndexError):
        P.coalesce(torch.tensor([[0, 7], [1, 2]], device=DEV), 5)
    # with a vertex map the RAW endpoints index the map: outside [0, len(map)) raises before map[] is read (far out of
    # bounds on purpose: an unchecked read would fault or pick up garbage), inside it the mapped pair is what is coalesced
    vm = torch.tensor([2, 0, 1, 1], device=DEV)
    got = P.coalesce(torch.tensor([[0, 3, 2], [1, 0, 0]], device=DEV), 3, vertex_map=vm).cpu().numpy()
    assert np.array_equal(got, np.array([[1, 2], [2, 0]]))
    for bad in (4, -1, 1 << 40):
        with pytest.raises(IndexError):
            P.coalesce(torch.tensor([[0, bad], [1, 2]], device=DEV), 3, vertex_map=vm)
    for c, v in ((np.array([[0.3, -0.2, 5.0]]), 0.5),
                 (rng.uniform(-50, 50, (3000, 3)), 1.0),
                 (np.round(rng.uniform(-4, 4, (2000, 3)) * 4) / 4, 0.25),          # coordinates exactly on voxel boundaries
                 (rng.uniform(0, 1e-3, (500, 3)), 10.0),                            # one voxel
                 (np.arange(900, dtype=np.float64).reshape(300, 3) * 7.0, 0.1)):    # every vertex alone
        n = c.shape[0]
        ei = rng.integers(0, n, (2, 4 * n))
        wc, wt, we = D.vertex_clustering(c, ei, v)
        gc, gt, ge = P.vertex_clustering(torch.from_numpy(c).to(DEV), torch.from_numpy(ei).to(DEV), v)
        assert np.array_equal(gt.cpu().numpy(), wt) and np.array_equal(ge.cpu().numpy().reshape(-1, 2), we.reshape(-1, 2))
        assert np.allclose(gc.cpu().numpy(), wc, rtol=1e-6, atol=1e-6)
        again = P.vertex_clustering(torch.from_numpy(c).to(DEV), torch.from_numpy(ei).to(DEV), v)
        assert torch.equal(again[0], gc) and torch.equal(again[1], gt) and torch.equal(again[2], ge)      # deterministic
    with pytest.raises(ValueError):
        P.vertex_clustering(torch.tensor([[0.0, 0.0, 0.0], [1e9, 0.0, 0.0]], device=DEV), torch.zeros(2, 0, dtype=torch.long, device=DEV), 1e-3)
