"""CPU-only checks of the host logic: the C-ABI library loads and exports every symbol the header
declares, the ctypes table matches the header, collation rules, synthetic mesh invariants, and the
module surface (state_dict keys / seeded init / parameter counts) against the golden records."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from _golden import GOLDEN, ModelFixture, load_npz
from surface_texture_inpainting_net_amd import _lib
from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S
from surface_texture_inpainting_net_amd.data import HierarchicalBatch, collate
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'stin_hip.h')


def _header_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(stin_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), 'build the extension first: python -c "import __graft_entry__ as g; g.build()"'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 18
    for name in syms:
        assert hasattr(lib, name), 'libstin_hip.so lacks %s' % name
    assert set(syms) == set(_lib.SIGNATURES.keys())


def test_library_host_only_entry_points():
    lib = _lib.load()                       # no GPU needed for these calls
    assert lib.stin_version() == 100
    assert lib.stin_error_string(0) == b'ok'
    assert b'workspace' in lib.stin_error_string(-4)
    assert lib.stin_colreduce_workspace_bytes(64, 1) >= 1024 * 2 * 64 * 8
    assert lib.stin_colreduce_workspace_bytes(0, 1) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.StinLibraryError):
        _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'surface_texture_inpainting_net_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('# oracle', ''), '%s mentions the oracle' % f


def test_state_dict_layout_and_param_counts():
    rec = json.load(open(os.path.join(GOLDEN, 'param_counts.json')))
    base = dict(output_nc=3, ngf=64, norm='instance', pooling_type='max')
    n = S.define_G(input_nc=10, filter_type='edgeconvtransinv', n_blocks=9, n_levels=2, **base)
    assert sum(p.numel() for p in n.parameters()) == rec['3d_transinv_nl2_nb9'] == 4202051
    assert {k: list(v.shape) for k, v in n.state_dict().items()} == rec['3d_state_dict_keys']
    n1 = S.define_G(input_nc=4, filter_type='edgeconv', n_blocks=9, n_levels=1, **base)
    assert sum(p.numel() for p in n1.parameters()) == rec['c1_edgeconv_nl1_nb9'] == 1050691
    n3 = S.define_G(input_nc=10, filter_type='edgeconvtransinv', n_blocks=9, n_levels=3, **base)
    assert sum(p.numel() for p in n3.parameters()) == rec['3d_transinv_nl3_nb9'] == 16794947
    for m in n.modules():
        if isinstance(m, torch.nn.Linear) and m.bias is not None:
            assert float(m.bias.abs().max()) == 0.0


@pytest.mark.parametrize('name', ['g1_imagegraph_edgeconv', 'g2_3level_transinv_max', 'g5_sageconv', 'g6_graphnorm'])
def test_reference_state_dicts_load(name):
    fx = ModelFixture(name)
    net = S.define_G(**fx.cfg)
    missing, unexpected = net.load_state_dict(fx.state_dict, strict=True)
    assert not missing and not unexpected


def test_error_behaviour_matches_reference():
    with pytest.raises(NotImplementedError):
        S.define_G(input_nc=3, output_nc=3, ngf=8, filter_type='gatconv')
    with pytest.raises(AssertionError):
        S.SurfaceTextureInpaintingNet(3, 3, 'edgeconv', n_blocks=-1)
    net = S.SurfaceTextureInpaintingNet(3, 3, 'edgeconv', ngf=8, pooling_type='median')
    with pytest.raises(ValueError):
        net._pooling(torch.zeros(2, 8), None)
    s = HierarchicalBatch(x=torch.zeros(2, 3))
    with pytest.raises(KeyError):
        s['hierarchy_trace_index_1']
    with pytest.raises(AttributeError):
        s.edge_index


def test_fused_weight_restructure_is_exact_algebra():
    """A_i + B_j must equal W1 [x_i ; x_j - x_i] + b1 (and W1 (x_j - x_i) + b1 for TransInv)."""
    from surface_texture_inpainting_net_amd import modules as M
    torch.manual_seed(3)
    x = torch.randn(50, 6, dtype=torch.float64)
    i, j = torch.randint(0, 50, (200,)), torch.randint(0, 50, (200,))
    for mod, din in ((None, True), (M.EdgeConvTransInv, False)):
        f = M.get_gcn_filter(6, 8, module=mod, double_input=din).double()
        torch.nn.init.normal_(f.nn[0].bias)
        sc = torch.nn.Linear(6, 8).double()
        wcat, bcat, w2e = f.fused_weights(sc)
        Y = x @ wcat.t() + bcat
        H = 16
        feat = (x[j] - x[i]) if mod is not None else torch.cat([x[i], x[j] - x[i]], 1)
        want = feat @ f.nn[0].weight.t() + f.nn[0].bias
        assert torch.allclose(Y[i, :H] + Y[j, H:2 * H], want, atol=1e-12)
        assert torch.allclose(Y[:, 2 * H:], sc(x), atol=1e-12)
        assert torch.equal(w2e[:, :H], f.nn[2].weight) and torch.equal(w2e[:, H], f.nn[2].bias)
        assert float(w2e[:, H + 1:].abs().max()) == 0.0


def test_collate_matches_reference_collation():
    """g3_graphs.npz holds the two single graphs, g3_batch2_unequal.npz the batch the REFERENCE's
    HierarchicalData.__inc__ + PyG collate produced from them."""
    z = load_npz('g3_graphs')
    graphs = []
    for gi in range(2):
        graphs.append(HierarchicalBatch(**{k.split('.', 1)[1]: torch.from_numpy(v) for k, v in z.items()
                                           if k.startswith('g%d.' % gi)}))
    got = collate(graphs)
    want = ModelFixture('g3_batch2_unequal').sample()
    for k in want.keys():
        assert torch.equal(got[k], want[k]), k
    assert got.num_vertices.shape == (2, 3) and got.num_vertices.dtype == torch.int32


def test_collate_dilated_offsets_fixed_vs_reference_quirk():
    a = make_synthetic_mesh(150, 2, seed=1, dilations=(2,))
    b = make_synthetic_mesh(200, 2, seed=2, dilations=(2,))
    key = 'hierarchy_dil_2_edge_index_1'
    fixed = collate([a, b])
    quirk = collate([a, b], fix_dilated_offsets=False)
    n1_a, n0_a = int(a.num_vertices[0, 1]), int(a.num_vertices[0, 0])
    ea = a[key].shape[1]
    assert torch.equal(fixed[key][:, ea:], b[key] + n1_a)          # correct: offset by N_level
    assert torch.equal(quirk[key][:, ea:], b[key] + n0_a)          # reference: offset by N0
    assert int(fixed[key].max()) < int(fixed.num_vertices.sum(0)[1])


def test_synthetic_mesh_invariants():
    s = make_synthetic_mesh(2500, 3, seed=5, dilations=(2, 4))
    nv = s.num_vertices[0].tolist()
    assert nv[0] == 2500 and nv[1] == int(0.3 * nv[0]) and nv[2] == int(0.3 * nv[1])
    ei = s.edge_index
    assert ei.dtype == torch.int64 and int(ei.max()) < nv[0] and bool((ei[0] != ei[1]).all())
    key = ei[0] * nv[0] + ei[1]
    assert torch.equal(key, torch.sort(key).values), 'edges grouped by source, sorted'
    rev = torch.sort(ei[1] * nv[0] + ei[0]).values
    assert torch.equal(rev, key), 'symmetric edge set'
    for lvl in (1, 2):
        tr = s['hierarchy_trace_index_%d' % lvl]
        assert tr.shape[0] == nv[lvl - 1] and int(tr.max()) == nv[lvl] - 1
        assert int(torch.bincount(tr, minlength=nv[lvl]).min()) >= 1
        e = s['hierarchy_edge_index_%d' % lvl]
        assert int(e.max()) < nv[lvl] and bool((e[0] != e[1]).all())
    assert s.x.shape == (2500, 10) and s.mask.shape == (2500, 1) and s.color.shape == (2500, 3)
    assert 0.15 < float((s.mask > 0).float().mean()) < 0.35
    big = make_synthetic_mesh(200_000, 1, seed=0, dilations=())
    assert big.x.shape[0] == 200_704 and big.edge_index.shape[1] == 1_200_642


def test_scene_io_round_trip_in_reference_schema(tmp_path):
    """Write a synthetic scene in the reference's graphs/<scene>.pt + masks/...npz schema and read it back the way
    ScanNetGraphColorDataSet.__getitem__ assembles a sample (feature layout, key names, dilation fall-back)."""
    from surface_texture_inpainting_net_amd.scene_io import load_scene, save_scene_like_reference
    s = make_synthetic_mesh(400, 3, seed=9, dilations=(2, 4))
    gp, mp = str(tmp_path / 'scene0000_00.pt'), str(tmp_path / '0.npz')
    save_scene_like_reference(s, gp, mp, dilation_dists=(2, 4, 8))      # dist 8 is empty -> falls back to dist 4
    t = load_scene(gp, mp, end_level=3)
    assert t['name'] == 'scene0000_00'
    for k in ('edge_index', 'hierarchy_edge_index_1', 'hierarchy_edge_index_2', 'hierarchy_trace_index_1',
              'hierarchy_trace_index_2', 'hierarchy_dil_2_edge_index_2', 'hierarchy_dil_4_edge_index_2', 'mask', 'batch'):
        assert torch.equal(t[k], s[k]), k
    assert torch.equal(t['hierarchy_dil_8_edge_index_2'], s['hierarchy_dil_4_edge_index_2'])
    assert torch.equal(t.num_vertices, s.num_vertices) and t.num_vertices.dtype == torch.int32
    assert torch.allclose(t.color, s.color, atol=1e-6)
    assert torch.allclose(t.x, s.x, atol=1e-6)          # [rgb*known, normal, pos/1.5, known]
    t2 = load_scene(gp, mp, end_level=2)
    assert t2.num_vertices.shape == (1, 2) and 'hierarchy_trace_index_2' not in t2


def test_scene_reader_locality_order_is_a_consistent_renumbering(tmp_path):
    """load_scene(locality_order=True): the same scene with the vertices of every level renumbered (Morton order of the
    positions, first-child order above).  A relabelling must be invisible to the network: the CPU oracle on the renumbered
    sample gives the rows of the original output in the new order (same weights), edge lists keep their order, and the
    numbering is more local than the file's."""
    from oracle import stin_oracle
    from surface_texture_inpainting_net_amd.scene_io import load_scene, save_scene_like_reference
    s = make_synthetic_mesh(900, 3, seed=12, dilations=(2,))
    gp, mp = str(tmp_path / 'scene0001_00.pt'), str(tmp_path / '0.npz')
    save_scene_like_reference(s, gp, mp, dilation_dists=(2,))
    a = load_scene(gp, mp, end_level=3)
    b = load_scene(gp, mp, end_level=3, locality_order=True)
    order = b['vertex_order']
    assert sorted(order.tolist()) == list(range(a.x.shape[0]))
    assert torch.equal(b.x, a.x[order]) and torch.equal(b.color, a.color[order]) and torch.equal(b.mask, a.mask[order])
    rank = torch.empty_like(order)
    rank[order] = torch.arange(order.numel())
    assert torch.equal(b.edge_index, rank[a.edge_index])                       # same edges, same order, new names
    spread = lambda e: float((e[0] - e[1]).abs().float().mean())
    assert spread(b.edge_index) < 0.5 * spread(a.edge_index)
    assert spread(b['hierarchy_edge_index_1']) < 0.6 * spread(a['hierarchy_edge_index_1'])
    cfg = dict(input_nc=10, output_nc=3, ngf=8, filter_type='edgeconvtransinv', norm='instance', n_blocks=2, n_levels=2,
               pooling_type='max', dilations=[1, 2])
    torch.manual_seed(3)
    net = stin_oracle.define_G(**cfg)
    with torch.no_grad():
        ya, yb = net(a), net(b)
    assert float((yb - ya[order]).abs().max()) <= 2e-5


def test_cpu_tensors_are_rejected_not_silently_computed():
    """No CPU / eager fallback: a CPU sample must fail loudly, never produce an answer."""
    s = make_synthetic_mesh(100, 2, seed=3, dilations=())
    net = S.define_G(input_nc=10, output_nc=3, ngf=8, filter_type='edgeconv', norm='instance', n_blocks=1, n_levels=1,
                     pooling_type='max')
    with pytest.raises((AssertionError, TypeError)):
        net(s)


def test_environment_switches_are_the_documented_ones():
    """Round 6 pruned the STIN_* environment switches from 73 to 29: every switch the product reads is in
    that list, and the list names nothing the product no longer reads."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, 'surface_texture_inpainting_net_amd')
    found = set()
    for path in glob.glob(os.path.join(pkg, 'csrc', '*')):
        if path.endswith(('.hip', '.inc', '.h')):
            found |= set(re.findall(r'getenv\("(STIN_[A-Z0-9_]+)"\)', open(path).read()))
    for path in glob.glob(os.path.join(pkg, '*.py')):
        found |= set(re.findall(r"environ(?:\.get)?[\(\[]\s*'(STIN_[A-Z0-9_]+)'", open(path).read()))
    design = open(os.path.join(root, 'DESIGN.md')).read()
    sec = design[design.index('### 5b. Environment switches'):design.index('## 6. Measurement')]
    documented = set(re.findall(r'`(STIN_[A-Z0-9_]+)`', sec))
    assert found == documented, (sorted(found - documented), sorted(documented - found))
    assert len(found) <= 30
