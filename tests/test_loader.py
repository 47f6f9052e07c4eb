"""Scene loader in front of the hot path: rank sharding, prefetch thread, resident graph cache."""
import pytest
import torch

from surface_texture_inpainting_net_amd.loader import SceneLoader, shard_indices
from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh


def test_shard_indices_cover_every_item_equally_on_all_ranks():
    for n, world in ((10, 1), (10, 4), (7, 8), (1201, 8)):
        for epoch in (0, 1):
            parts = [shard_indices(n, epoch, seed=3, shuffle=True, rank=r, world_size=world) for r in range(world)]
            assert len({len(p) for p in parts}) == 1, 'every rank runs the same number of steps'
            seen = sorted(i for p in parts for i in p)
            assert set(seen) == set(range(n)) and len(seen) == (n + world - 1) // world * world
        a = shard_indices(n, 0, 3, True, 0, world)
        assert a == shard_indices(n, 0, 3, True, 0, world) and (n < 3 or a != shard_indices(n, 1, 3, True, 0, world))
    assert shard_indices(5, 0, shuffle=False) == [0, 1, 2, 3, 4]


def test_loader_on_cpu_orders_batches_and_propagates_worker_errors():
    scenes = [make_synthetic_mesh(60 + 10 * i, 2, seed=i, dilations=()) for i in range(5)]
    ld = SceneLoader(scenes, 'cpu', batch_size=2, shuffle=False)
    got = list(ld.epoch(0))
    assert len(got) == 3 == ld.steps_per_epoch()
    assert [int(b.num_vertices.shape[0]) for b in got] == [2, 2, 1]
    assert got[0].x.shape[0] == scenes[0].x.shape[0] + scenes[1].x.shape[0]
    order = shard_indices(5, 1, seed=9, shuffle=True)
    ld = SceneLoader(scenes, 'cpu', batch_size=1, shuffle=True, seed=9)
    assert [b.x.shape[0] for b in ld.epoch(1)] == [scenes[i].x.shape[0] for i in order]

    def boom():
        raise RuntimeError('broken scene file')
    with pytest.raises(RuntimeError, match='broken scene'):
        list(SceneLoader([scenes[0], boom], 'cpu', shuffle=False).epoch(0))


@pytest.mark.gpu
def test_resident_graph_cache_reuses_plans_and_gives_identical_results():
    from surface_texture_inpainting_net_amd import surfacetextureinpaintingnet as S
    dev = 'cuda:0'
    cfg = dict(input_nc=10, output_nc=3, ngf=64, filter_type='edgeconvtransinv', norm='instance', n_blocks=2, n_levels=2,
               pooling_type='max', dilations=[1, 2])
    torch.manual_seed(0)
    net = S.define_G(**cfg).to(dev)
    scenes = [make_synthetic_mesh(4000 + 700 * i, 3, seed=20 + i, dilations=(2,)) for i in range(3)]
    with torch.no_grad():
        want = [net(s.to(dev)) for s in scenes]
        ld = SceneLoader(scenes, dev, shuffle=False, cache_bytes=1 << 30)
        plans = {}
        for epoch in range(3):
            for i, b in enumerate(ld.epoch(epoch)):
                out = net(b)
                assert torch.equal(out, want[i])
                plans.setdefault(i, []).append(b._plan_cache)
        assert len(ld.cache) == 3 and ld.cache.hits >= 6
        for i in range(3):
            assert plans[i][1] is plans[i][0] and plans[i][2] is plans[i][0], 'second / third visit reuse the resident plan'
        # masks change between visits while the graph stays: features are re-uploaded, graph + plan are not
        scenes[1]['x'] = scenes[1].x * 0.5
        b = list(ld.epoch(0))[1]
        assert torch.equal(net(b), net(scenes[1].to(dev)))
        # with a model the loader builds each batch's plan ahead (side streams, after the upload, joined at first use),
        # with and without the resident cache
        # (round 6) a plan that stays resident is built with the vertices renumbered by locality (locality_order, the default): paid
        # once per scene, invisible at the boundary - colours come back in the scene's vertex order and equal the file-order run up to
        # the rounding of the instance-norm column sums (other row order); locality_order=False keeps the bits
        for cache_bytes, loc in ((0, True), (1 << 30, True), (1 << 30, False)):
            ahead = SceneLoader(scenes, dev, shuffle=False, cache_bytes=cache_bytes, model=net, locality_order=loc)
            for epoch in range(2):
                for i, b in enumerate(ahead.epoch(epoch)):
                    assert b._plan_cache is not None
                    assert epoch == 1 and cache_bytes or b._plan_cache._pending, 'a fresh plan is handed over un-joined'
                    out = net(b)
                    renumbered = bool(cache_bytes) and loc
                    assert (b._plan_cache.order0 is not None) == renumbered
                    if renumbered:
                        assert i == 1 or float((out - want[i]).abs().max()) <= 2e-5
                        assert sorted(b._plan_cache.order0.tolist()) == list(range(b.x.shape[0]))
                    else:
                        assert i == 1 or torch.equal(out, want[i])
                    assert not b._plan_cache._pending
        # a cache too small for anything must behave like no cache
        tiny = SceneLoader(scenes, dev, shuffle=False, cache_bytes=1024)
        for epoch in range(2):
            for i, b in enumerate(tiny.epoch(epoch)):
                assert i == 1 or torch.equal(net(b), want[i])
        assert len(tiny.cache) == 0


def test_loader_keeps_parsed_scene_files_in_host_memory(tmp_path, monkeypatch):
    """File items (the reference's graph .pt + mask .npz): parsed once, revisits served from the host LRU; a cache too
    small for a scene behaves like none."""
    from surface_texture_inpainting_net_amd import loader as L
    from surface_texture_inpainting_net_amd import scene_io
    items = []
    for i in range(3):
        s = make_synthetic_mesh(300 + 50 * i, 3, seed=i, dilations=(2,))
        gp, mp = str(tmp_path / ('g%d.pt' % i)), str(tmp_path / ('m%d.npz' % i))
        scene_io.save_scene_like_reference(s, gp, mp, dilation_dists=(2,))
        items.append((gp, mp))
    calls = []
    real = L.load_scene
    monkeypatch.setattr(L, 'load_scene', lambda *a, **k: (calls.append(a[0]), real(*a, **k))[1])
    ld = SceneLoader(items, 'cpu', shuffle=False)
    first = [b.x.clone() for b in ld.epoch(0)]
    again = [b.x.clone() for b in ld.epoch(1)]
    assert len(calls) == 3 and all(torch.equal(a, b) for a, b in zip(first, again))
    calls.clear()
    tiny = SceneLoader(items, 'cpu', shuffle=False, host_cache_bytes=1024)
    for e in range(2):
        list(tiny.epoch(e))
    assert len(calls) == 6
