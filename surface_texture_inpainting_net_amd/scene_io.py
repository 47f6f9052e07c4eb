"""Reader for the reference's on-disk scene format -> HierarchicalBatch.

The reference's offline preprocessing (preprocessing/graph_level_generation.py:492-536) writes per scene a
``graphs/<scene>.pt`` dict with
    vertices      list[L]  level 0: [N0, 10] = pos(0:3) rgb in [0,1](3:6) normal(6:9) original-id(9); others [N_l, 3]
    edges         list[L]  int64 [E_l, 2]   (row-wise pairs: source, target)
    traces        list[L]  int64; traces[0] maps to the ORIGINAL mesh, traces[l] (l >= 1): level l-1 -> level l
    dilated_edges list[L]  None or list over dilation_dists of int64 [E_d, 2]  (dilated node, centre)
    dilation_dists list[int]
and per mask a ``masks/<mask_name>/<scene>/<id>.npz`` with ``vertex_mask`` (0 = known, > 0 = inpaint, the value
being the distance to the nearest known vertex).  ``load_scene`` assembles exactly the sample that
datasets/scannetcolorgraph_dataloader.py:83-156 hands to the model (same feature layout, same key names, same
fall-back to the previous dilation distance when one is empty), plus the CoordsNormalization transform
(transform/coords_normalization.py:16: x[:, 6:9] /= max_sizes).  Everything stays on the CPU (worker side);
``sample.to(device)`` and the GPU plan build happen in the training process.
"""
import numpy as np
import torch

from .data import HierarchicalBatch


def sample_from_tensors(saved, vertex_mask, end_level, cropped=False, coords_max_sizes=(1.5, 1.5, 1.5), name=None):
    """saved: the dict of a graphs/<scene>.pt file; vertex_mask: int array [N0]."""
    coords = [v.clone() if torch.is_tensor(v) else torch.as_tensor(v) for v in saved['vertices'][:end_level]]
    coords[0][:, 3:6] = coords[0][:, 3:6] * 2.0 - 1.0                       # colour to [-1, 1] (:95)
    edges = saved['edges'][:end_level]
    dil = saved.get('dilated_edges')
    dists = saved.get('dilation_dists')
    if dil is not None and dists is not None:
        dil = dil[:end_level]
    else:
        dil, dists = None, None
    mask = torch.as_tensor(np.asarray(vertex_mask)).unsqueeze(1)
    known = (mask == 0)
    x = torch.cat([coords[0][:, 3:6] * known, coords[0][:, 6:9], coords[0][:, :3], known], dim=-1).float()   # (:115)
    s = HierarchicalBatch(x=x, color=coords[0][:, 3:6].float(), mask=mask.long(),
                          edge_index=torch.as_tensor(edges[0]).t().contiguous().long())
    if name is not None:
        s['name'] = name
    # full scenes carry the trace back to the original mesh at position 0, crops do not (:124-128)
    traces = saved['traces'][:end_level - 1] if cropped else saved['traces'][1:end_level]
    nv = [int(coords[0].shape[0])]
    for lvl in range(1, len(edges)):
        s['hierarchy_edge_index_%d' % lvl] = torch.as_tensor(edges[lvl]).t().contiguous().long()
        if dil is not None and dil[lvl] is not None:
            for i, d in enumerate(dists):
                cur = dil[lvl][i]
                if len(cur) > 0:
                    s['hierarchy_dil_%s_edge_index_%d' % (d, lvl)] = torch.as_tensor(cur).t().contiguous().long()
                elif i > 0:                                                  # empty set: reuse the previous distance (:143-145)
                    s['hierarchy_dil_%s_edge_index_%d' % (d, lvl)] = torch.as_tensor(dil[lvl][i - 1]).t().contiguous().long()
        tr = torch.as_tensor(traces[lvl - 1]).long()
        s['hierarchy_trace_index_%d' % lvl] = tr
        nv.append(int(tr.max()) + 1)
    s['num_vertices'] = torch.tensor([nv], dtype=torch.int32)
    s['batch'] = torch.zeros(nv[0], dtype=torch.long)
    if coords_max_sizes is not None:                                         # CoordsNormalization on the position channels
        s['x'][:, 6:9] = s['x'][:, 6:9] / torch.tensor(coords_max_sizes, dtype=s['x'].dtype)
    return s


def load_scene(graph_path, mask_path, end_level=3, cropped=False, coords_max_sizes=(1.5, 1.5, 1.5), locality_order=False):
    """graphs/<scene>.pt + masks/<name>/<scene>/<id>.npz -> HierarchicalBatch (CPU).
    locality_order=True (not in the reference): renumber the vertices of every level so that memory order follows space
    (synthetic.renumber_by_locality: Morton order of the positions at level 0, first-child order above; edge lists and traces
    keep their order, index values are relabelled) - the GPU kernels' neighbour gathers then hit L2 instead of crossing the
    fabric (3-4 % of a training step at 200 k vertices).  The sample carries `vertex_order` (new row -> row of the scene file)
    so per-vertex outputs can be mapped back: out_file_order[sample.vertex_order] = out."""
    saved = torch.load(graph_path, map_location='cpu', weights_only=False)
    with open(mask_path, 'rb') as f:
        vertex_mask = np.load(f, allow_pickle=True)['vertex_mask']
    name = str(graph_path).rsplit('/', 1)[-1].rsplit('.', 1)[0]
    sample = sample_from_tensors(saved, vertex_mask, end_level, cropped, coords_max_sizes, name)
    if locality_order:
        from .synthetic import renumber_by_locality
        sample, order = renumber_by_locality(sample)
        sample['vertex_order'] = order
    return sample


def save_scene_like_reference(sample, graph_path, mask_path, dilation_dists=(2, 4, 8, 16)):
    """Write a single-graph HierarchicalBatch in the reference's on-disk schema (test / synthetic-data helper;
    the inverse of load_scene up to the normalisations)."""
    L = int(sample.num_vertices.shape[-1])
    x = sample.x
    n0 = x.shape[0]
    v0 = torch.zeros(n0, 10)
    v0[:, 0:3] = x[:, 6:9] * 1.5
    v0[:, 3:6] = (sample.color + 1.0) / 2.0
    v0[:, 6:9] = x[:, 3:6]
    v0[:, 9] = torch.arange(n0)
    vertices, edges, traces, dilated = [v0], [sample.edge_index.t().contiguous()], [torch.arange(n0)], [None]
    for lvl in range(1, L):
        n = int(sample.num_vertices.reshape(-1)[lvl])
        vertices.append(torch.zeros(n, 3))
        edges.append(sample['hierarchy_edge_index_%d' % lvl].t().contiguous())
        traces.append(sample['hierarchy_trace_index_%d' % lvl])
        sets = []
        for d in dilation_dists:
            k = 'hierarchy_dil_%s_edge_index_%d' % (d, lvl)
            sets.append(sample[k].t().contiguous() if k in sample else [])
        dilated.append(sets if any(len(z) > 0 for z in sets) else None)
    torch.save({'vertices': vertices, 'edges': edges, 'traces': traces, 'dilated_edges': dilated,
                'dilation_dists': list(dilation_dists)}, graph_path)
    np.savez(mask_path, vertex_mask=sample.mask.reshape(-1).numpy())
