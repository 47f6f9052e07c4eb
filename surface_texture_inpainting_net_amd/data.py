"""Host-side sample container and collation for the STINet hot path.

Mirrors what the reference model reads from a PyG ``Batch`` of
``HierarchicalData`` (reference utils/data_utils.py:11-42,
datasets/scannetcolorgraph_dataloader.py:114-151): ``sample.x``,
``sample.edge_index``, ``sample.batch``, ``sample.num_vertices`` ([B, L] int32),
``sample["hierarchy_edge_index_{l}"]``, ``sample["hierarchy_trace_index_{l}"]``,
``sample["hierarchy_dil_{d}_edge_index_{l}"]`` - read through BOTH attribute
and item access (models/surfacetextureinpaintingnet.py:404-455).
"""
import re

import torch

_LEVEL_KEY = re.compile(r'^hierarchy_(?:dil_\d+_)?(?:edge|trace)_index_(\d+)$')
_DIL_KEY = re.compile(r'^hierarchy_dil_(\d+)_edge_index_(\d+)$')


def sample_keys(sample):
    """The key list of a sample object: `keys` is a METHOD on this build's HierarchicalBatch (and on PyG >= 2.4) and a
    PROPERTY on the PyG 2.0.x Data / Batch objects the reference feeds its model (utils/data_utils.py:11-42)."""
    k = sample.keys
    return list(k() if callable(k) else k)


class HierarchicalBatch:
    """Attribute + item bag of tensors; ``.to(device)`` moves every tensor."""

    def __init__(self, **tensors):
        object.__setattr__(self, '_store', {})
        object.__setattr__(self, '_plan_cache', None)
        object.__setattr__(self, '_nv_host', None)          # host copy of num_vertices (level sizes without a device sync)
        for k, v in tensors.items():
            self._store[k] = v

    # attribute / item access -------------------------------------------------
    def __getattr__(self, key):
        store = object.__getattribute__(self, '_store')
        if key in store:
            return store[key]
        raise AttributeError(key)

    def __setattr__(self, key, value):
        if key in ('_store', '_plan_cache', '_nv_host'):
            object.__setattr__(self, key, value)
        else:
            self._store[key] = value

    def __getitem__(self, key):
        return self._store[key]

    def __setitem__(self, key, value):
        self._store[key] = value

    def __contains__(self, key):
        return key in self._store

    def keys(self):
        return list(self._store.keys())

    @property
    def num_graphs(self):
        nv = self._store['num_vertices']
        return int(nv.shape[0]) if nv.dim() == 2 else 1

    def to(self, device, non_blocking=False):
        out = HierarchicalBatch()
        for k, v in self._store.items():
            out._store[k] = v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v
        nv = self._store.get('num_vertices')
        if torch.is_tensor(nv):                              # the plan needs the level sizes on the host: keep them there
            out._nv_host = self._nv_host if self._nv_host is not None else (nv if not nv.is_cuda else None)
        return out

    def pin_memory(self):
        out = HierarchicalBatch()
        for k, v in self._store.items():
            out._store[k] = v.pin_memory() if torch.is_tensor(v) else v
        return out


def _increment(key, num_vertices, fix_dilated_offsets):
    """Per-graph index offset for ``key`` (reference utils/data_utils.py:29-42).

    ``edge_index`` += N0; ``hierarchy_{edge,trace}_index_l`` += num_vertices[l];
    every other key containing 'index' falls through to PyG's default
    ``num_nodes`` = N0 - which is what the reference does to
    ``hierarchy_dil_*`` keys and is WRONG for B > 1.  With
    ``fix_dilated_offsets`` (default) those keys use num_vertices[level]."""
    if key == 'edge_index':
        return int(num_vertices[0])
    m = _DIL_KEY.match(key)
    if m is not None:
        return int(num_vertices[int(m.group(2))]) if fix_dilated_offsets else int(num_vertices[0])
    m = _LEVEL_KEY.match(key)
    if m is not None:
        return int(num_vertices[int(m.group(1))])
    if 'index' in key:
        return int(num_vertices[0])
    return 0


def collate(samples, fix_dilated_offsets=True):
    """List of single-graph HierarchicalBatch -> one batched HierarchicalBatch
    with the PyG collate semantics the reference relies on: tensors whose key
    contains 'index' are concatenated along the LAST dim, others along dim 0;
    ``num_vertices`` is stacked to [B, L]; ``batch`` [N0] int64 is added."""
    keys = sample_keys(samples[0])
    out = HierarchicalBatch()
    offsets = {k: 0 for k in keys}
    parts = {k: [] for k in keys}
    batch_vec = []
    for g, s in enumerate(samples):
        nv = s['num_vertices'].reshape(-1)
        for k in keys:
            v = s[k]
            if k == 'num_vertices':
                parts[k].append(nv)
            elif k == 'batch':
                continue
            elif torch.is_tensor(v):
                parts[k].append(v + offsets[k] if offsets[k] else v)
                offsets[k] += _increment(k, nv, fix_dilated_offsets)
            else:
                parts[k].append(v)
        batch_vec.append(torch.full((int(nv[0]),), g, dtype=torch.long))
    for k in keys:
        if k == 'batch':
            continue
        v0 = samples[0][k]
        if k == 'num_vertices':
            out[k] = torch.stack(parts[k], 0)
        elif torch.is_tensor(v0):
            out[k] = torch.cat(parts[k], dim=-1 if 'index' in k else 0)
        else:
            out[k] = parts[k]
    out['batch'] = torch.cat(batch_vec)
    return out
