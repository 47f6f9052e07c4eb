"""Minimal pure-torch restatement of the torch_geometric (PyG 2.0.x) symbols the
STINet reference imports.  TEST INFRASTRUCTURE ONLY (oracle); see
oracle/README.md.  PyG is a third-party dependency that is absent from
/root/reference and unpinned there (reference README.md:38-41).  Semantics are
restated from PyG's published behaviour, not copied:

* MessagePassing(flow='source_to_target'): ``x_j = x[edge_index[0]]``,
  ``x_i = x[edge_index[1]]``, aggregation at ``edge_index[1]`` with
  ``dim_size = N``.
* EdgeConv.message = nn(cat([x_i, x_j - x_i], -1)).
* SAGEConv = lin_l(mean_j x_j) + lin_r(x); lin_l has the bias, lin_r none.
* BatchNorm(in_channels) wraps torch.nn.BatchNorm1d as ``.module``.
* utils.degree = bincount; utils.coalesce = sort by (row, col) + dedup.
* Data collate rules: ``__cat_dim__`` is -1 for keys containing 'index' else 0,
  ``__inc__`` is ``num_nodes`` for keys containing 'index' else 0; a ``None``
  cat-dim means "stack".
"""
from . import typing, utils, nn, data  # noqa: F401
__version__ = '2.0.4-shim'
