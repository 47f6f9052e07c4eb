"""Import-name shim: `torch_scatter` -> oracle.scatter_ops (see that file)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.scatter_ops import scatter, scatter_sum, scatter_mean, scatter_max  # noqa: F401,E402
scatter_add = scatter_sum
