"""Import-only shim: the reference imports SparseTensor/matmul
(models/modules/sage_conv_filter.py:14) but never reaches them with a dense
``edge_index`` tensor."""


class SparseTensor:  # pragma: no cover - never instantiated on the path
    def __init__(self, *a, **k):
        raise NotImplementedError('torch_sparse is not on the STINet hot path')


def matmul(*a, **k):  # pragma: no cover
    raise NotImplementedError('torch_sparse is not on the STINet hot path')
