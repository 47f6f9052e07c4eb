import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # TrainStep(graph=True) tests: the ROCm 7.2 graph-replay workaround must be in the environment BEFORE the first HIP call
    # of the process (torch.cuda.is_available() below is one) - an opt-in of the package, see its __init__
    import surface_texture_inpainting_net_amd as pkg
    pkg.enable_graph_replay()
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (build container only); skipped elsewhere')


def pytest_collection_modifyitems(config, items):
    import torch
    has_gpu = torch.cuda.is_available()
    has_ref = os.path.isdir('/root/reference/models')
    for item in items:
        if 'gpu' in item.keywords and not has_gpu:
            item.add_marker(pytest.mark.skip(reason='no GPU in this container'))
        if 'reference' in item.keywords and not has_ref:
            item.add_marker(pytest.mark.skip(reason='/root/reference not present'))
