"""Import the reference's own hot-path classes in THIS container only.

TEST INFRASTRUCTURE (oracle pinning).  /root/reference is read-only, is never
copied, and does not exist on the GPU box: this module is used only by
oracle/make_golden.py and by the `reference`-marked CPU tests, which skip when
/root/reference is absent.

Recipe: put oracle/pyg_shim first on sys.path (build-owned
restatement of the absent third-party torch_geometric/torch_scatter/torch_sparse
symbols), register `models`, `models.modules`, ... as bare namespace packages so
the reference's auto-importing __init__ files (torchvision etc.) are skipped, then
import the reference modules by name.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = '/root/reference'
_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'models'))


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def setup():
    if not available():
        raise RuntimeError('reference tree not present (expected only in the build container)')
    sys.dont_write_bytecode = True  # the reference tree is read-only
    for p in (_REPO, os.path.join(_HERE, 'pyg_shim')):
        if p not in sys.path:
            sys.path.insert(0, p)
    if REFERENCE_ROOT not in sys.path:
        sys.path.append(REFERENCE_ROOT)
    for name in ('models', 'models.modules', 'datasets', 'trainers', 'preprocessing'):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REFERENCE_ROOT, name.replace('.', '/'))]
            sys.modules[name] = m


def load_model_module():
    """-> the reference's models.surfacetextureinpaintingnet module."""
    setup()
    for f in ('edge_conv_filter', 'edge_conv_translation_invariance', 'sage_conv_filter',
              'fastinstancenorm', 'singlebatchgroupnorm'):
        importlib.import_module('models.modules.' + f)
    return importlib.import_module('models.surfacetextureinpaintingnet')


def load_module(name):
    """Any other reference module (utils.data_utils, utils.metrics.graph_metrics, ...);
    the reference's `utils/__init__` pulls open3d etc., which are stubbed."""
    setup()
    _stub('open3d')
    _stub('termcolor', colored=lambda s, *a, **k: s)
    _stub('git', Repo=object)
    return importlib.import_module(name)


def load_singleconvmeshnet_module():
    """-> the reference's models.singleconvmeshnet module (its BaseModel comes from the reference's `base` package)."""
    load_model_module()
    if 'base' not in sys.modules:
        m = types.ModuleType('base')
        m.__path__ = [os.path.join(REFERENCE_ROOT, 'base')]
        sys.modules['base'] = m
    return importlib.import_module('models.singleconvmeshnet')


def load_preprocessing():
    """-> (preprocessing.graph_dilation, preprocessing.graph_level_generation) with the mesh-IO libraries they import
    but the graph functions do not need (open3d, plyfile) stubbed and tqdm silenced."""
    setup()
    _stub('open3d')
    _stub('plyfile', PlyData=object)
    _stub('termcolor', colored=lambda s, *a, **k: s)
    _stub('git', Repo=object)
    gd = importlib.import_module('preprocessing.graph_dilation')
    gd.tqdm = lambda it: it
    return gd, importlib.import_module('preprocessing.graph_level_generation')


def load_imagegraph_dataset_class():
    """-> datasets.imagegraph_dataloader.ImageGraphTextureDataSet under stubs for the
    image libraries it imports but does not need for the index maps."""
    setup()
    class _Easy(dict):
        __getattr__ = dict.get
    _stub('cv2')
    _stub('open3d')
    sk = _stub('skimage', img_as_float32=lambda x: x)
    sk.io = _stub('skimage.io')
    tv = _stub('torchvision')
    tv.transforms = _stub('torchvision.transforms', Compose=lambda ts: ts)
    _stub('easydict', EasyDict=_Easy)
    _stub('transform')
    return importlib.import_module('datasets.imagegraph_dataloader')


def load_trainer3d_module():
    setup()
    _stub('open3d')
    _stub('termcolor', colored=lambda s, *a, **k: s)
    _stub('git', Repo=object)
    return importlib.import_module('trainers.inpainting3d_trainer')


def load_scannet_color_dataset_module():
    """-> datasets.scannetcolorgraph_dataloader (the reference's 3-D inpainting dataset) under stubs for the libraries it
    imports but __getitem__ does not use (open3d, torchvision, easydict, PyG loaders); `transform` is the reference's own
    package (CoordsNormalization is applied to the sample), `utils.data_utils.HierarchicalData` its own sample class."""
    setup()
    class _Easy(dict):
        __getattr__ = dict.get
    _stub('open3d')
    _stub('termcolor', colored=lambda s, *a, **k: s)
    _stub('git', Repo=object)
    tv = _stub('torchvision')
    tv.transforms = _stub('torchvision.transforms', Compose=lambda ts: ts)
    _stub('easydict', EasyDict=_Easy)
    import torch_geometric.data as tgd
    if not hasattr(tgd, 'DataListLoader'):
        tgd.DataListLoader = object
    # (load_imagegraph_dataset_class leaves an EMPTY stub `transform` behind - the image dataset never calls it; this
    # dataset applies transform.CoordsNormalization, so a stub without it is replaced by the reference's own package)
    if not hasattr(sys.modules.get('transform'), 'CoordsNormalization'):
        m = types.ModuleType('transform')
        m.__path__ = [os.path.join(REFERENCE_ROOT, 'transform')]
        sys.modules['transform'] = m
        for f in ('coords_normalization',):
            sub = importlib.import_module('transform.' + f)
            for k, v in sub.__dict__.items():
                if not k.startswith('_'):
                    setattr(m, k, v)
    return importlib.import_module('datasets.scannetcolorgraph_dataloader')
