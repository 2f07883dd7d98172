"""Import shim for the upstream reference (lives at /root/reference, read-only).

Only used in the build container to GENERATE golden vectors and to validate the
oracle restatement.  Nothing under tests/ -m gpu, bench.py or smoke() imports this:
/root/reference does not exist on the GPU box.

Recipe follows SURVEY.md section 8(c): stub the un-vendored third-party modules the
reference imports at module scope but never uses on the XLSTM_HVED path, neutralise the
hard-coded `.cuda()` (RA_HVED.py:520) and import RA_HVED directly.
"""
import contextlib
import io
import sys
import types

REF_ROOT = "/root/reference"


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__getattr__ = lambda attr: _Anything  # any "from x import y" resolves
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent:
        setattr(_stub(parent), child, m)
    return m


def load_reference():
    """Returns the imported reference modules as a namespace: .RA_HVED, .utils, .metrics, .loss"""
    import torch

    for name in [
        "dynamic_network_architectures",
        "dynamic_network_architectures.building_blocks",
        "dynamic_network_architectures.building_blocks.helper",
        "dynamic_network_architectures.building_blocks.residual",
        "dynamic_network_architectures.initialization",
        "dynamic_network_architectures.initialization.weight_init",
        "nnunetv2",
        "nnunetv2.utilities",
        "nnunetv2.utilities.plans_handling",
        "nnunetv2.utilities.plans_handling.plans_handler",
        "nnunetv2.utilities.network_initialization",
        "h5py",
        "skimage",
        "skimage.segmentation",
        "torchsummary",
        "nibabel",
        "SimpleITK",
        "medpy",
        "medpy.metric",
        "torchmetrics",
    ]:
        _stub(name)
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self  # RA_HVED.py:520
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    ns = types.SimpleNamespace()
    with contextlib.redirect_stdout(io.StringIO()):
        import RA_HVED
        import utils as ref_utils
        import metrics as ref_metrics
    ns.RA_HVED = RA_HVED
    ns.utils = ref_utils
    ns.metrics = ref_metrics
    return ns


TRAIN_KWARGS = dict(multi_stream=4, fusion_level=4, shared_recon=True, recon_skip=True,
                    MVAE_reduction=True, final_sigmoid=True, f_maps=4, layer_order='ilc')


def build_reference_model(ns, seed=1, cls="XLSTM_HVED", dtype=None, **overrides):
    """train.py:75,142-145: seed -> ctor -> model.apply(init_weights)."""
    import torch

    torch.manual_seed(seed)
    kw = dict(TRAIN_KWARGS)
    kw.update(overrides)
    with contextlib.redirect_stdout(io.StringIO()):
        model = getattr(ns.RA_HVED, cls)(1, 3, **kw)
        model.apply(ns.utils.init_weights)
    if dtype is not None:
        model = model.to(dtype)
    return model
