"""Import the read-only reference (/root/reference) in the BUILD CONTAINER ONLY, to generate golden
vectors.  Nothing under tests/ imports this at test time; the reference never travels.

Packages the reference imports but that are not installed here are stubbed.
None of the stubs carries hot-path arithmetic EXCEPT:
  * torchvision.transforms.CenterCrop -- restated from torchvision's documented behaviour (A3);
  * shapely.geometry.Polygon          -- backed by oracle/oracle_ref.c so that nms_rotated's CONTROL
    FLOW (box_utils.py:693-738) can run; the GEOS arithmetic itself stays unpinned.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _CenterCrop:
    def __init__(self, size):
        self.size = size

    def __call__(self, x):
        th, tw = self.size
        H, W = x.shape[-2:]
        if th > H or tw > W:
            pl = (tw - W) // 2 if tw > W else 0
            pt = (th - H) // 2 if th > H else 0
            pr = (tw - W + 1) // 2 if tw > W else 0
            pb = (th - H + 1) // 2 if th > H else 0
            x = torch.nn.functional.pad(x, [pl, pr, pt, pb])
            H, W = x.shape[-2:]
        t = int(round((H - th) / 2.0))
        l = int(round((W - tw) / 2.0))
        return x[..., t:t + th, l:l + tw]


class _PassThrough:
    def __init__(self, *a, **k):
        pass

    def __call__(self, x):
        return x


class _Area:
    def __init__(self, a):
        self.area = a


class _Polygon:
    """Stand-in for shapely.geometry.Polygon: convex quads only, areas from the C oracle."""

    def __init__(self, pts):
        self.q = np.asarray(pts, np.float32).reshape(4, 2)

    def _parts(self, other):
        from oracle import cref
        a, b = self.q.astype(np.float64), other.q.astype(np.float64)

        def area(p):
            x, y = p[:, 0], p[:, 1]
            return abs(0.5 * np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))
        sa, sb = area(a), area(b)
        iou = float(cref.quad_iou(self.q[None], other.q[None])[0, 0])
        if np.isnan(iou):
            return 0.0, 0.0
        inter = iou * (sa + sb) / (1.0 + iou)
        return inter, sa + sb - inter

    def intersection(self, other):
        return _Area(self._parts(other)[0])

    def union(self, other):
        return _Area(self._parts(other)[1])


def install(stub_opencood_packages=True):
    """stub_opencood_packages=False: third-party stubs only; the reference's own package __init__ files then run (used by
    the overlay test, where heal_amd's modules stand in for the ones that need spconv / the CUDA extensions)."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present: golden vectors can only be generated in the build container")
    _stub("icecream", ic=lambda *a, **k: None)
    _stub("termcolor", colored=lambda s, *a, **k: s)
    _stub("cv2")
    _stub("open3d")
    sh = _stub("shapely")
    sh.geometry = _stub("shapely.geometry", Polygon=_Polygon, Point=object, MultiPoint=object)
    _stub("pyquaternion", Quaternion=object)
    _stub("pypcd").pypcd = _stub("pypcd.pypcd")          # .pcd file reader (disk I/O, not on the path)
    _stub("efficientnet_pytorch", EfficientNet=object)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms", CenterCrop=_CenterCrop, Compose=_PassThrough,
                          Normalize=_PassThrough, ToTensor=_PassThrough, ToPILImage=_PassThrough)
    tv.models = _stub("torchvision.models")
    tv.models.resnet = _stub("torchvision.models.resnet", resnet101=None, resnet18=None)
    _stub("timm")
    _stub("timm.models")
    _stub("timm.models.layers", DropPath=nn.Identity)
    _stub("spconv", SparseSequential=nn.Sequential, SubMConv3d=object, SparseConv3d=object,
          SparseInverseConv3d=object, SparseConvTensor=object)
    # package __init__ files that drag in unrelated detectors / visualisation: register the
    # packages with their real __path__ but without executing __init__.py
    for pkg in (("opencood.data_utils.post_processor", "opencood.data_utils.pre_processor", "opencood.visualization")
                if stub_opencood_packages else ("opencood.visualization",)):
        m = _stub(pkg)
        m.__path__ = [os.path.join(REF, *pkg.split("."))]
    _stub("opencood.visualization.vis_utils")
    _stub("opencood.visualization.debug_plot", plot_feature=lambda *a, **k: None)
    # Cython extension used by label generation only (not on the inference path): the reference's own .pyx compiled by
    # oracle/Makefile.ref when available (oracle/_ref/box_overlaps*.so), a stub otherwise
    ref_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle", "_ref")
    real = None
    if os.path.isdir(ref_dir) and any(f.startswith("box_overlaps") and f.endswith(".so") for f in os.listdir(ref_dir)):
        sys.path.insert(0, ref_dir)
        try:
            real = importlib.import_module("box_overlaps")
        finally:
            sys.path.remove(ref_dir)
    if real is not None:
        sys.modules["opencood.utils.box_overlaps"] = real
    else:
        _stub("opencood.utils.box_overlaps", bbox_overlaps=None)


def ref(module):
    install()
    return importlib.import_module(module)
