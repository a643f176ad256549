"""Expose heal_amd.opencood.* under the reference's package name `opencood.*`, so that drivers written
against HEAL (`tools/inference.py`, `tools/train.py` style code, `create_model(hypes)`) import the
MI355X implementation unchanged."""
import importlib
import pkgutil
import sys


def install_as_opencood(force=False):
    """Register every heal_amd.opencood module as opencood.<same path>.  Refuses to shadow a real
    `opencood` installation unless force=True."""
    if "opencood" in sys.modules and not force:
        mod = sys.modules["opencood"]
        if not getattr(mod, "__heal_amd__", False):
            raise RuntimeError("a different `opencood` package is already imported; pass force=True to shadow it")
        return mod
    import heal_amd.opencood as root
    root.__heal_amd__ = True
    sys.modules["opencood"] = root
    for info in pkgutil.walk_packages(root.__path__, prefix="heal_amd.opencood."):
        try:
            m = importlib.import_module(info.name)
        except Exception:  # a module that needs the GPU library at import time is registered lazily
            continue
        sys.modules["opencood." + info.name[len("heal_amd.opencood."):]] = m
    return root


# ---- overlay on a HEAL checkout ------------------------------------------------------------------------------
# install_as_opencood() answers to the name `opencood` alone: enough for drivers that only touch the mirrored modules
# (examples/inference_synthetic.py).  The reference's unmodified tools/inference.py / tools/train.py also import its
# datasets, evaluation and visualisation code.  overlay_reference() keeps the reference's package and replaces, under
# the reference's module names, only what this repo implements: the model files, the losses, the pcdet IoU/NMS API and
# the voxel pre / post processors.  Nothing of spconv, the CUDA extensions or the Cython module is imported any more.
_OVERLAY_LEAVES = ("pcdet_utils.iou3d_nms.iou3d_nms_utils",)
_OVERLAY_TREES = ("models", "loss")


def _lazy_subclass_module(name, ours_module, class_name, base_module, base_class):
    """A module object for `name` whose `class_name` is our class with the reference's base class appended to its
    bases (the datasets call base-class helpers such as generate_object_center / project_points_to_bev_map).  Built on
    first attribute access (PEP 562): the reference's package __init__ imports this module while it is itself being
    imported, and only then is its base module importable."""
    import types
    mod = types.ModuleType(name)
    mod.__heal_amd__ = True
    cache = {}

    def __getattr__(attr):
        if attr == class_name:
            if attr not in cache:
                ours = getattr(ours_module, class_name)
                base = getattr(importlib.import_module(base_module), base_class)
                cache[attr] = type(class_name, (ours, base), {"__module__": name})
            return cache[attr]
        return getattr(ours_module, attr)
    mod.__getattr__ = __getattr__
    return mod


def overlay_reference(reference_root=None):
    """Make `import opencood...` resolve to the HEAL checkout at `reference_root` (or wherever `opencood` is already
    importable from) with this repo's modules in place of the ones it implements.  Call before importing `opencood`.
    Returns the list of overlaid module names."""
    if "opencood" in sys.modules:
        raise RuntimeError("overlay_reference() must run before `opencood` is imported")
    if reference_root is not None and reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    import heal_amd.opencood as root
    names = []
    for info in pkgutil.walk_packages(root.__path__, prefix="heal_amd.opencood."):
        rel = info.name[len("heal_amd.opencood."):]
        if info.ispkg:
            continue
        if rel in _OVERLAY_LEAVES or rel.split(".")[0] in _OVERLAY_TREES:
            sys.modules["opencood." + rel] = importlib.import_module(info.name)
            names.append("opencood." + rel)
    from heal_amd.opencood.data_utils.post_processor import voxel_postprocessor as post
    from heal_amd.opencood.data_utils.pre_processor import sp_voxel_preprocessor as pre
    for name, ours, cls, base_mod, base_cls in (
            ("opencood.data_utils.post_processor.voxel_postprocessor", post, "VoxelPostprocessor",
             "opencood.data_utils.post_processor.base_postprocessor", "BasePostprocessor"),
            ("opencood.data_utils.pre_processor.sp_voxel_preprocessor", pre, "SpVoxelPreprocessor",
             "opencood.data_utils.pre_processor.base_preprocessor", "BasePreprocessor")):
        sys.modules[name] = _lazy_subclass_module(name, ours, cls, base_mod, base_cls)
        names.append(name)
    return names
