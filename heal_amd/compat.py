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
