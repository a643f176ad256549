"""heal_amd.compat.overlay_reference(): the reference's own package, unmodified, with this repo's modules standing in
for the ones it implements (build container only; runs in a subprocess so that the overlaid `opencood` never leaks into
the test session).  Third-party packages the image lacks (cv2, open3d, shapely, h5py, tensorboardX, ...) are stubbed; no
spconv / CUDA-extension / Cython module of the reference is imported."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/opencood"), reason="reference tree not present")

SCRIPT = r'''
import importlib, sys
sys.path.insert(0, %r)
from tests.golden import ref_import as R
R.install(stub_opencood_packages=False)
for name in ("h5py", "tensorboardX"):
    R._stub(name, SummaryWriter=object)
from heal_amd import compat
names = compat.overlay_reference("/root/reference")
assert len(names) > 30 and "opencood.models.heter_pyramid_collab" in names

tu = importlib.import_module("opencood.tools.train_utils")
yu = importlib.import_module("opencood.hypes_yaml.yaml_utils")
assert tu.__file__.startswith("/root/reference/") and yu.__file__.startswith("/root/reference/")   # the reference's files
for rel, cls in (("LiDAROnly/lidar_pyramid.yaml", "HeterPyramidCollab"), ("LiDAROnly/lidar_v2xvit.yaml", "HeterModelBaseline"),
                 ("MoreModality/HEAL/final_infer/m1m2m3m4.yaml", "HeterPyramidCollab")):
    path = "/root/reference/opencood/hypes_yaml/opv2v/" + rel
    import os
    if not os.path.exists(path):
        continue
    hy = yu.load_yaml(path)
    model = tu.create_model(hy)                     # the reference's discovery finds this repo's class
    assert type(model).__name__ == cls and type(model).__module__.startswith("heal_amd.opencood.models."), type(model)
crit = tu.create_loss(hy)
assert type(crit).__module__.startswith("heal_amd.opencood.loss.")

post = importlib.import_module("opencood.data_utils.post_processor").build_postprocessor(hy["postprocess"], train=False)
mro = [c.__module__ for c in type(post).__mro__]
assert mro[1] == "heal_amd.opencood.data_utils.post_processor.voxel_postprocessor"
assert mro[2] == "opencood.data_utils.post_processor.base_postprocessor"
assert post.generate_anchor_box().shape[-2:] == (2, 7) and hasattr(post, "generate_object_center")
pre = importlib.import_module("opencood.data_utils.pre_processor").build_preprocessor(hy["preprocess"], train=False)
assert [c.__module__ for c in type(pre).__mro__][1] == "heal_amd.opencood.data_utils.pre_processor.sp_voxel_preprocessor"
iou = importlib.import_module("opencood.pcdet_utils.iou3d_nms.iou3d_nms_utils")
assert iou.__name__ == "heal_amd.opencood.pcdet_utils.iou3d_nms.iou3d_nms_utils"

assert hasattr(importlib.import_module("opencood.data_utils.datasets"), "build_dataset")   # the reference's dataset stack
inference = importlib.import_module("opencood.tools.inference")                            # the reference's driver, unmodified
train = importlib.import_module("opencood.tools.train")
assert inference.__file__ == "/root/reference/opencood/tools/inference.py" and hasattr(inference, "main")
assert train.__file__ == "/root/reference/opencood/tools/train.py"
bad = [m for m in sys.modules if m.startswith(("spconv.", "opencood.utils.iou3d", "opencood.pcdet_utils.iou3d_nms.iou3d_nms_cuda"))]
assert not bad, bad
print("OVERLAY-OK", len(names))
'''


def test_reference_drivers_import_and_discover_our_modules_under_the_overlay():
    res = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], capture_output=True, text=True, timeout=600,
                         cwd=ROOT, env={**os.environ, "PYTHONPATH": ROOT})
    assert res.returncode == 0 and "OVERLAY-OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


def test_overlay_refuses_to_run_after_opencood_was_imported():
    code = ("import sys, types; sys.path.insert(0, %r); sys.modules['opencood'] = types.ModuleType('opencood')\n"
            "from heal_amd import compat\n"
            "try:\n    compat.overlay_reference()\nexcept RuntimeError as e:\n    print('REFUSED')\n" % ROOT)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert "REFUSED" in res.stdout, res.stdout + res.stderr
