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


DATASET_SCRIPT = r'''
import importlib, json, os, sys, tempfile
import numpy as np
sys.path.insert(0, %r)
os.environ["HEAL_DEFER_VOXELIZE"] = "1"      # DataLoader-worker mode: neither K1 nor the label assignment may touch the GPU
os.environ["HEAL_INFERENCE_ONLY"] = "1"      # what examples/run_reference_tool.py sets for tools/inference*.py (labels never read)
from tests.golden import ref_import as R
R.install(stub_opencood_packages=False)
for name in ("h5py", "tensorboardX"):
    R._stub(name, SummaryWriter=object)
from heal_amd import compat, synth
compat.overlay_reference("/root/reference")
import torch, yaml

# a two-vehicle OPV2V scenario on disk: <root>/test/<scenario>/<cav>/<timestamp>.yaml + .pcd
root = tempfile.mkdtemp(prefix="opv2v_")
scen = os.path.join(root, "test", "2021_08_18_19_48_05")
poses = {"641": [10.0, 5.0, 1.9, 0.0, 12.0, 0.0], "650": [22.0, -3.0, 1.9, 0.0, -170.0, 0.0]}
spots = {101: (18, 6, 10), 102: (30, -8, -160), 103: (5, 12, 95), 104: (150, 150, 0)}     # 104 is out of range
vehicles = {k: {"angle": [0.0, float(a), 0.0], "center": [0.0, 0.0, 0.8], "extent": [2.3, 1.0, 0.8],
                "location": [float(x), float(y), 0.05], "speed": 3.0} for k, (x, y, a) in spots.items()}
clouds = {}
for i, (cav, pose) in enumerate(poses.items()):
    d = os.path.join(scen, cav)
    os.makedirs(d)
    meta = {"lidar_pose": pose, "true_ego_pos": pose, "predicted_ego_pos": pose, "ego_speed": 5.0, "vehicles": vehicles}
    for c in range(4):
        meta["camera%%d" %% c] = {"cords": pose, "extrinsic": np.eye(4).tolist(), "intrinsic": np.eye(3).tolist()}
    yaml.safe_dump(meta, open(os.path.join(d, "000068.yaml"), "w"))
    open(os.path.join(d, "000068.pcd"), "w").write("placeholder: the .pcd reader (open3d) is replaced below")
    clouds[os.path.join(d, "000068.pcd")] = synth.lidar_frame(40 + i)
assign = os.path.join(root, "assign.json")
json.dump({"2021_08_18_19_48_05": {"641": "m1", "650": "m1"}}, open(assign, "w"))
importlib.import_module("opencood.utils.pcd_utils").pcd_to_np = lambda f: clouds[f]

yu = importlib.import_module("opencood.hypes_yaml.yaml_utils")            # everything below is the REFERENCE's code ...
hy = yu.load_yaml("/root/reference/opencood/hypes_yaml/opv2v/LiDAROnly/lidar_pyramid.yaml")
hy["validate_dir"] = hy["test_dir"] = os.path.join(root, "test")
hy["heter"]["assignment_path"] = assign
dataset = importlib.import_module("opencood.data_utils.datasets").build_dataset(hy, visualize=False, train=False)
assert type(dataset.pre_processor_m1).__mro__[1].__module__.startswith("heal_amd.")   # ... around this repo's processors
batch = dataset.collate_batch_test([dataset[0]])
tu = importlib.import_module("opencood.tools.train_utils")
batch = tu.to_device(batch, torch.device("cpu"))
ego = batch["ego"]

# the input contract of the model mirrors (SURVEY 8b)
assert ego["agent_modality_list"] == ["m1", "m1"] and ego["record_len"].tolist() == [2]
assert tuple(ego["pairwise_t_matrix"].shape) == (1, 5, 5, 4, 4) and ego["pairwise_t_matrix"].dtype == torch.float64
assert tuple(ego["anchor_box"].shape) == (256, 256, 2, 7)
inp = ego["inputs_m1"]
assert set(inp) == {"points", "max_points_per_voxel", "max_voxels"} and inp["max_voxels"] == 70000
assert len(inp["points"]) == 2 and all(p.dtype == torch.float32 and p.shape[1] == 4 and p.shape[0] > 1000 for p in inp["points"])
for p in inp["points"]:      # the dataset's own shuffle_points + mask_ego_points ran before our preprocess (:144-146)
    assert not bool(((p[:, 0] >= -1.95) & (p[:, 0] <= 2.95) & (p[:, 1] >= -1.1) & (p[:, 1] <= 1.1)).any())
assert {"pos_equal_one", "neg_equal_one", "targets"} <= set(ego["label_dict"])

# this repo's model (built by the reference's create_model) accepts exactly this dictionary
from tests.test_glue_cpu import _stub_model
model = _stub_model(tu.create_model(hy), 64, lidar_hw=(256, 256))
with torch.no_grad():
    out = model(ego)
assert out["pyramid"] == "collab" and out["cls_preds"].shape[1] == 2 and out["occ_single_list"][0].shape[0] == 2
gt = dataset.post_processor.generate_gt_bbx(batch)      # evaluation side: 3 of the 4 vehicles are in range, seen twice
assert tuple(gt.shape) == (3, 8, 3), gt.shape
print("DATASET-OK")
'''


def test_reference_dataset_on_a_synthetic_opv2v_scenario_feeds_the_model_contract():
    """The reference's OPV2VBaseDataset + IntermediateheterFusionDataset, unmodified, read a synthetic two-vehicle OPV2V
    scenario from disk under the overlay with the processors in deferred (DataLoader-worker) mode, and what
    `collate_batch_test` + `to_device` hand over is what this repo's model mirror reads."""
    res = subprocess.run([sys.executable, "-c", DATASET_SCRIPT % ROOT], capture_output=True, text=True, timeout=600,
                         cwd=ROOT, env={**os.environ, "PYTHONPATH": ROOT})
    assert res.returncode == 0 and "DATASET-OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


HETERO_SCRIPT = r'''
import importlib, json, os, sys, tempfile, warnings
import numpy as np
warnings.simplefilter("ignore")
sys.path.insert(0, %r)
os.environ["HEAL_DEFER_VOXELIZE"] = "1"
os.environ["HEAL_INFERENCE_ONLY"] = "1"
from tests.golden import ref_import as R
R.install(stub_opencood_packages=False)
for name in ("h5py", "tensorboardX"):
    R._stub(name, SummaryWriter=object)
import torch
from PIL import Image

# the image has no torchvision / opencv: the three transforms and two cv2 calls the camera branch of the dataset uses are
# restated from their documented behaviour, for this test only
class ToTensor:
    def __call__(self, pic):
        a = np.array(pic)
        a = a[:, :, None] if a.ndim == 2 else a
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        return t.float().div(255) if t.dtype == torch.uint8 else t
class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)
    def __call__(self, x):
        return (x - self.mean) / self.std
class Compose:
    def __init__(self, ts):
        self.ts = ts
    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x
tvt = sys.modules["torchvision.transforms"]
tvt.ToTensor, tvt.Normalize, tvt.Compose = ToTensor, Normalize, Compose
cv2 = sys.modules["cv2"]
cv2.imread = lambda f, *a: np.array(Image.open(f))
cv2.cvtColor = lambda img, code: img if img.ndim == 2 else img[..., 0]
cv2.COLOR_BGR2GRAY = 6

from heal_amd import compat, synth
compat.overlay_reference("/root/reference")
import yaml

root = tempfile.mkdtemp(prefix="opv2v_h_")
scen = os.path.join(root, "test", "2021_08_18_19_48_05")
poses = {"641": [10.0, 5.0, 1.9, 0.0, 12.0, 0.0], "650": [22.0, -3.0, 1.9, 0.0, -170.0, 0.0],
         "659": [15.0, 14.0, 1.9, 0.0, 80.0, 0.0], "668": [2.0, -9.0, 1.9, 0.0, 40.0, 0.0]}
assignment = {"641": "m1", "650": "m2", "659": "m3", "668": "m4"}      # PointPillars, LSS-EfficientNet, SECOND, LSS-ResNet
spots = {101: (18, 6, 10), 102: (30, -8, -160), 103: (5, 12, 95), 104: (150, 150, 0)}
vehicles = {k: {"angle": [0.0, float(a), 0.0], "center": [0.0, 0.0, 0.8], "extent": [2.3, 1.0, 0.8],
                "location": [float(x), float(y), 0.05], "speed": 3.0} for k, (x, y, a) in spots.items()}
rig = synth.camera_rig(0, 4, 600, 800)
clouds = {}
for i, (cav, pose) in enumerate(poses.items()):
    d = os.path.join(scen, cav)
    os.makedirs(d)
    meta = {"lidar_pose": pose, "true_ego_pos": pose, "predicted_ego_pos": pose, "ego_speed": 5.0, "vehicles": vehicles}
    for c in range(4):
        ext = np.eye(4)
        ext[:3, :3], ext[:3, 3] = rig["rots"][c], rig["trans"][c]
        meta["camera%%d" %% c] = {"cords": pose, "extrinsic": ext.tolist(), "intrinsic": rig["intrins"][c].tolist()}
        g = np.random.default_rng(10 * i + c)
        Image.fromarray(g.integers(0, 255, (600, 800, 3), dtype=np.uint8)).save(os.path.join(d, "000068_camera%%d.png" %% c))
        Image.fromarray(g.integers(0, 255, (600, 800), dtype=np.uint8)).save(os.path.join(d, "000068_depth%%d.png" %% c))
    Image.fromarray(np.zeros((256, 256), np.uint8)).save(os.path.join(d, "000068_bev_visibility.png"))
    yaml.safe_dump(meta, open(os.path.join(d, "000068.yaml"), "w"))
    open(os.path.join(d, "000068.pcd"), "w").write("placeholder")
    clouds[os.path.join(d, "000068.pcd")] = synth.lidar_frame(40 + i)
    clouds[os.path.join(d, "000068_32.pcd")] = synth.lidar_frame(40 + i)[::2]     # lidar_channels_dict: m3 reads the 32-line file
assign = os.path.join(root, "assign.json")
json.dump({"2021_08_18_19_48_05": assignment}, open(assign, "w"))
importlib.import_module("opencood.utils.pcd_utils").pcd_to_np = lambda f: clouds[f]

yu = importlib.import_module("opencood.hypes_yaml.yaml_utils")
tu = importlib.import_module("opencood.tools.train_utils")
hy = yu.load_yaml("/root/reference/opencood/hypes_yaml/opv2v/MoreModality/HEAL/final_infer/m1m2m3m4.yaml")
hy["validate_dir"] = hy["test_dir"] = os.path.join(root, "test")
hy["heter"]["assignment_path"] = assign
dataset = importlib.import_module("opencood.data_utils.datasets").build_dataset(hy, visualize=False, train=False)
ego = tu.to_device(dataset.collate_batch_test([dataset[0]]), torch.device("cpu"))["ego"]
assert ego["agent_modality_list"] == ["m1", "m2", "m3", "m4"] and ego["record_len"].tolist() == [4]
for m, P in (("m1", 32), ("m3", 5)):        # PointPillars / SECOND: raw clouds + the caps of that modality's preprocess block
    inp = ego["inputs_" + m]
    assert len(inp["points"]) == 1 and inp["points"][0].dtype == torch.float32 and inp["max_points_per_voxel"] == P, (m, inp)
for m, (H, W) in (("m2", (384, 512)), ("m4", (336, 448))):
    inp = ego["inputs_" + m]
    assert tuple(inp["imgs"].shape) == (1, 4, 4, H, W)                    # RGB + depth channel; CamEncode reads [:, :3]
    for k, shp in (("rots", (1, 4, 3, 3)), ("trans", (1, 4, 3)), ("intrins", (1, 4, 3, 3)), ("post_rots", (1, 4, 3, 3)),
                   ("post_trans", (1, 4, 3))):
        assert tuple(inp[k].shape) == shp and inp[k].dtype == torch.float32, (m, k)

from tests.test_glue_cpu import _stub_model
model = tu.create_model(hy)
assert type(model).__module__ == "heal_amd.opencood.models.heter_pyramid_collab"
model = _stub_model(model, 64, camera_hw=(128, 128), lidar_hw=(256, 256))
with torch.no_grad():
    out = model(ego)
assert out["pyramid"] == "collab" and out["occ_single_list"][0].shape[0] == 4 and out["reg_preds"].shape[1] == 14
print("HETERO-OK")
'''


def test_reference_dataset_four_modalities_feed_the_model_contract():
    """Same as above with the reference's 4-modality HEAL YAML (final_infer/m1m2m3m4.yaml): one PointPillars, one
    Lift-Splat/EfficientNet, one SECOND and one Lift-Splat/ResNet agent, camera PNGs and depth maps on disk."""
    res = subprocess.run([sys.executable, "-c", HETERO_SCRIPT % ROOT], capture_output=True, text=True, timeout=600,
                         cwd=ROOT, env={**os.environ, "PYTHONPATH": ROOT})
    assert res.returncode == 0 and "HETERO-OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
