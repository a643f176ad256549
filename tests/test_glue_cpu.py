"""Python glue of the model mirrors without a GPU: the per-modality encoders, backbones and fusion operators (everything
that launches a HIP kernel) are replaced by shape-correct stand-ins AFTER construction, and `forward(data_dict)` runs on
CPU tensors.  What this exercises is the host-side control flow this repo owns -- modality bookkeeping, camera crop,
depth-item plumbing, scene-order re-assembly, head wiring, output dictionary keys -- so that a slip there is caught by
the CPU suite and not first on the GPU box.  It says nothing about numerics (the GPU parity tests do)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from heal_amd import configs


class _Encoder(nn.Module):
    """[n, C, H, W] random features for the agents of one modality; remembers depth items like the camera encoders."""

    def __init__(self, channels, hw, camera):
        super().__init__()
        self.channels, self.hw, self.camera, self.depth_items = channels, hw, camera, None

    def forward(self, data_dict, m):
        inp = data_dict[f"inputs_{m}"]
        n = int(inp["imgs"].shape[0]) if self.camera else len(inp["points"])
        if self.camera:
            self.depth_items = (torch.zeros(n * 4, 48, 6, 8), torch.zeros(n * 4, 6, 8, dtype=torch.long))
        g = torch.Generator().manual_seed(n)
        return torch.randn((n, self.channels) + self.hw, generator=g)


class _Backbone(nn.Module):
    def forward(self, batch_dict):
        return {"spatial_features_2d": batch_dict["spatial_features"]}


class _Pyramid(nn.Module):
    def forward_single(self, x):
        return x.repeat(1, 4, 1, 1)[:, :256], [x[:, :1], x[:, :1, ::2, ::2], x[:, :1, ::4, ::4]]

    def forward_collab(self, x, record_len, affine_matrix, agent_modality_list=None, cam_crop_info=None, grid_f64=True, cam_boxes=None):
        assert x.shape[0] == sum(record_len) and affine_matrix.shape[-2:] == (2, 3)
        return x[:1].repeat(1, 4, 1, 1)[:, :256], [x[:, :1], x[:, :1, ::2, ::2], x[:, :1, ::4, ::4]]


class _Layers(nn.Module):
    def get_layer_i_feature(self, x, layer_i):
        return x

    def decode_multiscale_feature(self, xs):
        assert len(xs) == 3                      # layer 0 output + layers 1, 2
        return torch.cat(xs, 1)[:, :256]         # the late model's heads take 256 channels


class _Fusion(nn.Module):
    def forward(self, x, record_len, affine_matrix):
        return x[:1]


def _stub_model(model, channels, camera_hw=(128, 128), lidar_hw=(128, 128)):
    for m in model.modality_name_list:
        camera = model.sensor_type_dict[m] == "camera"
        setattr(model, f"encoder_{m}", _Encoder(channels, camera_hw if camera else lidar_hw, camera))
        setattr(model, f"backbone_{m}", _Backbone())
        for name in (f"aligner_{m}", f"shrinker_{m}", f"shrink_conv_{m}"):
            if hasattr(model, name):
                setattr(model, name, nn.Identity())
        if hasattr(model, f"layers_{m}"):
            setattr(model, f"layers_{m}", _Layers())
    for name, fake in (("pyramid_backbone", _Pyramid()), ("fusion_net", _Fusion()), ("shrink_conv", nn.Identity())):
        if hasattr(model, name):
            setattr(model, name, fake)
    return model.eval()


def _scene_inputs(mods):
    d = {"agent_modality_list": list(mods), "record_len": torch.tensor([len(mods)]),
         "pairwise_t_matrix": torch.from_numpy(np.tile(np.eye(4), (1, 5, 5, 1, 1)))}
    for m in sorted(set(mods)):
        n = mods.count(m)
        d[f"inputs_{m}"] = {"imgs": torch.zeros(n, 4, 3, 8, 8)} if m in ("m2", "m4") else {"points": [None] * n}
    return d


@torch.no_grad()
def test_heter_pyramid_collab_glue():
    from heal_amd.opencood.tools.train_utils import create_model
    mods = ["m1", "m2", "m1", "m4", "m1"]
    # LiDAR maps: 204.8 m at 0.8 m/px = 256; camera grid: +-51.2 m = 128 px, zero-padded to the LiDAR range (ratio 2)
    model = _stub_model(create_model(configs.heal_heter(("m1", "m2", "m4"))), 64, camera_hw=(128, 128), lidar_hw=(256, 256))
    out = model(_scene_inputs(mods))
    assert out["pyramid"] == "collab" and tuple(out["cls_preds"].shape) == (1, 2, 256, 256)
    assert tuple(out["reg_preds"].shape) == (1, 14, 256, 256) and tuple(out["dir_preds"].shape) == (1, 4, 256, 256)
    assert len(out["occ_single_list"]) == 3 and out["occ_single_list"][0].shape[0] == 5
    assert set(k for k in out if k.startswith("depth_items_")) == {"depth_items_m2", "depth_items_m4"}
    # camera maps are cropped to the LiDAR range before they meet the LiDAR maps (same H, W after the crop)
    feats = model.encode_modality(_scene_inputs(mods), "m2")
    ratio = model.crop_ratio_H_m2
    assert feats.shape[-1] == int(128 * ratio) and feats.shape[0] == 1


@torch.no_grad()
def test_single_late_and_baseline_glue():
    from heal_amd.opencood.tools.train_utils import create_model
    single = _stub_model(create_model(configs.m1_single_pyramid()), 64)
    out = single({"inputs_m1": {"points": [None]}})
    assert out["pyramid"] == "single" and tuple(out["cls_preds"].shape)[:2] == (1, 2) and len(out["occ_single_list"]) == 3
    late = _stub_model(create_model(configs.m1_late()), 128)
    out = late({"inputs_m1": {"points": [None]}})
    assert set(out) == {"cls_preds", "reg_preds", "dir_preds"} and out["reg_preds"].shape[1] == 14
    with pytest.raises(AssertionError):
        late({"inputs_m1": {"points": [None]}, "inputs_m2": {}})
    for method in ("v2xvit", "att", "max"):
        base = _stub_model(create_model(configs.lidar_baseline(method)), 256)
        out = base(_scene_inputs(["m1", "m1", "m1"]))
        assert tuple(out["cls_preds"].shape)[:2] == (1, 2) and tuple(out["dir_preds"].shape)[:2] == (1, 4)
        assert not any(k.startswith("depth_items") for k in out)


@torch.no_grad()
def test_oldstyle_pointpillar_glue(monkeypatch):
    from heal_amd.opencood.models import point_pillar as pp
    from heal_amd.opencood.tools.train_utils import create_model
    monkeypatch.setattr(pp, "head", lambda conv, x: conv(x))
    for fusion in (None, "max", "att"):
        model = create_model(configs.oldstyle_pointpillar(fusion, compression=0 if fusion is None else 4)).eval()
        n = 1 if fusion is None else 3
        monkeypatch.setattr(type(model), "encode_processed_lidar", lambda self, d, n=n: torch.randn(n, 64, 32, 32))
        model.backbone = _Backbone()
        model.shrink_conv = nn.Conv2d(64, 256, 1)
        if fusion is not None:
            model.fusion_net = _Fusion()
            model.naive_compressor = nn.Identity()
            assert model.compression
        out = model({"record_len": torch.tensor([n]), "pairwise_t_matrix": torch.from_numpy(np.tile(np.eye(4), (1, 5, 5, 1, 1)))})
        assert tuple(out["cls_preds"].shape) == (1, 2, 32, 32) and tuple(out["reg_preds"].shape) == (1, 14, 32, 32)
        assert tuple(out["dir_preds"].shape) == (1, 4, 32, 32)
    frozen = create_model({"model": {"core_method": "point_pillar_baseline",
                                     "args": dict(configs.oldstyle_pointpillar("max")["model"]["args"], backbone_fix=True)}})
    assert not any(p.requires_grad for p in frozen.backbone.parameters())
    assert not any(p.requires_grad for p in frozen.cls_head.parameters())


def test_no_runtime_memset_or_d2d_copy_in_the_library():
    """Round-5 rule (include/heal_amd.h, heal_fill_bytes): the library initialises device memory with its own fill KERNEL.  A
    hipMemsetAsync / hipMemcpyAsync captured into a HIP graph becomes a runtime-executed node, and the r4 memory fault was such a node
    writing a wrong pattern (profiles/r05_k1_memset_node_dump.txt).  No source file may call either."""
    import glob
    import os
    import re
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "heal_amd", "csrc")
    hits = []
    for path in sorted(glob.glob(os.path.join(src, "*")) + glob.glob(os.path.join(src, "experimental", "*"))):
        if os.path.isdir(path):
            continue
        text = open(path).read()
        text = re.sub(r"//.*", "", text)
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b(hipMemset\w*|hipMemcpy\w*Async)\s*\(", text):     # (a blocking D2H read in a debug branch is no graph node)
            hits.append((os.path.basename(path), m.group(0)))
    assert not hits, hits


def test_collection_order_names_exist():
    """tests/conftest.py runs one parity test per BASELINE config FIRST (by name) and the rank-spawning file LAST: a renamed test or file
    would silently fall out of that order.  Every name in `_FIRST` must be a collected test, and they must lead the GPU collection."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "--collect-only", "-q"], cwd=root,
                         capture_output=True, text=True, timeout=600).stdout
    ids = [line.strip() for line in out.splitlines() if "::" in line]
    from tests import conftest
    want = ["tests/" + name for name in conftest._FIRST]
    assert ids[:len(want)] == want, ids[:len(want)]
    files = [i.split("::")[0] for i in ids]
    assert files[-1] == "tests/test_gpu_dist.py" and "tests/test_gpu_dist.py" not in files[:files.index("tests/test_gpu_dist.py")]
    for f in conftest._FILE_RANK:
        assert os.path.exists(os.path.join(root, "tests", f)), f
