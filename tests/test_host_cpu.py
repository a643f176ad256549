"""CPU-side checks: C-ABI surface, host logic, configs, oracle known-answer tests."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from oracle import cref
from oracle import oracle_np as O
from tests.golden.detfill import fill_module

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL_RANGE = [-25.6, -25.6, -3, 25.6, 25.6, 1]


# ------------------------------------------------------------------------------------------- C ABI
def test_library_loads_and_exports_every_declared_symbol():
    from heal_amd import _capi, build
    build.build()
    names = _capi.declared_symbols()
    assert "heal_voxelize" in names and "heal_decode_nms" in names
    L = ctypes.CDLL(_capi.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/heal_amd.h but not exported: {missing}"
    assert set(names) <= set(_capi._SIGNATURES), "ctypes signature table is missing a declared symbol"
    assert _capi.lib().heal_abi_version() == _capi.abi_version_of_header() >= 2
    # the measured-negative kernels are NOT part of the shipped ABI: declared in their own header, bound by the
    # signature table, exported only by a HEAL_BUILD_EXPERIMENTAL=1 library -- all of them or none
    exp = _capi.declared_symbols(experimental=True)
    assert exp and not set(exp) & set(names) and set(exp) <= set(_capi._SIGNATURES)
    have = [n for n in exp if hasattr(L, n)]
    assert have == [] or have == exp, f"half an experimental build: {have}"
    import os
    if os.environ.get("HEAL_BUILD_EXPERIMENTAL", "0") != "1":
        assert have == [], "the default build must not carry the experimental kernels"


def test_product_has_no_cpu_path():
    from heal_amd import _capi, ops
    with pytest.raises(_capi.HealAmdError):
        ops.voxelize(torch.zeros(8, 4), SMALL_RANGE, [0.4, 0.4, 4], 32, 100)
    with pytest.raises(_capi.HealAmdError):
        ops.warp_fuse(torch.zeros(1, 4, 8, 8), torch.zeros(1, 1, 8, 8), np.zeros((1, 2, 3)))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under heal_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "heal_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text, os.path.join(dirpath, f)
                assert "/root/reference" not in text, os.path.join(dirpath, f)


# ------------------------------------------------------------------------------------- host mirror
def test_state_dict_keys_match_reference():
    from heal_amd import configs
    from heal_amd.opencood.tools.train_utils import create_model
    keys = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")))
    for name, hy in (("collab", configs.lidar_pyramid()), ("single", configs.m1_single_pyramid()),
                     ("late", configs.m1_late()), ("baseline_v2xvit", configs.lidar_baseline("v2xvit"))):
        sd = create_model(hy).state_dict()
        assert {k: list(v.shape) for k, v in sd.items()} == keys[name], name
    # old-style models
    for name, hy in (("point_pillar", configs.oldstyle_pointpillar()),
                     ("point_pillar_baseline_max", configs.oldstyle_pointpillar("max", compression=4)),
                     ("point_pillar_baseline_att", configs.oldstyle_pointpillar("att", compression=4))):
        sd = create_model(hy).state_dict()
        assert {k: list(v.shape) for k, v in sd.items()} == keys[name], name


def test_oldstyle_lift_splat_shoot_keys():
    """SURVEY 8f-3, opencood/models/lift_splat_shoot.py: torchvision / efficientnet_pytorch are not installed, so there is
    no generated key list; the names below are the packages' published parameter names the reference's checkpoints use."""
    from heal_amd import configs
    from heal_amd.opencood.tools.train_utils import create_model
    model = create_model(configs.oldstyle_lss())
    assert type(model).__name__ == "LiftSplatShoot"
    sd = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert sd["bevencode.conv1.weight"] == (64, 128, 7, 7)
    assert sd["bevencode.layer2.0.downsample.0.weight"] == (128, 64, 1, 1)
    assert sd["bevencode.layer3.1.conv2.weight"] == (256, 256, 3, 3)
    assert sd["bevencode.up1.conv.0.weight"] == (256, 320, 3, 3)
    assert sd["bevencode.up2.1.weight"] == (128, 256, 3, 3) and sd["bevencode.up2.4.bias"] == (128,)
    assert sd["camencode.trunk._conv_stem.weight"] == (32, 3, 3, 3) and sd["camencode.depth_head.weight"] == (48, 512, 1, 1)
    assert sd["shrink_conv.layers.0.double_conv.0.weight"] == (128, 128, 3, 3)
    assert sd["cls_head.weight"] == (2, 128, 1, 1) and sd["reg_head.weight"] == (14, 128, 1, 1) and sd["dir_head.weight"] == (4, 128, 1, 1)
    n_bev = sum(1 for k in sd if k.startswith("bevencode.") and not k.endswith("num_batches_tracked"))
    # stem 1+4, 6 BasicBlocks x 10, 2 downsamples x 5, up1 2x5, up2 1+4+2
    assert n_bev == 5 + 60 + 10 + 10 + 7
    with pytest.raises(KeyError):
        create_model({"model": {"core_method": "lift_splat_shoot", "args": {"grid_conf": {}}}})


def test_generate_gt_bbx_and_collate_batch_match_reference(golden):
    """Evaluation-side companions of post_process: BasePostprocessor.generate_gt_bbx on two cavs with shared
    object ids and out-of-range boxes, VoxelPostprocessor.collate_batch -- equal to the imported reference's outputs."""
    from heal_amd.opencood.data_utils.post_processor.voxel_postprocessor import VoxelPostprocessor
    g = golden("gt")
    post = VoxelPostprocessor({"order": str(g["order"]), "gt_range": g["gt_range"].tolist(), "anchor_args": {"num": 2}},
                              train=False)
    data = {name: {"object_bbx_center": torch.from_numpy(g[f"{name}_center"]),
                   "object_bbx_mask": torch.from_numpy(g[f"{name}_mask"]),
                   "object_ids": g[f"{name}_ids"].tolist(),
                   "transformation_matrix_clean": torch.from_numpy(g[f"{name}_tfm"])} for name in ("ego", "cav1")}
    got = post.generate_gt_bbx(data)
    assert got.dtype == torch.float32 and tuple(got.shape) == g["gt_box"].shape
    np.testing.assert_allclose(got.numpy(), g["gt_box"], rtol=0, atol=1e-5)
    frames = [{n: g[f"frame{k}_{n}"] for n in ("pos_equal_one", "neg_equal_one", "targets")} for k in range(3)]
    col = VoxelPostprocessor.collate_batch(frames)
    for n in ("pos_equal_one", "neg_equal_one", "targets"):
        assert col[n].dtype == torch.float64 and np.array_equal(col[n].numpy(), g[f"col_{n}"])


def test_pcd_utils_host_mirror_matches_reference(golden):
    """opencood/utils/pcd_utils.py filters / projection (numpy contract) against the imported reference's outputs,
    including points exactly on the range and ego-box faces and a NaN point."""
    from heal_amd.opencood.utils import pcd_utils
    g = golden("pcd")
    pts, rng = g["points"], g["lidar_range"].tolist()
    ego = pcd_utils.mask_ego_points(pts)
    assert np.array_equal(ego, g["ego"], equal_nan=True)
    both = pcd_utils.mask_points_by_range(ego, rng)
    assert np.array_equal(both, g["ego_range"])
    assert np.array_equal(pcd_utils.mask_points_by_range(pts, rng), g["only_range"])
    proj = pcd_utils.lidar_project(both, g["tfm"])
    assert proj.dtype == g["projected"].dtype and np.array_equal(proj, g["projected"])
    assert np.array_equal(pcd_utils.projected_lidar_stack([both[:5], ego[:3]]), g["stacked"])
    assert pcd_utils.shuffle_points(both).shape == both.shape


@pytest.mark.grad
def test_pyramid_loss_matches_reference_golden(golden):
    """SURVEY 8f-2 (training side): PointPillarPyramidLoss value and gradients against the imported reference's
    (tests/golden/loss.npz), fused heads / per-agent occupancy pass / single-agent model, with and without the depth
    foreground mask."""
    from heal_amd import configs
    from heal_amd.opencood.tools.train_utils import create_loss
    from tests.test_reference_live import _leafs, _loss_inputs
    g = golden("loss")
    for fg in (0, 1):
        hy = configs.lidar_pyramid()
        hy["loss"]["args"]["depth"]["use_fg_mask"] = bool(fg)
        crit = create_loss(hy)
        for seed, mode, suffix in ((0, "collab", ""), (1, "collab", "_single"), (2, "single", "")):
            out, tgt = _loss_inputs(seed)
            out["pyramid"] = mode
            leafs = _leafs(out)
            loss = crit(out, tgt, suffix)
            loss.backward()
            tag = f"fg{fg}_s{seed}"
            np.testing.assert_allclose(loss.detach().numpy(), g[f"{tag}_loss"], rtol=1e-6)
            for k, leaf in enumerate(leafs):
                assert (leaf.grad is not None) == (f"{tag}_grad{k}" in g.files), (tag, k)
                if leaf.grad is not None:
                    np.testing.assert_allclose(leaf.grad.numpy(), g[f"{tag}_grad{k}"], rtol=1e-5, atol=1e-9)


def test_preprocessor_deferred_mode_needs_no_gpu_and_survives_the_dataset_merges(monkeypatch):
    """SpVoxelPreprocessor with `defer_to_device`: preprocess() runs in forked DataLoader workers, so it must not touch the
    GPU; what it returns has to pass through the datasets' merge_features_to_dict (values gathered into lists, twice) and
    collate_batch into the encoder's device-points input."""
    from collections import OrderedDict
    from heal_amd import ops
    from heal_amd.opencood.data_utils.pre_processor.sp_voxel_preprocessor import SpVoxelPreprocessor
    monkeypatch.setattr(ops, "voxelize", lambda *a, **k: (_ for _ in ()).throw(AssertionError("GPU touched")))
    params = {"cav_lidar_range": [-102.4, -102.4, -3, 102.4, 102.4, 1],
              "args": {"voxel_size": [0.4, 0.4, 4], "max_points_per_voxel": 32, "max_voxel_train": 32000,
                       "max_voxel_test": 70000, "defer_to_device": True}}
    pre = SpVoxelPreprocessor(params, train=False)
    rng = np.random.default_rng(0)
    clouds = [rng.standard_normal((n, c)) for n, c in ((100, 4), (50, 5), (7, 4))]
    per_cav = [pre.preprocess(c) for c in clouds]
    assert all(d["points"].dtype == np.float32 and d["points"].shape[1] == 4 for d in per_cav)

    def merge(dicts):  # common_utils.merge_features_to_dict (:48-92) without the image branches
        out = OrderedDict()
        for d in dicts:
            for k, v in d.items():
                out.setdefault(k, [])
                out[k] += v if isinstance(v, list) else [v]
        return out
    sample0, sample1 = merge(per_cav[:2]), merge(per_cav[2:])          # per scene, then across the batch
    batch = pre.collate_batch(merge([sample0, sample1]))
    assert [tuple(t.shape) for t in batch["points"]] == [(100, 4), (50, 4), (7, 4)]
    assert batch["max_points_per_voxel"] == 32 and batch["max_voxels"] == 70000
    np.testing.assert_array_equal(batch["points"][1].numpy(), clouds[1][:, :4].astype(np.float32))
    assert len(pre.collate_batch(per_cav)["points"]) == 3               # list form (late fusion datasets)
    assert SpVoxelPreprocessor(params, train=True).preprocess(clouds[0])["max_voxels"] == 32000
    monkeypatch.setenv("HEAL_DEFER_VOXELIZE", "1")
    params["args"].pop("defer_to_device")
    assert "points" in SpVoxelPreprocessor(params, train=False).preprocess(clouds[0])


def test_deferred_labels_pack_in_the_worker_and_resolve_at_the_loss(monkeypatch):
    """VoxelPostprocessor with `defer_to_device`: generate_label / collate_batch (DataLoader worker side) must not touch
    the GPU and must carry everything the assignment needs; resolve_deferred_labels (loss side) adds the three label
    tensors once.  The assignment itself is the ordinary generate_label (GPU-tested); here it is replaced by a recorder."""
    from heal_amd import configs
    from heal_amd.opencood.data_utils.post_processor import voxel_postprocessor as vp
    hy = configs.lidar_pyramid(SMALL_RANGE)
    post = vp.VoxelPostprocessor(dict(hy["postprocess"], defer_to_device=True), train=True)
    anchors = post.generate_anchor_box()
    H, W, A = anchors.shape[:3]
    rng = np.random.default_rng(3)
    gts = [rng.standard_normal((20, 7)).astype(np.float32) for _ in range(2)]
    masks = [np.r_[np.ones(k), np.zeros(20 - k)].astype(np.float32) for k in (5, 0)]
    frames = [post.generate_label(gt_box_center=g, anchors=anchors, mask=m) for g, m in zip(gts, masks)]   # no GPU here
    batch = post.collate_batch(frames)
    assert tuple(batch["deferred_gt_box_center"].shape) == (2, 20, 7) and tuple(batch["deferred_anchors"].shape) == anchors.shape
    assert batch["deferred_pos_threshold"] == 0.6 and batch["deferred_neg_threshold"] == 0.45
    from heal_amd.opencood.tools.train_utils import to_device
    assert to_device(batch, "cpu")["deferred_pos_threshold"] == 0.6          # floats pass through to_device
    calls = []

    def fake_generate_label(self, **kw):
        assert not self.defer
        calls.append((kw["gt_box_center"].copy(), kw["mask"].copy(), kw["anchors"].shape))
        return {"pos_equal_one": np.zeros((H, W, A)), "neg_equal_one": np.ones((H, W, A)), "targets": np.zeros((H, W, 7 * A))}
    monkeypatch.setattr(vp.VoxelPostprocessor, "generate_label", fake_generate_label)
    out = vp.resolve_deferred_labels(batch)
    assert out is batch and len(calls) == 2 and np.array_equal(calls[1][0], gts[1]) and np.array_equal(calls[0][1], masks[0])
    assert tuple(batch["pos_equal_one"].shape) == (2, H, W, A) and batch["targets"].dtype == torch.float64
    vp.resolve_deferred_labels(batch)
    assert len(calls) == 2                                                    # resolved once
    plain = {"pos_equal_one": torch.zeros(1)}
    assert vp.resolve_deferred_labels(plain) is plain and len(calls) == 2


def test_yaml_loader_round_trip(tmp_path):
    from heal_amd import configs
    from heal_amd.opencood.hypes_yaml import yaml_utils
    hy = configs.lidar_pyramid()
    p = tmp_path / "config.yaml"
    configs.dump_yaml(hy, str(p))
    back = yaml_utils.load_yaml(str(p))
    assert back["postprocess"]["anchor_args"]["W"] == 512 and back["postprocess"]["anchor_args"]["H"] == 512
    assert back["model"]["core_method"] == "heter_pyramid_collab"
    # the float resolver accepts exponent forms without a dot (yaml_utils.py:34-44)
    q = tmp_path / "f.yaml"
    q.write_text("a: 1e-10\nb: 2.5e3\nc: [0.4, 0.4, 4]\n")
    d = yaml_utils.load_yaml(str(q))
    assert isinstance(d["a"], float) and d["a"] == 1e-10 and d["b"] == 2500.0

    class Opt:
        model_dir = str(tmp_path)
    assert yaml_utils.load_yaml(None, Opt())["name"] == hy["name"]
    small = yaml_utils.update_ranges(back, SMALL_RANGE)
    assert small["model"]["args"]["m1"]["encoder_args"]["lidar_range"] == SMALL_RANGE


def test_center_crop_matches_torchvision_spec():
    from heal_amd.opencood.models._heter_common import center_crop
    x = torch.arange(2 * 3 * 4 * 6, dtype=torch.float32).reshape(2, 3, 4, 6)
    y = center_crop(x, 8, 12)  # pad 128^2 -> 256^2 style
    assert y.shape == (2, 3, 8, 12) and torch.equal(y[..., 2:6, 3:9], x) and y.sum() == x.sum()
    z = center_crop(x, 2, 4)
    assert torch.equal(z, x[..., 1:3, 1:5])
    o = center_crop(x, 5, 7)  # odd pad: left/top floor, right/bottom ceil
    assert o.shape == (2, 3, 5, 7) and torch.equal(o[..., 0:4, 0:6], x)


def test_model_ref_matches_reference_golden(golden):
    """The CPU port used as bench.py's cpu_baseline reproduces the reference end to end."""
    from heal_amd import configs
    from heal_amd.opencood.tools.train_utils import create_model
    from oracle import model_ref
    g = golden("collab_small")
    hy = configs.lidar_pyramid(SMALL_RANGE)
    sd = fill_module(create_model(hy)).state_dict()
    out = model_ref.heter_pyramid_collab_m1(sd, hy["model"]["args"], g["a2_voxel_features"], g["a2_voxel_coords"],
                                            g["a2_voxel_num_points"], 2, g["a2_pairwise"])
    for k, name in (("cls_preds", "cls"), ("reg_preds", "reg"), ("dir_preds", "dir")):
        ref = g[f"a2_{name}"]
        assert np.abs(out[k] - ref).max() / np.abs(ref).max() < 1e-4, k
    for i in range(3):
        np.testing.assert_allclose(out["occ_single_list"][i], g[f"a2_occ{i}"], rtol=1e-3, atol=1e-4)


# ------------------------------------------------------------------- oracle known-answer: voxelize
def test_oracle_voxelize_known_answers():
    R = [0.0, 0.0, 0.0, 4.0, 2.0, 1.0]
    V = [1.0, 1.0, 1.0]  # grid 4 x 2 x 1
    pts = np.array([
        [0.5, 0.5, 0.5, 1],     # voxel 0 = cell (z0,y0,x0)
        [3.5, 1.5, 0.5, 2],     # voxel 1 = (0,1,3)
        [0.6, 0.4, 0.1, 3],     # voxel 0, slot 1
        [4.0, 0.5, 0.5, 4],     # x == max -> dropped (upper bound exclusive)
        [0.0, 0.0, 0.0, 5],     # lower bound inclusive -> voxel 0, slot 2 (over P=2 -> dropped)
        [-0.0001, 0.5, 0.5, 6],  # below range
        [1.5, 0.5, 0.5, 7],     # voxel 2
        [2.5, 0.5, 0.5, 8],     # would be voxel 3 but max_voxels = 3 -> dropped
        [1.2, 0.9, 0.9, 9],     # existing voxel 2 still accepts points after the cap
    ], np.float32)
    v, c, n = cref.voxelize(pts, R, V, max_points=2, max_voxels=3)
    assert c.tolist() == [[0, 0, 0], [0, 1, 3], [0, 0, 1]]
    assert n.tolist() == [2, 1, 2]
    assert v[0, :, 3].tolist() == [1, 3] and v[1, :, 3].tolist() == [2, 0] and v[2, :, 3].tolist() == [7, 9]
    # batch index prepend (collate) and empty input
    _, cb, _ = cref.voxelize(pts, R, V, 2, 3, batch_idx=4)
    assert cb[:, 0].tolist() == [4, 4, 4]
    v0, c0, n0 = cref.voxelize(np.zeros((0, 4), np.float32), R, V, 2, 3)
    assert v0.shape == (0, 2, 4) and n0.shape == (0,)


def test_oracle_voxelize_float32_floor_semantics():
    """(p - min) / size is evaluated in float32: a point a hair below a cell edge in real arithmetic
    can land on the edge after rounding -- the restatement must follow fp32, not fp64."""
    R = [-102.4, -102.4, -3, 102.4, 102.4, 1]
    V = [0.4, 0.4, 4]
    xs = np.float32(-102.4) + np.arange(1, 512, dtype=np.float32) * np.float32(0.4)
    pts = np.stack([xs, np.zeros_like(xs), np.zeros_like(xs), np.zeros_like(xs)], 1).astype(np.float32)
    _, c, _ = cref.voxelize(pts, R, V, 4, 1000)
    want = np.floor((xs - np.float32(-102.4)) / np.float32(0.4)).astype(np.int32)
    got = {}
    # voxels are in first-appearance order; rebuild per-point cell from the oracle run point by point
    for i in range(len(xs)):
        _, ci, _ = cref.voxelize(pts[i:i + 1], R, V, 4, 10)
        got[i] = ci[0, 2]
    assert [got[i] for i in range(len(xs))] == want.tolist()


# ------------------------------------------------------------------------ oracle known-answer: IoU
def _sq(cx, cy, s, th=0.0):
    p = np.array([[s / 2, -s / 2], [s / 2, s / 2], [-s / 2, s / 2], [-s / 2, -s / 2]])
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    return (p @ R.T + [cx, cy]).astype(np.float32)


def test_oracle_quad_iou_analytic_cases():
    a = _sq(0, 0, 2)
    assert cref.quad_iou(a, a)[0, 0] == 1.0
    assert cref.quad_iou(a, _sq(10, 10, 2))[0, 0] == 0.0
    assert abs(cref.quad_iou(a, _sq(1, 0, 2))[0, 0] - 1 / 3) < 1e-7           # half-shifted
    inter = 2 * np.sqrt(2) - 2                                                 # unit squares at 45 deg
    assert abs(cref.quad_iou(_sq(0, 0, 1), _sq(0, 0, 1, np.pi / 4))[0, 0] - inter / (2 - inter)) < 1e-6
    assert cref.quad_iou(a, a[::-1].copy())[0, 0] == 1.0                       # winding does not matter
    assert np.isnan(cref.quad_iou(np.zeros((4, 2), np.float32), np.zeros((4, 2), np.float32))[0, 0])
    assert cref.quad_iou(a, _sq(2, 0, 2))[0, 0] == 0.0                         # touching edges


def test_oracle_quad_iou_against_scipy_halfspaces():
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(0)

    def inter_area(p, q):
        # brute force: vertices of the intersection = p-in-q, q-in-p and edge crossings
        def inside(pt, poly):
            s = []
            for i in range(4):
                a, b = poly[i], poly[(i + 1) % 4]
                s.append((b[0] - a[0]) * (pt[1] - a[1]) - (b[1] - a[1]) * (pt[0] - a[0]))
            return all(v >= -1e-12 for v in s) or all(v <= 1e-12 for v in s)
        pts = [pt for pt in p if inside(pt, q)] + [pt for pt in q if inside(pt, p)]
        for i in range(4):
            for j in range(4):
                a, b, c, d = p[i], p[(i + 1) % 4], q[j], q[(j + 1) % 4]
                den = (b[0] - a[0]) * (d[1] - c[1]) - (b[1] - a[1]) * (d[0] - c[0])
                if abs(den) < 1e-14:
                    continue
                t = ((c[0] - a[0]) * (d[1] - c[1]) - (c[1] - a[1]) * (d[0] - c[0])) / den
                u = ((c[0] - a[0]) * (b[1] - a[1]) - (c[1] - a[1]) * (b[0] - a[0])) / den
                if 0 <= t <= 1 and 0 <= u <= 1:
                    pts.append(a + t * (b - a))
        if len(pts) < 3:
            return 0.0
        try:
            return ConvexHull(np.array(pts)).volume
        except Exception:
            return 0.0
    for _ in range(200):
        p = _sq(rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(1, 4), rng.uniform(-3, 3)).astype(np.float64)
        q = _sq(rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(1, 4), rng.uniform(-3, 3)).astype(np.float64)
        p32, q32 = p.astype(np.float32), q.astype(np.float32)
        ia = inter_area(p32.astype(np.float64), q32.astype(np.float64))
        sa = ConvexHull(p32.astype(np.float64)).volume
        sb = ConvexHull(q32.astype(np.float64)).volume
        want = ia / (sa + sb - ia)
        assert abs(cref.quad_iou(p32, q32)[0, 0] - want) < 1e-5


def test_oracle_nms_control_flow():
    quads = np.stack([_sq(0, 0, 2), _sq(0.1, 0, 2), _sq(5, 5, 2), _sq(5.05, 5, 2), _sq(-7, 3, 2)])
    scores = np.array([0.9, 0.8, 0.3, 0.95, 0.5], np.float32)
    order = O.nms_order(scores)
    assert order.tolist() == [3, 0, 1, 4, 2]
    assert cref.nms_rotated(quads, order, 0.15).tolist() == [3, 0, 4]
    assert cref.nms_rotated(quads, order[:2], 0.15).tolist() == [3, 0]          # top-k truncation
    assert cref.nms_rotated(quads, order, 0.99).tolist() == [3, 0, 1, 4, 2]
    tie = np.array([0.5, 0.5, 0.5], np.float32)
    assert O.nms_order(tie).tolist() == [2, 1, 0]                               # documented tie rule


def test_synth_scene_is_deterministic():
    from heal_amd import synth
    a, b = synth.lidar_frame(5, n_azimuth=256), synth.lidar_frame(5, n_azimuth=256)
    assert np.array_equal(a, b) and a.dtype == np.float32 and a.shape[1] == 4
    pw = synth.pairwise_t_matrix(synth.agent_poses(1, 3), 5)
    assert pw.shape == (5, 5, 4, 4) and pw.dtype == np.float64
    np.testing.assert_allclose(pw[0, 1] @ pw[1, 0], np.eye(4), atol=1e-9)


def test_oracle_sparse_conv_known_answers():
    """spconv is absent (parity unpinned): the oracle's active-site rules are pinned by hand-made cases.
    Submanifold: outputs only at input sites, neighbours that are inactive contribute nothing.
    Regular (strided): an output site is active iff any input lies in its receptive field."""
    import torch
    from oracle import oracle_np as O
    feats = np.array([[1.0, 2.0], [10.0, 20.0]], np.float32)          # two sites, C_in = 2
    idx = np.array([[0, 1, 1, 1], [0, 1, 1, 2]], np.int64)            # (b, z, y, x): x-neighbours
    dense, mask = O.densify(feats, idx, (4, 4, 4), 1)
    w = np.zeros((3, 3, 3, 2, 1), np.float32)
    w[1, 1, 1] = [[1.0], [1.0]]                                       # centre tap: sum of channels
    w[1, 1, 2] = [[0.5], [0.0]]                                       # +x neighbour: half of channel 0
    one, zero = np.ones(1, np.float32), np.zeros(1, np.float32)
    y, m = O.sparse_conv_dense(dense, mask, w, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, one, zero, zero, one - 1e-3)
    assert m.sum() == 2 and torch.equal(m, mask)                      # same active set
    np.testing.assert_allclose(float(y[0, 0, 1, 1, 1]), 3.0 + 0.5 * 10.0, rtol=1e-6)   # centre + right neighbour
    np.testing.assert_allclose(float(y[0, 0, 1, 1, 2]), 30.0, rtol=1e-6)               # right neighbour is inactive
    assert float(y.abs().sum()) == pytest.approx(38.0, rel=1e-6)      # nothing leaks to inactive cells
    # strided 3x3x3, stride 2, padding 1: output cell o covers inputs 2o-1 .. 2o+1 per axis
    y2, m2 = O.sparse_conv_dense(dense, mask, np.ones((3, 3, 3, 2, 1), np.float32), (3, 3, 3), (2, 2, 2), (1, 1, 1),
                                 False, one, zero, zero, one - 1e-3)
    act = {tuple(int(v) for v in t) for t in torch.nonzero(m2[0, 0])}
    # input (1,1,1) -> outputs with each coord in {0,1}; input (1,1,2) -> z,y in {0,1}, x in {1}
    assert act == {(z, y_, x) for z in (0, 1) for y_ in (0, 1) for x in (0, 1)}
    np.testing.assert_allclose(float(y2[0, 0, 0, 0, 0]), 3.0, rtol=1e-6)     # sees only the first site
    np.testing.assert_allclose(float(y2[0, 0, 0, 0, 1]), 33.0, rtol=1e-6)    # sees both


def test_pose_algebra_mirror_matches_reference_golden(golden):
    """opencood/utils/transformation_utils.py:21-66,264-334 (host numpy): x_to_world, x1_to_x2,
    get_pairwise_transformation -- the producers of `pairwise_t_matrix` -- against the reference's outputs; the
    synthetic scene generator must agree with them too."""
    from heal_amd import synth
    from heal_amd.opencood.utils import transformation_utils as tu
    g = golden("pose")
    poses = g["poses"]
    for k, p in enumerate(poses):
        np.testing.assert_array_equal(tu.x_to_world(p.tolist()), g["x_to_world"][k])
        np.testing.assert_allclose(synth.x_to_world(p.tolist()), g["x_to_world"][k], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(tu.x1_to_x2(poses[1].tolist(), poses[2].tolist()), g["x1_to_x2"])
    base = {k: {"params": {"lidar_pose": poses[k].tolist()}} for k in range(4)}
    np.testing.assert_array_equal(tu.get_pairwise_transformation(base, 5, False), g["pairwise"])
    np.testing.assert_array_equal(tu.get_pairwise_transformation(base, 5, True), g["pairwise_proj_first"])
    np.testing.assert_allclose(synth.pairwise_t_matrix([p.tolist() for p in poses], 5), g["pairwise"], rtol=1e-12, atol=1e-12)
    import torch
    parts = tu.regroup(torch.arange(10).view(5, 2), torch.tensor([2, 3]))
    assert [tuple(p.shape) for p in parts] == [(2, 2), (3, 2)]


def test_inference_utils_mirror_contract():
    """opencood/tools/inference_utils.py: the model-calling helpers only need model(dict) and dataset.post_process."""
    from heal_amd.opencood.tools import inference_utils as iu

    class DS:
        def post_process(self, batch, out):
            return ("boxes", sorted(out.keys()), "gt")

        def post_process_no_fusion(self, batch, out):
            return ("boxes_nf", sorted(batch.keys()), "gt")

    model = lambda d: {"cls_preds": d["x"], "depth_items": 7} if "d" in d else {"cls_preds": d["x"]}
    batch = {"ego": {"x": 1, "d": 1}, "cav1": {"x": 2}}
    r = iu.inference_late_fusion(batch, model, DS())
    assert r == {"pred_box_tensor": "boxes", "pred_score": ["cav1", "ego"], "gt_box_tensor": "gt"}
    r = iu.inference_intermediate_fusion(batch, model, DS())
    assert r["pred_score"] == ["ego"] and r["depth_items"] == 7
    assert iu.inference_no_fusion(batch, model, DS(), single_gt=True)["pred_score"] == ["ego"]
    assert iu.inference_no_fusion(batch, model, DS())["pred_score"] == ["cav1", "ego"]


def test_deferred_validation_labels_are_not_zeroed_without_the_inference_switch(monkeypatch):
    """tools/train.py validates with train=False datasets and computes the loss on their labels: the deferred mode must
    pack the label inputs there too; only the explicit inference-only switch may hand out all-zero labels."""
    from heal_amd import configs
    from heal_amd.opencood.data_utils.post_processor import voxel_postprocessor as vp
    monkeypatch.delenv("HEAL_INFERENCE_ONLY", raising=False)
    hy = configs.lidar_pyramid(SMALL_RANGE)
    gt = np.zeros((20, 7), np.float32)
    gt[0] = [3, 4, -1, 1.6, 1.8, 4.0, 0.3]
    mask = np.r_[1.0, np.zeros(19)].astype(np.float32)
    val = vp.VoxelPostprocessor(dict(hy["postprocess"], defer_to_device=True), train=False)
    anchors = val.generate_anchor_box()
    out = val.generate_label(gt_box_center=gt, anchors=anchors, mask=mask)
    assert "deferred_gt_box_center" in out and "pos_equal_one" not in out
    inf = vp.VoxelPostprocessor(dict(hy["postprocess"], defer_to_device=True, inference_only=True), train=False)
    out = inf.generate_label(gt_box_center=gt, anchors=anchors, mask=mask)
    assert float(np.abs(out["pos_equal_one"]).sum()) == 0 and out["targets"].shape[-1] == 14


def test_pack_rig_views_share_one_buffer():
    """pipeline.pack_rig: the five rig tensors of a camera agent are views of ONE flat tensor, so loading a frame into the
    static buffers of a captured graph is one copy for the rig; values and shapes are those of the separate tensors."""
    from heal_amd import synth
    from heal_amd.pipeline import RIG_KEYS, pack_rig
    rig = {k: torch.from_numpy(v) for k, v in synth.camera_rig(3, 4, 96, 128).items()}
    packed = pack_rig(rig, "cpu")
    flat = packed["_rig"]
    assert flat.dim() == 1 and flat.numel() == sum(rig[k].numel() for k in RIG_KEYS)
    for k in RIG_KEYS:
        assert packed[k].shape == rig[k].shape and torch.equal(packed[k], rig[k].float())
        assert packed[k].data_ptr() >= flat.data_ptr() and packed[k].data_ptr() < flat.data_ptr() + 4 * flat.numel()
    other = pack_rig({k: torch.zeros_like(v) for k, v in rig.items()}, "cpu")
    other["_rig"].copy_(flat)                                   # what StaticInputs.load does
    assert all(torch.equal(other[k], packed[k]) for k in RIG_KEYS)


def test_grad_path_switch():
    """bev_blocks.grad_path: off under no_grad whatever the module state; on with autograd when the input carries gradient, a
    module trains, or a module has trainable parameters (a frozen eval-mode block with a gradient-free input stays on the
    inference operators)."""
    import torch.nn as nn
    from heal_amd.opencood.models.sub_modules.bev_blocks import grad_path
    conv = nn.Conv2d(4, 4, 1)
    x = torch.zeros(1, 4, 2, 2)
    with torch.no_grad():
        assert not grad_path(x, conv) and not grad_path(x.requires_grad_(False), conv.train())
    with torch.enable_grad():
        conv.eval()
        assert grad_path(x, conv)                                # trainable parameters
        for p in conv.parameters():
            p.requires_grad_(False)
        assert not grad_path(x, conv)                            # frozen, eval, input without gradient
        assert grad_path(x.clone().requires_grad_(True), conv)   # gradient flows through
        assert grad_path(x, conv.train())                        # training mode (BatchNorm statistics etc.)


def test_conv1x1_ksplit_policy_never_leaves_an_empty_split(monkeypatch):
    """ops.conv1x1_ksplit: 1 for big grids / shallow reductions; otherwise a split count that the kernel accepts -- every split
    owns at least one K chunk (heal_conv1x1_splitk rejects the rest), n * ksplit fits the launch grid."""
    from heal_amd import ops
    monkeypatch.delenv("HEAL_C1_KSPLIT", raising=False)
    assert ops.conv1x1_ksplit(5, 256, 512, 64 * 64) == 1 and ops.conv1x1_ksplit(4, 16, 96, 192 * 256) == 1
    assert ops.conv1x1_ksplit(4, 1152, 192, 12 * 16) > 1
    for n in (1, 4, 700):
        for cin in (33, 256, 300, 1152, 1153):
            for want in (None, 2, 5, 7, 36, 1000):
                if want is None:
                    monkeypatch.delenv("HEAL_C1_KSPLIT", raising=False)
                else:
                    monkeypatch.setenv("HEAL_C1_KSPLIT", str(want))
                ks = ops.conv1x1_ksplit(n, cin, 192, 12 * 16)
                chunks = (cin + 31) // 32
                assert 1 <= ks <= chunks and n * ks <= 65535
                if ks > 1:
                    per = -(-chunks // ks)
                    assert (ks - 1) * per < chunks, (n, cin, want, ks)


def test_grouped_small_fragment_layout():
    """ops.grouped_small_fragments: [C, cg, 3, 3] -> [C/16][tap][ci][16]: element (sg, tap, ci, co) = W[sg*16 + co][ci][tap]
    (what k_gconv_small reads as `wq[(sg * 9 cg + tap * cg + ci) * 16 + lane % 16]`)."""
    from heal_amd import ops
    for C, cg in ((128, 4), (256, 8), (512, 16)):
        w = torch.arange(C * cg * 9, dtype=torch.float32).reshape(C, cg, 3, 3)
        f = ops.grouped_small_fragments(w, cg)
        assert tuple(f.shape) == (C // 16, 9, cg, 16) and f.is_contiguous()
        for sg, tap, ci, co in ((0, 0, 0, 0), (C // 16 - 1, 8, cg - 1, 15), (3, 5, 1, 9)):
            assert float(f[sg, tap, ci, co]) == float(w[sg * 16 + co, ci, tap // 3, tap % 3])


def _pairs_near_threshold(n, seed, thr=0.15):
    """Pairs of car-sized rotated boxes whose IoU lies around `thr`: B = A shifted along A's heading by the offset that gives exactly
    `thr` for an aligned pair, then perturbed (offset, lateral shift, yaw) by amounts from 1e-7 to 1e-2 -- so the set holds pairs from
    far on either side of the threshold down to a few ulps of it."""
    from oracle import exact_iou as E
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        cx, cy = rng.uniform(-100, 100, 2)
        L, W = rng.uniform(3.5, 5.0), rng.uniform(1.6, 2.2)
        yaw = rng.uniform(-np.pi, np.pi)
        dx = L * (1 - thr) / (1 + thr)                      # aligned pair: iou = (L - dx) / (L + dx)
        eps = 10.0 ** rng.uniform(-7, -2) * rng.choice([-1, 1])
        lat = 10.0 ** rng.uniform(-7, -2) * rng.choice([-1, 0, 1])
        dyaw = 10.0 ** rng.uniform(-7, -2) * rng.choice([-1, 0, 1])
        bx = cx + (dx + eps) * np.cos(yaw) - lat * np.sin(yaw)
        by = cy + (dx + eps) * np.sin(yaw) + lat * np.cos(yaw)
        out.append((E.box_quad(cx, cy, L, W, yaw), E.box_quad(bx, by, L, W, yaw + dyaw)))
    return out


def test_oracle_quad_iou_against_exact_rational_arithmetic():
    """the NMS survivors are bit-exact by contract, the reference's IoU arithmetic (GEOS) is absent, so the oracle's
    fp64 clip is checked against the TRUE IoU -- exact rational arithmetic on the fp32 corners (oracle/exact_iou.py) -- on 10^4 pairs
    drawn around the 0.15 threshold: (a) the oracle's value is the true value rounded to fp32 (one ulp of slack for a true value on a
    rounding boundary), (b) its suppression decision `iou > 0.15f` equals the true one for EVERY pair, (c) pairs whose true IoU lies
    within 1e-9 of the threshold -- the only ones where GEOS's robust predicates, or any other correct fp64 evaluation, could decide
    differently -- are counted and reported."""
    from fractions import Fraction
    from oracle import cref
    from oracle import exact_iou as E
    from tests.report import note
    pairs = _pairs_near_threshold(10000, seed=7)
    thr32 = np.float32(0.15)
    thr = Fraction(float(thr32))
    got = np.array([cref.quad_iou(a[None], b[None])[0, 0] for a, b in pairs], np.float32)
    close = flips = off_ulp = 0
    worst = 0.0
    for (a, b), g in zip(pairs, got):
        t = E.exact_quad_iou(a, b)
        tf = float(t)
        worst = max(worst, abs(float(g) - tf))
        if np.float32(tf) != g:
            off_ulp += 1
            assert abs(float(g) - tf) <= np.spacing(np.float32(tf)) + 1e-10, (float(g), tf)
        if abs(t - thr) < Fraction(1, 10 ** 9):
            close += 1
        # the decision the reference takes: fp32(iou) > fp32(0.15); the true value decides the same way unless rounding to fp32
        # carries it across the threshold -- which the comparison below would expose
        if bool(g > thr32) != bool(np.float32(tf) > thr32):
            flips += 1
    note("nms_iou_exact_rational", pairs=len(pairs), within_1e9_of_threshold=close, decision_flips=flips,
         not_the_rounded_true_value=off_ulp, max_abs_error=worst)
    assert flips == 0
    assert worst < 1e-7


def test_exact_rational_iou_known_answers():
    """The checker itself: identical boxes 1, disjoint 0, half-shifted axis-aligned unit squares 1/3, a 45-degree rotated unit square
    over a unit square (2 sqrt 2 - 2) / (4 - 2 sqrt 2) (irrational: only approximately), containment = area ratio -- exact where rational."""
    from fractions import Fraction
    from oracle import exact_iou as E
    sq = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    assert E.exact_quad_iou(sq, sq) == 1
    assert E.exact_quad_iou(sq, sq + np.float32(2)) == 0
    assert E.exact_quad_iou(sq, sq + np.array([0.5, 0], np.float32)) == Fraction(1, 3)
    assert E.exact_quad_iou(sq, sq[::-1].copy()) == 1                      # orientation does not matter
    big = np.array([[-1, -1], [3, -1], [3, 3], [-1, 3]], np.float32)
    assert E.exact_quad_iou(sq, big) == Fraction(1, 16)
    assert E.exact_quad_iou(sq, sq + np.array([1, 0], np.float32)) == 0    # shared edge only
    assert E.exact_quad_iou(np.zeros((4, 2), np.float32), np.zeros((4, 2), np.float32)) is None


def test_camera_crop_walk_is_exact_on_the_torch_path():
    """Round 6 (fuse_modules/pyramid_fuse.py): camera agents' zero-padded maps go through the pyramid stages on the crop their content can
    influence, the rest of every stage output is the stack's response to an all-zero map.  The claim is about LOCALITY, not about
    kernels, so it is checked here on the CPU with the modules' torch path (parameters require grad -> conv / BatchNorm as torch operators):
    random weights and BatchNorm statistics, two LiDAR-like agents with dense maps + two camera agents that are zero outside a box,
    both agent orders; every level must equal the plain walk EVERYWHERE (inside and outside the pasted box) to fp32 rounding.  Also the
    interval arithmetic the crop is planned with."""
    from heal_amd.opencood.models.fuse_modules.pyramid_fuse import PyramidFusion, _stage_influence, _stage_needs
    cfg = {"layer_nums": [3, 5, 8], "num_filters": [16, 32, 64], "layer_strides": [1, 2, 2], "upsample_strides": [1, 2, 4],
           "num_upsample_filter": [32, 32, 32], "resnext": True, "inplanes": 16}
    torch.manual_seed(0)
    pf = PyramidFusion(cfg).eval()
    for m in pf.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.5); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.5)
    H = W = 128
    box = (48, 80, 52, 84)
    for cams_first in (False, True):
        x = torch.randn(4, 16, H, W)
        cam = (0, 2) if cams_first else (2, 4)
        keep = torch.zeros(1, 1, H, W)
        keep[..., box[0]:box[1], box[2]:box[3]] = 1
        x[cam[0]:cam[1]] *= keep
        with torch.enable_grad():                       # (the torch path; no autograd graph is needed afterwards)
            # the cached zero-input response, computed here on the torch path (the model computes it under no_grad on the device)
            pf._bg = [f.detach() for f in pf.resnet(torch.zeros(1, 16, H, W))]
            pf._bg_key = pf._zero_response_key(x)
            plain = [f.detach() for f in pf.get_multiscale_feature(x)]
            plan = pf._camcrop_plan(x, cam, box)
            crop = [f.detach() for f in pf.get_multiscale_feature_camcrop(x, cam, box)]
        assert plan[0] is not None and plan[1] is not None          # levels 0 and 1 are cropped at this size
        for i, (a, b) in enumerate(zip(plain, crop)):
            assert a.shape == b.shape
            assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()), (cams_first, i)
    l0, l1 = [type("B", (), {"stride": 1})()] * 3, [type("B", (), {"stride": 2})()] + [type("B", (), {"stride": 1})()] * 4
    assert _stage_influence(64, 192, l0, 256) == (61, 195, 256) and _stage_needs(61, 195, l0, 256) == (58, 198)
    assert _stage_influence(61, 195, l1, 256) == (26, 102, 128) and _stage_needs(26, 102, l1, 256) == (43, 212)
    assert _stage_influence(0, 10, l0, 256)[0] == 0 and _stage_needs(0, 5, l0, 256)[0] == 0      # clipped at the map's border


def test_bench_refuses_profile_files_of_another_library_build(tmp_path, monkeypatch):
    """bench.py quotes `roofline.traffic` / the in-graph rocprof duration only from committed profile files that carry the build stamp of
    the library that is loaded: a summary of another build is refused with the reason in `traffic_source`."""
    import importlib
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    stamp = bench.lib_stamp()
    assert stamp and len(stamp) == 64
    body = {"kernels": {"heal::k_conv1x1<64, 64, 32, 1>": {"dispatches": 4, "bytes_per_dispatch": 1000.0}}}
    for tag, recorded, want in (("other", "0" * 64, None), ("mine", stamp, 1000.0)):
        bench._PMC, bench._PMC_NOTE = None, {}
        json.dump(dict(body, lib_stamp=recorded), open(tmp_path / "profiles" / "r06_pmc_traffic_scene5.json", "w"))
        assert bench.pmc_traffic("scene5", "heal::k_conv1x1<") == want, tag
        note = bench._PMC_NOTE["scene5"]
        assert ("refused" in note) == (want is None), note
    bench._KSTATS.clear()
    open(tmp_path / "profiles" / "r06_kernel_stats_scene5.csv", "w").write(
        '"Name","Calls","TotalDurationNs","AverageNs"\n"void heal::k_conv1x1<64, 64, 32, 1>(float const*)",10,300000,30000\n')
    open(tmp_path / "profiles" / "r06_kernel_stats_scene5.stamp", "w").write("0" * 64)
    assert bench.rocprof_mean_us("scene5", "void heal::k_conv1x1<") is None
    bench._KSTATS.clear()
    open(tmp_path / "profiles" / "r06_kernel_stats_scene5.stamp", "w").write(stamp)
    assert bench.rocprof_mean_us("scene5", "void heal::k_conv1x1<") == 30.0


def test_winograd_split_k_policy_never_leaves_an_empty_split(monkeypatch):
    """ops.conv3x3_winograd_ksplit (host logic of heal_conv3x3_winograd_splitk): splits only small grids with a deep reduction, and what it
    returns always satisfies the C entry point's contract -- 2 <= ksplit <= chunks and (ksplit - 1) * ceil(chunks / ksplit) < chunks."""
    from heal_amd import ops
    monkeypatch.delenv("HEAL_C3_KSPLIT", raising=False)
    assert ops.conv3x3_winograd_ksplit(4, 432, 512, 24, 32, 4) == 4      # the camera Up block: 192 blocks of 54 chunks
    assert ops.conv3x3_winograd_ksplit(4, 512, 512, 48, 64, 4) == 1      # 768 blocks: the grid fills the chip
    assert ops.conv3x3_winograd_ksplit(1, 64, 64, 16, 16, 4) == 1        # 8 chunks: too shallow
    assert ops.conv3x3_winograd_ksplit(1, 512, 64, 15, 15, 4) == 1       # H * W % 4 != 0: the float4 reduce does not apply
    rng = np.random.default_rng(0)
    for _ in range(300):
        n, cin, cout = int(rng.integers(1, 9)), int(rng.integers(1, 130)) * 8, int(rng.integers(1, 9)) * 64
        H, W, waves = int(rng.integers(1, 40)) * 2, int(rng.integers(1, 40)) * 2, int(rng.choice([4, 8]))
        for env in (None, "2", "3", "5", "64", "1000"):
            if env is None:
                monkeypatch.delenv("HEAL_C3_KSPLIT", raising=False)
            else:
                monkeypatch.setenv("HEAL_C3_KSPLIT", env)
            ks, chunks = ops.conv3x3_winograd_ksplit(n, cin, cout, H, W, waves), (cin + 7) // 8
            assert ks == 1 or (2 <= ks <= chunks and (ks - 1) * -(-chunks // ks) < chunks), (n, cin, cout, H, W, waves, env, ks)
