"""GPU twin of tests/test_train_path.py: the gradient path on the device, against the inference operators and
through a training step whose LiDAR input goes through the K1 voxeliser."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.test_train_path import hetero_small, synthetic_targets

pytestmark = [pytest.mark.gpu, pytest.mark.grad]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def in_fresh_process(test_name):
    """Run one test of this file in its own interpreter.  The two tests that call backward() go through MIOpen's backward
    convolutions (torch autograd on the gradient path; none of this repo's kernels run in them).  At the end of the full
    `pytest -m gpu` session -- ~230 tests, a dozen captured graphs and their pools behind it -- one of those library kernels hit
    `Memory access fault by GPU node` (an address on a 2 MiB segment boundary) in two consecutive full runs, while the same
    tests pass alone and after any single test file.  A fault aborts the whole pytest process, so they are isolated: a fresh
    process has a fresh allocator layout, and a fault there fails ONE test instead of killing the session."""
    if os.environ.get("HEAL_TRAIN_TEST_INPROC") == "1":
        return False
    res = subprocess.run([sys.executable, "-m", "pytest", f"{os.path.abspath(__file__)}::{test_name}", "-q", "-m", "gpu",
                          "-p", "no:cacheprovider"], cwd=ROOT, capture_output=True, text=True, timeout=900,
                         env={**os.environ, "HEAL_TRAIN_TEST_INPROC": "1", "PYTHONPATH": ROOT})
    assert res.returncode == 0 and " passed" in res.stdout, res.stdout[-3000:] + res.stderr[-3000:]
    return True


def test_gradient_path_equals_inference_operators():
    """Same model, same inputs, eval mode: gradients enabled -> torch-operator path, torch.no_grad() -> HIP operators.
    Heads and occupancy maps agree to 1e-3 of their scale (the north-star tolerance); the inference outputs carry no graph."""
    _, model, data, _ = hetero_small("cuda")
    model.eval()
    ref = model(data)
    assert ref["cls_preds"].requires_grad
    with torch.no_grad():
        got = model(data)
    assert not got["cls_preds"].requires_grad
    for k in ("cls_preds", "reg_preds", "dir_preds"):
        err = float((got[k] - ref[k]).abs().max() / ref[k].abs().max())
        assert err < 1e-3, (k, err)
    for i in range(3):
        a, b = got["occ_single_list"][i], ref["occ_single_list"][i]
        assert float((a - b).abs().max() / b.abs().max()) < 1e-3, i


def test_training_steps_on_the_device():
    """train.py's iteration on the GPU: forward (gradient path), pyramid loss, backward, Adam -- finite and decreasing."""
    if in_fresh_process("test_training_steps_on_the_device"):
        return
    from heal_amd.opencood.tools.train_utils import create_loss
    hypes, model, data, _ = hetero_small("cuda")
    model.train()
    criterion = create_loss(hypes)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    losses = []
    for _step in range(5):
        opt.zero_grad()
        out = model(data)
        tgt = synthetic_targets(out, device="cuda")
        n_agents = out["occ_single_list"][0].shape[0]
        single = {k: v.expand(n_agents, *v.shape[1:]).contiguous() for k, v in tgt.items()}
        loss = criterion(out, tgt) + criterion(out, single, suffix="_single")
        assert bool(torch.isfinite(loss))
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses


def test_training_from_raw_point_clouds():
    """LiDAR agents given as device point clouds: K1 voxelises (no gradient into the points), the PFN / scatter / backbone /
    fusion run on the gradient path; the canvas equals the inference operator's (K2) for the same clouds."""
    if in_fresh_process("test_training_from_raw_point_clouds"):
        return
    from heal_amd import configs
    from heal_amd.opencood.tools.train_utils import create_model
    from heal_amd.pipeline import Scene
    from tests.golden.detfill import fill_module
    hypes = configs.lidar_pyramid([-25.6, -25.6, -3, 25.6, 25.6, 1])
    model = fill_module(create_model(hypes)).cuda()
    scene = Scene(2, seed=3, device="cuda:0", modalities=["m1", "m1"])
    data = scene.model_input()
    model.eval()
    enc = model.encoder_m1
    canvas_g = enc(data, "m1")                      # gradient path (parameters require grad)
    with torch.no_grad():
        canvas_i = enc(data, "m1")                  # K1 + K2 (round 6: ops.PillarBEV when the backbone reads the pillars itself)
        canvas_i = canvas_i.dense() if hasattr(canvas_i, "dense") else canvas_i
    assert canvas_g.requires_grad and not canvas_i.requires_grad
    assert float((canvas_g - canvas_i).abs().max() / canvas_i.abs().max()) < 1e-4
    model.train()
    out = model(data)
    loss = sum(out[k].square().mean() for k in ("cls_preds", "reg_preds", "dir_preds"))
    loss.backward()
    for name, p in model.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name


@pytest.mark.parametrize("pitch", [0.0, 0.12])
def test_lift_pool_backward_kernel_vs_torch_autograd(pitch):
    """K4 backward (heal_bev_pool_backward behind LiftSplatShoot's autograd Function) against torch's autograd of the
    reference composition (get_geometry + softmax + outer product + index_add): same forward, gradients of the depth logits
    and of the image features within 1e-4 of their scale.  Level and pitched rigs (a pitched camera spreads a column over
    several cells).  The two paths evaluate the frustum geometry in a different fp32 order, so a point that sits on a cell
    boundary to the last bit may be binned differently: at most a handful of pixels may differ, and they are counted."""
    from heal_amd import configs, synth
    from heal_amd.opencood.models.heter_encoders import LiftSplatShoot, _LiftPool
    args = configs.heal_heter(("m2",), [-25.6, -25.6, -3, 25.6, 25.6, 1], cam_bound=12.8, cam_dims={"m2": (96, 128)})
    enc = LiftSplatShoot(args["model"]["args"]["m2"]["encoder_args"]).cuda()
    H, W = 96, 128
    rig = synth.camera_rig(7, 4, H, W)
    inp = {k: torch.from_numpy(v)[None].cuda() for k, v in rig.items()}
    if pitch:
        c, s_ = float(np.cos(pitch)), float(np.sin(pitch))
        R = torch.tensor([[1, 0, 0], [0, c, -s_], [0, s_, c]], dtype=inp["rots"].dtype, device="cuda")
        inp["rots"] = inp["rots"] @ R
    g = torch.Generator().manual_seed(3)
    D, C, fH, fW = enc.D, enc.camC, H // enc.downsample, W // enc.downsample
    logit = torch.randn((4, D, fH, fW), generator=g).cuda().requires_grad_(True)
    feat = torch.randn((4, C, fH, fW), generator=g).cuda().requires_grad_(True)
    ref = enc.lift_pool_autograd(logit, feat, inp, 1, 4)
    wgt = torch.randn(ref.shape, generator=g).cuda()
    (ref * wgt).sum().backward()
    g_logit_ref, g_feat_ref = logit.grad.clone(), feat.grad.clone()
    logit.grad = feat.grad = None
    with torch.no_grad():
        cam = enc.camera_matrices(inp["rots"], inp["trans"], inp["intrins"], inp["post_rots"], inp["post_trans"])
    got = _LiftPool.apply(logit, feat, enc.frustum(logit.device), cam, 1, 4, enc.dx_host, enc.bx_host, enc.nx_host)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-4
    (got * wgt).sum().backward()
    for name, a, b in (("logit", logit.grad, g_logit_ref), ("feat", feat.grad, g_feat_ref)):
        err = (a - b).abs() / b.abs().max()
        bad = int((err.amax(dim=1) > 1e-4).sum())            # pixels (camera, v, u) with any channel / bin off
        assert bad <= 4, (name, bad, float(err.max()))
    assert bool(torch.isfinite(logit.grad).all()) and bool(torch.isfinite(feat.grad).all())
    # finite differences in FLOAT64 with the ORACLE's cell assignment (oracle_np.lss_geometry + the trunc-to-cell / bounds rule of
    # oracle_np.bev_pool; the cell of every frustum point is fixed, so the pooled map is smooth in logits and features):
    # directional derivative along a random direction against <device gradient, direction> (the reference above
    # shares the device and fp32 with the kernel under test)
    from oracle import oracle_np as O
    rigs = {k: v.detach().cpu().numpy() for k, v in inp.items()}
    fr = O.create_frustum(enc.data_aug_conf["final_dim"], enc.downsample, enc.grid_conf["ddiscr"], enc.grid_conf["mode"])
    dx, bx, nx = O.gen_dx_bx(enc.grid_conf["xbound"], enc.grid_conf["ybound"], enc.grid_conf["zbound"])
    geom = O.lss_geometry(fr, rigs["rots"], rigs["trans"], rigs["intrins"], rigs["post_rots"], rigs["post_trans"])   # [1,4,D,fH,fW,3]
    cell = np.trunc((np.asarray(geom, np.float32) - (bx - dx / np.float32(2.0))) / dx).astype(np.int64)[0]            # [4,D,fH,fW,3]
    ok = ((cell >= 0) & (cell < np.asarray(nx)[None, None, None, None, :])).all(-1)
    w64 = wgt.double().cpu().numpy()[0]                                   # [C * nz, ny, nx], nz = 1
    assert int(nx[2]) == 1
    G = np.where(ok[..., None], w64[:, np.clip(cell[..., 1], 0, nx[1] - 1), np.clip(cell[..., 0], 0, nx[0] - 1)]
                 .transpose(1, 2, 3, 4, 0), 0.0)                          # weight of the point's cell per channel [4,D,fH,fW,C]

    def value(lg, ft):
        e = np.exp(lg - lg.max(1, keepdims=True)); p = e / e.sum(1, keepdims=True)                  # [4,D,fH,fW]
        return float(np.einsum("ndvu,ncvu,ndvuc->", p, ft, G))
    l0, f0 = logit.detach().double().cpu().numpy(), feat.detach().double().cpu().numpy()
    rng = np.random.default_rng(1)
    dl, df = rng.standard_normal(l0.shape), rng.standard_normal(f0.shape)
    eps = 1e-6
    fd = (value(l0 + eps * dl, f0 + eps * df) - value(l0 - eps * dl, f0 - eps * df)) / (2 * eps)
    an = float((logit.grad.double().cpu().numpy() * dl).sum() + (feat.grad.double().cpu().numpy() * df).sum())
    assert abs(fd - an) <= 1e-3 * max(abs(fd), 1.0), (fd, an)


@pytest.mark.parametrize("n,f64,with_crop", [(5, True, False), (3, False, True), (1, True, False)])
def test_warp_fuse_backward_kernel_vs_torch_autograd(n, f64, with_crop):
    """K5 backward (heal_warp_fuse_backward behind the autograd Function of weighted_fuse) against torch's autograd of the
    reference composition (affine_grid + grid_sample x 2, masked softmax over agents, weighted sum): gradients of the agents'
    maps and of the occupancy logits within 1e-4 of their scale; rotated / translated poses, an agent out of view, crops."""
    from heal_amd.opencood.models.fuse_modules.pyramid_fuse import weighted_fuse, weighted_fuse_autograd
    g = np.random.default_rng(50 + n)
    C, H, W = 24, 40, 36
    x = torch.from_numpy(g.standard_normal((n, C, H, W)).astype(np.float32)).cuda().requires_grad_(True)
    occ = torch.from_numpy((g.standard_normal((n, 1, H, W)) * 2).astype(np.float32)).cuda().requires_grad_(True)
    rows = np.zeros((1, n, n, 2, 3))
    for a in range(n):
        th = 0.0 if a == 0 else g.uniform(-np.pi, np.pi)
        rows[0, 0, a] = [[np.cos(th), -np.sin(th), 0.0 if a == 0 else g.uniform(-0.7, 0.7)],
                         [np.sin(th), np.cos(th), 0.0 if a == 0 else g.uniform(-0.7, 0.7)]]
    if n > 2:
        rows[0, 0, n - 1, :, 2] = [4.0, -3.0]
    if not f64:
        rows = rows.astype(np.float32)
    crops = [None if a % 2 == 0 else (H // 4, 3 * H // 4, W // 5, 4 * W // 5) for a in range(n)] if with_crop else None
    wgt = torch.from_numpy(g.standard_normal((1, C, H, W)).astype(np.float32)).cuda()
    ref = weighted_fuse_autograd(x, occ, [n], rows, crops)
    (ref * wgt).sum().backward()
    gx_ref, go_ref = x.grad.clone(), occ.grad.clone()
    x.grad = occ.grad = None
    got = weighted_fuse(x, occ, [n], rows, grid_f64=f64, crops=crops)
    assert got.requires_grad and float((got - ref).abs().max() / ref.abs().max()) < 1e-4
    (got * wgt).sum().backward()
    assert float((x.grad - gx_ref).abs().max() / gx_ref.abs().max()) < 1e-4
    # one agent: the softmax is constant, the logits get no gradient at all (in either implementation)
    assert float((occ.grad - go_ref).abs().max()) <= 1e-4 * float(go_ref.abs().max())
    # independent of the device and of fp32: the same composition differentiated on the CPU in FLOAT64 (the check
    # above shares the device and the precision with the kernel under test)
    x64 = x.detach().double().cpu().requires_grad_(True)
    o64 = occ.detach().double().cpu().requires_grad_(True)
    ref64 = weighted_fuse_autograd(x64, o64, [n], rows.astype(np.float64), crops)
    (ref64 * wgt.double().cpu()).sum().backward()
    assert float((got.detach().double().cpu() - ref64.detach()).abs().max() / ref64.abs().max()) < 1e-4
    assert float((x.grad.double().cpu() - x64.grad).abs().max() / x64.grad.abs().max()) < 1e-4
    assert float((occ.grad.double().cpu() - o64.grad).abs().max()) <= 1e-4 * max(float(o64.grad.abs().max()), 1e-30)


def test_second_gradient_path_equals_sparse_kernels(monkeypatch):
    """SECOND on the gradient path against the inference path (K3: rulebooks + gather-GEMM kernels) on a small grid -- the same BEV
    map within 1e-3 of its scale, the same set of non-zero cells -- and the SPARSE backward (heal_sp_conv on the transposed
    rulebook for the feature gradients, per-tap gathered matrix products for the weight gradients; round 3) against torch's
    autograd of the dense masked conv3d restatement of the same encoder: every parameter gradient within 1e-3 of its scale."""
    from heal_amd import configs
    from heal_amd.opencood.models.heter_encoders import SECOND
    from tests.golden.detfill import fill_module
    rng = [-6.4, -6.4, -3, 6.4, 6.4, 1]
    enc = fill_module(SECOND(configs._second_modality(rng)["encoder_args"])).cuda().train()
    for mod in enc.modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
            mod.eval()
    g = np.random.default_rng(9)
    B, D, H, W = 2, 40, 128, 128
    flat = g.choice(B * D * H * W, size=3000, replace=False)
    coords = np.stack(np.unravel_index(flat, (B, D, H, W)), 1).astype(np.int32)
    coords = coords[np.argsort(coords[:, 0], kind="stable")]
    num = g.integers(1, 6, size=coords.shape[0]).astype(np.int32)
    voxels = g.standard_normal((coords.shape[0], 5, 4)).astype(np.float32) * (np.arange(5)[None, :, None] < num[:, None, None])
    data = {"inputs_m3": {"voxel_features": torch.from_numpy(voxels).cuda(), "voxel_coords": torch.from_numpy(coords).cuda(),
                          "voxel_num_points": torch.from_numpy(num).cuda(), "n_agents": B}}
    grads = {}
    outs = {}
    wgt = None
    for mode in ("dense", "sparse"):
        monkeypatch.setenv("HEAL_SP_GRAD", mode)
        enc.zero_grad(set_to_none=True)
        out = enc(data, "m3")
        assert out.requires_grad
        if wgt is None:
            wgt = torch.from_numpy(g.standard_normal(tuple(out.shape)).astype(np.float32)).cuda()
        (out * wgt).sum().backward()
        outs[mode] = out.detach()
        grads[mode] = {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}
    assert float((outs["sparse"] - outs["dense"]).abs().max() / outs["dense"].abs().max()) < 1e-3
    assert set(grads["sparse"]) == set(grads["dense"]) and any("conv" in n for n in grads["dense"])
    for n, gd in grads["dense"].items():
        assert float((grads["sparse"][n] - gd).abs().max()) <= 1e-3 * float(gd.abs().max()) + 1e-7, n
    ref = outs["sparse"]
    with torch.no_grad():
        got = enc.eval()(data, "m3")
    assert got.shape == ref.shape
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-3
    assert int(((got != 0) != (ref != 0)).sum()) == 0


@pytest.mark.parametrize("training,P", [(True, 32), (False, 32), (True, 5)])
def test_pfn_forward_backward_kernels_vs_torch(training, P):
    """K2's training path on the device (heal_pfn_moments / heal_pfn_features / heal_pfn_backward behind _PFNFunction): pillar
    features, running-statistics update and the gradients of the Linear weight and the BatchNorm affine against the torch
    composition it replaces (PillarVFE.pillar_features) -- evaluated on the CPU in FLOAT64 on the same pillars, so the reference
    shares neither the device nor the precision with the kernels.  Batch statistics (training mode: all M x P rows, zeroed
    padding rows included) and running statistics (eval mode with trainable parameters); pillars that are full, partly filled
    and single-point."""
    import copy
    from heal_amd import ops, synth
    from heal_amd.opencood.models.sub_modules.pillar_vfe import PillarVFE
    pts = torch.from_numpy(synth.lidar_frame(5)).cuda()
    R, V = [-102.4, -102.4, -3.5, 102.4, 102.4, 1.5], [0.4, 0.4, 5.0]
    v, c, n = ops.voxelize(pts, R, V, P, 32000)
    assert int((n == P).sum()) > 0 and int((n == 1).sum()) > 0 and int(((n > 1) & (n < P)).sum()) > 0
    vfe = PillarVFE({"use_norm": True, "with_distance": False, "use_absolute_xyz": True, "num_filters": [64]}, 4, V, R).cuda()
    torch.manual_seed(3)
    with torch.no_grad():
        vfe.pfn_layers[0].linear.weight.normal_(0, 0.5)
        vfe.pfn_layers[0].norm.weight.uniform_(0.5, 1.5)
        vfe.pfn_layers[0].norm.bias.normal_(0, 0.3)
        vfe.pfn_layers[0].norm.running_mean.normal_(0, 0.5)
        vfe.pfn_layers[0].norm.running_var.uniform_(0.5, 2.0)
    ref = copy.deepcopy(vfe).double().cpu()
    vfe.train(training)
    ref.train(training)
    gout = torch.randn((v.shape[0], 64), generator=torch.Generator().manual_seed(1))
    out = vfe.pillar_features_kernels(v, c, n)
    (out * gout.cuda()).sum().backward()
    want = ref.pillar_features(v.double().cpu(), c.cpu(), n.cpu())
    (want * gout.double()).sum().backward()
    scale = float(want.abs().max())
    assert float((out.double().cpu() - want).abs().max()) / scale < 1e-5
    for name in ("linear.weight", "norm.weight", "norm.bias"):
        got = dict(vfe.pfn_layers[0].named_parameters())[name].grad.double().cpu()
        exp = dict(ref.pfn_layers[0].named_parameters())[name].grad
        err = float((got - exp).abs().max() / (exp.abs().max() + 1e-30))
        assert err < 2e-4, (name, err)
    bn, rbn = vfe.pfn_layers[0].norm, ref.pfn_layers[0].norm
    assert float((bn.running_mean.double().cpu() - rbn.running_mean).abs().max()) < 1e-5
    assert float((bn.running_var.double().cpu() - rbn.running_var).abs().max() / rbn.running_var.abs().max()) < 1e-5
    assert int(bn.num_batches_tracked) == int(rbn.num_batches_tracked)


@pytest.mark.parametrize("cin,cout,ksize", [(4, 16, (3, 3, 3)), (16, 32, (3, 3, 3)), (32, 32, (3, 3, 3)), (64, 64, (3, 3, 3)),
                                             (64, 64, (3, 1, 1)), (32, 64, (3, 3, 3))])
def test_sparse_weight_gradient_kernel_vs_float64(cin, cout, ksize):
    """heal_sp_wgrad (pair-compacted gather + MFMA over the pair index, one partial per 2048 output rows) against the per-tap
    gather + matmul it replaces, evaluated in float64: submanifold and strided rulebooks of a real sweep (several chunks, taps
    with few and with many pairs), every channel combination of the encoder; two runs bit-identical."""
    from heal_amd import ops, synth
    R = [-102.4, -102.4, -3.0, 102.4, 102.4, 1.0]
    pts = torch.from_numpy(synth.lidar_frame(77)).cuda()
    v, c, n = ops.voxelize(pts, R, [0.1, 0.1, 0.1], 5, 70000)
    x = ops.SparseTensor.from_unsorted(torch.randn((v.shape[0], cin), device="cuda"), c.int().contiguous(), [41, 2048, 2048], 1)
    stride = (1, 1, 1) if ksize == (3, 3, 3) and cin == cout else (2, 2, 2) if ksize == (3, 3, 3) else (2, 1, 1)
    pad = tuple(k // 2 for k in ksize) if stride == (1, 1, 1) else ((1, 1, 1) if ksize == (3, 3, 3) else (0, 0, 0))
    if stride == (1, 1, 1):
        oi, oshape = x.indices, x.spatial_shape
    else:
        oi, oshape, _, _ = x.out_sites_ex(ksize, stride, pad)
    nbr = x.neighbors(oi, oshape, ksize, stride, pad)
    assert nbr.shape[0] > 2 * 2048
    g = torch.randn((nbr.shape[0], cout), device="cuda")
    got = ops.sp_wgrad(x.features, g, nbr)
    ref = torch.zeros((nbr.shape[1], cin, cout), dtype=torch.float64, device="cuda")
    for t in range(nbr.shape[1]):
        o = (nbr[:, t] >= 0).nonzero(as_tuple=True)[0]
        if o.numel():
            ref[t] = x.features.double().index_select(0, nbr[o, t].long()).t() @ g.double().index_select(0, o)
    err = float((got.double() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err
    assert torch.equal(ops.sp_wgrad(x.features, g, nbr), got)


def test_inference_operator_refuses_autograd_activations():
    """An activation with autograd history must never reach a HIP operator silently (its result would drop out of the graph)."""
    from heal_amd import _capi, ops
    x = torch.randn((1, 64, 8, 8), device="cuda", requires_grad=True) * 2.0
    w = torch.randn((64, 64, 1, 1), device="cuda")
    with pytest.raises(_capi.HealAmdError, match="autograd history"):
        ops.conv1x1(x, w)
    with torch.no_grad():
        ops.conv1x1(x, w)   # fine without a graph


@pytest.mark.parametrize("heads,L,agent_major,out_rows,masked", [(8, 5, True, 5, False), (1, 3, False, 1, False),
                                                                  (16, 8, True, 8, True), (4, 2, False, 2, False),
                                                                  (8, 1, True, 1, False)])
def test_agent_attention_backward_kernel_vs_float64_autograd(heads, L, agent_major, out_rows, masked):
    """K6 backward (heal_agent_attention_backward behind ops.AgentAttention) against autograd of the reference's composition
    (hmsa.py:131-146: scores -> masked softmax over agents -> weighted sum; fusion_in_one.py:37-44 for one head and the ego row)
    differentiated on the CPU in FLOAT64: output and the three gradients within 1e-5 of their scale."""
    from heal_amd import ops
    g = np.random.default_rng(7 * heads + L)
    P, C = 203, 256
    d = C // heads
    scale = d ** -0.5
    shape = (L, P, C) if agent_major else (P, L, C)
    q, k, v = (torch.from_numpy((g.standard_normal(shape) * 1.5).astype(np.float32)).cuda().requires_grad_(True) for _ in range(3))
    mask = None
    if masked:
        mask_np = np.ones(L, np.int32)
        mask_np[-2:] = 0
        mask = torch.from_numpy(mask_np).cuda()
    oshape = (out_rows, P, C) if agent_major else (P, out_rows, C)
    wgt = torch.from_numpy(g.standard_normal(oshape).astype(np.float32)).cuda()
    if mask is None:
        out = ops.AgentAttention.apply(q, k, v, heads, scale, out_rows, agent_major)
        assert out.requires_grad
        (out * wgt).sum().backward()
        got = (out.detach(), q.grad, k.grad, v.grad)
    else:   # the masked form has no autograd wrapper (HEAL never pads under training): forward + backward entry points directly
        with torch.no_grad():
            out = ops.agent_attention(q.detach(), k.detach(), v.detach(), heads, scale, key_mask=mask, out_rows=out_rows,
                                      agent_major=agent_major)
            got = (out,) + ops.agent_attention_backward(q.detach(), k.detach(), v.detach(), wgt, heads, scale, key_mask=mask,
                                                        agent_major=agent_major)

    def pm(t):     # -> [P, L, heads, d] float64 on the host
        t = t.detach().double().cpu()
        return (t.permute(1, 0, 2) if agent_major else t).reshape(P, -1, heads, d)
    q64, k64, v64 = (pm(t).requires_grad_(True) for t in (q, k, v))
    att = torch.einsum("pihd,pjhd->phij", q64, k64) * scale
    if masked:
        att = att.masked_fill(torch.from_numpy(mask_np == 0)[None, None, None, :], float("-inf"))
    ref = torch.einsum("phij,pjhd->pihd", att.softmax(-1), v64)[:, :out_rows]
    (ref * pm(wgt)).sum().backward()
    for name, a, b in zip(("out", "grad_q", "grad_k", "grad_v"), got, (ref, q64.grad, k64.grad, v64.grad)):
        b = b.detach()
        # one agent: the softmax is constant, q and k get no gradient at all (0 against 0)
        err = float((pm(a) - b).abs().max() / max(float(b.abs().max()), 1e-30))
        assert err < 1e-5, (name, err)
    if out_rows < L:
        assert float(got[1].reshape(shape)[(slice(out_rows, None),) if agent_major else (slice(None), slice(out_rows, None))].abs().max()) == 0.0


def test_attention_modules_gradient_path_kernel_equals_torch(monkeypatch):
    """HGTCavAttention and AttFusion under autograd: the device path (K6 forward + backward kernels) and the torch composition
    (HEAL_ATTN_GRAD=torch) give the same outputs and parameter / input gradients."""
    from heal_amd.opencood.models.fuse_modules.fusion_in_one import AttFusion
    from heal_amd.opencood.models.sub_modules.v2xvit_basic import HGTCavAttention
    torch.manual_seed(11)
    att = HGTCavAttention(256, heads=8, dim_head=32, dropout=0.0).cuda().train()
    x = torch.randn(3, 12, 16, 256, device="cuda", requires_grad=True)
    fus = AttFusion(256).cuda()
    e = torch.randn(4, 256, 10, 12, device="cuda", requires_grad=True)
    res = {}
    for mode in ("kernel", "torch"):
        monkeypatch.setenv("HEAL_ATTN_GRAD", mode)
        att.zero_grad(); x.grad = None; e.grad = None
        y = att(x)
        (y * y).sum().backward()
        z = fus.fuse_warped(e)
        z.square().sum().backward()
        res[mode] = [y.detach(), x.grad.clone(), att.relation_att.grad.clone(), att.q_linears[0].weight.grad.clone(),
                     att.v_linears[0].weight.grad.clone(), z.detach(), e.grad.clone()]
    for a, b in zip(res["kernel"], res["torch"]):
        assert float((a - b).abs().max() / b.abs().max()) < 2e-4


@pytest.mark.parametrize("ws,d,m,use_bias", [(4, 16, 16, True), (8, 32, 8, True), (16, 64, 4, True), (8, 32, 8, False)])
def test_window_attention_backward_kernel_vs_float64_autograd(ws, d, m, use_bias):
    """K6b backward (heal_window_attention_backward behind ops.WindowAttention; opt-in in the modules, HEAL_WATTN_GRAD=kernel)
    against autograd of the reference's composition (mswin.py:64-78: window re-layout, scores + position bias, softmax, weighted
    sum) differentiated on the CPU in FLOAT64: output, grad_qkv and grad_bias within 2e-5 of their scale."""
    from heal_amd import ops
    g = np.random.default_rng(100 + ws)
    L, H, W = 2, 16, 32
    C = m * d
    T = ws * ws
    scale = d ** -0.5
    qkv = torch.from_numpy((g.standard_normal((L, H, W, 3 * C)) * 0.8).astype(np.float32)).cuda().requires_grad_(True)
    bias = torch.from_numpy(g.standard_normal((T, T)).astype(np.float32)).cuda().requires_grad_(True) if use_bias else None
    wgt = torch.from_numpy(g.standard_normal((L, H, W, C)).astype(np.float32)).cuda()
    out = ops.WindowAttention.apply(qkv, bias, m, d, ws, scale)
    (out * wgt).sum().backward()

    q64 = qkv.detach().double().cpu().requires_grad_(True)
    b64 = bias.detach().double().cpu().requires_grad_(True) if use_bias else None
    nh, nw = H // ws, W // ws
    t = q64.view(L, nh, ws, nw, ws, 3, m, d).permute(5, 0, 6, 1, 3, 2, 4, 7).reshape(3, L * m * nh * nw, T, d)
    dots = torch.matmul(t[0], t[1].transpose(1, 2)) * scale
    if use_bias:
        dots = dots + b64[None]
    ref = torch.matmul(dots.softmax(-1), t[2]).view(L, m, nh, nw, ws, ws, d).permute(0, 2, 4, 3, 5, 1, 6).reshape(L, H, W, C)
    (ref * wgt.double().cpu()).sum().backward()
    pairs = [("out", out.detach(), ref.detach()), ("grad_qkv", qkv.grad, q64.grad)]
    if use_bias:
        pairs.append(("grad_bias", bias.grad, b64.grad))
    for name, a, b in pairs:
        err = float((a.double().cpu() - b).abs().max() / b.abs().max())
        assert err < 2e-5, (name, err)


def test_window_attention_module_gradient_path_kernel_equals_torch(monkeypatch):
    """BaseWindowAttention under autograd: the opt-in device path (HEAL_WATTN_GRAD=kernel) and the library composition give the
    same output and the same gradients of the input, the projections and the relative-position table."""
    from heal_amd.opencood.models.sub_modules.v2xvit_basic import BaseWindowAttention
    torch.manual_seed(13)
    att = BaseWindowAttention(256, heads=8, dim_head=32, drop_out=0.0, window_size=8, relative_pos_embedding=True).cuda().train()
    x = torch.randn(2, 16, 24, 256, device="cuda", requires_grad=True)
    res = {}
    for mode in ("kernel", "torch"):
        monkeypatch.setenv("HEAL_WATTN_GRAD", mode)
        att.zero_grad(); x.grad = None
        y = att(x)
        (y * y).sum().backward()
        res[mode] = [y.detach(), x.grad.clone(), att.to_qkv.weight.grad.clone(), att.pos_embedding.grad.clone(),
                     att.to_out[0].weight.grad.clone()]
    for a, b in zip(res["kernel"], res["torch"]):
        assert float((a - b).abs().max() / b.abs().max()) < 2e-4
