"""oracle_np -- TEST INFRASTRUCTURE ONLY.

numpy restatement (fp32 unless noted, same operation order as the reference's torch code) of the
floating-point parts of HEAL's perception hot path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; heal_amd/ (the product) never does.

Every function cites the reference file:line it restates (paths relative to the reference root).
Parity: the functions marked [pinned] are checked against outputs of the imported reference
(tests/golden/*.npz, produced by tests/golden/gen_golden.py in the build container);
[unpinned] ones restate third-party arithmetic that is absent from the reference tree and are
pinned by analytic known-answer tests only (see oracle/oracle_ref.c header and DESIGN.md).
"""
import math

import numpy as np

from . import cref

F32 = np.float32


# --------------------------------------------------------------------------------------------------
# K2: PillarVFE + PFNLayer + PointPillarScatter                                           [pinned]
# --------------------------------------------------------------------------------------------------
def pillar_features(voxels, coords, num_points, voxel_size, lidar_range):
    """pillar_vfe.py:105-149 -- decorate points: [xyzI, xyz - mean, xyz - pillar centre], mask pads.

    voxels [M,P,4] f32, coords [M,4] (b,z,y,x) int, num_points [M] int -> [M,P,10] f32."""
    voxels = np.asarray(voxels, F32)
    M, P, _ = voxels.shape
    vx, vy, vz = (F32(v) for v in voxel_size)
    # offsets are python floats in the reference (pillar_vfe.py:89-91), used as fp32 scalars
    x_off = F32(voxel_size[0] / 2 + lidar_range[0])
    y_off = F32(voxel_size[1] / 2 + lidar_range[1])
    z_off = F32(voxel_size[2] / 2 + lidar_range[2])
    n = np.asarray(num_points).astype(F32).reshape(-1, 1, 1)
    mean = voxels[:, :, :3].sum(axis=1, keepdims=True, dtype=F32) / n
    f_cluster = voxels[:, :, :3] - mean
    f_center = np.zeros_like(voxels[:, :, :3])
    c = np.asarray(coords)
    f_center[:, :, 0] = voxels[:, :, 0] - (c[:, 3].astype(F32)[:, None] * vx + x_off)
    f_center[:, :, 1] = voxels[:, :, 1] - (c[:, 2].astype(F32)[:, None] * vy + y_off)
    f_center[:, :, 2] = voxels[:, :, 2] - (c[:, 1].astype(F32)[:, None] * vz + z_off)
    feats = np.concatenate([voxels, f_cluster, f_center], axis=-1)
    mask = (np.asarray(num_points).astype(np.int64)[:, None] > np.arange(P)[None, :]).astype(F32)
    return feats * mask[:, :, None]


def pfn_layer(feats, weight, bn_gamma, bn_beta, bn_mean, bn_var, eps=1e-3):
    """pillar_vfe.py:31-53 (last layer): Linear(no bias) -> eval BatchNorm1d -> ReLU -> max over P
    (padded rows take part in the max).  feats [M,P,10] -> [M,C]."""
    x = feats.astype(F32) @ np.asarray(weight, F32).T  # [M,P,C]
    inv = (F32(1.0) / np.sqrt(np.asarray(bn_var, F32) + F32(eps))).astype(F32)
    x = (x - np.asarray(bn_mean, F32)) * inv * np.asarray(bn_gamma, F32) + np.asarray(bn_beta, F32)
    x = np.maximum(x, F32(0))
    return x.max(axis=1)


def scatter_to_canvas(pillars, coords, n_agents, ny, nx):
    """point_pillar_scatter.py:19-76 -- canvas[b, :, z + y*nx + x] = pillar; -> [n,C,ny,nx]."""
    C = pillars.shape[1]
    canvas = np.zeros((n_agents, C, ny * nx), F32)
    c = np.asarray(coords).astype(np.int64)
    for b in range(n_agents):
        m = c[:, 0] == b
        idx = c[m, 1] + c[m, 2] * nx + c[m, 3]
        canvas[b][:, idx] = pillars[m].T
    return canvas.reshape(n_agents, C, ny, nx)


def pfn_scatter(voxels, coords, num_points, weight, bn_gamma, bn_beta, bn_mean, bn_var, voxel_size,
                lidar_range, n_agents, ny, nx, eps=1e-3):
    f = pillar_features(voxels, coords, num_points, voxel_size, lidar_range)
    p = pfn_layer(f, weight, bn_gamma, bn_beta, bn_mean, bn_var, eps)
    return scatter_to_canvas(p, coords, n_agents, ny, nx), p


# --------------------------------------------------------------------------------------------------
# K5: normalize_pairwise_tfm + warp_affine_simple + weighted_fuse                          [pinned]
# --------------------------------------------------------------------------------------------------
def normalize_pairwise_tfm(pairwise_t_matrix, H, W, discrete_ratio, downsample_rate=1):
    """transformation_utils.py:68-92; keeps the input dtype (float64 when it comes from numpy)."""
    t = np.asarray(pairwise_t_matrix)
    a = t[:, :, :, [0, 1], :][:, :, :, :, [0, 1, 3]].copy()
    a[..., 0, 1] = a[..., 0, 1] * H / W
    a[..., 1, 0] = a[..., 1, 0] * W / H
    a[..., 0, 2] = a[..., 0, 2] / (downsample_rate * discrete_ratio * W) * 2
    a[..., 1, 2] = a[..., 1, 2] / (downsample_rate * discrete_ratio * H) * 2
    return a


def affine_grid(theta, H, W):
    """F.affine_grid(theta,[N,C,H,W],align_corners=False) (torch_transformation_utils.py:328-330).
    Computed in theta's dtype, as torch does.  theta [N,2,3] -> grid [N,H,W,2] (x,y)."""
    theta = np.asarray(theta)
    dt = theta.dtype
    xs = (np.linspace(-1, 1, W, dtype=dt) * dt.type(W - 1) / dt.type(W)) if W > 1 else np.zeros(1, dt)
    ys = (np.linspace(-1, 1, H, dtype=dt) * dt.type(H - 1) / dt.type(H)) if H > 1 else np.zeros(1, dt)
    gx = (theta[:, 0, 0, None, None] * xs[None, None, :] + theta[:, 0, 1, None, None] * ys[None, :, None]
          + theta[:, 0, 2, None, None])
    gy = (theta[:, 1, 0, None, None] * xs[None, None, :] + theta[:, 1, 1, None, None] * ys[None, :, None]
          + theta[:, 1, 2, None, None])
    return np.stack([gx, gy], axis=-1)


def grid_sample_bilinear(src, grid):
    """F.grid_sample(src, grid, mode='bilinear', padding_mode='zeros', align_corners=False).
    src [N,C,H,W] f32, grid [N,Ho,Wo,2] -> [N,C,Ho,Wo] f32 (torch_transformation_utils.py:332)."""
    src = np.asarray(src, F32)
    grid = np.asarray(grid).astype(F32)  # `.to(src)` cast
    N, C, H, W = src.shape
    x = grid[..., 0]
    y = grid[..., 1]
    ix = ((x + F32(1)) * F32(W) - F32(1)) / F32(2)
    iy = ((y + F32(1)) * F32(H) - F32(1)) / F32(2)
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    x1 = x0 + F32(1)
    y1 = y0 + F32(1)
    w_nw = (x1 - ix) * (y1 - iy)
    w_ne = (ix - x0) * (y1 - iy)
    w_sw = (x1 - ix) * (iy - y0)
    w_se = (ix - x0) * (iy - y0)
    out = np.zeros((N, C) + x.shape[1:], F32)
    for (xx, yy, ww) in ((x0, y0, w_nw), (x1, y0, w_ne), (x0, y1, w_sw), (x1, y1, w_se)):
        xi = xx.astype(np.int64)
        yi = yy.astype(np.int64)
        ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
        xi = np.clip(xi, 0, W - 1)
        yi = np.clip(yi, 0, H - 1)
        for n in range(N):
            v = src[n][:, yi[n], xi[n]]  # [C,Ho,Wo]
            out[n] += v * (ww[n] * ok[n]).astype(F32)[None]
    return out


def warp_affine_simple(src, M, dsize):
    return grid_sample_bilinear(src, affine_grid(M, dsize[0], dsize[1]))


def camera_crop_mask(n_agents, H, W, agent_modality_list, cam_crop_info):
    """pyramid_fuse.py:147-162 (eval mode): 1 inside the kept centre window of camera agents, 0
    outside; all ones for other agents.  Returns [n,1,H,W] f32."""
    mask = np.ones((n_agents, 1, H, W), F32)
    for cam_modality, info in cam_crop_info.items():
        crop_H = H / info[f"crop_ratio_H_{cam_modality}"] - 4
        crop_W = W / info[f"crop_ratio_W_{cam_modality}"] - 4
        start_h = int(H // 2 - crop_H // 2)
        end_h = int(H // 2 + crop_H // 2)
        start_w = int(W // 2 - crop_W // 2)
        end_w = int(W // 2 + crop_W // 2)
        for a, mod in enumerate(agent_modality_list):
            if mod == cam_modality:
                m = np.ones((1, H, W), F32)
                m[:, start_h:end_h, start_w:end_w] = 0
                mask[a] = 1 - m
    return mask


def weighted_fuse(x, score, affine_row):
    """pyramid_fuse.py:17-63 for ONE scene: x [n,C,H,W], score [n,1,H,W], affine_row [n,2,3] =
    affine_matrix[b][0,:n] -> [C,H,W]."""
    n, C, H, W = x.shape
    feat = warp_affine_simple(x, affine_row, (H, W))
    s = warp_affine_simple(score, affine_row, (H, W))
    s = np.where(s == 0, -np.inf, s).astype(F32)
    with np.errstate(invalid="ignore"):
        mx = s.max(axis=0, keepdims=True)
        e = np.exp(s - mx)
        p = e / e.sum(axis=0, keepdims=True)
    p = np.where(np.isnan(p), F32(0), p).astype(F32)
    return (feat * p).sum(axis=0, dtype=F32)


def occ_to_score(occ, crop_mask=None):
    """pyramid_fuse.py:145,162 -- sigmoid(occ) + 1e-4, times the camera crop mask."""
    s = (F32(1) / (F32(1) + np.exp(-np.asarray(occ, F32)))).astype(F32) + F32(1e-4)
    if crop_mask is not None:
        s = s * crop_mask
    return s.astype(F32)


# --------------------------------------------------------------------------------------------------
# K8: anchors, decode, filters, rotated NMS                   [decode pinned; NMS geometry unpinned]
# --------------------------------------------------------------------------------------------------
def generate_anchor_box(lidar_range, voxel_w, voxel_h, W, H, l, w, h, r_deg, feature_stride=2,
                        order="hwl"):
    """voxel_postprocessor.py:30-83 -> [H//fs, W//fs, A, 7] float64."""
    A = len(r_deg)
    r = [math.radians(e) for e in r_deg]
    x = np.linspace(lidar_range[0] + voxel_w, lidar_range[3] - voxel_w, W // feature_stride)
    y = np.linspace(lidar_range[1] + voxel_h, lidar_range[4] - voxel_h, H // feature_stride)
    cx, cy = np.meshgrid(x, y)
    cx = np.tile(cx[..., None], A)
    cy = np.tile(cy[..., None], A)
    cz = np.ones_like(cx) * -1.0
    ww = np.ones_like(cx) * w
    ll = np.ones_like(cx) * l
    hh = np.ones_like(cx) * h
    rr = np.ones_like(cx)
    for i in range(A):
        rr[..., i] = r[i]
    if order == "hwl":
        return np.stack([cx, cy, cz, hh, ww, ll, rr], axis=-1)
    return np.stack([cx, cy, cz, ll, hh, ww, rr], axis=-1)


def delta_to_boxes3d(deltas, anchors):
    """voxel_postprocessor.py:407-453.  deltas [1,7A,H,W] f32, anchors [H,W,A,7] -> [H*W*A,7] f32."""
    d = np.asarray(deltas, F32)
    N = d.shape[0]
    d = d.transpose(0, 2, 3, 1).reshape(N, -1, 7)[0]
    a = np.asarray(anchors).reshape(-1, 7).astype(F32)
    ad = np.sqrt(a[:, 4] ** 2 + a[:, 5] ** 2).astype(F32)
    b = np.zeros_like(d)
    b[:, 0] = d[:, 0] * ad + a[:, 0]
    b[:, 1] = d[:, 1] * ad + a[:, 1]
    b[:, 2] = d[:, 2] * a[:, 3] + a[:, 2]
    b[:, 3:6] = np.exp(d[:, 3:6]) * a[:, 3:6]
    b[:, 6] = d[:, 6] + a[:, 6]
    return b


def limit_period(val, offset, period):
    """common_utils.py:104-113 in fp32 (python scalars enter as fp32, like torch)."""
    val = np.asarray(val, F32)
    return (val - np.floor(val / F32(period) + F32(offset)) * F32(period)).astype(F32)


def boxes_to_corners_3d_hwl(boxes):
    """box_utils.py:152-204 with order='hwl'.  boxes [K,7] (x,y,z,h,w,l,yaw) f32 -> [K,8,3] f32."""
    b = np.asarray(boxes, F32)[:, [0, 1, 2, 5, 4, 3, 6]]  # -> x,y,z,l,w,h,yaw
    template = np.array([[1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, -1],
                         [1, -1, 1], [1, 1, 1], [-1, 1, 1], [-1, -1, 1]], F32) / F32(2)
    corners = b[:, None, 3:6] * template[None]
    cosa = np.cos(b[:, 6]).astype(F32)
    sina = np.sin(b[:, 6]).astype(F32)
    zeros = np.zeros_like(cosa)
    ones = np.ones_like(cosa)
    rot = np.stack([cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones], axis=1).reshape(-1, 3, 3)
    corners = np.matmul(corners, rot).astype(F32)  # common_utils.py:139-161
    return corners + b[:, None, 0:3]


def project_box3d(corners, T):
    """box_utils.py:278-316: T(4x4) @ [corners;1]."""
    c = np.asarray(corners, F32)
    T = np.asarray(T, F32)
    h = np.concatenate([c.transpose(0, 2, 1), np.ones((c.shape[0], 1, 8), F32)], axis=1)  # [K,4,8]
    p = np.matmul(T[None], h).astype(F32)
    return p[:, :3, :].transpose(0, 2, 1)


def decode_candidates(cls, reg, dirp, anchors, score_thr, dir_offset, num_bins, tfm):
    """voxel_postprocessor.py:295-380: sigmoid, decode, threshold mask, direction fix, corners,
    projection, large-box and abnormal-z filters.  Returns (corners [K,8,3] f32, scores [K] f32,
    anchor index [K]) in anchor order."""
    cls = np.asarray(cls, F32)
    prob = (F32(1) / (F32(1) + np.exp(-cls.transpose(0, 2, 3, 1)))).astype(F32).reshape(-1)
    boxes = delta_to_boxes3d(reg, anchors)
    mask = prob > F32(score_thr)
    idx = np.nonzero(mask)[0]
    boxes = boxes[mask]
    scores = prob[mask]
    if dirp is not None and len(boxes):
        dm = np.asarray(dirp, F32).transpose(0, 2, 3, 1).reshape(-1, num_bins)[mask]
        labels = np.argmax(dm, axis=-1)  # first max wins, as torch.max
        period = 2 * np.pi / num_bins
        dir_rot = limit_period(boxes[:, 6] - F32(dir_offset), 0, period)
        boxes[:, 6] = dir_rot + F32(dir_offset) + F32(period) * labels.astype(F32)
        boxes[:, 6] = limit_period(boxes[:, 6], 0.5, 2 * np.pi)
    if len(boxes) == 0:
        return np.zeros((0, 8, 3), F32), np.zeros((0,), F32), idx
    corners = project_box3d(boxes_to_corners_3d_hwl(boxes), tfm)
    # remove_large_pred_bbx (box_utils.py:840-869; its z_len term is computed from the y column and
    # only used as a truthy value, i.e. y_len != 0) and remove_bbx_abnormal_z (:872-890)
    x_len = corners[:, :, 0].max(1) - corners[:, :, 0].min(1)
    y_len = corners[:, :, 1].max(1) - corners[:, :, 1].min(1)
    keep = (x_len <= 6) & (y_len <= 6) & (y_len != 0)
    keep &= (corners[:, :, 2].min(1) >= -3) & (corners[:, :, 2].max(1) <= 1)
    return corners[keep], scores[keep], idx[keep]


def nms_order(scores, top=1000):
    """box_utils.py:712-714: scores.argsort()[::-1][:top].  numpy's default argsort is not stable;
    the oracle fixes the tie order as 'stable ascending, then reversed' (larger index first)."""
    return np.argsort(np.asarray(scores), kind="stable")[::-1][:top].astype(np.int32)


def post_process(cls, reg, dirp, anchors, score_thr, dir_offset, num_bins, nms_thr, tfm, gt_range,
                 top=1000):
    """voxel_postprocessor.py:245-405 for one cav (intermediate fusion): returns
    (corners [K,8,3] f32, scores [K] f32) after NMS and the range mask, or (None, None)."""
    corners, scores, _ = decode_candidates(cls, reg, dirp, anchors, score_thr, dir_offset, num_bins, tfm)
    if len(corners) == 0:
        return None, None
    order = nms_order(scores, top)
    keep = cref.nms_rotated(corners[:, :4, :2], order, nms_thr)
    corners = corners[keep]
    scores = scores[keep]
    r = np.asarray(gt_range, F32)
    inside = ((corners >= r[0:3]) & (corners <= r[3:6])).all(axis=2).sum(axis=1) >= 8  # box_utils.py:384-421
    return corners[inside], scores[inside]


# --------------------------------------------------------------------------------------------------
# K4: Lift-Splat geometry + BEV pooling                                                   [pinned]
# --------------------------------------------------------------------------------------------------
def gen_dx_bx(xbound, ybound, zbound):
    """camera_utils.py:129-134 (torch.Tensor -> fp32; nx int64)."""
    dx = np.array([row[2] for row in (xbound, ybound, zbound)], F32)
    bx = np.array([F32(row[0]) + F32(row[2]) / F32(2.0) for row in (xbound, ybound, zbound)], F32)
    nx = np.array([int((F32(row[1]) - F32(row[0])) / F32(row[2])) for row in (xbound, ybound, zbound)],
                  np.int64)
    return dx, bx, nx


def depth_discretization(depth_min, depth_max, num_bins, mode):
    """camera_utils.py:187-196."""
    if mode == "UD":
        return np.linspace(depth_min, depth_max, num_bins + 1)[:-1]
    if mode == "LID":
        bin_size = 2 * (depth_max - depth_min) / (num_bins * (1 + num_bins))
        idx = np.arange(0, num_bins)
        return depth_min + bin_size / 8 * ((2 * idx + 1) ** 2 - 1)
    raise ValueError(mode)


def create_frustum(final_dim, downsample, ddiscr, mode):
    """heter_encoders.py:110-123 -> [D,fH,fW,3] f32 (u, v, depth)."""
    ogfH, ogfW = final_dim
    fH, fW = ogfH // downsample, ogfW // downsample
    ds = np.asarray(depth_discretization(*ddiscr, mode), np.float64).astype(F32)
    D = ds.shape[0]
    xs = np.linspace(0, ogfW - 1, fW, dtype=np.float64).astype(F32) if fW > 1 else np.zeros(1, F32)
    ys = np.linspace(0, ogfH - 1, fH, dtype=np.float64).astype(F32) if fH > 1 else np.zeros(1, F32)
    fr = np.zeros((D, fH, fW, 3), F32)
    fr[..., 0] = xs[None, None, :]
    fr[..., 1] = ys[None, :, None]
    fr[..., 2] = ds[:, None, None]
    return fr


def lss_geometry(frustum, rots, trans, intrins, post_rots, post_trans):
    """heter_encoders.py:125-147 -> [B,N,D,fH,fW,3] f32."""
    fr = np.asarray(frustum, F32)
    B, N, _ = trans.shape
    pts = fr[None, None] - np.asarray(post_trans, F32).reshape(B, N, 1, 1, 1, 3)
    ipr = np.linalg.inv(np.asarray(post_rots, F32)).astype(F32).reshape(B, N, 1, 1, 1, 3, 3)
    pts = np.matmul(ipr, pts[..., None]).astype(F32)
    pts = np.concatenate([pts[..., :2, :] * pts[..., 2:3, :], pts[..., 2:3, :]], axis=-2)
    comb = np.matmul(np.asarray(rots, F32), np.linalg.inv(np.asarray(intrins, F32)).astype(F32)).astype(F32)
    pts = np.matmul(comb.reshape(B, N, 1, 1, 1, 3, 3), pts).astype(F32)[..., 0]
    return pts + np.asarray(trans, F32).reshape(B, N, 1, 1, 1, 3)


def bev_pool(geom, x, dx, bx, nx, accumulate=np.float64):
    """heter_encoders.py:161-217 (voxel_pooling): trunc-to-cell, bounds filter, per-cell sum,
    final[b,:,z,y,x], z folded into channels.  geom [B,N,D,H,W,3], x [B,N,D,H,W,C] -> [B,C*nz,ny,nx].
    The reference's cumsum trick accumulates fp32 error; the oracle sums each cell in `accumulate`
    precision (fp64 by default) -- the sum per cell, not its order, is what is defined."""
    B, N, D, H, W, C = x.shape
    Np = B * N * D * H * W
    xf = np.asarray(x, F32).reshape(Np, C)
    dx = np.asarray(dx, F32); bx = np.asarray(bx, F32)
    g = ((np.asarray(geom, F32) - (bx - dx / F32(2.0))) / dx)
    g = np.trunc(g).astype(np.int64).reshape(Np, 3)  # .long() truncates toward zero
    bix = np.repeat(np.arange(B), Np // B)
    kept = ((g[:, 0] >= 0) & (g[:, 0] < nx[0]) & (g[:, 1] >= 0) & (g[:, 1] < nx[1])
            & (g[:, 2] >= 0) & (g[:, 2] < nx[2]))
    g = g[kept]; xf = xf[kept]; bix = bix[kept]
    final = np.zeros((B, int(nx[2]), int(nx[1]), int(nx[0]), C), accumulate)
    np.add.at(final, (bix, g[:, 2], g[:, 1], g[:, 0]), xf.astype(accumulate))
    final = final.transpose(0, 4, 1, 2, 3)  # B,C,Z,Y,X
    return np.concatenate([final[:, :, z] for z in range(int(nx[2]))], axis=1).astype(F32)


def lift(depth_logit, feat):
    """lss_submodule.py:129-134: softmax over depth (x) features.  depth_logit [BN,D,fH,fW],
    feat [BN,C,fH,fW] -> [BN,C,D,fH,fW] f32."""
    d = np.asarray(depth_logit, F32)
    d = d - d.max(axis=1, keepdims=True)
    e = np.exp(d)
    p = (e / e.sum(axis=1, keepdims=True)).astype(F32)
    return p[:, None] * np.asarray(feat, F32)[:, :, None]


# --------------------------------------------------------------------------------------------------
# K3: MeanVFE + sparse 3-D convolution (spconv semantics)             [unpinned]
# The arithmetic lives in the third-party spconv (not vendored, not installed): PARITY UNPINNED.
# Restated as a DENSE conv3d on the densified grid, masked by spconv's active-site rules:
#   submanifold: outputs only at the input's active sites;
#   strided:     an output site is active iff some active input lies in its receptive field.
# Only practical on small grids -- which is what the tests use.
# --------------------------------------------------------------------------------------------------
def mean_vfe(voxels, num_points):
    """mean_vfe.py:13-31: sum over all P rows / clamp_min(num, 1)."""
    v = np.asarray(voxels, F32)
    n = np.maximum(np.asarray(num_points).astype(F32), F32(1.0))
    return (v.sum(axis=1, dtype=F32) / n[:, None]).astype(F32)


def densify(features, indices, shape, batch):
    import torch
    C = features.shape[1]
    D, H, W = shape
    dense = torch.zeros((batch, C, D, H, W), dtype=torch.float32)
    mask = torch.zeros((batch, 1, D, H, W), dtype=torch.float32)
    idx = torch.as_tensor(np.asarray(indices)).long()
    dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = torch.as_tensor(np.asarray(features, F32))
    mask[idx[:, 0], 0, idx[:, 1], idx[:, 2], idx[:, 3]] = 1.0
    return dense, mask


def sparse_conv_dense(dense, mask, weight, ksize, stride, padding, subm, bn_gamma, bn_beta, bn_mean, bn_var,
                      eps=1e-3):
    """One SparseSequential(conv, BatchNorm1d, ReLU) block (sparse_backbone_3d.py:11-30) on dense
    tensors.  weight [kz,ky,kx,Cin,Cout] (spconv 1.2.1 layout).  Returns (dense_out, mask_out)."""
    import torch
    import torch.nn.functional as Fn
    w = torch.as_tensor(np.asarray(weight, F32)).permute(4, 3, 0, 1, 2).contiguous()
    if subm:
        pad = tuple(k // 2 for k in ksize)
        y = Fn.conv3d(dense, w, None, 1, pad)
        mask_out = mask
    else:
        y = Fn.conv3d(dense, w, None, tuple(stride), tuple(padding))
        ones = torch.ones((1, 1) + tuple(ksize))
        mask_out = (Fn.conv3d(mask, ones, None, tuple(stride), tuple(padding)) > 0).float()
    g = torch.as_tensor(np.asarray(bn_gamma, F32)).view(1, -1, 1, 1, 1)
    b = torch.as_tensor(np.asarray(bn_beta, F32)).view(1, -1, 1, 1, 1)
    mu = torch.as_tensor(np.asarray(bn_mean, F32)).view(1, -1, 1, 1, 1)
    var = torch.as_tensor(np.asarray(bn_var, F32)).view(1, -1, 1, 1, 1)
    y = torch.relu((y - mu) / torch.sqrt(var + eps) * g + b)
    return y * mask_out, mask_out  # BN + ReLU act on the active rows only; inactive cells stay zero


SECOND_LAYERS = [  # (state_dict prefix, ksize, stride, padding, subm)  sparse_backbone_3d.py:48-91
    ("conv_input", (3, 3, 3), (1, 1, 1), (1, 1, 1), True),
    ("conv1.0", (3, 3, 3), (1, 1, 1), (1, 1, 1), True),
    ("conv2.0", (3, 3, 3), (2, 2, 2), (1, 1, 1), False), ("conv2.1", (3, 3, 3), (1, 1, 1), (1, 1, 1), True),
    ("conv2.2", (3, 3, 3), (1, 1, 1), (1, 1, 1), True),
    ("conv3.0", (3, 3, 3), (2, 2, 2), (1, 1, 1), False), ("conv3.1", (3, 3, 3), (1, 1, 1), (1, 1, 1), True),
    ("conv3.2", (3, 3, 3), (1, 1, 1), (1, 1, 1), True),
    ("conv4.0", (3, 3, 3), (2, 2, 2), (0, 1, 1), False), ("conv4.1", (3, 3, 3), (1, 1, 1), (1, 1, 1), True),
    ("conv4.2", (3, 3, 3), (1, 1, 1), (1, 1, 1), True),
    ("conv_out", (3, 1, 1), (2, 1, 1), (0, 0, 0), False),
]


def second_backbone(sd, prefix, vfe_features, indices, sparse_shape, batch):
    """VoxelBackBone8x.forward + HeightCompression (sparse_backbone_3d.py:114-130,
    height_compression.py:10-26) -> dense [B, C*D, H, W] numpy."""
    dense, mask = densify(vfe_features, indices, sparse_shape, batch)
    for name, k, s, p, subm in SECOND_LAYERS:
        cp = f"{prefix}{name}.0." if name in ("conv_input", "conv_out") else f"{prefix}{name}.0."
        bp = f"{prefix}{name}.1."
        dense, mask = sparse_conv_dense(dense, mask, np.asarray(sd[cp + "weight"]), k, s, p, subm,
                                        np.asarray(sd[bp + "weight"]), np.asarray(sd[bp + "bias"]),
                                        np.asarray(sd[bp + "running_mean"]), np.asarray(sd[bp + "running_var"]))
    B, C, D, H, W = dense.shape
    return dense.reshape(B, C * D, H, W).numpy()


def _sp_keys(idx, shape):
    D, H, W = shape
    idx = np.asarray(idx).astype(np.int64)
    return ((idx[:, 0] * D + idx[:, 1]) * H + idx[:, 2]) * W + idx[:, 3]


def _sp_lookup(sorted_keys, query, valid):
    """Row of every query key in the sorted key list, -1 where absent (or where `valid` is False)."""
    pos = np.searchsorted(sorted_keys, query)
    pos = np.minimum(pos, len(sorted_keys) - 1)
    hit = valid & (sorted_keys[pos] == query)
    return np.where(hit, pos, -1)


def sparse_conv_rules(in_idx, in_shape, ksize, stride, padding, subm):
    """spconv's indice pairs restated on sorted coordinate lists (the rules of the comment block above; same semantics as
    `sparse_conv_dense`, without the dense grid): -> (out_idx [n_out, 4] sorted by linear coordinate, out_shape, nbr [n_out, K])
    with nbr[o][tap] = row of the input site at o * stride - padding + tap, or -1; taps enumerated (kz, ky, kx) row-major."""
    in_idx = np.asarray(in_idx).astype(np.int64)
    in_keys = _sp_keys(in_idx, in_shape)
    assert np.all(np.diff(in_keys) > 0), "input sites must be sorted and unique"
    k, s, p = (np.asarray(v, np.int64) for v in (ksize, stride, padding))
    if subm:
        out_shape, out_idx = list(in_shape), in_idx
        p = k // 2
        s = np.ones(3, np.int64)
    else:
        out_shape = [int((in_shape[d] + 2 * p[d] - k[d]) // s[d] + 1) for d in range(3)]
        cands = []
        for kz in range(k[0]):
            for ky in range(k[1]):
                for kx in range(k[2]):
                    num = in_idx[:, 1:] + p - np.array([kz, ky, kx])          # o * s = i + p - tap
                    ok = np.all((num % s == 0) & (num >= 0), axis=1)
                    o = num // s
                    ok &= np.all(o < np.array(out_shape), axis=1)
                    cands.append(np.concatenate([in_idx[ok, :1], o[ok]], 1))
        c = np.concatenate(cands)
        keys = np.unique(_sp_keys(c, out_shape))
        D, H, W = out_shape
        out_idx = np.stack([keys // (D * H * W), keys // (H * W) % D, keys // W % H, keys % W], 1)
    K = int(k[0] * k[1] * k[2])
    nbr = np.full((out_idx.shape[0], K), -1, np.int64)
    t = 0
    for kz in range(k[0]):
        for ky in range(k[1]):
            for kx in range(k[2]):
                c = out_idx[:, 1:] * s - p + np.array([kz, ky, kx])
                ok = np.all((c >= 0) & (c < np.array(in_shape)), axis=1)
                q = _sp_keys(np.concatenate([out_idx[:, :1], np.where(ok[:, None], c, 0)], 1), in_shape)
                nbr[:, t] = _sp_lookup(in_keys, q, ok)
                t += 1
    return out_idx, out_shape, nbr


def second_backbone_sparse(sd, prefix, vfe_features, indices, sparse_shape, batch, return_sites=False):
    """VoxelBackBone8x.forward + HeightCompression (sparse_backbone_3d.py:114-130, height_compression.py:10-26) with the sparse
    convolutions evaluated on their rule pairs instead of a dense grid -- the same arithmetic as `second_backbone` (which the
    tests hold it against on small grids), practical at the reference's +-102.4 m / 0.1 m grid (1.7e8 cells per agent).
    out[o] = ReLU(BN(sum_tap W[tap]^T x[nbr[o][tap]])), taps added in (kz, ky, kx) order in fp32.  -> dense [B, C*D, H, W]."""
    idx = np.asarray(indices).astype(np.int64)
    order = np.argsort(_sp_keys(idx, sparse_shape), kind="stable")
    idx, x = idx[order], np.asarray(vfe_features, F32)[order]
    shape = list(sparse_shape)
    cache = {}
    for name, k, s, p, subm in SECOND_LAYERS:
        w = np.asarray(sd[f"{prefix}{name}.0.weight"], F32)
        w = w.reshape(-1, w.shape[3], w.shape[4])
        key = (id(idx), tuple(k)) if subm else None
        if subm and key in cache:
            out_idx, out_shape, nbr = cache[key]
        else:
            out_idx, out_shape, nbr = sparse_conv_rules(idx, shape, k, s, p, subm)
            if subm:
                cache[(id(out_idx), tuple(k))] = (out_idx, out_shape, nbr)
        y = np.zeros((out_idx.shape[0], w.shape[2]), F32)
        for t in range(w.shape[0]):
            rows = np.nonzero(nbr[:, t] >= 0)[0]
            if rows.size:
                y[rows] += x[nbr[rows, t]] @ w[t]
        bp = f"{prefix}{name}.1."
        g, b = np.asarray(sd[bp + "weight"], F32), np.asarray(sd[bp + "bias"], F32)
        mu, var = np.asarray(sd[bp + "running_mean"], F32), np.asarray(sd[bp + "running_var"], F32)
        x = np.maximum((y - mu) / np.sqrt(var + F32(1e-3)) * g + b, F32(0.0)).astype(F32)
        idx, shape = out_idx, out_shape
    D, H, W = shape
    C = x.shape[1]
    dense = np.zeros((batch, C, D, H, W), F32)
    dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = x
    out = dense.reshape(batch, C * D, H, W)
    return (out, idx, x) if return_sites else out


# ---- training-side label path -------------------------------------------------------------------------
def bbox_overlaps(boxes, query_boxes):
    """opencood/utils/box_overlaps.pyx:17-57 (Fast R-CNN bbox_overlaps, the `+1` pixel convention), fp32 throughout:
    boxes [N,4], query_boxes [K,4] (x1,y1,x2,y2) -> IoU [N,K].  PINNED: bit-exact against the reference's compiled
    Cython routine (oracle/_ref/box_overlaps*.so, tests/golden/label.npz)."""
    # Arithmetic types as Cython generates them (oracle/_ref/box_overlaps.c): differences of two float32 are float32,
    # every `+ 1` is `+ 1.0` in DOUBLE, products of such terms are double, and the result is rounded once when stored
    # in the float32 variable; `iw * ih` is a float32 product; the final division is float32.
    b = np.asarray(boxes, F32)[:, None, :]
    q = np.asarray(query_boxes, F32)[None, :, :]
    D = np.float64
    box_area = (((q[..., 2] - q[..., 0]).astype(D) + 1.0) * ((q[..., 3] - q[..., 1]).astype(D) + 1.0)).astype(F32)
    iw = ((np.minimum(b[..., 2], q[..., 2]) - np.maximum(b[..., 0], q[..., 0])).astype(D) + 1.0).astype(F32)
    ih = ((np.minimum(b[..., 3], q[..., 3]) - np.maximum(b[..., 1], q[..., 1])).astype(D) + 1.0).astype(F32)
    inter = (iw * ih).astype(F32)
    ua = (((b[..., 2] - b[..., 0]).astype(D) + 1.0) * ((b[..., 3] - b[..., 1]).astype(D) + 1.0)
          + box_area.astype(D) - inter.astype(D)).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        ov = (inter / ua).astype(F32)
    return np.where((iw > 0) & (ih > 0), ov, F32(0)).astype(F32)


def standup_boxes(boxes_center, order="hwl"):
    """boxes_to_corners_3d + corner2d_to_standup_box (box_utils.py:152-204,225-248): axis-aligned (x1,y1,x2,y2) of the
    rotated footprint, float64 like the reference's np.zeros default."""
    assert order == "hwl"
    c = boxes_to_corners_3d_hwl(np.asarray(boxes_center))
    return np.stack([c[:, :, 0].min(1), c[:, :, 1].min(1), c[:, :, 0].max(1), c[:, :, 1].max(1)], 1).astype(np.float64)


def generate_label(gt_box_center, anchors, mask, pos_threshold, neg_threshold, order="hwl"):
    """VoxelPostprocessor.generate_label (voxel_postprocessor.py:85-207): anchors [H,W,A,7], gt_box_center [max,7],
    mask [max] -> pos_equal_one [H,W,A], neg_equal_one [H,W,A], targets [H,W,7A] (float64 like the reference)."""
    H, W, A = anchors.shape[:3]
    an = np.asarray(anchors).reshape(-1, 7)
    gt_all = np.asarray(gt_box_center)
    gt = gt_all[np.asarray(mask) == 1]
    an_d = np.sqrt(an[:, 4] ** 2 + an[:, 5] ** 2)
    iou = bbox_overlaps(standup_boxes(an, order).astype(F32), standup_boxes(gt, order).astype(F32)) \
        if len(gt) else np.zeros((len(an), 0), F32)
    pos = np.zeros(len(an))
    neg = np.zeros(len(an))
    tgt = np.zeros((len(an), 7))
    assigned = np.full(len(an), -1, np.int64)
    # (1) anchors above the positive threshold take the SMALLEST gt index above it (np.where order + np.unique first hit)
    above = iou > pos_threshold
    has = above.any(1) if iou.shape[1] else np.zeros(len(an), bool)
    if has.any():
        assigned[has] = above[has].argmax(1)
    # (2) the best anchor of every gt (first maximum, only if its IoU is > 0) becomes positive too; an anchor that is
    #     the best of several gts and not already positive keeps the smallest such gt
    best = []
    for g in range(iou.shape[1]):
        a = int(iou[:, g].argmax())
        if iou[a, g] > 0:
            best.append(a)
            if assigned[a] < 0:
                assigned[a] = g
    p = np.nonzero(assigned >= 0)[0]
    pos[p] = 1
    neg[(iou < neg_threshold).all(1)] = 1
    if best:
        neg[best] = 0
    # NOTE the reference indexes gt_box_center (all rows), not the masked subset, with the masked gt index
    # (voxel_postprocessor.py:172-190): identical when the valid boxes come first, as the datasets produce them
    g = gt_all[assigned[p]]
    tgt[p, 0] = (g[:, 0] - an[p, 0]) / an_d[p]
    tgt[p, 1] = (g[:, 1] - an[p, 1]) / an_d[p]
    tgt[p, 2] = (g[:, 2] - an[p, 2]) / an[p, 3]
    tgt[p, 3] = np.log(g[:, 3] / an[p, 3])
    tgt[p, 4] = np.log(g[:, 4] / an[p, 4])
    tgt[p, 5] = np.log(g[:, 5] / an[p, 5])
    tgt[p, 6] = g[:, 6] - an[p, 6]
    return pos.reshape(H, W, A), neg.reshape(H, W, A), tgt.reshape(H, W, A * 7)
