"""Synthetic OPV2V-shaped scenes: seeded LiDAR frames, camera rigs and agent poses.

There is no dataset on the GPU box; bench.py, smoke() and the tests build their inputs here.
Everything is numpy + a seed, so the same scene can be regenerated anywhere.
"""
import math

import numpy as np


def lidar_frame(seed, n_rings=64, n_azimuth=1024, sensor_height=1.9, n_boxes=40, max_range=120.0,
                dropout=0.05):
    """One spinning-LiDAR sweep: rays hit the ground plane or one of `n_boxes` car-sized boxes.

    Returns float32 [N,4] (x, y, z, intensity) in the sensor frame, after the reference's
    ego-point mask (opencood/utils/pcd_utils.py:70-88) and a seeded shuffle (:91-95).
    """
    rng = np.random.default_rng(seed)
    elev = np.deg2rad(np.linspace(-25.0, 2.0, n_rings))
    azim = np.linspace(-np.pi, np.pi, n_azimuth, endpoint=False)
    el, az = np.meshgrid(elev, azim, indexing="ij")
    dirs = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], -1).reshape(-1, 3)
    # ground plane z = -sensor_height
    with np.errstate(divide="ignore", invalid="ignore"):
        t_ground = np.where(dirs[:, 2] < -1e-6, -sensor_height / dirs[:, 2], np.inf)
    t = np.minimum(t_ground, max_range)
    # boxes: axis-aligned in their own yaw frame, 4.5 x 2 x 1.6 m, sitting on the ground
    centers = rng.uniform(-100.0, 100.0, size=(n_boxes, 2))
    yaws = rng.uniform(-np.pi, np.pi, size=n_boxes)
    half = np.array([2.25, 1.0, 0.8])
    for c, yaw in zip(centers, yaws):
        ca, sa = math.cos(yaw), math.sin(yaw)
        R = np.array([[ca, sa, 0.0], [-sa, ca, 0.0], [0.0, 0.0, 1.0]])  # world -> box
        o = R @ np.array([-c[0], -c[1], sensor_height - 0.8])  # ray origin in the box frame
        d = dirs @ R.T
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (-half - o) / d
            t2 = (half - o) / d
        tn = np.nanmax(np.minimum(t1, t2), axis=1)
        tf = np.nanmin(np.maximum(t1, t2), axis=1)
        hit = (tn <= tf) & (tn > 0.5)
        t = np.where(hit & (tn < t), tn, t)
    keep = np.isfinite(t) & (t < max_range) & (rng.uniform(size=t.shape) >= dropout)
    pts = dirs[keep] * t[keep, None]
    pts = pts + rng.normal(0.0, 0.01, size=pts.shape)
    inten = rng.uniform(0.0, 1.0, size=(pts.shape[0], 1))
    pcd = np.concatenate([pts, inten], 1).astype(np.float32)
    # mask_ego_points
    ego = (pcd[:, 0] >= -1.95) & (pcd[:, 0] <= 2.95) & (pcd[:, 1] >= -1.1) & (pcd[:, 1] <= 1.1)
    pcd = pcd[~ego]
    rng.shuffle(pcd, axis=0)
    return np.ascontiguousarray(pcd)


def x_to_world(pose):
    """pose = [x, y, z, roll, yaw, pitch] (degrees), the reference's convention
    (opencood/utils/transformation_utils.py:264-307) -> 4x4 float64."""
    x, y, z, roll, yaw, pitch = pose
    c_y, s_y = math.cos(math.radians(yaw)), math.sin(math.radians(yaw))
    c_r, s_r = math.cos(math.radians(roll)), math.sin(math.radians(roll))
    c_p, s_p = math.cos(math.radians(pitch)), math.sin(math.radians(pitch))
    m = np.identity(4)
    m[0, 3], m[1, 3], m[2, 3] = x, y, z
    m[0, 0] = c_p * c_y
    m[0, 1] = c_y * s_p * s_r - s_y * c_r
    m[0, 2] = -c_y * s_p * c_r - s_y * s_r
    m[1, 0] = s_y * c_p
    m[1, 1] = s_y * s_p * s_r + c_y * c_r
    m[1, 2] = -s_y * s_p * c_r + c_y * s_r
    m[2, 0] = s_p
    m[2, 1] = -c_p * s_r
    m[2, 2] = c_p * c_r
    return m


def agent_poses(seed, n_agents, r_min=10.0, r_max=60.0):
    """Ego at the origin; agent k at radius U(r_min,r_max), bearing 2*pi*k/n, yaw U(-180,180)."""
    rng = np.random.default_rng(seed)
    poses = [[0.0, 0.0, 0.0, 0.0, 0.0, 0.0]]
    for k in range(1, n_agents):
        r = rng.uniform(r_min, r_max)
        b = 2.0 * math.pi * k / n_agents
        poses.append([r * math.cos(b), r * math.sin(b), 0.0, 0.0, rng.uniform(-180.0, 180.0), 0.0])
    return poses


def pairwise_t_matrix(poses, max_cav):
    """[L,L,4,4] float64, entry [i,j] = T_{j<-i} = inv(T_j) T_i (transformation_utils.py:21-66,
    proj_first = False); identity beyond the present agents."""
    L = max_cav
    out = np.tile(np.eye(4), (L, L, 1, 1))
    T = [x_to_world(p) for p in poses]
    for i in range(len(T)):
        for j in range(len(T)):
            if i != j:
                out[i, j] = np.linalg.solve(T[j], T[i])
    return out


def camera_rig(seed, n_cams=4, H=384, W=512):
    """Four cameras yawed 0/100/-100/180 deg, 90-degree FoV, no post-augmentation.
    Returns dict of float32 arrays: rots [n,3,3], trans [n,3], intrins [n,3,3], post_rots, post_trans."""
    yaws = np.deg2rad([0.0, 100.0, -100.0, 180.0])[:n_cams]
    rots, trans = [], []
    # camera frame (x right, y down, z forward) -> ego frame (x forward, y left, z up)
    base = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
    for yw in yaws:
        c, s = math.cos(yw), math.sin(yw)
        Rz = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
        rots.append(Rz @ base)
        trans.append(Rz @ np.array([1.0, 0.0, 1.5]))
    K = np.array([[W / 2.0, 0.0, W / 2.0], [0.0, W / 2.0, H / 2.0], [0.0, 0.0, 1.0]])
    return {
        "rots": np.stack(rots).astype(np.float32),
        "trans": np.stack(trans).astype(np.float32),
        "intrins": np.tile(K, (n_cams, 1, 1)).astype(np.float32),
        "post_rots": np.tile(np.eye(3), (n_cams, 1, 1)).astype(np.float32),
        "post_trans": np.zeros((n_cams, 3), np.float32),
    }
