"""ctypes view of oracle/liboracle.so (oracle_ref.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "oracle_ref.c")
LIB = os.path.join(HERE, "liboracle.so")
_lib = None


def build(force=False):
    """gcc -O2 -ffp-contract=off -shared -fPIC oracle_ref.c -> liboracle.so (rebuilt when stale)."""
    if (not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC)):
        return LIB
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", SRC, "-o", LIB, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("gcc failed:\n" + r.stderr)
    return LIB


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_voxelize.restype = ctypes.c_int
        _lib.oracle_quad_iou.restype = ctypes.c_float
        _lib.oracle_nms_rotated.restype = ctypes.c_int
        _lib.oracle_pcdet_overlap.restype = ctypes.c_float
        _lib.oracle_pcdet_iou_bev.restype = ctypes.c_float
        _lib.oracle_pcdet_nms.restype = ctypes.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def voxelize(points, lidar_range, voxel_size, max_points, max_voxels, batch_idx=None):
    """Returns (voxels [M,P,4] f32, coords [M,3] (z,y,x) or [M,4] (b,z,y,x) i32, num [M] i32)."""
    pts = np.ascontiguousarray(points, np.float32)
    assert pts.ndim == 2 and pts.shape[1] == 4
    n = pts.shape[0]
    cap = max(1, min(n, max_voxels))
    voxels = np.zeros((cap, max_points, 4), np.float32)
    coords = np.zeros((cap, 4), np.int32)
    num = np.zeros((cap,), np.int32)
    rng = np.asarray(lidar_range, np.float32)
    vs = np.asarray(voxel_size, np.float32)
    m = lib().oracle_voxelize(_p(pts, ctypes.c_float), n, _p(rng, ctypes.c_float), _p(vs, ctypes.c_float),
                              int(max_points), int(max_voxels), int(batch_idx or 0),
                              _p(voxels, ctypes.c_float), _p(coords, ctypes.c_int32), _p(num, ctypes.c_int32))
    if m < 0:
        raise MemoryError("oracle_voxelize")
    c = coords[:m] if batch_idx is not None else coords[:m, 1:]
    return voxels[:m].copy(), np.ascontiguousarray(c), num[:m].copy()


def quad_iou(qa, qb):
    a = np.ascontiguousarray(qa, np.float32).reshape(-1, 8)
    b = np.ascontiguousarray(qb, np.float32).reshape(-1, 8)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    lib().oracle_quad_iou_matrix(_p(a, ctypes.c_float), a.shape[0], _p(b, ctypes.c_float), b.shape[0],
                                 _p(out, ctypes.c_float))
    return out


def nms_rotated(quads, order, thr):
    """quads [N,4,2] f32, order = candidate indices sorted by descending score -> kept indices."""
    q = np.ascontiguousarray(quads, np.float32).reshape(-1, 8)
    o = np.ascontiguousarray(order, np.int32)
    keep = np.zeros((max(1, o.shape[0]),), np.int32)
    k = lib().oracle_nms_rotated(_p(q, ctypes.c_float), _p(o, ctypes.c_int32), int(o.shape[0]),
                                 ctypes.c_float(thr), _p(keep, ctypes.c_int32))
    return keep[:k].copy()


# ---- pcdet rotated BEV IoU / NMS -------------------------------------------------------------
REF_LIB = os.path.join(HERE, "_ref", "libpcdet_iou_ref.so")
_ref = None


def build_ref(reference_root="/root/reference"):
    """Compile the reference's own iou3d_cpu.cpp where it lies (oracle/Makefile.ref).  Returns the .so path, or
    None when the reference tree is absent (GPU box: the prebuilt file travels with the snapshot)."""
    if not os.path.isdir(reference_root):
        return REF_LIB if os.path.exists(REF_LIB) else None
    r = subprocess.run(["make", "-f", os.path.join("oracle", "Makefile.ref"), f"REF={reference_root}"],
                       cwd=os.path.dirname(HERE), capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("reference build failed:\n" + r.stdout + r.stderr)
    return REF_LIB


def ref_lib():
    """ctypes handle of oracle/_ref/libpcdet_iou_ref.so (None if it was never built)."""
    global _ref
    if _ref is None and os.path.exists(REF_LIB):
        import torch  # noqa: F401  (the reference routine takes at::Tensor: libtorch must be loaded first)
        _ref = ctypes.CDLL(REF_LIB)
        _ref.ref_boxes_iou_bev_cpu.restype = ctypes.c_int
    return _ref


def ref_boxes_iou_bev_cpu(boxes_a, boxes_b):
    """The reference's boxes_iou_bev_cpu (iou3d_cpu.cpp:233-252) on [N,7] / [M,7] f32 -> [N,M] f32."""
    a = np.ascontiguousarray(boxes_a, np.float32)
    b = np.ascontiguousarray(boxes_b, np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    ref_lib().ref_boxes_iou_bev_cpu(_p(a, ctypes.c_float), a.shape[0], _p(b, ctypes.c_float), b.shape[0],
                                    _p(out, ctypes.c_float))
    return out


def pcdet_matrix(boxes_a, boxes_b, mode):
    """mode 'overlap' | 'iou' | 'iou_normal' -> [N,M] f32 (oracle restatement)."""
    a = np.ascontiguousarray(boxes_a, np.float32)
    b = np.ascontiguousarray(boxes_b, np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    lib().oracle_pcdet_matrix(_p(a, ctypes.c_float), a.shape[0], _p(b, ctypes.c_float), b.shape[0],
                              {"overlap": 0, "iou": 1, "iou_normal": 2}[mode], _p(out, ctypes.c_float))
    return out


def pcdet_nms(boxes_sorted, thr, rotated=True):
    """Greedy NMS over boxes already in descending-score order -> kept indices (int64)."""
    b = np.ascontiguousarray(boxes_sorted, np.float32)
    keep = np.zeros((b.shape[0],), np.int64)
    k = lib().oracle_pcdet_nms(_p(b, ctypes.c_float), b.shape[0], ctypes.c_float(thr), int(bool(rotated)),
                               _p(keep, ctypes.c_int64))
    return keep[:k]
