"""Exact-rational IoU of two convex quadrilaterals (TEST INFRASTRUCTURE -- tightens the one decision of the path that is bit-exact by
contract: which candidates rotated NMS keeps).

The reference computes the IoU with shapely / GEOS (opencood/utils/common_utils.py:230-270, box_utils.py:693-738), which is absent from
the reference tree and from this image: `oracle_quad_iou` (oracle/oracle_ref.c) restates it as an fp64 Sutherland-Hodgman clip -- "parity
unpinned".  The corner coordinates are fp32 values, i.e. exact rationals, and the intersection of two convex polygons has rational
vertices, so the TRUE IoU of a pair is a rational number.  This module computes it with `fractions.Fraction` (no rounding anywhere).
A suppression decision `iou > thr` can only depend on the floating-point arithmetic used (GEOS's, the oracle's, the GPU kernel's) when the
true IoU lies within that arithmetic's error of the threshold; the tests count such pairs and check that the oracle decides every other
pair exactly like the true value does.
"""
from fractions import Fraction

import numpy as np


def _area2(poly):
    """Twice the signed area (shoelace), exact."""
    s = Fraction(0)
    n = len(poly)
    for i in range(n):
        x0, y0 = poly[i]
        x1, y1 = poly[(i + 1) % n]
        s += x0 * y1 - x1 * y0
    return s


def _clip(poly, a, b):
    """Sutherland-Hodgman step: the part of convex `poly` on the left of the directed line a -> b (counter-clockwise clip polygon)."""
    out = []
    n = len(poly)
    ax, ay = a
    dx, dy = b[0] - ax, b[1] - ay

    def side(p):
        return dx * (p[1] - ay) - dy * (p[0] - ax)
    for i in range(n):
        p, q = poly[i], poly[(i + 1) % n]
        sp, sq = side(p), side(q)
        if sp >= 0:
            out.append(p)
        if (sp > 0 and sq < 0) or (sp < 0 and sq > 0):
            t = sp / (sp - sq)
            out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return out


def exact_quad_iou(qa, qb):
    """qa, qb: [4, 2] float32 corners in ring order (either orientation) -> Fraction IoU (None for a degenerate pair: zero union)."""
    a = [(Fraction(float(x)), Fraction(float(y))) for x, y in np.asarray(qa, np.float32).reshape(4, 2)]
    b = [(Fraction(float(x)), Fraction(float(y))) for x, y in np.asarray(qb, np.float32).reshape(4, 2)]
    if _area2(a) < 0:
        a.reverse()
    if _area2(b) < 0:
        b.reverse()
    sa, sb = _area2(a), _area2(b)
    cur = a
    for e in range(4):
        if not cur:
            break
        cur = _clip(cur, b[e], b[(e + 1) % 4])
    inter = _area2(cur) if len(cur) >= 3 else Fraction(0)
    if inter < 0:
        inter = Fraction(0)
    uni = sa + sb - inter
    if uni == 0:
        return None
    return inter / uni


def box_quad(cx, cy, length, width, yaw):
    """fp32 corners of a rotated rectangle (ring order), the way the decoded boxes reach nms_rotated: computed in fp32."""
    c, s = np.float32(np.cos(np.float32(yaw))), np.float32(np.sin(np.float32(yaw)))
    hl, hw = np.float32(length) / np.float32(2), np.float32(width) / np.float32(2)
    pts = []
    for sx, sy in ((1, 1), (1, -1), (-1, -1), (-1, 1)):
        lx, ly = np.float32(sx) * hl, np.float32(sy) * hw
        pts.append((np.float32(cx) + lx * c - ly * s, np.float32(cy) + lx * s + ly * c))
    return np.asarray(pts, np.float32)
