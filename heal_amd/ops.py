"""Tensor-level wrappers over the C ABI (include/heal_amd.h).

torch is used for device memory and the current HIP stream only; all arithmetic happens in
libheal_amd.so.  Every function requires CUDA(HIP) tensors and raises otherwise -- there is no CPU
path in the product.
"""
import ctypes
import os

import numpy as np
import torch

from . import _capi

_WS = {}

# Optional per-operator timing with HIP events on the launch stream (bench.py turns it on):
# TIMING = {} enables it; every wrapper then appends (start_event, end_event) under its name.
TIMING = None
# SP_TRACE = [] additionally records, per sparse convolution, what its roofline needs: channel counts, taps,
# the (device) input / output row counts and the neighbour table (-> rule pairs R), with the launch's timing events.
SP_TRACE = None
# LAST_CALLS = {} keeps, per MULTI-LAUNCH operator (K1: memset + 5 kernels, K8: memset + 4 kernels), a closure that repeats its last
# call on the same tensors.  An event pair around such a chain launched from the host mostly times the host (6 launches a ~10 us
# apart); bench.py replays the closure inside a captured graph (graph_period_ms) -- how the chain runs in the timed region.
LAST_CALLS = None


def _remember(name, fn):
    if LAST_CALLS is not None:
        LAST_CALLS[name] = fn


def graph_period_ms(fn, reps=20, iters=10):
    """Device time per call of `fn` (launches only, outputs dropped) as a captured graph runs it: `reps` back-to-back calls captured
    once on the current (non-default) stream, replayed `iters` times; median of replay time / reps."""
    global TIMING, LAST_CALLS
    saved = (TIMING, LAST_CALLS)
    TIMING, LAST_CALLS = None, None          # no events inside a capture
    try:
        st = torch.cuda.current_stream()
        for _ in range(3):
            fn()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
        ts = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            g.replay()
            e1.record(st)
            st.synchronize()
            ts.append(e0.elapsed_time(e1) / reps)
        ts.sort()
        return ts[len(ts) // 2]
    finally:
        TIMING, LAST_CALLS = saved


class _Timed:
    """work: optional (flops, algorithmic bytes) of this launch, accumulated per name for the roofline report.
    kernel_events=True (operators that are ONE kernel whose C entry point supports heal_next_launch_events): the events are
    stamped with the kernel's own begin / end (what a rocprofv3 kernel trace reports) instead of being recorded around the launch,
    which adds the dispatch and marker latencies (3-5 us: a quarter of K4's 13 us)."""

    def __init__(self, name, flops=0.0, nbytes=0.0, kernel_events=False):
        self.name = name
        self.flops = float(flops)
        self.nbytes = float(nbytes)
        self.kernel_events = kernel_events

    def __enter__(self):
        if TIMING is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            st = torch.cuda.current_stream()
            self.e0.record(st)
            if self.kernel_events and not torch.cuda.is_current_stream_capturing():
                self.e1.record(st)          # instantiates the underlying hipEvent_t; the launch re-stamps both
                _capi.call("heal_next_launch_events", ctypes.c_void_p(self.e0.cuda_event), ctypes.c_void_p(self.e1.cuda_event))
            else:
                self.kernel_events = False
        return self

    def __exit__(self, *exc):
        if TIMING is not None:
            if not self.kernel_events:
                self.e1.record(torch.cuda.current_stream())
            else:
                _capi.call("heal_next_launch_events", None, None)   # disarm if the launch never happened (an error above)
            TIMING.setdefault(self.name, []).append((self.e0, self.e1, self.flops, self.nbytes))
        return False


def timing_summary():
    """name -> (calls, mean milliseconds); call after torch.cuda.synchronize()."""
    out = {}
    for name, evs in (TIMING or {}).items():
        ms = [e[0].elapsed_time(e[1]) for e in evs]
        out[name] = (len(ms), sum(ms) / max(len(ms), 1))
    return out


def work_summary():
    """name -> dict(calls, total_ms, flops, bytes) summed over the recorded launches (call after a synchronize)."""
    out = {}
    for name, evs in (TIMING or {}).items():
        out[name] = {"calls": len(evs), "total_ms": sum(e[0].elapsed_time(e[1]) for e in evs),
                     "flops": sum(e[2] for e in evs), "bytes": sum(e[3] for e in evs)}
    return out


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _need(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _capi.HealAmdError(f"{name} must be a CUDA/HIP tensor (heal_amd has no CPU path)")
    if t.dtype != dtype:
        raise _capi.HealAmdError(f"{name} must have dtype {dtype}, got {t.dtype}")
    if t.grad_fn is not None and torch.is_grad_enabled():
        # an activation of a recorded autograd graph: the HIP operators have no backward, the result would silently drop out
        # of the graph.  The modules route such calls to their gradient path (bev_blocks.grad_path); reaching this is a bug
        # or an operator without one (sparse 3-D encoder, attention fusions).
        raise _capi.HealAmdError(f"{name} carries autograd history: this HIP operator is inference-only (no backward). "
                                 "Run under torch.no_grad(), or train a configuration whose blocks have a gradient path")
    return t.contiguous()


_WS_RETIRED = []


def _retire_cache(cache):
    """Evict a weight-layout cache WITHOUT freeing its device tensors: a captured HIP graph may have their addresses baked in
    (same policy as `_workspace`); every tensor found in the cached values moves to the keep-alive list."""
    for v in cache.values():
        for t in (v if isinstance(v, (tuple, list)) else (v,)):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                _WS_RETIRED.append(t)
    cache.clear()


def _workspace(key, nbytes, device):
    """Grow-only per-(op, device) scratch buffer (256-B aligned by the caching allocator).  A buffer that is outgrown is
    RETIRED, not freed: a captured HIP graph may have its address baked in, and handing the memory back to the caching
    allocator would let a later replay scribble over somebody else's tensor."""
    # one scratch per (operator, device, STREAM): modalities encoded on concurrent streams must not share scratch
    k = (key, device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(k)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _WS_RETIRED.append(buf)
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _WS[k] = buf
    return buf


# ---- capacity checks of the no-host-sync sparse path ---------------------------------------------------------------------
# Strided sparse layers that run without a host round trip size their outputs by a capacity bound and record
# (device counter, capacity) here.  Whoever synchronises with the host next on the same stream (decode_nms(sync=True),
# ScenePipeline.replay) verifies them, so a frame denser than the bound raises instead of silently dropping sites.
_SPARSE_CHECKS = []


# Sticky form: a captured graph re-creates every n_out counter on each replay, so a check every N replays only sees
# the LAST frame.  The rank-path out_sites kernel also atomicMax-es an overflowing count into this per-device word, which no
# graph ever resets: verify_sparse_capacity() reads it, so an overflow on ANY replayed frame is reported at the next check.
_SPARSE_OVERFLOW = {}


def sparse_overflow_flag(device):
    """int32 [1] on `device`, persistent: max over all strided sparse layers of an n_out that exceeded its capacity (0 = none).
    Allocated at the first (eager) use, i.e. outside any graph capture."""
    key = str(device)
    flag = _SPARSE_OVERFLOW.get(key)
    if flag is None:
        flag = _SPARSE_OVERFLOW[key] = torch.zeros((1,), dtype=torch.int32, device=device)
    return flag


def take_sparse_checks():
    """Hand over (and forget) the pending checks, e.g. to keep those recorded inside a captured graph."""
    global _SPARSE_CHECKS
    out, _SPARSE_CHECKS = _SPARSE_CHECKS, []
    return out


def verify_sparse_capacity(checks=None):
    """Raise if any recorded strided sparse layer produced more active sites than its capacity (synchronises: one small
    D2H copy).  checks=None: verify and clear the pending list."""
    pending = take_sparse_checks() if checks is None else checks
    for key, flag in _SPARSE_OVERFLOW.items():
        worst = int(flag.item())
        if worst:
            flag.zero_()
            raise _capi.HealAmdError(
                f"sparse conv: a strided layer produced {worst} active sites on {key}, more than its capacity in the "
                "no-host-sync mode, on this or an earlier (replayed) frame (sites beyond the capacity were dropped: the result "
                "is invalid).  Feed exact-size voxel inputs or raise the capacity policy (SparseTensor.out_sites)")
    if not pending:
        return
    counts = torch.cat([c.reshape(1) for c, _ in pending]).cpu().tolist()
    for n, (_, cap) in zip(counts, pending):
        if n > cap:
            raise _capi.HealAmdError(
                f"sparse conv: a strided layer produced {n} active sites, more than its capacity {cap} in the no-host-sync "
                "mode (sites beyond the capacity were dropped: the result is invalid).  Feed exact-size voxel inputs or "
                "raise the capacity policy (SparseTensor.out_sites)")


def _host_array(values, ctype):
    arr = (ctype * len(values))(*values)
    return arr


# ------------------------------------------------------------------------------------------------
# K1's per-cell table is self-cleaning (round 6): a workspace that a finished call left behind needs no fill.  One workspace per
# (stream, layout) -- a call with other sizes carves the buffer differently -- and the set of buffers known to be clean.
_VOX_CLEAN = set()


def _vox_cells(lidar_range, voxel_size):
    """Cells of the voxel grid, as the library counts them (round((max - min) / size) per axis in fp64)."""
    c = 1
    for j in range(3):
        c *= int(round((float(np.float32(lidar_range[3 + j])) - float(np.float32(lidar_range[j]))) / float(np.float32(voxel_size[j]))))
    return max(c, 0)


def _vox_workspace(kind, nbytes, dev, layout):
    ws = _workspace(("voxelize", kind) + tuple(layout), nbytes, dev)
    return ws, (ws.data_ptr(), ws.numel(), kind) + tuple(layout)


def _vox_call(name, key, *args):
    """Call a voxelize entry point with tables_clean from the bookkeeping; any failure leaves the workspace 'unknown'."""
    clean = key in _VOX_CLEAN
    _VOX_CLEAN.discard(key)
    _capi.call(name, *args[:-1], 1 if clean else 0, args[-1])
    # A call RECORDED into a graph has not run: if it carries the initial fill (workspace first seen inside a capture), the tables are
    # clean only once that graph has been replayed -- later calls must not rely on it, so the workspace stays 'unknown' (every capture
    # then carries its own fill; the first eager call settles it).
    if clean or not torch.cuda.is_current_stream_capturing():
        _VOX_CLEAN.add(key)


def voxelize(points, lidar_range, voxel_size, max_points, max_voxels, batch_idx=0, sync=True):
    """K1.  points [N,4] f32 cuda -> (voxels [M,P,4], coords [M,4] (b,z,y,x) i32, num_points [M] i32).

    With sync=True (the SpVoxelPreprocessor contract: exact-size outputs) the voxel count is read
    back and the outputs are sliced; with sync=False the full-capacity buffers and the device
    counter are returned: (voxels, coords, num_points, n_voxels_dev)."""
    points = _need(points, torch.float32, "points")
    if points.dim() != 2 or points.shape[1] != 4:
        raise _capi.HealAmdError("points must be [N,4]")
    n = int(points.shape[0])
    cap = max(1, min(n, int(max_voxels)))
    dev = points.device
    voxels = torch.empty((cap, max_points, 4), dtype=torch.float32, device=dev)
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    num = torch.empty((cap,), dtype=torch.int32, device=dev)
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    cells = _vox_cells(lidar_range, voxel_size)
    nbytes = _capi.query("heal_voxelize_workspace", n, int(max_points), int(max_voxels), cells)
    ws, key = _vox_workspace("one", nbytes, dev, (n, int(max_points), int(max_voxels), cells))
    rng = _host_array([float(v) for v in lidar_range], ctypes.c_float)
    vs = _host_array([float(v) for v in voxel_size], ctypes.c_float)
    with _Timed("voxelize"):
        _vox_call("heal_voxelize", key, _ptr(points), n, rng, vs, int(max_points), int(max_voxels), int(batch_idx),
                  _ptr(voxels), _ptr(coords), _ptr(num), _ptr(count), None, None, _ptr(ws), ws.numel(), _stream())
    if not sync:
        return voxels, coords, num, count
    m = int(count.item())
    return voxels[:m], coords[:m], num[:m]


def mask_points(points, limit_range=None, mask_ego=True, out=None):
    """pcd_utils.mask_points_by_range / mask_ego_points on the device: points [N,4] f32 cuda -> same-shape tensor in
    which dropped points are NaN (the voxeliser skips them; order and length are kept, so there is no host sync)."""
    points = _need(points, torch.float32, "points")
    if points.dim() != 2 or points.shape[1] != 4:
        raise _capi.HealAmdError("points must be [N,4]")
    if out is None:
        out = torch.empty_like(points)
    rng = _host_array([float(v) for v in limit_range], ctypes.c_float) if limit_range is not None else None
    _capi.call("heal_mask_points", _ptr(points), int(points.shape[0]), rng, 1 if mask_ego else 0, _ptr(out), _stream())
    return out


def voxelize_collated(point_list, lidar_range, voxel_size, max_points, max_voxels):
    """K1 for every agent of a modality into ONE set of collated buffers (collate_batch_list,
    sp_voxel_preprocessor.py:110-147) without a host round trip: agent b's rows follow agent b-1's, the running row
    offset lives on the device.  -> (voxels [cap,P,4], coords [cap,4] (b,z,y,x), num_points [cap], offsets [n+1] i32
    device; offsets[b+1]-offsets[b] = voxels of agent b, offsets[n] = total); rows beyond the total are unspecified."""
    pts = [_need(p, torch.float32, "points") for p in point_list]
    for p in pts:
        if p.dim() != 2 or p.shape[1] != 4:
            raise _capi.HealAmdError("points must be [N,4]")
    dev = pts[0].device
    caps = [min(int(p.shape[0]), int(max_voxels)) for p in pts]
    cap = max(1, sum(caps))
    voxels = torch.empty((cap, max_points, 4), dtype=torch.float32, device=dev)
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    num = torch.empty((cap,), dtype=torch.int32, device=dev)
    offsets = torch.zeros((len(pts) + 1,), dtype=torch.int32, device=dev)
    rng = _host_array([float(v) for v in lidar_range], ctypes.c_float)
    vs = _host_array([float(v) for v in voxel_size], ctypes.c_float)
    if len(pts) <= 16:
        # ONE launch chain for all agents (heal_voxelize_batch): the clouds are concatenated, keys carry the agent
        bounds = [0]
        for p in pts:
            bounds.append(bounds[-1] + int(p.shape[0]))
        # clouds that already lie back to back in memory (pipeline.StaticInputs) are read in place
        adjacent = all(p.is_contiguous() for p in pts) and all(
            pts[i].data_ptr() + pts[i].numel() * 4 == pts[i + 1].data_ptr() for i in range(len(pts) - 1))
        allp = pts[0] if (len(pts) == 1 or adjacent) else torch.cat(pts, 0)
        axc = len(pts) * _vox_cells(lidar_range, voxel_size)
        nbytes = _capi.query("heal_voxelize_batch_workspace", bounds[-1], len(pts), int(max_points), int(max_voxels), axc)
        # (the carve depends on the total point count, the row capacity = sum of min(n_b, max_voxels), max_points and the table kind)
        ws, key = _vox_workspace("batch", nbytes, dev, (bounds[-1], cap, int(max_points), axc))
        with _Timed("voxelize"):
            _vox_call("heal_voxelize_batch", key, _ptr(allp), _host_array(bounds, ctypes.c_int32), len(pts), rng, vs,
                      int(max_points), int(max_voxels), _ptr(voxels), _ptr(coords), _ptr(num), _ptr(offsets), _ptr(ws),
                      ws.numel(), _stream())
        _remember("voxelize", lambda: voxelize_collated(point_list, lidar_range, voxel_size, max_points, max_voxels))
        return voxels, coords, num, offsets
    counts = torch.zeros((len(pts),), dtype=torch.int32, device=dev)
    for b, p in enumerate(pts):
        n = int(p.shape[0])
        cells = _vox_cells(lidar_range, voxel_size)
        nbytes = _capi.query("heal_voxelize_workspace", n, int(max_points), int(max_voxels), cells)
        ws, key = _vox_workspace("one", nbytes, dev, (n, int(max_points), int(max_voxels), cells))
        with _Timed("voxelize"):
            _vox_call("heal_voxelize", key, _ptr(p), n, rng, vs, int(max_points), int(max_voxels), b, _ptr(voxels),
                      _ptr(coords), _ptr(num), _ptr(counts[b:b + 1]), _ptr(offsets[b:b + 1]),
                      _ptr(offsets[b + 1:b + 2]), _ptr(ws), ws.numel(), _stream())
    return voxels, coords, num, offsets


def pfn_scatter(voxels, coords, num_points, weight, bn_scale, bn_shift, voxel_size, lidar_range,
                n_agents, ny, nx, n_voxels_dev=None, return_pillars=False, out=None):
    """K2.  -> canvas [n_agents,64,ny,nx] (and pillar features [M,64] when asked)."""
    voxels = _need(voxels, torch.float32, "voxels")
    coords = _need(coords, torch.int32, "coords")
    num_points = _need(num_points, torch.int32, "num_points")
    weight = _need(weight, torch.float32, "weight")
    bn_scale = _need(bn_scale, torch.float32, "bn_scale")
    bn_shift = _need(bn_shift, torch.float32, "bn_shift")
    M, P = int(voxels.shape[0]), int(voxels.shape[1])
    C = int(weight.shape[0])
    if weight.shape[1] != 10 or voxels.shape[2] != 4 or coords.shape[1] != 4:
        raise _capi.HealAmdError("pfn_scatter: expected voxels [M,P,4], coords [M,4], weight [C,10]")
    dev = voxels.device
    if out is None:
        canvas = torch.empty((n_agents, C, ny, nx), dtype=torch.float32, device=dev)
    else:
        canvas = out
        if (tuple(canvas.shape) != (n_agents, C, ny, nx) or canvas.dtype != torch.float32
                or not canvas.is_contiguous() or not canvas.is_cuda):
            raise _capi.HealAmdError("pfn_scatter: `out` must be a contiguous f32 cuda [n_agents,C,ny,nx]")
    pillars = torch.empty((max(M, 1), C), dtype=torch.float32, device=dev)
    nbytes = _capi.query("heal_pfn_scatter_workspace", M, n_agents, ny, nx, C)
    ws = _workspace("pfn_scatter", nbytes, dev)
    vx, vy, vz = (float(v) for v in voxel_size)
    xo = vx / 2 + lidar_range[0]
    yo = vy / 2 + lidar_range[1]
    zo = vz / 2 + lidar_range[2]
    with _Timed("pfn_scatter"):
        _capi.call("heal_pfn_scatter", _ptr(voxels), _ptr(coords), _ptr(num_points), M, _ptr(n_voxels_dev), P,
                   _ptr(weight), _ptr(bn_scale), _ptr(bn_shift), C, vx, vy, vz, xo, yo, zo,
                   int(n_agents), int(ny), int(nx), _ptr(canvas), _ptr(pillars), _ptr(ws), ws.numel(), _stream())
    if return_pillars:
        return canvas, pillars[:M]
    return canvas


class PillarBEV:
    """K2's result WITHOUT the dense canvas: pillar feature rows [M, 64] + the cell -> pillar-row map [n_agents, ny, nx] (-1 = empty).
    It stands for PointPillarScatter's [n_agents, 64, ny, nx] canvas (point_pillar_scatter.py:19-76; 96 % zeros) until somebody
    needs it: `stem_block(...)` -- the first BasicBlock of the PointPillars ResNetBEVBackbone reads the pillars through the map
    (heal_pillar_stem_block); `dense()` -- anybody else gets the reference's tensor.  Same protocol as PooledBEV (K4)."""

    def __init__(self, pillars, cell_map, n_agents, ny, nx):
        self.pillars, self.cell_map = pillars, cell_map
        self.n_agents, self.ny, self.nx_ = int(n_agents), int(ny), int(nx)
        self.channels = int(pillars.shape[1])
        self.device = pillars.device

    @property
    def shape(self):
        return (self.n_agents, self.channels, self.ny, self.nx_)

    @property
    def is_cuda(self):
        return True

    def dense(self):
        out = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        with _Timed("pillar_canvas", nbytes=4.0 * out.numel()):
            _capi.call("heal_pillar_canvas", _ptr(self.cell_map), _ptr(self.pillars), self.n_agents, self.channels, self.ny,
                       self.nx_, _ptr(out), _stream())
        return out

    def stem_supported(self, cout_main, cout_down):
        return (self.channels == 64 and cout_main == 64 and cout_down == 64 and ((self.nx_ - 1) // 2 + 1) % 4 == 0)

    # weight layout of stem_block: "lanes" (pillar_stem_fragments, the pixel-compacted kernel) | "tiles" (stem_fragments, v1: A/B)
    weight_layout = "tiles" if os.environ.get("HEAL_PILLAR_STEM", "2") == "1" else "lanes"

    def fragments(self, w_main, w_down):
        return pillar_stem_fragments(w_main, w_down) if self.weight_layout == "lanes" else stem_fragments(w_main, w_down)

    def stem_block(self, w_main, b_main, w_down, b_down):
        """relu(conv3x3_s2(canvas, W1) + b1), conv1x1_s2(canvas, Wd) + bd straight from the pillars; weights from
        self.fragments(...)."""
        Ho, Wo = (self.ny - 1) // 2 + 1, (self.nx_ - 1) // 2 + 1
        out_main = torch.empty((self.n_agents, 64, Ho, Wo), dtype=torch.float32, device=self.device)
        out_id = torch.empty_like(out_main)
        flops = 2.0 * self.n_agents * Ho * Wo * 64 * 64 * 10          # the dense convolutions it replaces
        with _Timed("pillar_stem_block", flops=flops, nbytes=8.0 * out_main.numel(), kernel_events=True):
            _capi.call("heal_pillar_stem_block", _ptr(self.pillars), _ptr(self.cell_map), self.n_agents, self.channels, self.ny,
                       self.nx_, _ptr(w_main), _ptr(b_main), _ptr(w_down), _ptr(b_down), 1 if self.weight_layout == "tiles" else 0,
                       _ptr(out_main), _ptr(out_id), _stream())
        return out_main, out_id


def pillar_stem_fragments(w_main, w_down):
    """Weights of heal_pillar_stem_block (layout 0): w_main [64, 64, 3, 3] -> [9][4][64][16] with frag[tap][w][16 lk + ln][ks] =
    W[16 w + ln][16 lk + ks][tap]; w_down [64, 64, 1, 1] -> [4][64][16] likewise (the 16x16x4 A operand of wave w with the
    reduction index permuted so that a lane's 16 k-steps are 16 consecutive input channels)."""
    if tuple(w_main.shape) != (64, 64, 3, 3) or tuple(w_down.shape) != (64, 64, 1, 1):
        raise _capi.HealAmdError("pillar_stem_fragments: expected [64, 64, 3, 3] and [64, 64, 1, 1]")
    # [co = (w, ln), ci = (lk, ks), tap] -> [tap, w, lk, ln, ks]
    wm = w_main.detach().to(torch.float32).reshape(4, 16, 4, 16, 9).permute(4, 0, 2, 1, 3).contiguous()
    wd = w_down.detach().to(torch.float32).reshape(4, 16, 4, 16).permute(0, 2, 1, 3).contiguous()
    return wm, wd


def pfn_pillars(voxels, coords, num_points, weight, bn_scale, bn_shift, voxel_size, lidar_range, n_agents, ny, nx,
                n_voxels_dev=None):
    """K2 without the canvas -> PillarBEV (pillar features [M, 64] + cell -> pillar map); see pfn_scatter for the arguments."""
    voxels = _need(voxels, torch.float32, "voxels")
    coords = _need(coords, torch.int32, "coords")
    num_points = _need(num_points, torch.int32, "num_points")
    weight = _need(weight, torch.float32, "weight")
    bn_scale = _need(bn_scale, torch.float32, "bn_scale")
    bn_shift = _need(bn_shift, torch.float32, "bn_shift")
    M, P = int(voxels.shape[0]), int(voxels.shape[1])
    C = int(weight.shape[0])
    if weight.shape[1] != 10 or voxels.shape[2] != 4 or coords.shape[1] != 4:
        raise _capi.HealAmdError("pfn_pillars: expected voxels [M,P,4], coords [M,4], weight [C,10]")
    dev = voxels.device
    pillars = torch.empty((max(M, 1), C), dtype=torch.float32, device=dev)
    cell_map = torch.empty((n_agents, ny, nx), dtype=torch.int32, device=dev)
    vx, vy, vz = (float(v) for v in voxel_size)
    xo, yo, zo = vx / 2 + lidar_range[0], vy / 2 + lidar_range[1], vz / 2 + lidar_range[2]
    with _Timed("pfn_pillars", nbytes=16.0 * M * P + 20.0 * M + 4.0 * C * M):
        _capi.call("heal_pfn_pillars", _ptr(voxels), _ptr(coords), _ptr(num_points), M, _ptr(n_voxels_dev), P,
                   _ptr(weight), _ptr(bn_scale), _ptr(bn_shift), C, vx, vy, vz, xo, yo, zo, int(n_agents), int(ny), int(nx),
                   _ptr(pillars), _ptr(cell_map), _stream())
    return PillarBEV(pillars, cell_map, n_agents, ny, nx)


def _pfn_geom(voxel_size, lidar_range):
    vx, vy, vz = (float(v) for v in voxel_size)
    return vx, vy, vz, vx / 2 + lidar_range[0], vy / 2 + lidar_range[1], vz / 2 + lidar_range[2]


def _pfn_inputs(voxels, coords, num_points):
    voxels = _need(voxels, torch.float32, "voxels")
    coords = _need(coords, torch.int32, "coords")
    num_points = _need(num_points, torch.int32, "num_points")
    if voxels.dim() != 3 or voxels.shape[2] != 4 or coords.shape[1] != 4 or int(voxels.shape[1]) > 32:
        raise _capi.HealAmdError("pfn (training kernels): expected voxels [M,P<=32,4], coords [M,4]")
    return voxels, coords, num_points


def pfn_train_supported(voxels, weight=None):
    """heal_pfn_features / _moments / _backward are written for Linear(10 -> 64) (PillarVFE with num_filters [64]); any other
    width must take the torch path (the kernels would read weight / bn rows out of bounds)."""
    if weight is not None and tuple(weight.shape) != (64, 10):
        return False
    return voxels.is_cuda and voxels.dim() == 3 and 1 <= int(voxels.shape[1]) <= 32 and int(voxels.shape[0]) >= 1


def pfn_moments(voxels, coords, num_points, voxel_size, lidar_range):
    """-> (s1 [10], S [10,10]) float64: sums over all M x P rows of the decorated, masked point features f and of f f^T."""
    voxels, coords, num_points = _pfn_inputs(voxels, coords, num_points)
    M, P = int(voxels.shape[0]), int(voxels.shape[1])
    nb = _capi.query("heal_pfn_train_blocks", M)
    part = torch.empty((nb, 65), dtype=torch.float32, device=voxels.device)
    _capi.call("heal_pfn_moments", _ptr(voxels), _ptr(coords), _ptr(num_points), M, P, *_pfn_geom(voxel_size, lidar_range),
               _ptr(part), _stream())
    tot = part.double().sum(0)
    S = torch.zeros((10, 10), dtype=torch.float64, device=voxels.device)
    iu = torch.triu_indices(10, 10, device=voxels.device)
    S[iu[0], iu[1]] = tot[10:]
    S = S + S.t() - torch.diag(torch.diag(S))
    return tot[:10], S


def pfn_features(voxels, coords, num_points, weight, bn_scale, bn_shift, voxel_size, lidar_range):
    """Pillar features [M,64] = max over the points of relu(scale (W f) + shift): K2's first stage without the canvas."""
    voxels, coords, num_points = _pfn_inputs(voxels, coords, num_points)
    M, P = int(voxels.shape[0]), int(voxels.shape[1])
    out = torch.empty((M, 64), dtype=torch.float32, device=voxels.device)
    if tuple(weight.shape) != (64, 10) or bn_scale.numel() != 64 or bn_shift.numel() != 64:
        raise _capi.HealAmdError("pfn_features: the kernel is written for Linear(10 -> 64) (weight [64, 10], 64 BN channels)")
    _capi.call("heal_pfn_features", _ptr(voxels), _ptr(coords), _ptr(num_points), M, P, _ptr(_need(weight, torch.float32, "weight")),
               _ptr(_need(bn_scale, torch.float32, "bn_scale")), _ptr(_need(bn_shift, torch.float32, "bn_shift")),
               *_pfn_geom(voxel_size, lidar_range), _ptr(out), _stream())
    return out


def pfn_backward(voxels, coords, num_points, weight, bn_scale, bn_shift, mean, rstd, voxel_size, lidar_range, grad_pillar):
    """-> (A [64,10], B [64], Cx [64]) float64: sum dy f_{p*}, sum dy, sum dy xhat_{p*} over the pillars (include/heal_amd.h)."""
    voxels, coords, num_points = _pfn_inputs(voxels, coords, num_points)
    M, P = int(voxels.shape[0]), int(voxels.shape[1])
    nb = _capi.query("heal_pfn_train_blocks", M)
    part = torch.empty((nb, 64, 12), dtype=torch.float32, device=voxels.device)
    if tuple(weight.shape) != (64, 10) or bn_scale.numel() != 64 or bn_shift.numel() != 64:
        raise _capi.HealAmdError("pfn_backward: the kernel is written for Linear(10 -> 64) (weight [64, 10], 64 BN channels)")
    _capi.call("heal_pfn_backward", _ptr(voxels), _ptr(coords), _ptr(num_points), M, P, _ptr(_need(weight, torch.float32, "weight")),
               _ptr(_need(bn_scale, torch.float32, "bn_scale")), _ptr(_need(bn_shift, torch.float32, "bn_shift")),
               _ptr(_need(mean, torch.float32, "mean")), _ptr(_need(rstd, torch.float32, "rstd")),
               *_pfn_geom(voxel_size, lidar_range), _ptr(_need(grad_pillar, torch.float32, "grad_pillar")), _ptr(part), _stream())
    tot = part.double().sum(0)
    return tot[:, :10], tot[:, 10], tot[:, 11]


def _affine_args(affine_rows, n):
    """-> (keep-alive object, host pointer, device pointer).  A CUDA tensor of affine rows stays on the device (no host
    round trip; a captured graph then reads the poses at replay time); anything else is passed by value from the host."""
    if isinstance(affine_rows, torch.Tensor) and affine_rows.is_cuda:
        a = affine_rows.detach()
        if a.dtype != torch.float64:
            a = a.double()
        a = a.reshape(-1, 6).contiguous()
        if int(a.shape[0]) != n:
            raise _capi.HealAmdError(f"affine rows: expected {n} x (2,3), got {tuple(affine_rows.shape)}")
        return a, ctypes.c_void_p(0), ctypes.c_void_p(a.data_ptr())
    if isinstance(affine_rows, torch.Tensor):
        affine_rows = affine_rows.detach().numpy()
    a = np.ascontiguousarray(np.asarray(affine_rows, dtype=np.float64).reshape(-1, 6))
    if a.shape[0] != n:
        raise _capi.HealAmdError(f"affine rows: expected {n} x (2,3), got {a.shape}")
    return a, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(0)


def _crop_host(crop, n):
    if crop is None:
        return None, ctypes.c_void_p(0)
    c = np.ascontiguousarray(np.asarray(crop, dtype=np.int32).reshape(n, 4))
    return c, c.ctypes.data_as(ctypes.c_void_p)


def warp_fuse(feats, occ, affine_rows, grid_f64=True, crop=None):
    """K5 fused.  feats [n,C,H,W], occ [n,1,H,W] logits, affine_rows [n,2,3] (host array, or a CUDA tensor that is then
    read on the device at run time) -> [C,H,W]."""
    feats = _need(feats, torch.float32, "feats")
    occ = _need(occ, torch.float32, "occ")
    n, C, H, W = (int(v) for v in feats.shape)
    out = torch.empty((C, H, W), dtype=torch.float32, device=feats.device)
    a, ap, adev = _affine_args(affine_rows, n)
    c, cp = _crop_host(crop, n)
    with _Timed(f"warp_fuse_c{C}", 0.0, 4.0 * H * W * (n * (C + 1) + C)):   # read every agent's map + score, write one
        _capi.call("heal_warp_fuse", _ptr(feats), _ptr(occ), n, C, H, W, ap, adev, int(bool(grid_f64)), cp,
                   _ptr(out), _stream())
    return out


def warp_fuse_levels(feats_list, occ_list, affine_rows, grid_f64=True, crops=None):
    """K5 for every pyramid level of one scene in ONE launch (heal_warp_fuse_levels: LDS-staged source footprints).
    feats_list[l] [n,C_l,H_l,W_l], occ_list[l] [n,1,H_l,W_l] logits, affine_rows [n,2,3] (host array or CUDA tensor, shared by
    the levels), crops[l] = per-agent (h0,h1,w0,w1) | None -> list of [C_l,H_l,W_l]."""
    L = len(feats_list)
    feats_list = [_need(f, torch.float32, "feats") for f in feats_list]
    occ_list = [_need(o, torch.float32, "occ") for o in occ_list]
    n = int(feats_list[0].shape[0])
    outs, nbytes = [], 0.0
    for f, o in zip(feats_list, occ_list):
        if int(f.shape[0]) != n or tuple(o.shape) != (n, 1, int(f.shape[2]), int(f.shape[3])):
            raise _capi.HealAmdError("warp_fuse_levels: inconsistent level shapes")
        C, H, W = (int(v) for v in f.shape[1:])
        outs.append(torch.empty((C, H, W), dtype=torch.float32, device=f.device))
        nbytes += 4.0 * H * W * (n * (C + 1) + C)
    a, ap, adev = _affine_args(affine_rows, n)
    cp, carr = ctypes.c_void_p(0), None
    if crops is not None and any(c is not None for c in crops):
        carr = np.zeros((L, n, 4), dtype=np.int32)
        for l, c in enumerate(crops):
            if c is not None:
                carr[l] = np.asarray([ci if ci is not None else (0, 0, 0, 0) for ci in c], dtype=np.int32).reshape(n, 4)
        cp = carr.ctypes.data_as(ctypes.c_void_p)
    fp = _host_array([f.data_ptr() for f in feats_list], ctypes.c_void_p)
    op = _host_array([o.data_ptr() for o in occ_list], ctypes.c_void_p)
    yp = _host_array([y.data_ptr() for y in outs], ctypes.c_void_p)
    with _Timed("warp_fuse_levels", 0.0, nbytes, kernel_events=True):
        _capi.call("heal_warp_fuse_levels", L, fp, op, n, _host_array([int(f.shape[1]) for f in feats_list], ctypes.c_int32),
                   _host_array([int(f.shape[2]) for f in feats_list], ctypes.c_int32),
                   _host_array([int(f.shape[3]) for f in feats_list], ctypes.c_int32), ap, adev, int(bool(grid_f64)), cp, yp,
                   _stream())
    return outs


def warp_fuse_backward(feats, occ, affine_rows, grad_out, grid_f64=True, crop=None):
    """Gradient of warp_fuse with respect to (feats, occ): grad_out [C,H,W] -> ([n,C,H,W], [n,1,H,W])."""
    feats = _need(feats, torch.float32, "feats")
    occ = _need(occ, torch.float32, "occ")
    grad_out = _need(grad_out, torch.float32, "grad_out")
    n, C, H, W = (int(v) for v in feats.shape)
    g_feats, g_occ = torch.zeros_like(feats), torch.zeros_like(occ)
    a, ap, adev = _affine_args(affine_rows, n)
    c, cp = _crop_host(crop, n)
    with _Timed(f"warp_fuse_backward_c{C}"):
        _capi.call("heal_warp_fuse_backward", _ptr(feats), _ptr(occ), n, C, H, W, ap, adev, int(bool(grid_f64)), cp,
                   _ptr(grad_out), _ptr(g_feats), _ptr(g_occ), _stream())
    return g_feats, g_occ


def warp_agent(feat, occ, affine_row, grid_f64=True, crop=None, out=None):
    """K5 split, rank-local half: feat [C,H,W], occ [1,H,W] -> (feat_ego [C,H,W], score_ego [1,H,W]).
    out = (feat_ego, score_ego): contiguous fp32 destinations to write into (the agent's row of the exchange buffer)."""
    feat = _need(feat, torch.float32, "feat")
    occ = _need(occ, torch.float32, "occ")
    C, H, W = (int(v) for v in feat.shape[-3:])
    if out is not None:
        feat_ego, score_ego = out
        if not (feat_ego.is_contiguous() and score_ego.is_contiguous() and feat_ego.dtype == score_ego.dtype == torch.float32
                and feat_ego.numel() == C * H * W and score_ego.numel() == H * W and feat_ego.is_cuda):
            raise _capi.HealAmdError("warp_agent: `out` must be contiguous fp32 CUDA tensors of C*H*W and H*W elements")
    else:
        feat_ego = torch.empty((C, H, W), dtype=torch.float32, device=feat.device)
        score_ego = torch.empty((1, H, W), dtype=torch.float32, device=feat.device)
    a, ap, adev = _affine_args(affine_row, 1)
    c, cp = _crop_host(crop, 1)
    _capi.call("heal_warp_agent", _ptr(feat), _ptr(occ), C, H, W, ap, adev, int(bool(grid_f64)), cp,
               _ptr(feat_ego), _ptr(score_ego), _stream())
    return feat_ego, score_ego


def warp_agents_pm(feats, affine_rows, grid_f64=True):
    """warp_affine_simple + permute(0, 2, 3, 1) for one scene: feats [n,C,H,W] -> ego-frame maps, token-major [n,H,W,C]."""
    feats = _need(feats, torch.float32, "feats")
    n, C, H, W = (int(v) for v in feats.shape)
    out = torch.empty((n, H, W, C), dtype=torch.float32, device=feats.device)
    a, ap, adev = _affine_args(affine_rows, n)
    with _Timed("warp_agents_pm", 0.0, 8.0 * n * C * H * W):
        _capi.call("heal_warp_agents_pm", _ptr(feats), n, C, H, W, ap, adev, int(bool(grid_f64)), _ptr(out), _stream())
    return out


def fuse_warped_rows(base, feat_offsets, score_offsets, C, H, W):
    """K5 split, post-exchange half on the exchange buffer IN PLACE: agent a's [C,H,W] features at base.flatten()[feat_offsets[a]:],
    scores at score_offsets[a] (element offsets, multiples of 4) -> [C,H,W].  No re-pack of the gathered rows."""
    base = _need(base, torch.float32, "base")
    n = len(feat_offsets)
    out = torch.empty((C, H, W), dtype=torch.float32, device=base.device)
    _capi.call("heal_fuse_warped_rows", _ptr(base), _host_array([int(v) for v in feat_offsets], ctypes.c_int64),
               _host_array([int(v) for v in score_offsets], ctypes.c_int64), n, int(C), int(H), int(W), _ptr(out), _stream())
    return out


def fuse_warped(feats_ego, scores_ego):
    """K5 split, post-all-gather half: [n,C,H,W], [n,1,H,W] -> [C,H,W]."""
    feats_ego = _need(feats_ego, torch.float32, "feats_ego")
    scores_ego = _need(scores_ego, torch.float32, "scores_ego")
    n, C, H, W = (int(v) for v in feats_ego.shape)
    out = torch.empty((C, H, W), dtype=torch.float32, device=feats_ego.device)
    _capi.call("heal_fuse_warped", _ptr(feats_ego), _ptr(scores_ego), n, C, H, W, _ptr(out), _stream())
    return out


def decode_nms(cls, reg, dirp, anchors, score_thr, dir_offset, num_bins, nms_thr, tfm, gt_range,
               nms_top=1000, sync=True):
    """K8.  cls [1,A,H,W], reg [1,7A,H,W], dirp [1,bins*A,H,W] or None, anchors [H,W,A,7] f32 cuda.
    -> (corners [K,8,3], scores [K]) or (None, None) when nothing survives (sync=True), or the
    full-capacity buffers plus the device count (sync=False)."""
    cls = _need(cls, torch.float32, "cls_preds")
    reg = _need(reg, torch.float32, "reg_preds")
    anchors = _need(anchors, torch.float32, "anchors")
    if dirp is not None:
        dirp = _need(dirp, torch.float32, "dir_preds")
    if cls.dim() == 4:
        if cls.shape[0] != 1:
            raise _capi.HealAmdError("decode_nms: batch size must be 1 (voxel_postprocessor.py:314)")
        cls, reg = cls[0], reg[0]
        dirp = dirp[0] if dirp is not None else None
    A, H, W = (int(v) for v in cls.shape)
    if anchors.numel() != H * W * A * 7:
        raise _capi.HealAmdError(f"decode_nms: anchors {tuple(anchors.shape)} do not match the {A} x {H} x {W} score map "
                                 "(anchor_args.feature_stride of the YAML vs. the model's output stride)")
    dev = cls.device
    out_c = torch.empty((nms_top, 8, 3), dtype=torch.float32, device=dev)
    out_s = torch.empty((nms_top,), dtype=torch.float32, device=dev)
    out_n = torch.zeros((1,), dtype=torch.int32, device=dev)
    nbytes = _capi.query("heal_decode_nms_workspace", A * H * W, int(nms_top))
    ws = _workspace("decode_nms", nbytes, dev)
    t = _host_array([float(v) for v in np.asarray(tfm, dtype=np.float32).reshape(-1)], ctypes.c_float)
    g = _host_array([float(v) for v in gt_range], ctypes.c_float)
    with _Timed("decode_nms"):
        _capi.call("heal_decode_nms", _ptr(cls), _ptr(reg), _ptr(dirp), _ptr(anchors), H, W, A, int(num_bins),
                   float(score_thr), float(dir_offset), float(nms_thr), int(nms_top), t, g,
                   _ptr(out_c), _ptr(out_s), _ptr(out_n), int(nms_top), _ptr(ws), ws.numel(), _stream())
    _remember("decode_nms", lambda: decode_nms(cls, reg, dirp, anchors, score_thr, dir_offset, num_bins, nms_thr, tfm, gt_range,
                                               nms_top, sync=False))
    if not sync:
        return out_c, out_s, out_n
    k = int(out_n.item())
    if _SPARSE_CHECKS and not torch.cuda.is_current_stream_capturing():
        verify_sparse_capacity()   # first host sync after a no-sync SECOND encoder: its capacity counters are final now
    if k == 0:
        return None, None
    return out_c[:k], out_s[:k]


def quad_iou(a, b):
    """Pairwise rotated IoU of quads a [n,4,2], b [m,4,2] (f32 cuda) -> [n,m]."""
    a = _need(a, torch.float32, "a")
    b = _need(b, torch.float32, "b")
    n, m = int(a.shape[0]), int(b.shape[0])
    out = torch.empty((n, m), dtype=torch.float32, device=a.device)
    _capi.call("heal_quad_iou", _ptr(a), n, _ptr(b), m, _ptr(out), _stream())
    return out


_WINDOW_ATTN_SHAPES = {(4, 16), (8, 32), (16, 64), (4, 32), (8, 16), (8, 64), (4, 64)}


def window_attention_supported(window, dim_head, H, W):
    return (int(window), int(dim_head)) in _WINDOW_ATTN_SHAPES and H % window == 0 and W % window == 0


def window_attention(qkv, pos_bias, heads, dim_head, window, scale, out=None):
    """Fused window attention: qkv [L,H,W,3*heads*dim_head] (packed q|k|v), pos_bias [T,T] or None -> [L,H,W,heads*dim_head]
    (written into `out` when given: a contiguous [L,H,W,heads*dim_head] view)."""
    qkv = _need(qkv, torch.float32, "qkv")
    L, H, W, C3 = (int(v) for v in qkv.shape)
    if C3 != 3 * heads * dim_head or not window_attention_supported(window, dim_head, H, W):
        raise _capi.HealAmdError(f"window_attention: unsupported shape (window {window}, dim_head {dim_head}, map {H}x{W})")
    if pos_bias is not None:
        pos_bias = _need(pos_bias, torch.float32, "pos_bias")
        if tuple(pos_bias.shape) != (window * window, window * window):
            raise _capi.HealAmdError("window_attention: pos_bias must be [window^2, window^2]")
    if out is None:
        out = torch.empty((L, H, W, heads * dim_head), dtype=torch.float32, device=qkv.device)
    elif not (out.is_contiguous() and out.numel() == L * H * W * heads * dim_head and out.dtype == torch.float32):
        raise _capi.HealAmdError("window_attention: `out` must be a contiguous f32 [L,H,W,heads*dim_head] buffer")
    # per token: q k^T and p v = 4 window^2 inner FLOPs; q | k | v read + the result written = 16 inner bytes
    inner = heads * dim_head
    with _Timed(f"window_attention_ws{window}", flops=4.0 * L * H * W * window * window * inner, nbytes=16.0 * L * H * W * inner):
        _capi.call("heal_window_attention", _ptr(qkv), _optr(pos_bias), L, H, W, int(heads), int(dim_head), int(window),
                   float(scale), _ptr(out), _stream())
    return out


def window_attention_backward(qkv, pos_bias, out, grad_out, heads, dim_head, window, scale):
    """Backward of window_attention -> (grad_qkv like qkv, grad_bias [T,T] | None).  Unmeasured so far: the modules use it only
    under HEAL_WATTN_GRAD=kernel."""
    qkv = _need(qkv, torch.float32, "qkv"); out = _need(out, torch.float32, "out"); grad_out = _need(grad_out, torch.float32, "grad_out")
    L, H, W, C3 = (int(v) for v in qkv.shape)
    if C3 != 3 * heads * dim_head or not window_attention_supported(window, dim_head, H, W):
        raise _capi.HealAmdError(f"window_attention_backward: unsupported shape (window {window}, dim_head {dim_head}, map {H}x{W})")
    grad_qkv = torch.empty_like(qkv)
    grad_bias = None
    if pos_bias is not None:
        pos_bias = _need(pos_bias, torch.float32, "pos_bias")
        grad_bias = torch.zeros_like(pos_bias)
    need = _capi.query("heal_window_attention_backward_workspace", L, H, W, int(heads))
    ws = _workspace("window_attention_bwd", need, qkv.device)
    _capi.call("heal_window_attention_backward", _ptr(qkv), _optr(pos_bias), _ptr(out), _ptr(grad_out), L, H, W, int(heads),
               int(dim_head), int(window), float(scale), _ptr(grad_qkv), _optr(grad_bias), _ptr(ws), need, _stream())
    return grad_qkv, grad_bias


class WindowAttention(torch.autograd.Function):
    """window_attention under autograd: forward = K6b, backward = heal_window_attention_backward (saves qkv, the bias table and
    the output; the T x T probabilities are recomputed per window)."""

    @staticmethod
    def forward(ctx, qkv, pos_bias, heads, dim_head, window, scale):
        qkv = qkv.contiguous()
        out = window_attention(qkv, pos_bias, heads, dim_head, window, scale)
        ctx.save_for_backward(qkv, pos_bias, out)
        ctx.cfg = (int(heads), int(dim_head), int(window), float(scale))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        qkv, pos_bias, out = ctx.saved_tensors
        heads, dim_head, window, scale = ctx.cfg
        gq, gb = window_attention_backward(qkv, pos_bias, out, grad_out.contiguous(), heads, dim_head, window, scale)
        return gq, gb, None, None, None, None


def label_assign(anchor_boxes, gt_boxes, pos_threshold, neg_threshold):
    """Anchor labelling core of generate_label: stand-up boxes [N,4] / [G,4] f32 cuda -> (assigned [N] i32: gt index of a
    positive anchor or -1, neg [N] u8)."""
    a = _need(anchor_boxes, torch.float32, "anchor_boxes")
    g = _need(gt_boxes, torch.float32, "gt_boxes")
    if a.dim() != 2 or a.shape[1] != 4 or g.dim() != 2 or g.shape[1] != 4:
        raise _capi.HealAmdError("label_assign: boxes must be [n,4]")
    n, k = int(a.shape[0]), int(g.shape[0])
    assigned = torch.empty((n,), dtype=torch.int32, device=a.device)
    neg = torch.empty((n,), dtype=torch.uint8, device=a.device)
    need = _capi.query("heal_label_assign_workspace", k)
    ws = _workspace("label_assign", need, a.device)
    _capi.call("heal_label_assign", _ptr(a), n, _ptr(g) if k else None, k, float(pos_threshold), float(neg_threshold),
               _ptr(assigned), _ptr(neg), _ptr(ws), need, _stream())
    return assigned, neg


_BEV_MODES = {"overlap": 0, "iou": 1, "iou_normal": 2}


def boxes_bev_matrix(boxes_a, boxes_b, mode="iou"):
    """pcdet BEV box ops: boxes [n,7] / [m,7] f32 cuda (x, y, z, dx, dy, dz, heading) -> [n,m] overlap area
    ('overlap'), rotated IoU ('iou') or axis-aligned IoU ('iou_normal')."""
    a = _need(boxes_a, torch.float32, "boxes_a")
    b = _need(boxes_b, torch.float32, "boxes_b")
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != 7 or b.shape[1] != 7:
        raise _capi.HealAmdError("boxes_bev_matrix: boxes must be [n,7]")
    n, m = int(a.shape[0]), int(b.shape[0])
    out = torch.zeros((n, m), dtype=torch.float32, device=a.device)
    _capi.call("heal_boxes_bev_matrix", _ptr(a), n, _ptr(b), m, _BEV_MODES[mode], _ptr(out), _stream())
    return out


def nms_bev(boxes_sorted, thresh, rotated=True):
    """Greedy pcdet NMS over boxes [n,7] already in descending-score order -> (keep [n] i64, count [1] i32), both on
    the device; keep[:count] are the surviving indices in ascending order."""
    b = _need(boxes_sorted, torch.float32, "boxes_sorted")
    if b.dim() != 2 or b.shape[1] != 7:
        raise _capi.HealAmdError("nms_bev: boxes must be [n,7]")
    n = int(b.shape[0])
    keep = torch.empty((n,), dtype=torch.int64, device=b.device)
    count = torch.zeros((1,), dtype=torch.int32, device=b.device)
    need = _capi.query("heal_nms_bev_workspace", n)
    ws = _workspace("nms_bev", need, b.device)
    _capi.call("heal_nms_bev", _ptr(b), n, float(thresh), int(bool(rotated)), _ptr(ws), need, _ptr(keep), _ptr(count),
               _stream())
    return keep, count


def camera_matrices(rots, trans, intrins, post_rots, post_trans):
    """[B,N,3,3] / [B,N,3] device tensors -> cam_mats [B*N,27] f32 (rots @ inv(intrins) | inv(post_rots) | post_trans |
    trans | 0): the 3x3 algebra of get_geometry in one launch."""
    f = lambda t, name: _need(t.float() if t.dtype != torch.float32 else t, torch.float32, name)
    rots, trans, intrins = f(rots, "rots"), f(trans, "trans"), f(intrins, "intrins")
    post_rots, post_trans = f(post_rots, "post_rots"), f(post_trans, "post_trans")
    n = int(trans.numel() // 3)
    if rots.numel() != 9 * n or intrins.numel() != 9 * n or post_rots.numel() != 9 * n or post_trans.numel() != 3 * n:
        raise _capi.HealAmdError("camera_matrices: inconsistent shapes")
    out = torch.empty((n, 27), dtype=torch.float32, device=trans.device)
    _capi.call("heal_camera_matrices", _ptr(rots), _ptr(trans), _ptr(intrins), _ptr(post_rots), _ptr(post_trans), n,
               _ptr(out), _stream())
    return out


def nms_quads(quads_sorted, thresh):
    """Greedy rotated NMS over quads [n,4,2] f32 cuda already in descending-score order -> (keep [n] i64, count [1] i32)
    on the device; keep[:count] are indices into the given order, in pick order."""
    q = _need(quads_sorted, torch.float32, "quads_sorted")
    if q.dim() != 3 or tuple(q.shape[1:]) != (4, 2):
        raise _capi.HealAmdError("nms_quads: quads must be [n,4,2]")
    n = int(q.shape[0])
    keep = torch.empty((n,), dtype=torch.int64, device=q.device)
    count = torch.zeros((1,), dtype=torch.int32, device=q.device)
    need = _capi.query("heal_nms_quads_workspace", n)
    ws = _workspace("nms_quads", need, q.device)
    _capi.call("heal_nms_quads", _ptr(q), n, float(thresh), _ptr(ws), need, _ptr(keep), _ptr(count), _stream())
    return keep, count


_ZWS = {}


def _workspace_zeroed(key, nbytes, device):
    """Scratch with a zero-on-entry / zero-on-exit contract (heal_bev_pool_pm): allocated zero-filled ONCE and then owned by
    the operator, which leaves it clean after every call.  Outgrown buffers are retired, not freed (captured graphs)."""
    k = (key, device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _ZWS.get(k)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _WS_RETIRED.append(buf)
        buf = torch.zeros(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ZWS[k] = buf
    return buf


_FRUSTUM_OK = {}


def _frustum_separable(frustum):
    """True if frustum[d][v][u] == (xs[u], ys[v], ds[d]) -- what create_frustum builds (heter_encoders.py:110-123) and what
    heal_bev_pool_pm assumes.  Checked once per tensor (a host sync; not during graph capture)."""
    key = (frustum.data_ptr(), frustum._version, tuple(frustum.shape))
    hit = _FRUSTUM_OK.get(key)
    if hit is None:
        if torch.cuda.is_current_stream_capturing():
            raise _capi.HealAmdError("bev_pool: first use of a frustum tensor inside a graph capture (run one eager step first)")
        D, fH, fW, _ = frustum.shape
        sep = torch.stack((frustum[0, 0, :, 0].view(1, 1, fW).expand(D, fH, fW),
                           frustum[0, :, 0, 1].view(1, fH, 1).expand(D, fH, fW),
                           frustum[:, 0, 0, 2].view(D, 1, 1).expand(D, fH, fW)), -1)
        if len(_FRUSTUM_OK) > 64:
            _FRUSTUM_OK.clear()
        hit = (bool(torch.equal(sep, frustum)), frustum)  # keep the tensor alive: the key is its address
        _FRUSTUM_OK[key] = hit
    return hit[0]


def bev_pool_pm_supported(D, fH, C):
    import os
    mtot = (D + 15) // 16 * 16
    fh4 = (fH + 3) // 4 * 4
    lds = (fh4 * (C + 16) + 2 * mtot * 66) * 4   # LssLds, bev_pool.hip
    return (os.environ.get("HEAL_LSS_PATH", "") != "sorted" and fH <= 64 and D <= 64 and D % 4 == 0 and C % 16 == 0
            and 16 <= C <= 256 and lds <= 150 * 1024)


class PooledBEV:
    """The SPARSE PIXEL-MAJOR BEV map heal_bev_pool_scatter leaves in its workspace: rows[cell][C] for the cells any lifted point
    fell into + a generation-tagged flag per cell (include/heal_amd.h, K4).  It stands for the reference's dense
    [n_agents, C*nz, ny, nx] `voxel_pooling` result (heter_encoders.py:161-217) until somebody needs it:

      * `stem_block(...)` -- the first BasicBlock of the camera ResNetBEVBackbone reads the rows through the flags, the dense
        canvas is never written;
      * `dense()` -- everybody else gets the reference's tensor (one streaming launch).

    Exactly one of the two may be called, once: the consumer hands the workspace half back to the next scatter."""

    def __init__(self, ws, n_agents, channels, nx):
        self.ws, self.n_agents, self.channels, self.nx = ws, int(n_agents), int(channels), [int(v) for v in nx]
        self.device = ws.device
        self._consumed = False

    @property
    def shape(self):
        return (self.n_agents, self.channels * self.nx[2], self.nx[1], self.nx[0])

    @property
    def is_cuda(self):
        return True

    def _take(self):
        if self._consumed:
            raise _capi.HealAmdError("PooledBEV: the pooled map was already consumed (one consumer per bev_pool scatter)")
        self._consumed = True

    def dense(self):
        self._take()
        out = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        with _Timed("bev_pool_emit"):
            _capi.call("heal_bev_pool_emit", self.n_agents, self.channels, _host_array(self.nx, ctypes.c_int32), _ptr(out),
                       _ptr(self.ws), self.ws.numel(), _stream())
        return out

    def stem_supported(self, cout_main, cout_down):
        return (self.nx[2] == 1 and self.channels % 32 == 0 and cout_main == 64 and cout_down == 64
                and ((self.nx[0] - 1) // 2 + 1) % 4 == 0)

    def stem_block(self, w_main, b_main, w_down, b_down):
        """relu(conv3x3_s2(x, W1) + b1), conv1x1_s2(x, Wd) + bd from the sparse map; w_main / w_down in the layouts of
        stem_fragments()."""
        self._take()
        Ho, Wo = (self.nx[1] - 1) // 2 + 1, (self.nx[0] - 1) // 2 + 1
        out_main = torch.empty((self.n_agents, 64, Ho, Wo), dtype=torch.float32, device=self.device)
        out_id = torch.empty_like(out_main)
        cin = self.channels
        flops = 2.0 * self.n_agents * Ho * Wo * 64 * cin * 10
        with _Timed("bev_stem_block", flops=flops, kernel_events=True):
            _capi.call("heal_bev_stem_block", self.n_agents, cin, _host_array(self.nx, ctypes.c_int32), _ptr(w_main),
                       _ptr(b_main), _ptr(w_down), _ptr(b_down), _ptr(out_main), _ptr(out_id), _ptr(self.ws), self.ws.numel(),
                       _stream())
        return out_main, out_id


def stem_fragments(w_main, w_down):
    """Weights of heal_bev_stem_block: w_main [64, C, 3, 3] -> [9][C/K][64][K] (tap = ky*3 + kx, K-channel chunk, cout,
    channel in chunk); w_down [64, C, 1, 1] -> [C/K][64][K]; K = 64 when C % 64 == 0, else 32."""
    co, c = int(w_main.shape[0]), int(w_main.shape[1])
    if co != 64 or c % 32 != 0 or tuple(w_main.shape[2:]) != (3, 3) or tuple(w_down.shape) != (64, c, 1, 1):
        raise _capi.HealAmdError("stem_fragments: expected [64, C, 3, 3] and [64, C, 1, 1] with C % 32 == 0")
    k = 64 if c % 64 == 0 else 32          # K chunk of k_bev_stem
    wm = w_main.detach().to(torch.float32).reshape(64, c // k, k, 9).permute(3, 1, 0, 2).contiguous()
    wd = w_down.detach().to(torch.float32).reshape(64, c // k, k).permute(1, 0, 2).contiguous()
    return wm, wd


def bev_pool_pm(head, C, D, fH, fW, frustum, cam_mats, n_agents, n_cams, dx, bx, nx, pooled=False):
    """K4, production path.  head [n_agents*n_cams, fH*fW, >= C + D] f32 cuda, PIXEL-MAJOR: per pixel the C image features
    followed by the D depth logits (what conv1x1(..., pixel_major=True) of the fused image_head | depth_head weight
    writes); frustum [D,fH,fW,3]; cam_mats [n_agents*n_cams,27] -> [n_agents, C*nz, ny, nx], or with pooled=True the
    PooledBEV hand-off (sparse pixel-major map; its consumer decides whether the dense tensor is ever written)."""
    head = _need(head, torch.float32, "head")
    frustum = _need(frustum, torch.float32, "frustum")
    cam_mats = _need(cam_mats, torch.float32, "cam_mats")
    BN, HW, CT = (int(v) for v in head.shape)
    if (BN != n_agents * n_cams or HW != fH * fW or CT < C + D or tuple(frustum.shape) != (D, fH, fW, 3)
            or tuple(cam_mats.shape) != (BN, 27)):
        raise _capi.HealAmdError("bev_pool_pm: inconsistent shapes")
    if not bev_pool_pm_supported(D, fH, C) or not _frustum_separable(frustum):
        raise _capi.HealAmdError("bev_pool_pm: shape / frustum outside the fused path (use bev_pool)")
    nxi = [int(v) for v in nx]
    dev = head.device
    nbytes = _capi.query("heal_bev_pool_pm_workspace", n_agents, C, nxi[0], nxi[1], nxi[2])
    # one scratch per problem shape: the two-half invariant holds for ONE carving of the buffer only
    ws = _workspace_zeroed(("bev_pool_pm", n_agents, C, nxi[0], nxi[1], nxi[2]), nbytes, dev)
    nbytes_alg = 4.0 * (BN * HW * (C + D)) + 4.0 * n_agents * C * nxi[0] * nxi[1] * nxi[2]
    with _Timed("bev_pool", nbytes=nbytes_alg, kernel_events=True):
        _capi.call("heal_bev_pool_scatter", _ptr(head), CT, _ptr(frustum), _ptr(cam_mats), n_agents, n_cams, D, fH, fW, C,
                   _host_array([float(v) for v in dx], ctypes.c_float),
                   _host_array([float(v) for v in bx], ctypes.c_float),
                   _host_array(nxi, ctypes.c_int32), _ptr(ws), ws.numel(), _stream())
    handle = PooledBEV(ws, n_agents, C, nxi)
    return handle if pooled else handle.dense()


def bev_pool_pm_multi(problems):
    """K4 for several independent problems in ONE launch (heal_bev_pool_scatter_multi): the camera agents of every camera modality of a
    scene.  problems: list of dicts with the arguments of bev_pool_pm (head, C, D, fH, fW, frustum, cam_mats, n_agents, n_cams, dx, bx,
    nx, pooled) -> list of PooledBEV | dense tensors in the same order."""
    if len(problems) == 1:
        return [bev_pool_pm(**problems[0])]
    dev = problems[0]["head"].device
    keep, wss, nbytes_alg = [], [], 0.0
    for i, q in enumerate(problems):
        head = _need(q["head"], torch.float32, "head")
        fr = _need(q["frustum"], torch.float32, "frustum")
        cm = _need(q["cam_mats"], torch.float32, "cam_mats")
        BN, HW, CT = (int(v) for v in head.shape)
        C, D, fH, fW = int(q["C"]), int(q["D"]), int(q["fH"]), int(q["fW"])
        if (BN != q["n_agents"] * q["n_cams"] or HW != fH * fW or CT < C + D or tuple(fr.shape) != (D, fH, fW, 3)
                or tuple(cm.shape) != (BN, 27)):
            raise _capi.HealAmdError("bev_pool_pm_multi: inconsistent shapes")
        if not bev_pool_pm_supported(D, fH, C) or not _frustum_separable(fr):
            raise _capi.HealAmdError("bev_pool_pm_multi: shape / frustum outside the fused path")
        nxi = [int(v) for v in q["nx"]]
        nb = _capi.query("heal_bev_pool_pm_workspace", int(q["n_agents"]), C, nxi[0], nxi[1], nxi[2])
        # one scratch per problem SLOT and shape: two problems of equal shape in one launch must not share rows / flags
        wss.append(_workspace_zeroed(("bev_pool_pm", i, int(q["n_agents"]), C, nxi[0], nxi[1], nxi[2]), nb, dev))
        keep.append((head, fr, cm, CT, nxi))
        nbytes_alg += 4.0 * (BN * HW * (C + D)) + 4.0 * q["n_agents"] * C * nxi[0] * nxi[1] * nxi[2]
    n = len(problems)
    PP = ctypes.c_void_p * n
    I32 = ctypes.c_int32 * n
    arr = lambda vals: I32(*[int(v) for v in vals])
    f3 = lambda key: (ctypes.c_float * (3 * n))(*[float(v) for q in problems for v in q[key]])
    with _Timed("bev_pool", nbytes=nbytes_alg, kernel_events=True):
        _capi.call("heal_bev_pool_scatter_multi", n,
                   ctypes.cast(PP(*[k[0].data_ptr() for k in keep]), ctypes.c_void_p), ctypes.cast(arr(k[3] for k in keep), ctypes.c_void_p),
                   ctypes.cast(PP(*[k[1].data_ptr() for k in keep]), ctypes.c_void_p), ctypes.cast(PP(*[k[2].data_ptr() for k in keep]), ctypes.c_void_p),
                   ctypes.cast(arr(q["n_agents"] for q in problems), ctypes.c_void_p), ctypes.cast(arr(q["n_cams"] for q in problems), ctypes.c_void_p),
                   ctypes.cast(arr(q["D"] for q in problems), ctypes.c_void_p), ctypes.cast(arr(q["fH"] for q in problems), ctypes.c_void_p),
                   ctypes.cast(arr(q["fW"] for q in problems), ctypes.c_void_p), ctypes.cast(arr(q["C"] for q in problems), ctypes.c_void_p),
                   ctypes.cast(f3("dx"), ctypes.c_void_p), ctypes.cast(f3("bx"), ctypes.c_void_p),
                   ctypes.cast((ctypes.c_int32 * (3 * n))(*[v for k in keep for v in k[4]]), ctypes.c_void_p),
                   ctypes.cast(PP(*[w.data_ptr() for w in wss]), ctypes.c_void_p),
                   ctypes.cast((ctypes.c_size_t * n)(*[w.numel() for w in wss]), ctypes.c_void_p), _stream())
    out = []
    for q, w, k in zip(problems, wss, keep):
        h = PooledBEV(w, q["n_agents"], q["C"], k[4])
        out.append(h if q.get("pooled") else h.dense())
    return out


def bev_pool(depth_logit, feat, frustum, cam_mats, n_agents, n_cams, dx, bx, nx):
    """K4 on the reference's NCHW tensors.  depth_logit [n_agents*n_cams,D,fH,fW], feat [n_agents*n_cams,C,fH,fW], frustum
    [D,fH,fW,3] (f32 cuda); cam_mats: f32 cuda [n_agents*n_cams,27] (combine 9, inv(post_rots) 9, post_trans 3,
    trans 3, pad 3); dx,bx host float[3], nx host int[3] -> [n_agents, C*nz, ny, nx].

    Shapes the fused path takes are re-laid pixel-major (one torch copy) and go through bev_pool_pm, the kernel the models
    run; everything else (and HEAL_LSS_PATH=sorted) takes the bit-reproducible radix-sort pipeline."""
    depth_logit = _need(depth_logit, torch.float32, "depth_logit")
    feat = _need(feat, torch.float32, "feat")
    frustum = _need(frustum, torch.float32, "frustum")
    BN, D, fH, fW = (int(v) for v in depth_logit.shape)
    C = int(feat.shape[1])
    if BN != n_agents * n_cams or tuple(feat.shape) != (BN, C, fH, fW) or tuple(frustum.shape) != (D, fH, fW, 3):
        raise _capi.HealAmdError("bev_pool: inconsistent shapes")
    cam_mats = _need(cam_mats, torch.float32, "cam_mats")
    if tuple(cam_mats.shape) != (BN, 27):
        raise _capi.HealAmdError("bev_pool: cam_mats must be [n_agents*n_cams, 27]")
    if bev_pool_pm_supported(D, fH, C) and D % 4 == 0 and _frustum_separable(frustum):
        head = torch.cat([feat, depth_logit], 1).permute(0, 2, 3, 1).reshape(BN, fH * fW, C + D).contiguous()
        return bev_pool_pm(head, C, D, fH, fW, frustum, cam_mats, n_agents, n_cams, dx, bx, nx)
    nxi = [int(v) for v in nx]
    dev = feat.device
    out = torch.empty((n_agents, C * nxi[2], nxi[1], nxi[0]), dtype=torch.float32, device=dev)
    nbytes = _capi.query("heal_bev_pool_workspace", n_agents, n_cams, D, fH, fW, C, nxi[0], nxi[1], nxi[2])
    ws = _workspace("bev_pool", nbytes, dev)
    with _Timed("bev_pool"):
        _capi.call("heal_bev_pool", _ptr(depth_logit), _ptr(feat), _ptr(frustum),
                   _ptr(cam_mats), n_agents, n_cams, D, fH, fW, C,
                   _host_array([float(v) for v in dx], ctypes.c_float),
                   _host_array([float(v) for v in bx], ctypes.c_float),
                   _host_array(nxi, ctypes.c_int32), _ptr(out), _ptr(ws), ws.numel(), _stream())
    return out


def bev_pool_backward(grad_out, depth_logit, feat, frustum, cam_mats, n_agents, n_cams, dx, bx, nx):
    """Gradient of bev_pool with respect to (depth_logit, feat).  grad_out [n_agents, C*nz, ny, nx] -> (grad_logit, grad_feat)
    with the shapes of depth_logit / feat."""
    depth_logit = _need(depth_logit, torch.float32, "depth_logit")
    feat = _need(feat, torch.float32, "feat")
    frustum = _need(frustum, torch.float32, "frustum")
    cam_mats = _need(cam_mats, torch.float32, "cam_mats")
    BN, D, fH, fW = (int(v) for v in depth_logit.shape)
    C = int(feat.shape[1])
    nxi = [int(v) for v in nx]
    if tuple(grad_out.shape) != (n_agents, C * nxi[2], nxi[1], nxi[0]) or tuple(frustum.shape) != (D, fH, fW, 3):
        raise _capi.HealAmdError("bev_pool_backward: inconsistent shapes")
    # cell-major rows: the gradient of one BEV cell is one contiguous row (the forward's scratch layout)
    gcells = _need(grad_out.reshape(n_agents, nxi[2], C, nxi[1], nxi[0]).permute(0, 1, 3, 4, 2).contiguous().view(-1, C),
                   torch.float32, "grad_out")
    g_logit, g_feat = torch.empty_like(depth_logit), torch.empty_like(feat)
    with _Timed("bev_pool_backward"):
        _capi.call("heal_bev_pool_backward", _ptr(gcells), _ptr(depth_logit), _ptr(feat), _ptr(frustum), _ptr(cam_mats),
                   n_agents, n_cams, D, fH, fW, C, _host_array([float(v) for v in dx], ctypes.c_float),
                   _host_array([float(v) for v in bx], ctypes.c_float), _host_array(nxi, ctypes.c_int32),
                   _ptr(g_logit), _ptr(g_feat), _stream())
    return g_logit, g_feat


# ------------------------------------------------------------------------------------------------ K3
def _i3(v):
    return _host_array([int(x) for x in v], ctypes.c_int32)


def _optr(t):
    return _ptr(t) if t is not None else None


def mean_vfe(voxels, num_points, n_dev=None):
    """MeanVFE: voxels [M,P,F], num_points [M] -> [M,F].  n_dev: optional device row count (M is then the capacity)."""
    voxels = _need(voxels, torch.float32, "voxels")
    num_points = _need(num_points, torch.int32, "num_points")
    M, P, F = (int(v) for v in voxels.shape)
    out = torch.empty((M, F), dtype=torch.float32, device=voxels.device)
    _capi.call("heal_mean_vfe", _ptr(voxels), _ptr(num_points), M, P, F, _ptr(out), _optr(n_dev), _stream())
    return out


_SP_FRAGS = {}


def sp_weight_fragments(weight):
    """[K,Cin,Cout] -> the fragment order of the pair-compacted kernel (heal_sp_weight_fragments), cached per weight version
    (a handful of entries: the 12 layers of the encoder).  None when the channel counts have no fragment form."""
    K, cin, cout = (int(v) for v in weight.shape)
    if cin % 4 or (cin >= 16 and cin % 16) or cout % 16 or cin < 4 or int(weight.shape[0]) > 27:
        return None
    key = (weight.data_ptr(), weight._version, K, cin, cout, str(weight.device))
    hit = _SP_FRAGS.get(key)
    if hit is None:
        if len(_SP_FRAGS) > 256:
            _WS_RETIRED.extend(v[0] for v in _SP_FRAGS.values())   # retired, not freed: a captured graph may hold the address
            _SP_FRAGS.clear()
        out = torch.empty_like(weight)
        _capi.call("heal_sp_weight_fragments", _ptr(weight), K, cin, cout, _ptr(out), _stream())
        hit = _SP_FRAGS[key] = (out, weight)  # keeps the source alive: data_ptr stays unique while cached
    return hit[0]


class PairTiles:
    """The rulebook of a thin sparse layer as pair tiles (include/heal_amd.h, heal_sp_neighbor_tiles): one fixed-stride slot per 64
    output sites, only the used prefix written.  Stands where the [n_out, 27] neighbour table stands in SparseTensor.conv for
    c_in <= 16; `.to_neighbors()` decodes it into that table (bit for bit what heal_sp_neighbors_rank gives)."""

    def __init__(self, buf, n_out, n_out_dev=None, slot_sites=64):
        self.buf, self.n_out, self.n_out_dev, self.slot_sites = buf, int(n_out), n_out_dev, int(slot_sites)
        self.shape = (self.n_out, 27)
        self.device = buf.device

    def to_neighbors(self):
        nbr = torch.full((self.n_out, 27), -1, dtype=torch.int32, device=self.buf.device)
        _capi.call("heal_sp_tiles_to_neighbors", _ptr(self.buf), self.n_out, self.slot_sites, _optr(self.n_out_dev), _ptr(nbr),
                   _stream())
        return nbr

    def __getitem__(self, key):     # bench.py counts the live pairs of a traced layer through the table
        return self.to_neighbors()[key]


def sp_tiles_enabled():
    """HEAL_SP_TILES=0: every sparse layer through the [n_out, K] neighbour table (A/B, and what training uses)."""
    return os.environ.get("HEAL_SP_TILES", "1") != "0"


class SparseTensor:
    """features [n,C] f32 + indices [n,4] i32 (b,z,y,x) sorted by linear coordinate + shape (D,H,W).

    `n_dev` (int32 [1] on the device, or None): when set, the buffers have CAPACITY rows and the live row count stays
    on the device -- no host round trip anywhere in the encoder (HIP-graph capturable).  Strided layers then size their
    outputs by a capacity bound (twice the voxel capacity, see out_sites) instead of the exact count; `overflow()` reports, after the
    fact, whether any layer produced more sites than its capacity."""

    def __init__(self, features, indices, spatial_shape, batch_size, n_dev=None, checks=None, root_cap=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(v) for v in spatial_shape]
        self.batch_size = int(batch_size)
        self.n_dev = n_dev
        self._checks = checks if checks is not None else []
        self._root_cap = int(root_cap) if root_cap is not None else int(indices.shape[0])  # rows of the voxel set
        self._table = None
        self._rank = None   # rank structure (bitmap + prefix counts) when this site set came out of out_sites_ex()
        self._rank_root = False   # ... or the two-level structure of the voxel set (from_unsorted)

    @property
    def n(self):
        """Rows of the buffers (= live rows when n_dev is None, capacity otherwise)."""
        return int(self.indices.shape[0])

    def overflow(self):
        """Host check (synchronises): True if a strided layer found more active sites than its capacity."""
        return any(int(c.item()) > cap for c, cap in self._checks)

    @staticmethod
    def from_unsorted(features, indices, spatial_shape, batch_size, n_dev=None):
        """Sort the sites by linear coordinate (K3 keeps them sorted for coherent tiles)."""
        features = _need(features, torch.float32, "features")
        indices = _need(indices, torch.int32, "indices")
        n = int(indices.shape[0])
        dev = indices.device
        sorted_idx = torch.empty_like(indices)
        perm = torch.zeros((n,), dtype=torch.int32, device=dev)  # zeros: padding rows gather row 0, harmlessly
        rank = None
        C = int(features.shape[1])
        feats = torch.zeros((n, C), dtype=torch.float32, device=dev) if n_dev is not None else torch.empty((n, C), dtype=torch.float32, device=dev)
        if os.environ.get("HEAL_SP_RULEBOOK", "rank") == "rank" and os.environ.get("HEAL_SP_ROOT", "rank") == "rank":
            # rank(linear coordinate) IS the sorted position: one scatter (sites, permutation and feature rows) instead of a radix
            # sort + gather, and the structure answers the neighbour queries of the layers that read the voxel set (no hash grid)
            nbytes = _capi.query("heal_sp_root_rank_bytes", _i3(spatial_shape), int(batch_size))
            rank = torch.empty((nbytes,), dtype=torch.uint8, device=dev)   # lives as long as the site set it describes
            with _Timed("sp_rulebook"):
                _capi.call("heal_sp_root_rank", _ptr(indices), n, _i3(spatial_shape), int(batch_size), _ptr(sorted_idx),
                           _ptr(perm), _ptr(features), C, _ptr(feats), _ptr(rank), nbytes, _optr(n_dev), _stream())
        else:
            ws = _workspace("sp_sort", _capi.query("heal_sp_sort_workspace", n), dev)
            with _Timed("sp_rulebook"):
                _capi.call("heal_sp_sort_sites", _ptr(indices), n, _i3(spatial_shape), int(batch_size), _ptr(sorted_idx),
                           _ptr(perm), _ptr(ws), ws.numel(), _optr(n_dev), _stream())
            with _Timed("sp_rulebook"):
                _capi.call("heal_sp_gather_rows", _ptr(features), _ptr(perm), n, C, _optr(n_dev), _ptr(feats), _stream())
        st = SparseTensor(feats, sorted_idx, spatial_shape, batch_size, n_dev)
        st._rank, st._rank_root = rank, rank is not None
        st._perm = perm   # row i of the sorted set = input row perm[i] (the gradient path re-applies it differentiably)
        return st

    def table(self):
        if self._table is None:
            cap = _capi.query("heal_sp_table_capacity", self.n)
            dev = self.indices.device
            keys = torch.empty((cap,), dtype=torch.int32, device=dev)
            vals = torch.empty((cap,), dtype=torch.int32, device=dev)
            with _Timed("sp_rulebook"):
                _capi.call("heal_sp_hash_build", _ptr(self.indices), self.n, _i3(self.spatial_shape), self.batch_size,
                           _ptr(keys), _ptr(vals), cap, _optr(self.n_dev), _stream())
            self._table = (keys, vals, cap)
        return self._table

    def neighbors(self, out_indices, out_shape, ksize, stride, padding, n_out_dev=None):
        n_out = int(out_indices.shape[0])
        K = int(ksize[0] * ksize[1] * ksize[2])
        nbr = torch.empty((n_out, K), dtype=torch.int32, device=self.indices.device)
        if self._rank is not None:
            with _Timed("sp_rulebook"):
                _capi.call("heal_sp_neighbors_root" if self._rank_root else "heal_sp_neighbors_rank",
                           _ptr(out_indices), n_out, _i3(ksize), _i3(stride), _i3(padding),
                           _i3(self.spatial_shape), _i3(out_shape), self.batch_size, _ptr(self._rank), self._rank.numel(),
                           self.n, _optr(self.n_dev), _ptr(nbr), _optr(n_out_dev), _stream())
            return nbr
        keys, vals, cap = self.table()
        with _Timed("sp_rulebook"):
            _capi.call("heal_sp_neighbors", _ptr(out_indices), n_out, _i3(ksize), _i3(stride), _i3(padding),
                       _i3(self.spatial_shape), _i3(out_shape), self.batch_size, _ptr(keys), _ptr(vals), cap,
                       _ptr(nbr), _optr(n_out_dev), _stream())
        return nbr

    def rulebook(self, out_indices, out_shape, ksize, stride, padding, cin, cout, n_out_dev=None):
        """What SparseTensor.conv needs for a (cin -> cout) layer on these output sites: pair tiles when the thin-layer kernel
        takes the layer (3 x 3 x 3, c_in <= 16, rank-structure path), the neighbour table otherwise."""
        if not self.tiles_ok(ksize, cin, cout):
            return self.neighbors(out_indices, out_shape, ksize, stride, padding, n_out_dev=n_out_dev)
        n_out = int(out_indices.shape[0])
        # sites per slot: 64.  128 fill the 16-pair tiles of a strided layer better (2.5 live taps of 27 per site on conv2's
        # SparseConv3d: -5 us on that convolution) but the rulebook kernel then has half the waves for the same lookups (+22 us):
        # HEAL_SP_SLOT_SITES=128 is the A/B switch of a HEAL_BUILD_EXPERIMENTAL=1 library (profiles/r06_k3_thin.json)
        sites = int(os.environ.get("HEAL_SP_SLOT_SITES", 0)) or 64
        buf = torch.empty((_capi.query("heal_sp_pair_tiles_words", n_out, sites),), dtype=torch.int32, device=self.indices.device)
        with _Timed("sp_rulebook"):
            _capi.call("heal_sp_neighbor_tiles", _ptr(out_indices), n_out, _i3(ksize), _i3(stride), _i3(padding),
                       _i3(self.spatial_shape), _i3(out_shape), self.batch_size, _ptr(self._rank), self._rank.numel(),
                       int(bool(self._rank_root)), self.n, _optr(self.n_dev), sites, _ptr(buf), _optr(n_out_dev), _stream())
        return PairTiles(buf, n_out, n_out_dev, sites)

    def tiles_ok(self, ksize, cin, cout):
        return (sp_tiles_enabled() and self._rank is not None and tuple(int(k) for k in ksize) == (3, 3, 3)
                and self.n < (1 << 24) and bool(_capi.query("heal_sp_conv_tiles_supported", int(cin), int(cout))))

    def out_sites(self, ksize, stride, padding):
        """Active output sites of a strided conv: (indices sorted, out_shape, n_out_dev).  Exact-size indices and
        n_out_dev None when this tensor carries host counts; capacity-size indices plus the device count otherwise."""
        return self.out_sites_ex(ksize, stride, padding)[:3]

    def out_sites_ex(self, ksize, stride, padding):
        """out_sites() plus the rank structure of the output site set (None on the hash + sort path, HEAL_SP_RULEBOOK=hash):
        hand it to the SparseTensor built on these sites (`._rank`) and its neighbour queries skip the hash grid."""
        out_shape = [(self.spatial_shape[d] + 2 * padding[d] - ksize[d]) // stride[d] + 1 for d in range(3)]
        K = int(ksize[0] * ksize[1] * ksize[2])
        dev = self.indices.device
        cells = self.batch_size * out_shape[0] * out_shape[1] * out_shape[2]
        worst = max(1, min(self.n * min(K, 8), cells))
        # capacity bound of the no-sync mode: twice the voxel capacity at every level.  Measured on 64-line sweeps at
        # 0.1 m voxels the site count goes x1.36 (isolated far-range voxels dilate), x0.68, x0.48, x0.51 through the
        # four strided layers; overflow() reports a violation.
        out_cap = worst if self.n_dev is None else max(1, min(worst, 2 * self._root_cap))
        out_idx = torch.empty((out_cap, 4), dtype=torch.int32, device=dev)
        n_out = torch.zeros((1,), dtype=torch.int32, device=dev)
        rank = None
        if os.environ.get("HEAL_SP_RULEBOOK", "rank") != "hash":
            nbytes = _capi.query("heal_sp_rank_bytes", _i3(out_shape), self.batch_size)
            rank = torch.empty((nbytes,), dtype=torch.uint8, device=dev)   # lives as long as the site set it describes
            with _Timed("sp_rulebook"):
                _capi.call("heal_sp_out_sites_rank", _ptr(self.indices), self.n, _i3(ksize), _i3(stride), _i3(padding),
                           _i3(self.spatial_shape), _i3(out_shape), self.batch_size, _ptr(out_idx), out_cap, _ptr(n_out),
                           _ptr(rank), nbytes, _optr(self.n_dev), _ptr(sparse_overflow_flag(dev)), _stream())
        else:
            ws = _workspace("sp_out_sites", _capi.query("heal_sp_out_sites_workspace", self.n, K), dev)
            with _Timed("sp_rulebook"):
                _capi.call("heal_sp_out_sites", _ptr(self.indices), self.n, _i3(ksize), _i3(stride), _i3(padding),
                           _i3(self.spatial_shape), _i3(out_shape), self.batch_size, _ptr(out_idx), out_cap, _ptr(n_out),
                           _ptr(ws), ws.numel(), _optr(self.n_dev), _stream())
        if self.n_dev is None:
            return out_idx[:int(n_out.item())], out_shape, None, rank
        self._checks.append((n_out, out_cap))
        _SPARSE_CHECKS.append((n_out, out_cap))
        if len(_SPARSE_CHECKS) > 4096:   # nobody verifies (no host sync on this path at all): keep the list bounded
            del _SPARSE_CHECKS[:2048]
        return out_idx, out_shape, n_out, rank

    def conv(self, nbr, weight, bn_scale, bn_shift, relu=True, n_out_dev=None):
        """Gather-GEMM: weight [K,Cin,Cout]; returns features [n_out,Cout]."""
        weight = _need(weight, torch.float32, "weight")
        K, cin, cout = (int(v) for v in weight.shape)
        frag = sp_weight_fragments(weight)
        n_out = int(nbr.shape[0])
        out = torch.empty((n_out, cout), dtype=torch.float32, device=nbr.device)
        # SURVEY 8d, K3 per layer: 4 (N_in C_in + N_out C_out) + 4 K C_in C_out + 8 R bytes, 2 R C_in C_out flops; with device
        # row counts the host only knows capacities: bench.py fills in the live N_in / N_out / R of its instrumented pass
        with _Timed(f"sp_conv_{cin}_{cout}") as tm:
            if isinstance(nbr, PairTiles):
                if K != 27 or not _capi.query("heal_sp_conv_tiles_supported", cin, cout):
                    raise ValueError(f"SparseTensor.conv: pair tiles do not serve a {K}-tap {cin} -> {cout} layer "
                                     "(SparseTensor.rulebook decides per layer)")
                _capi.call("heal_sp_conv_tiles", _ptr(self.features), _ptr(nbr.buf), n_out, nbr.slot_sites, cin, cout, _ptr(frag),
                           _ptr(_need(bn_scale, torch.float32, "bn_scale")), _ptr(_need(bn_shift, torch.float32, "bn_shift")),
                           int(bool(relu)), _ptr(out), _optr(n_out_dev), _stream())
            else:
                _capi.call("heal_sp_conv", _ptr(self.features), _ptr(nbr), n_out, K, cin, cout, _ptr(weight), _optr(frag),
                           _ptr(_need(bn_scale, torch.float32, "bn_scale")), _ptr(_need(bn_shift, torch.float32, "bn_shift")),
                           int(bool(relu)), _ptr(out), _optr(n_out_dev), _stream())
        if SP_TRACE is not None and TIMING is not None:
            SP_TRACE.append({"cin": cin, "cout": cout, "K": K, "n_in": self.n_dev if self.n_dev is not None else self.n,
                             "n_out": n_out_dev if n_out_dev is not None else n_out, "nbr": nbr, "events": (tm.e0, tm.e1)})
        return out

    def dense(self):
        """-> [B, C*D, H, W] with channel = c*D + z (height_compression.py:21-23)."""
        C = int(self.features.shape[1])
        D, H, W = self.spatial_shape
        dev = self.indices.device
        out = torch.empty((self.batch_size, C * D, H, W), dtype=torch.float32, device=dev)
        ws = _workspace("sp_to_bev", _capi.query("heal_sp_to_bev_workspace", self.batch_size, D, H, W), dev)
        with _Timed("sp_to_bev"):
            _capi.call("heal_sp_to_bev", _ptr(self.features), _ptr(self.indices), self.n, C, _i3(self.spatial_shape),
                       self.batch_size, _ptr(out), _ptr(ws), ws.numel(), _optr(self.n_dev), _stream())
        return out


_SP_IDENT = {}


def sp_conv_raw(features, nbr, weight):
    """The bare sparse convolution (no BatchNorm, no ReLU): out[o] = sum_tap weight[tap]^T features[nbr[o][tap]].  features
    [n_in, Cin], nbr [n_out, K] i32, weight [K, Cin, Cout] -> [n_out, Cout].  Building block of the gradient path (forward AND,
    with the transposed rulebook and weight[tap]^T, the gradient with respect to the input features)."""
    features = _need(features, torch.float32, "features")
    weight = _need(weight, torch.float32, "weight")
    K, cin, cout = (int(v) for v in weight.shape)
    key = (cout, str(features.device))
    ident = _SP_IDENT.get(key)
    if ident is None:
        ident = _SP_IDENT[key] = (torch.ones(cout, device=features.device), torch.zeros(cout, device=features.device))
    n_out = int(nbr.shape[0])
    out = torch.empty((n_out, cout), dtype=torch.float32, device=features.device)
    if n_out:
        _capi.call("heal_sp_conv", _ptr(features), _ptr(nbr), n_out, K, cin, cout, _ptr(weight), _optr(sp_weight_fragments(weight)),
                   _ptr(ident[0]), _ptr(ident[1]), 0, _ptr(out), None, _stream())
    return out


def sp_wgrad_supported(cin, cout):
    return 1 <= cin <= 64 and cout in (16, 32, 48, 64)


def sp_wgrad(features, grad_out, nbr):
    """Weight gradient of the bare sparse convolution: dW[tap] = sum over the rule pairs of x[i]^T g[o] -> [K, Cin, Cout]."""
    features = _need(features, torch.float32, "features")
    grad_out = _need(grad_out, torch.float32, "grad_out")
    nbr = _need(nbr, torch.int32, "nbr")
    n_out, K = (int(v) for v in nbr.shape)
    cin, cout = int(features.shape[1]), int(grad_out.shape[1])
    if n_out == 0:
        return torch.zeros((K, cin, cout), dtype=torch.float32, device=features.device)
    chunks = _capi.query("heal_sp_wgrad_chunks", n_out)
    part = torch.empty((chunks, K, cin, cout), dtype=torch.float32, device=features.device)
    _capi.call("heal_sp_wgrad", _ptr(features), _ptr(grad_out), _ptr(nbr), n_out, K, cin, cout, None, _ptr(part), _stream())
    return part.sum(0)


def sp_transpose_neighbors(nbr, n_in):
    """nbr [n_out, K] -> nbr_t [n_in, K]: nbr_t[i][tap] = o where nbr[o][tap] = i (else -1): the rulebook of the backward pass."""
    nbr = _need(nbr, torch.int32, "nbr")
    n_out, K = (int(v) for v in nbr.shape)
    nbr_t = torch.empty((int(n_in), K), dtype=torch.int32, device=nbr.device)
    _capi.call("heal_sp_transpose_neighbors", _ptr(nbr), n_out, K, int(n_in), _ptr(nbr_t), None, _stream())
    return nbr_t


# ------------------------------------------------------------------------------------------------ K6
def agent_attention(q, k, v, heads, scale, key_mask=None, out_rows=None, agent_major=False):
    """K6.  q,k,v [n_pix, L, 256] f32 cuda -> [n_pix, out_rows, 256] (softmax over the L agents per
    pixel and head; key_mask [L] int32 cuda marks real agents).  agent_major: q, k, v [L, n_pix, 256] -> [out_rows, n_pix, 256]."""
    q = _need(q, torch.float32, "q"); k = _need(k, torch.float32, "k"); v = _need(v, torch.float32, "v")
    if agent_major:
        L, n_pix, C = (int(x) for x in q.shape)
    else:
        n_pix, L, C = (int(x) for x in q.shape)
    rows = L if out_rows is None else int(out_rows)
    out = torch.empty((rows, n_pix, C) if agent_major else (n_pix, rows, C), dtype=torch.float32, device=q.device)
    if key_mask is not None:
        key_mask = _need(key_mask, torch.int32, "key_mask")
    with _Timed(f"agent_attention_h{heads}"):
        _capi.call("heal_agent_attention", _ptr(q), _ptr(k), _ptr(v), _optr(key_mask), n_pix, L, C, int(heads),
                   float(scale), rows, _ptr(out), int(bool(agent_major)), _stream())
    return out


def agent_attention_backward(q, k, v, grad_out, heads, scale, key_mask=None, agent_major=False):
    """Backward of agent_attention: grad_out [n_pix, out_rows, 256] (or [out_rows, n_pix, 256]) -> (grad_q, grad_k, grad_v), each
    shaped like q.  Rows >= out_rows of the forward output were not produced and carry no gradient."""
    q = _need(q, torch.float32, "q"); k = _need(k, torch.float32, "k"); v = _need(v, torch.float32, "v")
    grad_out = _need(grad_out, torch.float32, "grad_out")
    if agent_major:
        L, n_pix, C = (int(x) for x in q.shape)
        rows = int(grad_out.shape[0])
    else:
        n_pix, L, C = (int(x) for x in q.shape)
        rows = int(grad_out.shape[1])
    if key_mask is not None:
        key_mask = _need(key_mask, torch.int32, "key_mask")
    gq, gk, gv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    with _Timed(f"agent_attention_bwd_h{heads}"):
        _capi.call("heal_agent_attention_backward", _ptr(q), _ptr(k), _ptr(v), _optr(key_mask), _ptr(grad_out), n_pix, L, C,
                   int(heads), float(scale), rows, _ptr(gq), _ptr(gk), _ptr(gv), int(bool(agent_major)), _stream())
    return gq, gk, gv


class AgentAttention(torch.autograd.Function):
    """agent_attention under autograd (the gradient path of HGTCavAttention and AttFusion on the device): forward = K6, backward =
    heal_agent_attention_backward; saves q, k, v only (the L x L probabilities are recomputed per pixel)."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale, out_rows, agent_major):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        ctx.save_for_backward(q, k, v)
        ctx.cfg = (int(heads), float(scale), bool(agent_major))
        return agent_attention(q, k, v, heads, scale, out_rows=out_rows, agent_major=agent_major)

    @staticmethod
    def backward(ctx, grad_out):
        q, k, v = ctx.saved_tensors
        heads, scale, am = ctx.cfg
        gq, gk, gv = agent_attention_backward(q, k, v, grad_out.contiguous(), heads, scale, agent_major=am)
        return gq, gk, gv, None, None, None, None


def agent_attention_train_supported(q, heads):
    """The device gradient path: fp32 CUDA tensors, 256 channels, at most 8 agents, heads in {1, 4, 8, 16}."""
    return bool(q.is_cuda and q.dtype == torch.float32 and q.shape[-1] == 256 and heads in (1, 4, 8, 16))


# ------------------------------------------------------------------------------------------------ K6c
def ln_stats(x, eps):
    """(mean, rstd) of every token of x [..., C] -> [T, 2] f32: the statistics half of a LayerNorm whose application is
    folded into the consuming heal_linear."""
    x = _need(x, torch.float32, "x")
    C = int(x.shape[-1])
    T = x.numel() // C
    stats = torch.empty((T, 2), dtype=torch.float32, device=x.device)
    _capi.call("heal_ln_stats", _ptr(x), T, C, float(eps), _ptr(stats), _stream())
    return stats


def linear_supported(n_tokens, n_in, n_out):
    return n_out % 128 == 0 and n_in % 32 == 0 and n_in >= 32 and n_tokens > 0


_ACT = {None: 0, "none": 0, "gelu": 1, "relu": 2}


def linear(x, weight, bias=None, stats=None, act=None, residual=None, out=None, row_map=None, parts=1, colscale=None,
           colscale_part=0, group_rows=0, bias_per_group=False, x_parts=1):
    """heal_linear: out = act(norm(x) W^T * colscale + bias) + residual on the fp32 matrix cores.
    x [T, K] (any leading dims, contiguous), weight [N, K] (nn.Linear layout), stats from ln_stats() (gamma / beta already folded
    into weight / bias), row_map = (inner, outer): token t is written to row (t % inner) * outer + t // inner, parts: the N
    columns are split into `parts` equal column blocks written to out[p] -> returns [parts, T, N / parts] (parts > 1) or [T, N]."""
    x = _need(x, torch.float32, "x")
    weight = _need(weight, torch.float32, "weight")
    # x_parts > 1: x is [x_parts, T, K / x_parts] -- the reduction runs over the channel blocks of several tensors (the three
    # window-attention branches feeding the merged to_out projection) without a concatenated copy
    xk = int(x.shape[-1])
    K = xk * int(x_parts)
    T = x.numel() // K
    N = int(weight.shape[0])
    if int(weight.shape[1]) != K or not linear_supported(T, K, N) or N % parts or (N // parts) % 4:
        raise _capi.HealAmdError(f"linear: unsupported shape T={T} K={K} N={N} parts={parts}")
    pc = N // parts
    if out is None:
        out = torch.empty((parts, T, pc) if parts > 1 else (T, N), dtype=torch.float32, device=x.device)
    inner, outer = (int(row_map[0]), int(row_map[1])) if row_map is not None else (0, 0)
    if residual is not None:
        residual = _need(residual, torch.float32, "residual")
    with _Timed(f"linear_{K}_{N}", flops=2.0 * T * K * N, kernel_events=True):
        _capi.call("heal_linear", _ptr(x), xk, xk if x_parts > 1 else 0, T * xk, _optr(stats), _ptr(weight), _optr(bias),
                   int(bool(bias_per_group)),
                   _optr(colscale), int(colscale_part), int(group_rows), _optr(residual), N if residual is not None else 0,
                   _ptr(out), pc, T, N, K, inner, outer, pc, T * pc, _ACT[act], _stream())
    return out


def split_attn_weights(branches, groups, rows_per_group, w_out, b_out, fc1, ln_g, ln_b, eps, fc2):
    """branches [3, groups * rows_per_group, C] (window attention outputs before to_out) -> (scale [groups,3,C], bias [groups,C])."""
    branches = _need(branches, torch.float32, "branches")
    C = int(branches.shape[-1])
    dev = branches.device
    ws = _workspace("split_attn", _capi.query("heal_split_attn_workspace", groups, rows_per_group, C), dev)
    scale = torch.empty((groups, 3, C), dtype=torch.float32, device=dev)
    bias = torch.empty((groups, C), dtype=torch.float32, device=dev)
    _capi.call("heal_split_attn_weights", _ptr(branches), int(branches.stride(0)), groups, rows_per_group, C, _ptr(w_out),
               _ptr(b_out), _ptr(fc1), _ptr(ln_g), _ptr(ln_b), float(eps), _ptr(fc2), _ptr(ws), _ptr(scale), _ptr(bias),
               _stream())
    return scale, bias


def split_attn_colsum(branches, groups, rows_per_group):
    """First half of split_attn_weights for tokens spread over ranks: [groups, 3, ceil(rows / 512), C] per-chunk column sums of
    the local tokens of branches [3, groups * rows_per_group, C] (a fresh tensor: it is handed to a collective)."""
    branches = _need(branches, torch.float32, "branches")
    C = int(branches.shape[-1])
    out = torch.empty((groups, 3, (rows_per_group + 511) // 512, C), dtype=torch.float32, device=branches.device)
    _capi.call("heal_split_attn_colsum", _ptr(branches), int(branches.stride(0)), groups, rows_per_group, C, _ptr(out), _stream())
    return out


def split_attn_weights_from_colsum(colsum, rows_per_part, w_out, b_out, fc1, ln_g, ln_b, eps, fc2):
    """Second half: colsum [n_parts, groups, 3, chunks, C] (the all-gathered halves) -> (scale [groups,3,C], bias [groups,C])."""
    colsum = _need(colsum, torch.float32, "colsum")
    n_parts, groups, _three, chunks, C = (int(v) for v in colsum.shape)
    if chunks != (rows_per_part + 511) // 512:
        raise ValueError(f"split_attn_weights_from_colsum: {chunks} chunks do not match {rows_per_part} rows per part")
    dev = colsum.device
    scale = torch.empty((groups, 3, C), dtype=torch.float32, device=dev)
    bias = torch.empty((groups, C), dtype=torch.float32, device=dev)
    _capi.call("heal_split_attn_weights_from_colsum", _ptr(colsum), n_parts, groups, int(rows_per_part), C, _ptr(w_out),
               _ptr(b_out), _ptr(fc1), _ptr(ln_g), _ptr(ln_b), float(eps), _ptr(fc2), _ptr(scale), _ptr(bias), _stream())
    return scale, bias


# ------------------------------------------------------------------------------------------------ K7
_FRAGG_CACHE = {}


def grouped16_fragments(weight, cg):
    """[C, cg, 3, 3] grouped weight (cg = 16 | 8) -> MFMA A fragments of the 16-channel super-groups
    [C/16][tap][ks][lane] (block-diagonal zero padding for cg = 8), cached per storage + version."""
    key = (weight.data_ptr(), weight._version, tuple(weight.shape))
    hit = _FRAGG_CACHE.get(key)
    if hit is None:
        if len(_FRAGG_CACHE) > 512:
            _retire_cache(_FRAGG_CACHE)
        C = int(weight.shape[0])
        w = weight.detach().reshape(C // 16, 16, cg, 9)                   # [sg, co, ci_in_group, tap]
        if cg == 16:
            full = w
        else:                                                             # two groups of 8: rows 0-7 use ci 0-7, rows 8-15 ci 8-15
            full = torch.zeros((C // 16, 16, 16, 9), dtype=w.dtype, device=w.device)
            full[:, :8, :8] = w[:, :8]
            full[:, 8:, 8:] = w[:, 8:]
        # [sg, co(ln), ks, lk, tap] -> [sg, tap, ks, lk, ln]
        f = full.reshape(C // 16, 16, 4, 4, 9).permute(0, 4, 2, 3, 1).contiguous()
        hit = (f, weight)
        _FRAGG_CACHE[key] = hit
    return hit[0]


_FRAGQ_CACHE = {}


def grouped_small_fragments(weight, cg):
    """[C, cg, 3, 3] grouped weight (cg = 4 | 8 | 16) -> weight operands of heal_grouped_small_conv3x3: [C/16][tap][ci][16 output channels
    of the super-group], cached per storage + version."""
    key = (weight.data_ptr(), weight._version, tuple(weight.shape))
    hit = _FRAGQ_CACHE.get(key)
    if hit is None:
        if len(_FRAGQ_CACHE) > 512:
            _retire_cache(_FRAGQ_CACHE)
        C = int(weight.shape[0])
        f = weight.detach().reshape(C // 16, 16, cg, 9).permute(0, 3, 2, 1).contiguous()   # [sg, tap, ci, co]
        hit = (f, weight)
        _FRAGQ_CACHE[key] = hit
    return hit[0]


def grouped_conv3x3(x, weight, bias, groups, stride=1, relu=True):
    """32-group 3x3 conv (padding 1, stride 1 | 2) with fused bias + ReLU.  x [n,C,H,W], weight [C,C/groups,3,3].  4, 8 or 16
    channels per group run on the 16-block 4x4x1 MFMA (heal_grouped_small_conv3x3; measured 5 agents, us: 128 ch 256^2 166 ->
    98, 256 ch 128^2 84 -> 59, 512 ch 64^2 78 -> 49, stride 2: 256 ch 256^2 207 -> 107, 512 ch 128^2 122 -> 68);
    HEAL_GCONV_MFMA=16 | 8 selects the older 16x16x4 kernel (one m-tile per 16-channel group: 53 us; pairs of 8-channel groups
    with block-diagonal weights: 98 us), =0 the vector-ALU stencil, which also takes the shapes the MFMA kernels do not."""
    import os
    x = _need(x, torch.float32, "x")
    weight = _need(weight, torch.float32, "weight")
    n, C, H, W = (int(v) for v in x.shape)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    y = torch.empty((n, C, Ho, Wo), dtype=torch.float32, device=x.device)
    cg = C // groups
    mode = os.environ.get("HEAL_GCONV_MFMA", "1")
    if C % 16 == 0 and W % 4 == 0 and Wo % 4 == 0 and mode in ("1", "s") and cg in (4, 8, 16):
        frag = grouped_small_fragments(weight, cg)
        with _Timed(f"grouped_conv3x3_c{C}" + ("_s2" if stride == 2 else ""), 2.0 * 9 * n * C * cg * Ho * Wo,
                    4.0 * n * C * (H * W + Ho * Wo), kernel_events=True):
            _capi.call("heal_grouped_small_conv3x3", _ptr(x), _ptr(frag), _ptr(bias), n, C, cg, H, W, int(stride),
                       int(bool(relu)), _ptr(y), _stream())
        return y
    if stride == 1 and C % 16 == 0 and W % 4 == 0 and ((cg == 16 and mode != "0") or (cg == 8 and mode == "8")):   # mode "16" | "8"
        frag = grouped16_fragments(weight, cg)
        with _Timed(f"grouped_conv3x3_c{C}", 2.0 * 9 * n * C * cg * Ho * Wo, 4.0 * n * C * (H * W + Ho * Wo)):
            _capi.call("heal_grouped16_conv3x3", _ptr(x), _ptr(frag), _ptr(bias), n, C, H, W, int(bool(relu)), _ptr(y),
                       _stream())
        return y
    with _Timed(f"grouped_conv3x3_c{C}" + ("_s2" if stride == 2 else ""), 2.0 * 9 * n * C * (C // groups) * Ho * Wo,
                4.0 * n * C * (H * W + Ho * Wo)):
        _capi.call("heal_grouped_conv3x3", _ptr(x), _ptr(weight), _ptr(bias), n, C, int(groups), H, W, int(stride),
                   int(bool(relu)), _ptr(y), _stream())
    return y


def gconv_conv3_supported(width, cg, cout, H, W):
    return bool(_capi.query("heal_gconv_conv3_supported", int(width), int(cg), int(cout), int(H), int(W)))


def gconv_conv3(x, w2, b2, groups, w3, b3, residual=None, relu=True):
    """Back half of a ResNeXt bottleneck in one kernel: y = act(conv1x1(relu(gconv3x3(x, w2) + b2), w3) + b3 (+ residual)); the 2C-wide
    intermediate never reaches HBM.  x [n, width, H, W], w2 [width, width / groups, 3, 3], w3 [cout, width, 1, 1]."""
    x = _need(x, torch.float32, "x")
    n, width, H, W = (int(v) for v in x.shape)
    cg = width // groups
    cout = int(w3.shape[0])
    frag2 = grouped_small_fragments(_need(w2, torch.float32, "w2"), cg)
    key = (w3.data_ptr(), w3._version, "gc3")
    frag3 = _FRAG_CACHE.get(key)
    if frag3 is None:
        frag3 = _FRAG_CACHE[key] = mfma_a_fragments(w3.detach().reshape(cout, width))
    y = torch.empty((n, cout, H, W), dtype=torch.float32, device=x.device)
    flops = 2.0 * n * H * W * (9 * width * cg + width * cout)
    nbytes = 4.0 * n * H * W * (width + cout * (2 if residual is not None else 1))
    with _Timed(f"gconv_conv3_{width}_{cout}", flops, nbytes, kernel_events=True):
        _capi.call("heal_gconv_conv3", _ptr(x), _ptr(frag2), _ptr(_need(b2, torch.float32, "b2")), _ptr(frag3),
                   _ptr(_need(b3, torch.float32, "b3")), _optr(residual), n, width, cg, cout, H, W, int(bool(relu)), _ptr(y), _stream())
    return y


def bias_act_(x, bias=None, residual=None, relu=True):
    """In place: x = act(x + bias[c] + residual).  x [n,C,H,W] contiguous f32 cuda."""
    x = _need(x, torch.float32, "x")
    n, C = int(x.shape[0]), int(x.shape[1])
    HW = int(x.shape[2] * x.shape[3])
    if residual is not None:
        residual = _need(residual, torch.float32, "residual")
    _capi.call("heal_bias_act", _ptr(x), _ptr(bias), _ptr(residual), n, C, HW, int(bool(relu)), _stream())
    return x


def mfma_a_fragments(wm):
    """[M,K] matrix -> MFMA 16x16x4 A-fragment order [M/16, K/4, 64]: frag[mt][ks][lane] =
    wm[mt*16 + (lane & 15)][ks*4 + (lane >> 4)]."""
    M, K = wm.shape
    return wm.reshape(M // 16, 16, K // 4, 4).permute(0, 2, 3, 1).reshape(M // 16, K // 4, 64).contiguous()


_FRAG_CACHE = {}


def conv1x1_fragments(w):
    """Cached MFMA A-fragment layout of a [Cout,Cin,1,1] (or [Cout,Cin]) weight, zero-padded to [ceil64(Cout),
    ceil32(Cin)], keyed by storage + version."""
    key = (w.data_ptr(), w._version, tuple(w.shape))
    hit = _FRAG_CACHE.get(key)
    if hit is None:
        if len(_FRAG_CACHE) > 1024:
            _retire_cache(_FRAG_CACHE)
        cout, cin = int(w.shape[0]), int(w.shape[1])
        mpad, kpad = (cout + 63) // 64 * 64, (cin + 31) // 32 * 32
        wm = w.detach().reshape(cout, cin)
        if (mpad, kpad) != (cout, cin):
            wm = torch.nn.functional.pad(wm, (0, kpad - cin, 0, mpad - cout))
        fr = mfma_a_fragments(wm)            # [M/16, K/4, 64] -> four k-steps per lane contiguous: [M/16, K/16, 64, 4]
        fr = fr.reshape(mpad // 16, kpad // 16, 4, 64).permute(0, 1, 3, 2).contiguous()
        hit = (fr, w)  # keep w alive: the key is its address
        _FRAG_CACHE[key] = hit
    return hit[0]


def conv1x1_supported(cin, cout, hw, stride=1, out_w=None):
    """Shapes heal_conv1x1 takes: 16-byte pixel rows (stride 1: Ho*Wo % 4 == 0; stride 2: Wo % 4 == 0) and maps large
    enough to fill a few tiles."""
    if stride == 1:
        return hw % 4 == 0 and hw >= 64
    return stride == 2 and out_w is not None and out_w % 4 == 0 and hw >= 64


def conv1x1_ksplit(n, cin, cout, hw):
    """K splits for a stride-1 NCHW pointwise convolution: 1 unless the grid is small (< 128 blocks of 64 channels x 64 pixels)
    and the reduction deep (>= 8 chunks of 32 channels); then enough splits for ~256 blocks, at least 2 chunks each, none empty.
    HEAL_C1_KSPLIT forces a value (0 / 1: off)."""
    chunks = (cin + 31) // 32
    blocks = -(-cout // 64) * -(-hw // 64) * n
    env = os.environ.get("HEAL_C1_KSPLIT")
    want = int(env) if env is not None else (min(chunks // 2, max(2, 256 // blocks)) if blocks < 128 and chunks >= 8 else 1)
    if want < 2 or chunks < 2 or hw % 4:
        return 1
    want = min(want, chunks, 65535 // n)
    return -(-chunks // -(-chunks // want))       # ceil(chunks / ceil(chunks / want)): no empty split


_STEM_CACHE = {}


def stem7x7_fragments(w):
    """[64, cin, 7, 7] -> MFMA A-fragment order [4][ceil(cin * 49 / 4)][64] of heal_stem7x7 (cached per storage + version)."""
    key = (w.data_ptr(), w._version, tuple(w.shape))
    hit = _STEM_CACHE.get(key)
    if hit is None:
        if len(_STEM_CACHE) > 64:
            _WS_RETIRED.extend(v[0] for v in _STEM_CACHE.values())
            _STEM_CACHE.clear()
        cout, cin = int(w.shape[0]), int(w.shape[1])
        if cout != 64 or tuple(w.shape[2:]) != (7, 7) or not 1 <= cin <= 4:
            raise _capi.HealAmdError(f"stem7x7: weight must be [64, 1..4, 7, 7], got {tuple(w.shape)}")
        K = cin * 49
        ks = (K + 3) // 4
        wk = torch.nn.functional.pad(w.detach().to(torch.float32).reshape(64, K), (0, 4 * ks - K))      # [64, 4 ks]
        frag = wk.reshape(4, 16, ks, 4).permute(0, 2, 3, 1).contiguous()                               # [mt][ks][lk][ln]
        hit = _STEM_CACHE[key] = (frag, w)
    return hit[0]


def stem7x7(x, w, bias, pool=True):
    """relu(conv7x7/2(x[:, :cin], w) + bias) (+ 3x3/2 max-pool) in one kernel; x [n, >= cin, H, W] is read in place (the first
    cin = w.shape[1] channels of every image).  -> [n, 64, Hp, Wp]."""
    x = _need(x, torch.float32, "x")
    n, cx, H, W = (int(v) for v in x.shape)
    cin = int(w.shape[1])
    if cx < cin:
        raise _capi.HealAmdError(f"stem7x7: input has {cx} channels, the weight wants {cin}")
    frag = stem7x7_fragments(w)
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Ho, Wo = ((Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1) if pool else (Hc, Wc)
    y = torch.empty((n, 64, Ho, Wo), dtype=torch.float32, device=x.device)
    with _Timed(f"stem7x7_{cin}" + ("_pool" if pool else ""), 2.0 * n * 64 * cin * 49 * Hc * Wc, kernel_events=True):
        _capi.call("heal_stem7x7", _ptr(x), cx * H * W, n, cin, H, W, _ptr(frag),
                   _ptr(_need(bias, torch.float32, "bias")) if bias is not None else None, int(bool(pool)), _ptr(y), _stream())
    return y


def conv1x1_tiled_ok(n, cin, cout, hw):
    """Shapes the 128 x 128 x 32 core (heal_conv1x1_tiled, 32x32x2 fp32 MFMA) takes over from heal_conv1x1: 32-channel K chunks,
    64- or 128-channel M tiles, enough blocks to fill the chip, and a reduction deep enough to be MFMA- rather than HBM-bound.
    OPT-IN (HEAL_C1_TILED=1, =force for every shape it can take): measured at the scenes' shapes (scripts/c1t_bench.py,
    profiles/r04_c1t_bench.json) it is at parity with the 64 x 64 kernel -- 0.70-1.07x, ahead only on 256 -> 2048 and single-image
    256 -> 128; neither tile shape of either kernel moves the 85-105 TFLOP/s these 5-GFLOP launches reach in isolation."""
    mode = os.environ.get("HEAL_C1_TILED", "0")
    if mode == "0" or not experimental_build() or cin % 32 or cout % 64 or hw % 4 or hw < 128:
        return False
    bm = 128 if cout % 128 == 0 else 64
    blocks = (cout // bm) * -(-hw // 128) * n
    if mode == "force":
        return True
    return blocks >= 256 and cin >= 128


_FRAGS_CACHE = {}


def arith_products():
    """HEAL_ARITH: "" / "f32" (default: exact-fp32 MFMA everywhere) | "bf16x6" | "bf16x9" -- the OPT-IN split-bf16 evaluation of the
    pointwise convolutions (heal_conv1x1_split; fp32 in / out / accumulate, 6 or 9 bf16 partial products per fp32 product)."""
    a = os.environ.get("HEAL_ARITH", "")
    return {"bf16x6": 6, "bf16x9": 9}.get(a, 0)


def conv1x1_split_fragments(w):
    """[Cout, Cin(,1,1)] fp32 -> the three bf16 planes (w = h + m + l, round to nearest) in the fragment order of heal_conv1x1_split:
    [Cout/128][Cin/32][k16 step 2][row block 4][plane 3][lane 64][8] with element = W_p[128 ct + 32 rb + lane % 32][32 c + 16 s +
    8 (lane / 32) + e]; cached per storage + version."""
    key = (w.data_ptr(), w._version, tuple(w.shape))
    hit = _FRAGS_CACHE.get(key)
    if hit is None:
        if len(_FRAGS_CACHE) > 512:
            _retire_cache(_FRAGS_CACHE)
        cout, cin = int(w.shape[0]), int(w.shape[1])
        wm = w.detach().reshape(cout, cin).to(torch.float32)
        h = wm.to(torch.bfloat16)
        r1 = wm - h.float()
        m = r1.to(torch.bfloat16)
        lo = (r1 - m.float()).to(torch.bfloat16)
        planes = torch.stack([h, m, lo], 0)                                    # [3, cout, cin]
        f = planes.reshape(3, cout // 128, 4, 32, cin // 32, 2, 2, 8)          # [p, ct, rb, li, c, s, kb, e]
        f = f.permute(1, 4, 5, 2, 0, 6, 3, 7).contiguous()                     # [ct, c, s, rb, p, kb, li, e]
        hit = (f, w)
        _FRAGS_CACHE[key] = hit
    return hit[0]


def _w_rowmajor(w):
    w2 = w.detach().reshape(int(w.shape[0]), int(w.shape[1]))
    return w2 if w2.is_contiguous() else w2.contiguous()


def _out_or_empty(out, shape, device, who):
    """The caller's destination (a contiguous f32 tensor of exactly `shape`, e.g. a batch slice of a larger map) or a new one."""
    if out is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    if (tuple(out.shape) != tuple(shape) or out.dtype != torch.float32 or not out.is_cuda or not out.is_contiguous()):
        raise _capi.HealAmdError(f"{who}: `out` must be a contiguous f32 cuda tensor of shape {tuple(shape)}")
    return out


def conv1x1(x, w, bias=None, residual=None, act=0, in_scale=None, stride=1, pixel_major=False, out=None):
    """Pointwise convolution with fused prologue / epilogue: act(W (in_scale . x) + bias (+ residual));
    act 0 none | 1 ReLU | 2 SiLU; stride 1 | 2.  x [n,Cin,H,W] f32 cuda, w [Cout,Cin,1,1], in_scale [n,Cin].
    pixel_major=True: the result comes back as [n, Ho*Wo, Cout] (a pixel's channels contiguous) instead of NCHW.
    out: optional destination (NCHW result only), e.g. the batch slice of a stage's output that an agent chunk fills."""
    x = _need(x, torch.float32, "x")
    n, cin, H, W = (int(v) for v in x.shape)
    cout = int(w.shape[0])
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    hard_ok = stride in (1, 2) and ((Ho * Wo) % 4 == 0 if stride == 1 else Wo % 4 == 0)
    if int(w.shape[1]) != cin or not hard_ok:
        raise _capi.HealAmdError(f"conv1x1: unsupported shape Cin={cin} Cout={cout} HxW={H}x{W} stride={stride}")
    if out is not None and pixel_major:
        raise _capi.HealAmdError("conv1x1: `out` is for the NCHW result")
    nprod = arith_products()
    if (nprod and stride == 1 and not pixel_major and in_scale is None and w.dtype == torch.float32
            and _capi.query("heal_conv1x1_split_supported", cin, cout, H, W)):
        y = _out_or_empty(out, (n, cout, H, W), x.device, "conv1x1")
        if residual is not None:
            residual = _need(residual, torch.float32, "residual")
            if tuple(residual.shape) != tuple(y.shape):
                raise _capi.HealAmdError("conv1x1: residual shape mismatch")
        frag = conv1x1_split_fragments(w)
        with _Timed(f"conv1x1_{cin}_{cout}", 2.0 * n * cin * cout * H * W,
                    4.0 * n * (cin * H * W + cout * H * W * (2 if residual is not None else 1)), kernel_events=True):
            _capi.call("heal_conv1x1_split", _ptr(x), _ptr(frag), _ptr(bias) if bias is not None else None,
                       _ptr(residual) if residual is not None else None, n, cin, cout, H, W, int(act), nprod, _ptr(y), _stream())
        return y
    if (stride == 1 and not pixel_major and in_scale is None and conv1x1_tiled_ok(n, cin, cout, H * W)
            and w.dtype == torch.float32):
        y = _out_or_empty(out, (n, cout, H, W), x.device, "conv1x1")
        if residual is not None:
            residual = _need(residual, torch.float32, "residual")
            if tuple(residual.shape) != tuple(y.shape):
                raise _capi.HealAmdError("conv1x1: residual shape mismatch")
        wr = _w_rowmajor(w)
        with _Timed(f"conv1x1_{cin}_{cout}", 2.0 * n * cin * cout * H * W,
                    4.0 * n * (cin * H * W + cout * H * W * (2 if residual is not None else 1))):
            _capi.call("heal_conv1x1_tiled", _ptr(x), _ptr(wr), _ptr(bias) if bias is not None else None,
                       _ptr(residual) if residual is not None else None, n, cin, cout, H, W, int(act), 0, 0, 0, _ptr(y), _stream())
        return y
    frag = conv1x1_fragments(w)
    if pixel_major:
        if residual is not None or cout % 4 != 0:
            raise _capi.HealAmdError("conv1x1: pixel-major output needs Cout % 4 == 0 and takes no residual")
        y = torch.empty((n, Ho * Wo, cout), dtype=torch.float32, device=x.device)
        if bias is not None:
            bias = _need(bias, torch.float32, "bias")
    else:
        y = _out_or_empty(out, (n, cout, Ho, Wo), x.device, "conv1x1")
    if residual is not None:
        residual = _need(residual, torch.float32, "residual")
        if tuple(residual.shape) != tuple(y.shape):
            raise _capi.HealAmdError("conv1x1: residual shape mismatch")
    if in_scale is not None:
        in_scale = _need(in_scale.reshape(n, cin), torch.float32, "in_scale")
    ksplit = conv1x1_ksplit(n, cin, cout, H * W) if (stride == 1 and not pixel_major) else 1
    if ksplit > 1:
        nbytes = _capi.query("heal_conv1x1_splitk_workspace", n, cout, H, W, ksplit)
        ws = _workspace("conv1x1_splitk", nbytes, x.device)
        with _Timed(f"conv1x1_{cin}_{cout}", 2.0 * n * cin * cout * H * W,
                    4.0 * n * (cin * H * W + cout * H * W * (2 if residual is not None else 1)), kernel_events=True):
            _capi.call("heal_conv1x1_splitk", _ptr(x), _ptr(frag), _ptr(bias) if bias is not None else None,
                       _ptr(residual) if residual is not None else None, _ptr(in_scale) if in_scale is not None else None,
                       n, cin, cout, H, W, int(act), ksplit, _ptr(y), _ptr(ws), ws.numel(), _stream())
        return y
    with _Timed(f"conv1x1_{cin}_{cout}" + ("_s2" if stride == 2 else ""), 2.0 * n * cin * cout * Ho * Wo,
                4.0 * n * (cin * Ho * Wo + cout * Ho * Wo * (2 if residual is not None else 1)), kernel_events=True):
        _capi.call("heal_conv1x1", _ptr(x), _ptr(frag), _ptr(bias) if bias is not None else None,
                   _ptr(residual) if residual is not None else None, _ptr(in_scale) if in_scale is not None else None,
                   n, cin, cout, H, W, int(stride), int(act), int(bool(pixel_major)), _ptr(y), _stream())
    return y


def conv1x1_d2s(x, w, bias, act, k, dst, channel_offset):
    """Pointwise convolution to Cout = C k^2 channels whose epilogue writes depth-to-space into dst[:, off:off+C] (dst
    [n, Ctot, H k, W k], contiguous): a ConvTranspose2d(kernel = stride = k) deblock written straight into the concatenated
    tensor.  x [n,Cin,H,W], w [C k^2, Cin, 1, 1] (row c k^2 + dy k + dx = output channel c at sub-pixel (dy, dx))."""
    x = _need(x, torch.float32, "x")
    dst = _need(dst, torch.float32, "dst")
    n, cin, H, W = (int(v) for v in x.shape)
    cout = int(w.shape[0])
    if (not dst.is_contiguous() or int(dst.shape[0]) != n or int(dst.shape[2]) != H * k or int(dst.shape[3]) != W * k
            or cout % (k * k) or W % 4):
        raise _capi.HealAmdError(f"conv1x1_d2s: destination {tuple(dst.shape)} does not fit x {tuple(x.shape)} at k={k}")
    bias = _need(bias, torch.float32, "bias") if bias is not None else None
    if conv1x1_tiled_ok(n, cin, cout, H * W) and w.dtype == torch.float32:
        wr = _w_rowmajor(w)
        with _Timed(f"conv1x1_{cin}_{cout}", 2.0 * n * cin * cout * H * W, 4.0 * n * H * W * (cin + cout)):
            _capi.call("heal_conv1x1_tiled", _ptr(x), _ptr(wr), _ptr(bias) if bias is not None else None, None, n, cin, cout, H, W,
                       int(act), int(k), int(dst.shape[1]), int(channel_offset), _ptr(dst), _stream())
        return dst[:, channel_offset:channel_offset + cout // (k * k)]
    frag = conv1x1_fragments(w)
    with _Timed(f"conv1x1_{cin}_{cout}", 2.0 * n * cin * cout * H * W, 4.0 * n * H * W * (cin + cout), kernel_events=True):
        _capi.call("heal_conv1x1_d2s", _ptr(x), _ptr(frag), _ptr(bias) if bias is not None else None, n, cin, cout, H, W,
                   int(act), int(k), int(dst.shape[1]), int(channel_offset), _ptr(dst), _stream())
    return dst[:, channel_offset:channel_offset + cout // (k * k)]


_FRAG3_CACHE = {}


def conv3x3_fragments(w):
    """Cached MFMA A-fragment layout of a [Cout,Cin,3,3] weight for heal_conv3x3: zero-padded to [ceil64(Cout), ceil8(Cin)],
    ordered [Cout/64][Cin/8][tap][k-step][m-tile][lane] (see include/heal_amd.h); keyed by storage + version."""
    key = (w.data_ptr(), w._version, tuple(w.shape))
    hit = _FRAG3_CACHE.get(key)
    if hit is None:
        if len(_FRAG3_CACHE) > 512:
            _retire_cache(_FRAG3_CACHE)
        cout, cin = int(w.shape[0]), int(w.shape[1])
        mpad, kpad = (cout + 63) // 64 * 64, (cin + 7) // 8 * 8
        wm = w.detach().reshape(cout, cin, 9)
        if (mpad, kpad) != (cout, cin):
            wm = torch.nn.functional.pad(wm, (0, 0, 0, kpad - cin, 0, mpad - cout))
        # [mb, mt, ln, chunk, ks, lk, tap] -> [mb, chunk, tap, ks, mt, lk, ln]
        f = wm.reshape(mpad // 64, 4, 16, kpad // 8, 2, 4, 9).permute(0, 3, 6, 4, 1, 5, 2).contiguous()
        hit = (f, w)  # keep w alive: the key is its address
        _FRAG3_CACHE[key] = hit
    return hit[0]


def conv3x3_same(x, w, bias, stride, pad, act="none"):
    """Dense 3x3 convolution with explicit (left, right, top, bottom) zero padding where left / top are 0 | 1 and right / bottom
    at most 1 (TF-style "same" padding of the EfficientNet stem), bias and none | relu | silu fused.  x [n,Cin,H,W]."""
    x = _need(x, torch.float32, "x")
    n, cin, H, W = (int(v) for v in x.shape)
    cout = int(w.shape[0])
    pl, pr, pt, pb = (int(v) for v in pad)
    Ho, Wo = (H + pt + pb - 3) // stride + 1, (W + pl + pr - 3) // stride + 1
    if pl not in (0, 1) or pt not in (0, 1) or pr > 1 or pb > 1 or tuple(w.shape[1:]) != (cin, 3, 3):
        raise _capi.HealAmdError(f"conv3x3_same: unsupported padding {pad} / weight {tuple(w.shape)}")
    frag = conv3x3_fragments(w)
    y = torch.empty((n, cout, Ho, Wo), dtype=torch.float32, device=x.device)
    with _Timed(f"conv3x3_{cin}_{cout}_s{stride}same", 2.0 * 9 * n * cin * cout * Ho * Wo, 4.0 * n * (cin * H * W + cout * Ho * Wo),
                kernel_events=True):
        _capi.call("heal_conv3x3_same", _ptr(x), _ptr(frag), _ptr(_need(bias, torch.float32, "bias")) if bias is not None else None,
                   n, cin, cout, H, W, int(stride), pt, pl, Ho, Wo, {"none": 0, "relu": 1, "silu": 2}[act], _ptr(y), _stream())
    return y


_FRAGW_CACHE = {}


def conv3x3_winograd_waves(n=1, cout=64, H=256, W=256):
    """Waves per Winograd block: 8 (16x16-pixel tiles, one block per CU: best operand reuse) when that still gives every CU two
    or more blocks, else 4 (8x16-pixel tiles, two independent blocks per CU whose transform / MFMA phases overlap; measured
    crossover, scripts/conv3x3_bench.py); HEAL_WG_WAVES overrides."""
    import os
    e = os.environ.get("HEAL_WG_WAVES", "")
    if e in ("4", "8"):
        return int(e)
    blocks8 = n * ((cout + 63) // 64) * ((H + 15) // 16) * ((W + 15) // 16)
    return 8 if blocks8 >= 512 else 4


def conv3x3_winograd_kc(cin, waves, H=1, W=1):
    """Input channels per chunk of the Winograd K loop: 16 where the kernel has it (an experimental build, 8-wave blocks,
    cin % 16 == 0, a map whose 16-channel chunk stays below 2^31 bytes) and HEAL_WG_KC asks for it, else 8.  Measured in round 5
    (profiles/r05_wino_kc16.txt): 4-13 % slower, hence experimental."""
    import os
    want = os.environ.get("HEAL_WG_KC", _WG_KC_DEFAULT)
    ok16 = (want == "16" and waves == 8 and cin % 16 == 0 and 16 * H * W * 4 < 2 ** 31 and experimental_build())
    return 16 if ok16 else 8


def experimental_build():
    """True if libheal_amd.so was built with HEAL_BUILD_EXPERIMENTAL=1 (the measured-negative kernels of
    include/heal_amd_experimental.h are present)."""
    return hasattr(_capi.lib(), "heal_gconv_conv3")


_WG_KC_DEFAULT = "8"


def conv3x3_winograd_fragments(w, waves=8, kc=8):
    """Cached Winograd-domain weights U = G g G^T of a [Cout,Cin,3,3] filter bank in the lane-major fragment order
    heal_conv3x3_winograd[_kc] reads for `waves` waves per block and `kc` channels per chunk (include/heal_amd.h); keyed by storage +
    version."""
    key = (w.data_ptr(), w._version, tuple(w.shape), waves, kc)
    hit = _FRAGW_CACHE.get(key)
    if hit is None:
        if len(_FRAGW_CACHE) > 512:
            _retire_cache(_FRAGW_CACHE)
        cout, cin = int(w.shape[0]), int(w.shape[1])
        mpad, kpad = (cout + 63) // 64 * 64, (cin + kc - 1) // kc * kc
        G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64,
                         device=w.device)
        U = (G @ w.detach().double() @ G.t()).float().reshape(cout, cin, 16)      # xi = 4a + b
        if (mpad, kpad) != (cout, cin):
            U = torch.nn.functional.pad(U, (0, 0, 0, kpad - cin, 0, mpad - cout))
        # [mb, mt, ln, chunk, ks, lk, w, xi_i] -> [mb, chunk, w, lk, ln, xi_i, ks, mt]
        f = U.reshape(mpad // 64, 4, 16, kpad // kc, kc // 4, 4, waves, 16 // waves).permute(0, 3, 6, 5, 2, 7, 4, 1).contiguous()
        hit = (f, w)
        _FRAGW_CACHE[key] = hit
    return hit[0]


def conv3x3_winograd4_fragments(w):
    """Cached F(4x4,3x3) Winograd-domain weights U = G g G^T (6 x 6 per filter, formed in float64) of a [Cout,Cin,3,3] filter bank
    in the lane-major order heal_conv3x3_winograd4 reads: [Cout/32][Cin/16][wave 8][lane 64][xi_i 9][ks 4], value
    U[xi = 9 (wave & 3) + xi_i][co = 32 mb + 16 (wave >> 2) + (lane & 15)][ci = 16 chunk + 4 ks + (lane >> 4)], zero-padded."""
    key = (w.data_ptr(), w._version, tuple(w.shape), "f4")
    hit = _FRAGW_CACHE.get(key)
    if hit is None:
        if len(_FRAGW_CACHE) > 512:
            _retire_cache(_FRAGW_CACHE)
        cout, cin = int(w.shape[0]), int(w.shape[1])
        mpad, kpad = (cout + 31) // 32 * 32, (cin + 15) // 16 * 16
        G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                          [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64, device=w.device)
        U = (G @ w.detach().double() @ G.t()).float().reshape(cout, cin, 36)      # xi = 6 a + b
        if (mpad, kpad) != (cout, cin):
            U = torch.nn.functional.pad(U, (0, 0, 0, kpad - cin, 0, mpad - cout))
        # [mb, mt, ln, chunk, ks, lk, g, xi_i] -> [mb, chunk, mt, g, lk, ln, xi_i, ks]   (wave = 4 mt + g, lane = 16 lk + ln)
        f = U.reshape(mpad // 32, 2, 16, kpad // 16, 4, 4, 4, 9).permute(0, 3, 1, 6, 5, 2, 7, 4).contiguous()
        hit = (f, w)
        _FRAGW_CACHE[key] = hit
    return hit[0]


def conv3x3_winograd4_ok(n, cout, H, W):
    """F(4x4,3x3) (heal_conv3x3_winograd4) is OPT-IN: HEAL_C3_ALGO=winograd4.  Measured 0.80 - 1.03x of F(2x2,3x3) at the scenes'
    shapes (profiles/r03_wino_f44_vs_f22.json): a quarter of the multiplications instead of 4/9, but twice the transform work per
    output at 32 output channels per block."""
    import os
    return os.environ.get("HEAL_C3_ALGO", "") == "winograd4" and experimental_build()


def conv3x3_algo(stride, n=1, cout=64, H=256, W=256):
    """'winograd' | 'direct' for a shape; HEAL_C3_ALGO overrides for A/B.  Winograd F(2x2,3x3) is the stride-1 formulation
    and runs one 8-wave block per CU on a 16x16-pixel x 64-channel tile: below ~one block per CU the implicit GEMM with its
    smaller tiles fills the chip better (measured crossover ~100 blocks, scripts/conv3x3_bench.py)."""
    import os
    a = os.environ.get("HEAL_C3_ALGO", "")
    if stride != 1 or a == "direct":
        return "direct"
    if a in ("winograd", "winograd4"):
        return "winograd"
    blocks = n * ((cout + 63) // 64) * ((H + 15) // 16) * ((W + 15) // 16)
    return "winograd" if blocks >= 96 else "direct"


_TAPMAJOR_CACHE = {}


def conv_gemm_supported(cin, cout, Wo):
    import os
    return cout % 128 == 0 and cin % 32 == 0 and Wo % 4 == 0 and os.environ.get("HEAL_CONV_GEMM", "1") == "1"


def conv_gemm(x, w, bias=None, residual=None, relu=False, stride=1):
    """heal_conv_gemm: dense 3x3 (padding 1) or 1x1 convolution as an implicit GEMM on 128 x 128 x 32 tiles of the 32x32x2 fp32
    MFMA.  x [n,Cin,H,W], w [Cout,Cin,k,k] (re-laid [Cout, k*k, Cin] once, cached) -> [n,Cout,Ho,Wo]."""
    x = _need(x, torch.float32, "x")
    n, cin, H, W = (int(v) for v in x.shape)
    cout, ks = int(w.shape[0]), int(w.shape[2])
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    key = (w.data_ptr(), w._version, tuple(w.shape))
    hit = _TAPMAJOR_CACHE.get(key)
    if hit is None:
        if len(_TAPMAJOR_CACHE) > 256:
            _WS_RETIRED.extend(v[0] for v in _TAPMAJOR_CACHE.values())   # retired, not freed (captured graphs)
            _TAPMAJOR_CACHE.clear()
        hit = _TAPMAJOR_CACHE[key] = (w.detach().permute(0, 2, 3, 1).reshape(cout, ks * ks, cin).contiguous(), w)
    y = torch.empty((n, cout, Ho, Wo), dtype=torch.float32, device=x.device)
    if residual is not None:
        residual = _need(residual, torch.float32, "residual")
    name = (f"conv3x3_{cin}_{cout}" if ks == 3 else f"conv{ks}x{ks}_{cin}_{cout}") + ("_s2" if stride == 2 else "")
    with _Timed(name, 2.0 * ks * ks * n * cin * cout * Ho * Wo, 4.0 * n * (cin * H * W + cout * Ho * Wo), kernel_events=True):
        _capi.call("heal_conv_gemm", _ptr(x), _ptr(hit[0]), _optr(bias), _optr(residual), n, cin, cout, H, W, ks, int(stride),
                   int(bool(relu)), _ptr(y), _stream())
    return y


_PAD128_CACHE = {}


def conv7x7_s2_supported(cin, cout, W):
    return cin % 32 == 0 and cout <= 128 and ((W - 1) // 2 + 1) % 4 == 0 and os.environ.get("HEAL_CONV_GEMM", "1") == "1"


def conv7x7_s2(x, w, bias=None, relu=False):
    """Conv2d(Cin, Cout <= 128, kernel 7, stride 2, padding 3) (+ bias) (+ ReLU) on heal_conv_gemm's 128 x 128 x 32 implicit GEMM: BevEncode's
    stem of the old-style Lift-Splat model (lss_submodule.py:242).  The kernel wants 128-channel output tiles: the weight (and bias) are
    padded with zero rows once (cached), the first Cout channels of the result are returned."""
    cout = int(w.shape[0])
    key = (w.data_ptr(), w._version, None if bias is None else (bias.data_ptr(), bias._version))
    hit = _PAD128_CACHE.get(key)
    if hit is None:
        if len(_PAD128_CACHE) > 32:
            _retire_cache(_PAD128_CACHE)
        wp = torch.zeros((128,) + tuple(w.shape[1:]), dtype=torch.float32, device=w.device)
        wp[:cout] = w.detach()
        bp = None
        if bias is not None:
            bp = torch.zeros((128,), dtype=torch.float32, device=w.device)
            bp[:cout] = bias.detach()
        hit = _PAD128_CACHE[key] = (wp, bp, w)
    y = conv_gemm(x, hit[0], hit[1], None, relu, 2)
    return y[:, :cout].contiguous() if cout < 128 else y


def conv3x3_winograd_ksplit(n, cin, cout, H, W, waves):
    """K splits of a Winograd launch: 1 unless the grid is small (< 256 blocks: less than one per CU) and the reduction deep
    (>= 32 chunks of 8 input channels); then enough splits for ~768 blocks (three per CU: two resident 4-wave blocks + one queued),
    at least 8 chunks each, none empty.  The camera trunk's Up block (lss_submodule.py:33-50: 432 -> 512 and 512 -> 512 at
    4 x 24 x 32 pixels = 192 blocks of 54 / 64 chunks) is what this is for.  HEAL_C3_KSPLIT forces a value (0 / 1: off)."""
    chunks = (cin + 7) // 8
    blocks = -(-W // 16) * -(-H // (2 * waves)) * n * -(-cout // 64)
    env = os.environ.get("HEAL_C3_KSPLIT")
    want = int(env) if env is not None else (min(chunks // 8, max(2, -(-768 // blocks))) if blocks < 256 and chunks >= 32 else 1)
    if want < 2 or chunks < 2 or (H * W) % 4 or n * want > 65535:
        return 1
    want = min(want, chunks)
    return -(-chunks // -(-chunks // want))       # ceil(chunks / ceil(chunks / want)): no empty split


def conv3x3(x, w, bias=None, residual=None, relu=False, stride=1):
    """Dense 3x3 convolution, padding 1, stride 1 | 2, on the fp32 matrix cores with fused bias (+ residual) (+ ReLU).
    x [n,Cin,H,W] f32 cuda, w [Cout,Cin,3,3] -> [n,Cout,Ho,Wo].  Stride 1 runs the Winograd F(2x2,3x3) formulation
    (heal_conv3x3_winograd), stride 2 the implicit GEMM (heal_conv3x3)."""
    x = _need(x, torch.float32, "x")
    n, cin, H, W = (int(v) for v in x.shape)
    cout = int(w.shape[0])
    if tuple(w.shape[1:]) != (cin, 3, 3) or stride not in (1, 2):
        raise _capi.HealAmdError(f"conv3x3: unsupported weight {tuple(w.shape)} / stride {stride} for input {tuple(x.shape)}")
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((n, cout, Ho, Wo), dtype=torch.float32, device=x.device)
    if residual is not None:
        residual = _need(residual, torch.float32, "residual")
        if tuple(residual.shape) != tuple(y.shape):
            raise _capi.HealAmdError("conv3x3: residual shape mismatch")
    if bias is not None:
        bias = _need(bias, torch.float32, "bias")
    if stride == 1 and conv3x3_winograd4_ok(n, cout, H, W):
        frag = conv3x3_winograd4_fragments(w)
        with _Timed(f"conv3x3w_{cin}_{cout}", 2.0 * 9 * n * cin * cout * Ho * Wo, 4.0 * n * (cin * H * W + cout * Ho * Wo)):
            _capi.call("heal_conv3x3_winograd4", _ptr(x), _ptr(frag), _ptr(bias), _ptr(residual), n, cin, cout, H, W,
                       int(bool(relu)), _ptr(y), _stream())
        return y
    if conv3x3_algo(stride, n, cout, H, W) == "winograd":
        waves = conv3x3_winograd_waves(n, cout, H, W)
        kc = conv3x3_winograd_kc(cin, waves, H, W)
        frag = conv3x3_winograd_fragments(w, waves, kc)
        ksplit = conv3x3_winograd_ksplit(n, cin, cout, H, W, waves) if kc == 8 else 1
        with _Timed(f"conv3x3w_{cin}_{cout}", 2.0 * 9 * n * cin * cout * Ho * Wo, 4.0 * n * (cin * H * W + cout * Ho * Wo),
                    kernel_events=True):
            if ksplit > 1:
                ws = _workspace("conv3x3w_splitk", _capi.query("heal_conv3x3_winograd_splitk_workspace", n, cout, H, W, ksplit),
                                x.device)
                _capi.call("heal_conv3x3_winograd_splitk", _ptr(x), _ptr(frag), _ptr(bias), _ptr(residual), n, cin, cout, H, W,
                           int(bool(relu)), waves, ksplit, _ptr(y), _ptr(ws), ws.numel(), _stream())
            else:
                _capi.call("heal_conv3x3_winograd_kc", _ptr(x), _ptr(frag), _ptr(bias), _ptr(residual), n, cin, cout, H, W,
                           int(bool(relu)), waves, kc, _ptr(y), _stream())
        return y
    if stride == 2 and cin >= 128 and conv_gemm_supported(cin, cout, Wo) and n * Ho * Wo >= 65536:
        # the large stride-2 layers: the 128 x 128 x 32 implicit GEMM on 32x32x2 MFMA (heal_conv_gemm).  Measured
        # (scripts/conv_gemm_bench.py; round 4, after the kernel's weight staging stopped going through scratch memory):
        # 384 -> 256 @256^2 x 8: 2.19 vs 3.69 ms for heal_conv3x3 (MIOpen 2.63) = 106 TFLOP/s; 128 -> 256: 0.74 vs 1.02;
        # 128 -> 128 @256^2 x 5: 0.30 vs 0.31; 256 -> 256 @128^2 x 5 (20 480 pixels: below the bound) 0.35 vs 0.29 for MIOpen
        return conv_gemm(x, w, bias, residual, relu, stride)
    frag = conv3x3_fragments(w)
    with _Timed(f"conv3x3_{cin}_{cout}" + ("_s2" if stride == 2 else ""), 2.0 * 9 * n * cin * cout * Ho * Wo,
                4.0 * n * (cin * H * W + cout * Ho * Wo), kernel_events=True):
        _capi.call("heal_conv3x3", _ptr(x), _ptr(frag), _ptr(bias), _ptr(residual), n, cin, cout, H, W, int(stride),
                   int(bool(relu)), _ptr(y), _stream())
    return y


def channel_dot_supported(x):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and int(x.shape[2] * x.shape[3]) % 4 == 0


def channel_dot(x, weight, bias=None):
    """Pointwise convolution to one output channel: x [n,C,H,W], weight [1,C,1,1] (or [C]), bias [1] -> [n,1,H,W]."""
    x = _need(x, torch.float32, "x")
    n, C, H, W = (int(v) for v in x.shape)
    weight = _need(weight.reshape(-1), torch.float32, "weight")
    if int(weight.numel()) != C:
        raise _capi.HealAmdError(f"channel_dot: weight has {int(weight.numel())} elements for {C} channels")
    y = torch.empty((n, 1, H, W), dtype=torch.float32, device=x.device)
    with _Timed(f"channel_dot_{C}", 2.0 * n * C * H * W, 4.0 * n * (C + 1) * H * W):
        _capi.call("heal_channel_dot", _ptr(x), _ptr(weight), _ptr(_need(bias, torch.float32, "bias")) if bias is not None else None,
                   n, C, H * W, _ptr(y), _stream())
    return y


def layernorm_nchw(x, gamma, beta, eps):
    """LayerNorm over the channel axis of x [n,C,H,W] (biased variance, eps inside the sqrt)."""
    x = _need(x, torch.float32, "x")
    n, C, H, W = (int(v) for v in x.shape)
    y = torch.empty_like(x)
    _capi.call("heal_layernorm_nchw", _ptr(x), _ptr(_need(gamma, torch.float32, "gamma")),
               _ptr(_need(beta, torch.float32, "beta")), n, C, H * W, float(eps), _ptr(y), _stream())
    return y


_SE_T_CACHE = {}


def se_gate(mean, w_reduce, b_reduce, w_expand, b_expand, scale=1.0, tiles=1):
    """Squeeze-excite gate: mean [n,C] (any trailing 1-dims), w_reduce [S,C,1,1], w_expand [C,S,1,1] -> gate [n,C].
    tiles = T, scale = 1/(H*W): `mean` holds the per-tile sums [n,C,T] of depthwise_conv(channel_sums=True)."""
    n, C = int(mean.shape[0]), int(mean.shape[1])
    S = int(w_reduce.shape[0])
    mean = _need(mean.reshape(n, C, int(tiles)), torch.float32, "mean")
    w_expand = _need(w_expand, torch.float32, "w_expand")
    key = (w_expand.data_ptr(), w_expand._version)
    hit = _SE_T_CACHE.get(key)
    if hit is None:
        if len(_SE_T_CACHE) > 256:
            _SE_T_CACHE.clear()
        hit = (w_expand.detach().reshape(C, S).t().contiguous(), w_expand)  # [S,C]; keep the source alive (key = address)
        _SE_T_CACHE[key] = hit
    gate = torch.empty((n, C), dtype=torch.float32, device=mean.device)
    _capi.call("heal_se_gate", _ptr(mean), _ptr(_need(w_reduce, torch.float32, "w_reduce")),
               _ptr(_need(b_reduce, torch.float32, "b_reduce")), _ptr(hit[0]),
               _ptr(_need(b_expand, torch.float32, "b_expand")), n, C, S, float(scale), int(tiles), _ptr(gate),
               _stream())
    return gate


def resnext_bottleneck(x, w1_frag, b1, w2, b2, w3_frag, b3):
    """Fused stride-1 ResNeXt bottleneck (K7b).  x [n,C,H,W] -> y [n,C,H,W]."""
    x = _need(x, torch.float32, "x")
    n, C, H, W = (int(v) for v in x.shape)
    y = torch.empty_like(x)
    with _Timed(f"resnext_bottleneck_c{C}"):
        _capi.call("heal_resnext_bottleneck", _ptr(x), _ptr(w1_frag), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(w3_frag),
                   _ptr(b3), n, C, H, W, _ptr(y), _stream())
    return y


def upsample2x_bilinear(x):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True).  x [n,C,H,W] -> [n,C,2H,2W]."""
    x = _need(x, torch.float32, "x")
    n, C, H, W = (int(v) for v in x.shape)
    y = torch.empty((n, C, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    _capi.call("heal_upsample2x_bilinear", _ptr(x), n, C, H, W, _ptr(y), _stream())
    return y


def depthwise_tiles(Ho, Wo):
    """Output tiles (32 x 8 pixels) of one depthwise_conv channel = the last dimension of its `channel_sums` argument."""
    return ((Wo + 31) // 32) * ((Ho + 7) // 8)


def depthwise_conv(x, weight, bias, stride, pad, act="none", channel_sums=None):
    """Depthwise k x k conv; pad = (left, right, top, bottom) zero padding; act in none|relu|silu.  channel_sums=True: also
    return the per-tile sums [n, C, T] of the outputs (the squeeze of an SE stage, folded into this launch) -> (y, sums)."""
    x = _need(x, torch.float32, "x")
    weight = _need(weight, torch.float32, "weight")
    n, C, H, W = (int(v) for v in x.shape)
    k = int(weight.shape[-1])
    pl, pr, pt, pb = (int(v) for v in pad)
    Ho = (H + pt + pb - k) // stride + 1
    Wo = (W + pl + pr - k) // stride + 1
    y = torch.empty((n, C, Ho, Wo), dtype=torch.float32, device=x.device)
    sums = torch.empty((n, C, depthwise_tiles(Ho, Wo)), dtype=torch.float32, device=x.device) if channel_sums else None
    _capi.call("heal_depthwise_conv", _ptr(x), _ptr(weight), _ptr(bias), n, C, H, W, k, int(stride), pt, pl, Ho, Wo,
               {"none": 0, "relu": 1, "silu": 2}[act], _ptr(y), _ptr(sums) if sums is not None else None, _stream())
    return (y, sums) if channel_sums else y


def fill_bytes(t, byte):
    """t (contiguous CUDA tensor, size a multiple of 4 bytes) <- `byte` in every byte, by the library's fill KERNEL (heal_fill_bytes;
    include/heal_amd.h explains why not a memset)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or not t.is_contiguous():
        raise _capi.HealAmdError("fill_bytes: a contiguous CUDA tensor is required")
    _capi.call("heal_fill_bytes", _ptr(t), int(byte), t.numel() * t.element_size(), _stream())
    return t
