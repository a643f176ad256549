#!/usr/bin/env python
"""bench.py -- scenes/s of HEAL's per-frame perception hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload scene5|pair|single]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one synthetic OPV2V-shaped scene through the whole path with its inputs (device point
clouds, anchors) already resident in HBM: voxelise (K1) -> PFN+scatter (K2) -> BEV backbones ->
pyramid stages -> warp + occupancy-softmax fusion (K5) -> deblocks / shrink / heads -> decode +
rotated NMS (K8) -> host sees the boxes.  N = 1: everything on one GPU.  N > 1: the agents of the
scene are sharded one per rank with a single exchange of ego-frame maps (heal_amd/dist.py: gather to rank 0 by default);
`value` = scenes completed per second by the whole job.

Rank 0 prints ONE JSON line (metric/roofline/cpu_baseline as the contract in DESIGN.md describes).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP32_PEAK_TFLOPS = 157.3   # fp32 matrix/vector peak, same guide

WORKLOADS = {
    # name: (agent modalities in scene order, description)
    "scene5": (["m1", "m1", "m1", "m2", "m4"],
               "5-agent OPV2V-H scene (BASELINE config 4): 3x PointPillars LiDAR (m1) + Lift-Splat camera agents "
               "m2 (EfficientNet-b0, 4x384x512) and m4 (ResNet101 stem, 4x336x448), PyramidFusion, range +-102.4 m"),
    "scene5_lidar": (["m1"] * 5, "5-agent OPV2V scene, 5x PointPillars(m1) + PyramidFusion, range +-102.4 m"),
    "pair": (["m1", "m1"], "2-agent OPV2V scene, PointPillars + PyramidFusion (BASELINE config 3)"),
    "single": (["m1"], "single-agent PointPillars through the collaborative model (BASELINE config 2)"),
    # the YAMLs' NATIVE training range (hypes_yaml/opv2v/Single/m1_pointpillar_pretrain.yaml:17: [-96, -48, -3, 96, 48, 1] -> a
    # 480 x 240 pillar grid); tools/inference.py:34,54-73 widens every range to +-102.4 m, which is what the other workloads use
    "single_native": (["m1"], "single-agent PointPillars (BASELINE configs 1/2) at the YAML's native range [-96, -48, 96, 48]: "
                              "480 x 240 pillars"),
    "pair_native": (["m1", "m1"], "2-agent PointPillars + PyramidFusion (BASELINE config 3) at the YAML's native range: 480 x 240 pillars"),
    "scene8_second_v2xvit": (["m3"] * 8, "8-agent synthetic scene, SECOND (sparse conv) encoders + plain BEV backbone + "
                                         "V2X-ViT fusion, heter_model_baseline (BASELINE config 5)"),
}


NATIVE_RANGE = [-96, -48, -3, 96, 48, 1]
WORKLOAD_RANGE = {"single_native": NATIVE_RANGE, "pair_native": NATIVE_RANGE}


def k2_algorithmic_bytes(n_points_per_voxel_rows, n_voxels, ny, nx, channels=64):
    """16*M*P + 20*M + 4*C*ny*nx bytes per agent."""
    return 16 * n_voxels * n_points_per_voxel_rows + 20 * n_voxels + 4 * channels * ny * nx


def dense_flops(model, sample_input_fn):
    """FLOPs of the dense conv/linear part of one scene, counted by torch's flop counter."""
    try:
        from torch.utils.flop_counter import FlopCounterMode
        with FlopCounterMode(display=False) as fc:
            sample_input_fn()
        return float(fc.get_total_flops())
    except Exception:
        return None


def cpu_baseline(hypes, scene_cpu, cls_shift=0.0):
    """The oracle (CPU port of the reference algorithm, oracle/) timed on this box's host cores on a bounded sample: ONE
    scene of the SAME workload -- LiDAR agents through the C voxeliser + numpy PFN / scatter, camera agents through the
    plain-torch trunks (oracle/trunks.py), Up, heads, numpy lift + voxel pooling; backbones, ConvNeXt aligners, pyramid
    fusion (numpy warp + fuse), shrink head, heads as torch-CPU fp32 convolutions; decode + C rotated NMS.  Reported, never
    the thing measured above."""
    from heal_amd.opencood.tools.train_utils import create_model
    from heal_amd.pipeline import fill_deterministic
    from oracle import cref, model_ref
    from oracle import oracle_np as O
    # torch's CPU convolutions stop scaling (and regress) far below the 256 hardware threads of the
    # GPU box's host; use at most 32 threads and report that number
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sd = fill_deterministic(create_model(hypes), 0).state_dict()
    args = hypes["model"]["args"]
    r = args["lidar_range"]
    mods = list(scene_cpu.modalities)
    anchors = O.generate_anchor_box(r, 0.4, 0.4, int(round((r[3] - r[0]) / 0.4)), int(round((r[4] - r[1]) / 0.4)),
                                    3.9, 1.6, 1.56, [0, 90],
                                    feature_stride=hypes["postprocess"]["anchor_args"].get("feature_stride", 2))
    t0 = time.perf_counter()
    data = {"agent_modality_list": mods, "pairwise_t_matrix": np.asarray(scene_cpu.pairwise)}
    if hypes["model"]["core_method"] == "heter_model_baseline":
        # BASELINE config 5: C voxeliser at 0.1 m, numpy MeanVFE + the sparse SECOND encoder on its rule pairs, torch-CPU fp32
        # BaseBEVBackbone / shrinker / V2X-ViT (oracle/v2xvit_ref.py), heads, decode + C rotated NMS
        m = mods[0]
        enc = args[m]["encoder_args"]
        vs, cs, ns = [], [], []
        for b, k in enumerate(sorted(scene_cpu.points)):
            v, c, n = cref.voxelize(scene_cpu.points[k].numpy(), r, enc["voxel_size"], 5, 70000, batch_idx=b)
            vs.append(v); cs.append(c); ns.append(n)
        data[f"inputs_{m}"] = {"voxel_features": np.concatenate(vs), "voxel_coords": np.concatenate(cs),
                               "voxel_num_points": np.concatenate(ns)}
        out = model_ref.heter_model_baseline(sd, args, data)
        out["cls_preds"] = out["cls_preds"] + cls_shift
        O.post_process(out["cls_preds"], out["reg_preds"], out["dir_preds"], anchors, 0.2, 0.7853, 2, 0.15,
                       np.eye(4, dtype=np.float32), r)
        dt = time.perf_counter() - t0
        return {"value": 1.0 / dt, "unit": "scenes/s", "cores": cores, "kind": "port",
                "sample": f"1 scene of the same workload ({len(mods)} agents: {' '.join(mods)}) through oracle/ (C voxeliser, numpy "
                          f"MeanVFE + sparse SECOND encoder on its rule pairs, torch-CPU fp32 BEV backbone / shrinker / V2X-ViT / "
                          f"heads, C rotated NMS), {dt:.1f} s on {cores} threads"}
    vs, cs, ns = [], [], []
    for b, k in enumerate(sorted(scene_cpu.points)):
        v, c, n = cref.voxelize(scene_cpu.points[k].numpy(), r, [0.4, 0.4, 4], 32, 70000, batch_idx=b)
        vs.append(v); cs.append(c); ns.append(n)
    if vs:
        data["inputs_m1"] = {"voxel_features": np.concatenate(vs), "voxel_coords": np.concatenate(cs),
                             "voxel_num_points": np.concatenate(ns)}
    for m in sorted(set(mods)):
        ids = [i for i, mm in enumerate(mods) if mm == m and i in scene_cpu.cameras]
        if ids:
            data[f"inputs_{m}"] = {key: np.stack([scene_cpu.cameras[i][key].numpy() for i in ids])
                                   for key in ("imgs", "rots", "trans", "intrins", "post_rots", "post_trans")}
    out = model_ref.heter_pyramid_collab(sd, args, data)
    out["cls_preds"] = out["cls_preds"] + cls_shift  # same calibrated head bias as the GPU pipeline
    O.post_process(out["cls_preds"], out["reg_preds"], out["dir_preds"], anchors, 0.2, 0.7853, 2, 0.15,
                   np.eye(4, dtype=np.float32), r)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "scenes/s", "cores": cores, "kind": "port",
            "sample": f"1 scene of the same workload ({len(mods)} agents: {' '.join(mods)}) through oracle/ (C voxeliser, numpy "
                      f"PFN / lift / voxel pooling / warp + fuse, torch-CPU fp32 image trunks and convolutions, C rotated "
                      f"NMS), {dt:.1f} s on {cores} threads"}


_PMC = None
_PMC_NOTE = {}


def lib_stamp():
    """Identity of the loaded libheal_amd.so: the build stamp (SHA-256 of csrc + headers + flags) heal_amd.build wrote beside it."""
    try:
        from heal_amd import build
        return open(os.path.join(build.LIBDIR, "libheal_amd.stamp")).read().strip()
    except OSError:
        return None


def pmc_traffic(workload, *prefixes, per_launch_kernels=None):
    """HBM bytes per launch of a kernel family from the newest COMMITTED rocprofv3 PMC summary of THIS workload
    (profiles/r0N_pmc_traffic_<workload>.json, scripts/runs/r06_profile.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes, bytes =
    2 FETCH + WRITE per the guide's gfx950 correction).  A file read, not a measurement of this run -- so it is REFUSED (None, the
    reason in `traffic_source`) unless the summary records the build stamp of the library that is loaded now: counters of other
    kernels say nothing about these.  A family that is a SEQUENCE of kernels per launch sums the per-dispatch
    means of its kernels; a family of one kernel with many shapes takes its mean."""
    global _PMC
    if _PMC is None:
        _PMC = {}
    if workload not in _PMC:
        _PMC[workload] = {}
        _PMC_NOTE[workload] = "no committed PMC summary for this workload"
        for tag in ("r06", "r05", "r04", "r03"):      # the newest committed PMC summary of this workload (the box-local copy profile_round.sh writes counts too)
            path = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic_{workload}.json")
            if os.path.exists(path):
                j = json.load(open(path))
                if j.get("lib_stamp") and j.get("lib_stamp") == lib_stamp():
                    _PMC[workload] = j["kernels"]
                    _PMC_NOTE[workload] = (f"committed rocprofv3 PMC summary profiles/{tag}_pmc_traffic_{workload}.json of THIS library "
                                           f"build (stamp {j['lib_stamp'][:12]}), not measured in this run")
                else:
                    _PMC_NOTE[workload] = (f"refused: profiles/{tag}_pmc_traffic_{workload}.json was recorded for library build "
                                           f"{str(j.get('lib_stamp'))[:12]}, the loaded one is {str(lib_stamp())[:12]}")
                break
    ks = _PMC[workload]
    tot, hit = 0.0, False
    for pre in prefixes:
        m = [v for k, v in ks.items() if k.startswith(pre)]
        if m:
            hit = True
            d = sum(v["dispatches"] for v in m)
            tot += sum(v["bytes_per_dispatch"] * v["dispatches"] for v in m) / max(d, 1)
    return round(tot, 1) if hit else None


def k3_traffic(workload):
    """Mean HBM bytes per sparse-convolution launch over BOTH kernels of the family (k_sp_conv2: the table path, k_sp_tiles: the thin layers)."""
    pmc_traffic(workload, "heal::k_sp_conv2<")
    ks = [v for k, v in (_PMC.get(workload) or {}).items() if k.startswith("heal::k_sp_conv2<") or k.startswith("heal::k_sp_tiles<")]
    d = sum(v["dispatches"] for v in ks)
    return round(sum(v["bytes_per_dispatch"] * v["dispatches"] for v in ks) / d, 1) if d else None


_KSTATS = {}


def rocprof_mean_us(workload, prefix):
    """Mean duration (us) of the kernels whose name starts with `prefix` in the newest committed `rocprofv3 --kernel-trace --stats`
    summary of this workload's bench run (profiles/r0N_kernel_stats_<workload>.csv) -- the IN-GRAPH duration: the kernel as the
    timed region runs it, sharing the chip with the other modality streams and the second frame in flight -- or None.  Accepted only
    from a summary whose sidecar (.stamp) names the loaded library build."""
    if workload not in _KSTATS:
        _KSTATS[workload] = None
        for tag in ("r06",):
            path = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_{workload}.csv")
            side = path[:-4] + ".stamp"
            if os.path.exists(path) and os.path.exists(side) and open(side).read().strip() == lib_stamp():
                import csv
                _KSTATS[workload] = (tag, list(csv.DictReader(open(path))))
                break
    if not _KSTATS[workload]:
        return None
    tag, rows = _KSTATS[workload]
    calls = tot = 0.0
    for r in rows:
        name = r.get("Name") or r.get("KernelName") or ""
        if name.startswith(prefix):
            c = float(r.get("Calls") or 0)
            calls += c
            tot += float(r.get("TotalDurationNs") or 0)
    return round(tot / calls / 1e3, 2) if calls else None


def _entry(kernel, bound, achieved, launches, launch_ms, traffic=None, **extra):
    peak = HBM_PEAK_GBS if bound == "hbm" else FP32_PEAK_TFLOPS
    d = {"kernel": kernel, "bound": bound, "achieved": round(achieved, 2), "peak": peak,
         "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
         "traffic_source": _PMC_NOTE.get(_entry.workload, "no committed PMC summary for this workload"),
         "launches": launches, "launch_ms": round(launch_ms, 5)}
    if bound == "hbm" and traffic is not None and launch_ms > 0:
        # `frac` prices the kernel at SURVEY 8d's ALGORITHMIC bytes, which for a fused kernel include traffic it no
        # longer performs (K4: the dense cell map it stopped writing).  This is the same launch duration with the bytes the counters saw
        # cross the HBM interface: what the kernel actually sustains.  Both are printed; neither replaces the other.
        d["frac_counter_bytes"] = round(traffic / (launch_ms * 1e-3) / 1e9 / peak, 4)
    d.update(extra)
    return d


_entry.workload = None


def roofline_report(a, work, timing, scene, mods, m_per_agent, hypes, solo, world, n_agents, sp_trace, ny, nx, in_graph=None):
    """`roofline` = the hand-written kernel family that takes the most time in this workload; `roofline_other` = every other
    north-star kernel present.  achieved = algorithmic bytes (HBM-bound kernels) or FLOPs (MFMA-bound kernels)
    of the recorded launches / their summed HIP-event durations, i.e. per-launch work / average launch duration."""
    from heal_amd.dist import owned_agents
    from heal_amd.pipeline import Scene
    _entry.workload = a.workload
    fam = {}

    def add(name, kernel, bound, w):
        f = fam.setdefault(name, {"kernel": kernel, "bound": bound, "calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        f["calls"] += w["calls"]; f["ms"] += w["total_ms"]; f["flops"] += w["flops"]; f["bytes"] += w["bytes"]
    for name, w in work.items():
        if name.startswith("conv1x1_"):
            # ONE family entry: every launch of the kernel (the >= 1 GFLOP population alone flattered the headline).
            # The two populations -- fusion-backbone / trunk convolutions with >= 1 GFLOP per launch (MFMA-bound; most of the family's
            # time) and the small ones (image trunks at 1/16 .. 1/32 resolution, heads: launch / latency-bound) -- stay as sub-fields.
            add("conv1x1", "K7 heal_conv1x1, ALL launches (pointwise convolutions of the fusion backbone, the image trunks at 1/8..1/32 "
                           "resolution, the ConvNeXt aligners, deblocks and heads; fp32 MFMA, fused epilogues); `split` = the launches of "
                           ">= 1 GFLOP and those below", "mfma", w)
            add("conv1x1_big" if w["flops"] >= 1e9 * max(w["calls"], 1) else "conv1x1_small", "", "mfma", w)
        elif name.startswith("conv3x3w_"):
            add("conv3x3w", "K7 heal_conv3x3_winograd (dense 3x3 stride 1, F(2x2,3x3) on fp32 MFMA; achieved = EXECUTED matrix "
                            "FLOPs 2*16*Cin*Cout*tiles = direct/2.25, `direct_equiv_tflops` = the direct convolution's count)",
                "mfma", w)
        elif name.startswith("conv3x3_"):
            add("conv3x3", "K7 heal_conv3x3 (dense 3x3 implicit GEMM on fp32 MFMA: stride 2 and small maps)", "mfma", w)
        elif name.startswith("grouped_conv3x3"):
            add("grouped", "K7 heal_grouped_small_conv3x3 (32-group 3x3 of the ResNeXt bottlenecks on the 16-block 4x4x1 fp32 MFMA; "
                           "bytes = input read once + output written once)", "hbm", w)
        elif name.startswith("linear_"):
            add("linear", "K6c heal_linear (V2X-ViT token-major fp32 MFMA GEMM: LayerNorm prologue, bias / GELU / residual / "
                          "split-attention merge epilogues)", "mfma", w)
        elif name.startswith("window_attention_ws"):
            ws = int(name[len("window_attention_ws"):])
            # per token 4 ws^2 * 256 FLOPs against 4 KB of q | k | v | out: 16 FLOP/B at ws 4, 64 at ws 8 (below the fp32 ridge of
            # 20-30 FLOP/B of achievable HBM: HBM-bound), 256 at ws 16 (MFMA-bound)
            add(f"wattn{ws}", f"K6b heal_window_attention, window {ws} x {ws} (V2X-ViT PyramidWindowAttention branch: scores + relative "
                              f"position bias + softmax + p v in one kernel; q | k | v read once, result written once)",
                "mfma" if ws >= 16 else "hbm", w)
        elif name.startswith("warp_fuse"):
            add("k5", "K5 heal_warp_fuse_levels (warp + occupancy-softmax fusion, ALL pyramid levels in one launch, source footprints "
                      "staged through LDS; heal_warp_fuse per level where the model fuses level by level)", "hbm", w)
    entries = {}
    split1 = {}
    for key in ("conv1x1_big", "conv1x1_small"):
        f = fam.pop(key, None)
        if f and f["ms"] > 0:
            t_ = f["flops"] / (f["ms"] * 1e-3) / 1e12
            split1["ge_1_gflop" if key.endswith("big") else "lt_1_gflop"] = {
                "launches": f["calls"], "launch_ms": round(f["ms"] / max(f["calls"], 1), 5), "step_ms": round(f["ms"] / max(a.steps, 1), 4),
                "tflops": round(t_, 2), "frac": round(t_ / FP32_PEAK_TFLOPS, 4)}
    for key, f in fam.items():
        if f["ms"] <= 0:
            continue
        ach = (f["flops"] / (f["ms"] * 1e-3) / 1e12) if f["bound"] == "mfma" else (f["bytes"] / (f["ms"] * 1e-3) / 1e9)
        extra = {"step_ms": round(f["ms"] / max(a.steps, 1), 4)}
        if key in ("conv1x1", "conv3x3w", "conv3x3", "grouped", "k5", "linear"):
            # one kernel per launch: the events are stamped by the launch itself (heal_next_launch_events / hipExtLaunchKernelGGL)
            # with the kernel's own begin / end -- the duration a rocprofv3 kernel trace reports for it
            extra["duration"] = "kernel-own begin / end stamps"
        if key.startswith("wattn"):      # both roofs for the window-attention kernels
            extra["tflops"] = round(f["flops"] / (f["ms"] * 1e-3) / 1e12, 2)
            extra["hbm_gbs"] = round(f["bytes"] / (f["ms"] * 1e-3) / 1e9, 1)
        if key == "conv1x1":
            extra["split"] = split1
            # the same kernel inside the replayed graphs (two frames in flight, concurrent modality streams): rocprofv3's mean
            extra["rocprof_in_graph_mean_us"] = rocprof_mean_us(a.workload, "void heal::k_conv1x1<") or rocprof_mean_us(a.workload, "heal::k_conv1x1<")
        if key == "conv3x3w":   # the wrapper counts the direct convolution's FLOPs; the kernel executes 16/36 of them
            extra["direct_equiv_tflops"] = round(ach, 2)
            ach = ach / 2.25
        tr = pmc_traffic(a.workload, *{"conv1x1": ("heal::k_conv1x1<",),
                                       "conv3x3w": ("heal::k_conv3x3_wino<",), "conv3x3": ("heal::k_conv3x3<",),
                                       "grouped": ("heal::k_gconv_small<",), "k5": ("heal::k_warp_fuse",),
                                       "linear": ("heal::k_linear",)}.get(key, ()))
        entries[key] = (f["ms"], _entry(f["kernel"], f["bound"], ach, f["calls"], f["ms"] / max(f["calls"], 1), tr, **extra))
    # K2: algorithmic bytes of the collated LiDAR agents this rank encodes per launch
    if "pfn_scatter" in timing:
        calls, mean_ms = timing["pfn_scatter"]
        lidar_ids = [i for i, m in enumerate(mods) if m == "m1"]
        if not solo:
            lidar_ids = [i for i in lidar_ids if i in owned_agents(n_agents, 0, world)]
        order = sorted(scene.points)
        m_launch = [m_per_agent[order.index(i)] for i in lidar_ids if i in order]
        bytes_per_launch = float(sum(k2_algorithmic_bytes(32, m, ny, nx) for m in m_launch))
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_k2_traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch from the rocprofv3 PMC passes (scripts/profile_round.sh)
            tj = json.load(open(tpath))
            if int(tj.get("agents_per_launch", -1)) == len(m_launch):
                traffic = float(tj["k2_traffic_bytes_per_launch"])
        entries["k2"] = (calls * mean_ms, _entry(
            f"K2 heal_pfn_scatter, {len(m_launch)} collated LiDAR agents per launch (k_pfn + k_canvas)", "hbm",
            bytes_per_launch / (mean_ms * 1e-3) / 1e9, calls, mean_ms, traffic, bytes_per_launch=bytes_per_launch))
    if "pfn_pillars" in timing:
        # round 6: K2 WITHOUT the dense canvas (heal_pfn_pillars: pillar features + cell -> pillar map); its consumer, the first block of
        # the LiDAR backbone, reads the pillars (heal_pillar_stem_block).  `frac` prices the bytes this operator really has to move
        # (voxels + coords + counts read, pillar features + map written); SURVEY 8d's K2 formula -- which includes the 4*64*ny*nx canvas
        # this path no longer writes -- is printed beside it, as is the consumer.
        calls, mean_ms = timing["pfn_pillars"]
        lidar_ids = [i for i, m in enumerate(mods) if m == "m1"]
        if not solo:
            lidar_ids = [i for i in lidar_ids if i in owned_agents(n_agents, 0, world)]
        order = sorted(scene.points)
        m_launch = [m_per_agent[order.index(i)] for i in lidar_ids if i in order]
        moved = float(sum(16 * 32 * m + 20 * m + 256 * m for m in m_launch) + 4 * ny * nx * len(m_launch))
        survey = float(sum(k2_algorithmic_bytes(32, m, ny, nx) for m in m_launch))
        extra = {"bytes_per_launch": moved, "survey_8d_bytes_with_canvas": survey, "canvas_written": False}
        if "pillar_stem_block" in timing:
            extra["consumer"] = {"kernel": "heal_pillar_stem_block (3x3/2 conv1 + 1x1/2 downsample of the first LiDAR BasicBlock straight from "
                                           "the pillars: only output pixels that see a pillar are multiplied; replaces k_canvas + heal_conv3x3 "
                                           "stride 2 + heal_conv1x1 stride 2)", "launch_ms": round(timing["pillar_stem_block"][1], 5)}
        entries["k2"] = (calls * mean_ms, _entry(
            f"K2 heal_pfn_pillars, {len(m_launch)} collated LiDAR agents per launch (k_pfn4 + the cell map fill; no canvas)", "hbm",
            moved / (mean_ms * 1e-3) / 1e9, calls, mean_ms, pmc_traffic(a.workload, "heal::k_pfn"), **extra))
    # K4: one camera agent per launch (logits + features read, canvas written)
    if "bev_pool" in timing:
        cam_ids = [i for i, m in enumerate(mods) if m in Scene.CAMERA_DIMS and (solo or i in owned_agents(n_agents, 0, world))]
        if cam_ids:
            per_call = []
            margs = hypes["model"]["args"]
            for i in cam_ids:
                # from the model configuration of that modality (not constants): D depth bins, C image features, the BEV grid;
                # feature map = ceil(H / 8) x ceil(W / 8) (the trunks round up)
                ca = margs[mods[i]]["encoder_args"]
                gc = ca["grid_conf"]
                D, C = int(gc["ddiscr"][2]), int(ca["img_features"])
                nx = int(round((gc["xbound"][1] - gc["xbound"][0]) / gc["xbound"][2]))
                ny = int(round((gc["ybound"][1] - gc["ybound"][0]) / gc["ybound"][2]))
                nz = max(1, int(round((gc["zbound"][1] - gc["zbound"][0]) / gc["zbound"][2])))
                H, W = ca["data_aug_conf"]["final_dim"]
                ds = int(ca["img_downsample"])
                fhw = -(-H // ds) * -(-W // ds)
                n_cam = len(ca["data_aug_conf"].get("cams", [0, 1, 2, 3])) if isinstance(ca["data_aug_conf"].get("cams"), (list, tuple)) else 4
                per_call.append(4.0 * (n_cam * D * fhw + n_cam * C * fhw + C * nz * ny * nx))
            calls, mean_ms = timing["bev_pool"]
            # bytes per LAUNCH: one camera agent per launch, or (round 6, heal_bev_pool_scatter_multi) every camera agent of the scene in one
            launches_per_step = max(1, round(calls / max(a.steps, 1)))
            bts = sum(per_call) / launches_per_step
            extra = {}
            if "bev_stem_block" in timing:
                extra["consumer"] = {"kernel": "heal_bev_stem_block (first BasicBlock of the camera backbone reads the sparse map; "
                                               "fp32 MFMA, not part of K4's duration)", "launch_ms": round(timing["bev_stem_block"][1], 5)}
            if "bev_pool_emit" in timing:
                extra["dense_emit"] = {"kernel": "heal_bev_pool_emit (dense canvas for consumers that want it)",
                                       "launch_ms": round(timing["bev_pool_emit"][1], 5)}
            entries["k4"] = (calls * mean_ms, _entry(
                f"K4 heal_bev_pool_scatter{'_multi' if launches_per_step < len(per_call) else ''}, {len(per_call) // launches_per_step} camera agent(s) per launch (ONE kernel, k_lss_scatter: lift + splat into the sparse "
                "pixel-major BEV map; duration = the kernel's own begin/end stamps (hipExtLaunchKernelGGL events), mean over the "
                "scene's camera agents; bytes = SURVEY 8d: logits + features read + the dense [C,ny,nx] map the operator stands for, "
                "which is no longer written)",
                "hbm", bts / (mean_ms * 1e-3) / 1e9, calls, mean_ms,
                pmc_traffic(a.workload, "heal::k_lss_scatter"), bytes_per_launch=bts, **extra))
    # K1: 16 N + 16 M P + 20 M bytes per agent, all LiDAR agents of a modality in one launch chain
    if "voxelize" in timing and scene.points:
        calls, mean_ms = timing["voxelize"]
        P = 5 if a.workload == "scene8_second_v2xvit" else 32
        bts = float(sum(16 * int(scene.points[k].shape[0]) + 16 * m * P + 20 * m
                        for k, m in zip(sorted(scene.points), m_per_agent)))
        extra = {}
        if in_graph and "voxelize" in in_graph:      # the launch chain as the timed region runs it (captured graph), not host-paced
            extra = {"eager_event_pair_ms": round(mean_ms, 5), "duration": "per-call period of the launch chain inside a captured graph"}
            mean_ms = in_graph["voxelize"]
        entries["k1"] = (calls * mean_ms, _entry("K1 heal_voxelize_batch (all LiDAR agents of the scene: insert / assign + fill / select + write, self-cleaning tables)", "hbm",
                                                 bts / (mean_ms * 1e-3) / 1e9, calls, mean_ms,
                                                 pmc_traffic(a.workload, "heal::k_voxb_insert", "heal::k_vox_assign", "heal::k_vox_select_write"),
                                                 bytes_per_launch=bts, **extra))
    if "decode_nms" in timing:
        calls, mean_ms = timing["decode_nms"]
        hw = 256 * 256 if a.workload != "scene8_second_v2xvit" else 128 * 128
        bts = 4.0 * 20 * hw
        extra = {}
        if in_graph and "decode_nms" in in_graph:
            extra = {"eager_event_pair_ms": round(mean_ms, 5), "duration": "per-call period of the launch chain inside a captured graph"}
            mean_ms = in_graph["decode_nms"]
        entries["k8"] = (calls * mean_ms, _entry("K8 heal_decode_nms (decode + filters + rotated NMS: a 4-byte fill + four kernels; "
                                                 "latency-bound)", "hbm",
                                                 bts / (mean_ms * 1e-3) / 1e9, calls, mean_ms,
                                                 pmc_traffic(a.workload, "heal::k_decode_key", "heal::k_rank_prepare",
                                                             "heal::k_nms_mask", "heal::k_nms_reduce"),
                                                 bytes_per_launch=bts, **extra))
    # K3: per sparse layer N_in / N_out / R, bytes = 4 (N_in C_in + N_out C_out) + 4 K C_in C_out + 8 R, flops = 2 R C_in C_out
    if sp_trace:
        layers, tot_ms, tot_b, tot_f = [], 0.0, 0.0, 0.0
        for t in sp_trace:
            n_in = int(t["n_in"].item()) if hasattr(t["n_in"], "item") else int(t["n_in"])
            n_out = int(t["n_out"].item()) if hasattr(t["n_out"], "item") else int(t["n_out"])
            n_out = min(n_out, int(t["nbr"].shape[0]))
            R = int((t["nbr"][:n_out] >= 0).sum().item())
            ms = t["events"][0].elapsed_time(t["events"][1])
            bts = 4.0 * (n_in * t["cin"] + n_out * t["cout"]) + 4.0 * t["K"] * t["cin"] * t["cout"] + 8.0 * R
            fl = 2.0 * R * t["cin"] * t["cout"]
            layers.append({"cin": t["cin"], "cout": t["cout"], "K": t["K"], "N_in": n_in, "N_out": n_out, "R": R,
                           "us": round(ms * 1e3, 1), "GB/s": round(bts / (ms * 1e-3) / 1e9, 1),
                           "frac_hbm": round(bts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "TFLOP/s": round(fl / (ms * 1e-3) / 1e12, 2),
                           "frac_mfma": round(fl / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)})
            tot_ms += ms; tot_b += bts; tot_f += fl
        if tot_ms > 0:
            # which roof applies: 2 R Cin Cout flops over 4 (N_in Cin + N_out Cout) + 8 R bytes is 100-180 flop/B on the 32- and
            # 64-channel layers that carry 95 % of the time, against a machine balance of 157.3 TFLOP/s : 8 TB/s = 19.7 flop/B --
            # exact-fp32 sparse convolution is bound by the fp32 matrix cores, so `frac` is the MFMA fraction (useful flops:
            # the tile padding the kernel executes is not counted); the algorithmic-byte rate is reported beside it
            rb = timing.get("sp_rulebook", (0, 0.0))
            rb_ms = rb[0] * rb[1] / max(a.steps, 1)                       # site sort, hash / rank structures, neighbour tables, out sites
            entries["k3"] = (tot_ms * max(a.steps, 1), _entry(
                "K3 heal_sp_conv / heal_sp_conv_tiles (pair-compacted gather-GEMM on fp32 MFMA; the c_in <= 16 layers on the pair-tile rulebook; "
                "all sparse layers of one step, useful flops 2 R Cin Cout; `traffic` = mean HBM bytes per convolution launch over both kernels; "
                "`frac` prices the convolution launches; `rulebook_ms` = the rulebook launches of the same step (site sort, hash / rank "
                "structures, neighbour tables, output sites), `step_ms_with_rulebook` = both)", "mfma",
                tot_f / (tot_ms * 1e-3) / 1e12, len(layers), tot_ms / len(layers), k3_traffic(a.workload),
                step_ms=round(tot_ms, 4), rulebook_ms=round(rb_ms, 4), step_ms_with_rulebook=round(tot_ms + rb_ms, 4),
                tflops_with_rulebook=round(tot_f / ((tot_ms + rb_ms) * 1e-3) / 1e12, 2),
                frac_with_rulebook=round(tot_f / ((tot_ms + rb_ms) * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                hbm_gbs=round(tot_b / (tot_ms * 1e-3) / 1e9, 1),
                hbm_frac=round(tot_b / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                flop_per_byte=round(tot_f / max(tot_b, 1.0), 1), layers=layers))
    if not entries:
        return None, []
    order = sorted(entries, key=lambda k: -entries[k][0])
    return entries[order[0]][1], [entries[k][1] for k in order[1:]]


def preflight(dist, rank, world, dev, backend, limit_s=float(os.environ.get("HEAL_PREFLIGHT_S", "120"))):
    """First contact of the N > 1 path with the hardware: one all-reduce, one gather and one all-to-all of 1 KB each,
    bounded by a timer that ends the job with a clear message instead of a silent watchdog exit; every rank announces itself on stderr
    and rank 0 keeps who-runs-where for the JSON line, so the driver's record shows N ranks on N devices.  The limit (120 s, HEAL_PREFLIGHT_S)
    is generous on purpose: the ranks leave init_process_group together, but the first gather / all-to-all of a process group still set up
    their peer-to-peer channels lazily -- the timer is there to turn a dead job into a message, not to police a slow first contact."""
    import threading

    def _expire():
        print(f"[bench] rank {rank}: PREFLIGHT FAILED -- the {backend} process group of {world} ranks did not complete a 1-KB all_reduce / "
              f"gather / all_to_all_single within {limit_s:.0f} s (device {dev}; check HSA_ENABLE_IPC_MODE_LEGACY=0, one rank per GPU, "
              "MASTER_ADDR=127.0.0.1)", file=sys.stderr, flush=True)
        os._exit(4)
    timer = threading.Timer(limit_s, _expire)
    timer.daemon = True
    timer.start()
    t0 = time.perf_counter()
    x = torch.full((256,), float(rank + 1), device=dev)
    dist.all_reduce(x)
    assert float(x[0].item()) == world * (world + 1) / 2, "preflight: all_reduce returned a wrong sum"
    parts = [torch.empty(256, device=dev) for _ in range(world)] if rank == 0 else None
    dist.gather(torch.full((256,), float(rank), device=dev), parts, dst=0)
    if rank == 0:
        assert [float(p_[0].item()) for p_ in parts] == [float(r) for r in range(world)], "preflight: gather returned wrong rows"
    a2a_in = torch.arange(world * 64, device=dev, dtype=torch.float32) + 1000.0 * rank
    a2a_out = torch.empty_like(a2a_in)
    try:
        dist.all_to_all_single(a2a_out, a2a_in)
        a2a = bool(float(a2a_out[0].item()) == 64.0 * rank)
    except Exception as e:  # noqa: BLE001 - gloo has no all_to_all on device tensors: the striped tail then uses gathers
        a2a = f"unavailable on this backend ({type(e).__name__})"
    torch.cuda.synchronize()
    timer.cancel()
    me = {"rank": rank, "device_index": dev.index, "device": torch.cuda.get_device_name(dev), "pid": os.getpid()}
    print(f"[bench] rank {rank}/{world} on cuda:{dev.index} ({me['device']}), backend {backend}, preflight "
          f"{(time.perf_counter() - t0) * 1e3:.0f} ms", file=sys.stderr, flush=True)
    who = [None] * world
    dist.all_gather_object(who, me)
    out = {"ranks": who, "preflight": {"all_reduce": True, "gather": True, "all_to_all_single": a2a}}
    if backend == "nccl":
        try:
            out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            out["rccl_version"] = None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="scene5", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames", type=int, default=4,
                    help="distinct synthetic frames (same agent layout, different clouds / images / poses) resident in HBM; "
                         "the steps cycle through them, so every step sees a NEW frame like tools/inference.py's loop")
    ap.add_argument("--frames-in-flight", type=int, default=int(os.environ.get("HEAL_FRAMES_IN_FLIGHT", "2")),
                    help="captured copies of the step that run concurrently on their own streams (pipeline.FramesInFlight): frame "
                         "k + 1 is submitted while frame k runs and its boxes are read one step later; 1 = one frame at a time")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from the host instead of replaying a "
                    "captured HIP graph of the step (the heterogeneous scene is ~800 launches: host-bound when eager)")
    ap.add_argument("--parallel", default="agents", choices=["agents", "replicas"],
                    help="N>1: 'agents' (default, BASELINE north_star) shards the agents of ONE scene over the ranks with one "
                         "all-gather (strong scaling); 'replicas' runs one independent scene per rank, no collective "
                         "(weak scaling; the throughput upper bound of SURVEY 8e)")
    ap.add_argument("--watchdog-s", type=int, default=1200,
                    help="hard wall-clock limit of this process: a hung collective ends the job instead of the box")
    a = ap.parse_args()

    if a.watchdog_s > 0:
        import threading

        def _expire():
            print(f"[bench] watchdog: no result after {a.watchdog_s} s, exiting", file=sys.stderr, flush=True)
            os._exit(3)
        wd = threading.Timer(a.watchdog_s, _expire)
        wd.daemon = True
        wd.start()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and world == 1 and a.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # HEAL_DIST_BACKEND=gloo lets the N>1 code path be exercised on a 1-GPU box (ranks share cuda:0);
    # the real launch is one rank per GPU over RCCL ("nccl")
    backend = os.environ.get("HEAL_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    job = {"world_size": world, "backend": "none" if world == 1 else ("nccl (RCCL)" if backend == "nccl" else backend)}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        job.update(preflight(dist, rank, world, dev, backend))

    if os.environ.get("HEAL_MIOPEN_BENCHMARK", "0") == "1":
        torch.backends.cudnn.benchmark = True  # MIOpen find mode: time the applicable solvers once per shape
    from heal_amd import configs, ops
    from heal_amd.dist import make_sharded, owned_agents
    from heal_amd.pipeline import Scene, ScenePipeline, StaticInputs

    # everything runs on one non-default stream, so that an optional HIP-graph capture of the step reuses the
    # stream (and MIOpen state) of the eager warm-up
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    mods, desc = WORKLOADS[a.workload]
    n_agents = len(mods)
    lidar_only = all(m == "m1" for m in mods)
    baseline_model = a.workload == "scene8_second_v2xvit"
    replicas = world > 1 and a.parallel == "replicas"
    solo = world == 1 or replicas      # this rank runs whole scenes by itself
    if baseline_model:
        hypes = configs.lidar_baseline("v2xvit", max_cav=n_agents, modality="m3")
    elif lidar_only:
        hypes = configs.lidar_pyramid(WORKLOAD_RANGE.get(a.workload, configs.FULL_RANGE), max_cav=max(5, n_agents))
    else:
        hypes = configs.heal_heter(tuple(sorted(set(mods))), max_cav=max(5, n_agents))
    pipe = ScenePipeline(hypes, dev, seed=0)
    seed0 = 4 + (1000 * rank if replicas else 0)
    frames = [Scene(n_agents, seed=seed0 + 17 * i, device=dev, modalities=mods) for i in range(max(1, a.frames))]
    scene = frames[0]
    cls_shift = pipe.calibrate_cls_bias(scene)
    batch = {"ego": {"transformation_matrix": pipe.tfm, "anchor_box": pipe.anchor_box}}
    tick = [0]

    def next_frame():
        f = frames[tick[0] % len(frames)]
        tick[0] += 1
        return f

    use_graph = False
    ring = None          # pipeline.FramesInFlight when more than one frame is in flight
    if solo:
        def step():
            return pipe.step(next_frame())
        if not a.eager:
            try:
                pipe.capture(scene)
                use_graph = True

                def step():  # noqa: F811
                    return pipe.replay(next_frame())   # copies the frame into the graph's static input buffers first
                if a.frames_in_flight > 1:
                    from heal_amd.pipeline import FramesInFlight
                    ring = FramesInFlight(pipe, scene, depth=a.frames_in_flight)

                    def step():  # noqa: F811
                        return ring.step(next_frame())     # the boxes of the frame submitted `depth` steps earlier
            except Exception as e:  # a path with a host round trip (e.g. SECOND's site counts) cannot be captured
                print(f"[bench] HIP graph capture unavailable for this workload ({type(e).__name__}: {e}); "
                      "running eagerly", file=sys.stderr)
                torch.cuda.synchronize()
    else:
        wire = torch.float16 if os.environ.get("HEAL_WIRE", "fp32") == "fp16" else None  # opt-in half-size exchange
        sharded = make_sharded(pipe.model, rank, world, wire_dtype=wire)
        mine = owned_agents(n_agents, rank, world)
        # this rank's sensor inputs + the scene's pose matrices in fixed device buffers; every step loads the next frame
        static = StaticInputs(scene, agents=mine)
        local_inputs = static.inputs_for(mine)
        inp = static.scene_meta()

        def step():
            static.load(next_frame())
            out = sharded.forward(inp, n_agents, local_inputs)
            if rank == 0:
                return pipe.post.post_process(batch, {"ego": out})
            return None, None
        eager_step = step

        # Self-proof of the sharded step BEFORE anything is timed: frame 0 through the agent-sharded forward of
        # the whole job, and the same frame as ONE process on rank 0's GPU (every rank holds the whole model and the synthetic scene);
        # the boxes must agree -- same survivors, corners / scores to 1e-5 -- or the job ends here, on every rank.
        static.load(frames[0])
        out0 = sharded.forward(inp, n_agents, local_inputs)
        verdict = torch.ones(1, device=dev)
        if rank == 0:
            b_sh, s_sh = pipe.post.post_process(batch, {"ego": out0})
            b_1, s_1 = pipe.step(frames[0])
            same = (b_sh is None) == (b_1 is None)
            dev_box = dev_score = 0.0
            if same and b_1 is not None:
                # same survivors: equal count, every sharded box has its twin (nearest centre) within 5 mm with a score within 1e-3 --
                # the tolerance of tests/test_gpu_dist.py: the two paths sum the agents' contributions in different orders (warp per agent
                # + fuse vs one fused kernel; camera lift atomics), so the head maps agree to ~1e-5 of their scale, not bit for bit
                # (a candidate within rounding of the score threshold or of the 0.15 IoU may appear on one side only: up to 3 boxes without a
                #  twin are tolerated, as in tests/test_gpu_dist.py; every other box must have its twin)
                c_sh, c_1 = b_sh.mean(1), b_1.mean(1)
                twin = torch.cdist(c_sh, c_1, compute_mode="donot_use_mm_for_euclid_dist").argmin(1)
                # (the distance to the twin from the coordinates themselves: cdist's matrix-product form loses ~1e-2 m at |x| ~ 100 m)
                close = ((c_sh - c_1[twin]).norm(dim=1) < 5e-3) & ((s_sh - s_1[twin]).abs() < 1e-3)
                lone = int((~close).sum()) + max(0, int(b_1.shape[0]) - int(close.sum()))
                if bool(close.any()):
                    dev_box = float((b_sh[close] - b_1[twin[close]]).abs().max())
                    dev_score = float((s_sh[close] - s_1[twin[close]]).abs().max())
                same = bool(abs(int(b_sh.shape[0]) - int(b_1.shape[0])) <= 3 and lone <= 3
                            and len(set(twin[close].tolist())) == int(close.sum()))
            job["sharded_equals_single"] = bool(same)
            job["sharded_check"] = {"frame": 0, "boxes_sharded": 0 if b_sh is None else int(b_sh.shape[0]),
                                    "boxes_single_process": 0 if b_1 is None else int(b_1.shape[0]),
                                    "max_corner_deviation_m": dev_box, "max_score_deviation": dev_score,
                                    "tolerance": "twins within 5e-3 m / 1e-3 score; at most 3 boxes without a twin (tests/test_gpu_dist.py)",
                                    "bit_equal": bool(same and (b_1 is None or (torch.equal(b_sh, b_1) and torch.equal(s_sh, s_1))))}
            verdict.fill_(1.0 if same else 0.0)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
        if float(verdict.item()) != 1.0:
            raise SystemExit(f"[bench] rank {rank}: the agent-sharded step does NOT reproduce the single-process step on frame 0 "
                             f"({job.get('sharded_check')}): refusing to time it")

        if not a.eager:
            # graph(local stage) -> RCCL all-gather -> graph(fusion tail + decode/NMS on rank 0)
            dir_args = pipe.post.params.get("dir_args", {"dir_offset": 0.7853, "num_bins": 2})
            anchors_f32 = pipe.post._anchors_f32(pipe.anchor_box, dev)

            def post_fn(out):
                return ops.decode_nms(out["cls_preds"], out["reg_preds"], out.get("dir_preds"), anchors_f32,
                                      pipe.post.params["target_args"]["score_threshold"], dir_args["dir_offset"],
                                      dir_args["num_bins"], pipe.post.params["nms_thresh"],
                                      np.eye(4, dtype=np.float32), pipe.post.params["gt_range"], sync=False)
            if sharded.capture(inp, n_agents, local_inputs, post_fn):
                sharded.check_sparse_capacity()
                use_graph = True
            elif sharded._capture_error is not None:
                e = sharded._capture_error
                print(f"[bench] rank {rank}: HIP graph capture unavailable ({type(e).__name__}: {e}); running eagerly",
                      file=sys.stderr)
            if use_graph:

                def step():  # noqa: F811
                    static.load(next_frame())
                    res_ = sharded.replay()
                    if rank != 0:
                        return None, None
                    corners, scores, count = res_
                    k = int(count.item())
                    return (None, None) if k == 0 else (corners[:k], scores[:k])
                serial_step = step
                if a.frames_in_flight > 1:
                    # throughput mode of the sharded step: `depth` captured copies, local stage of frame k + 1 under the fusion
                    # tail of frame k on rank 0 (dist.ShardedFramesInFlight); exchanges stay in frame order
                    from heal_amd.dist import AgreedCaptureFailure, ShardedFramesInFlight
                    try:
                        ring = ShardedFramesInFlight(lambda: make_sharded(pipe.model, rank, world, wire_dtype=wire), scene, n_agents,
                                                     rank, world, depth=a.frames_in_flight, post_fn=post_fn)
                    except AgreedCaptureFailure as e:
                        # a slot that cannot be captured is AGREED between the ranks (dist._Sharded._agree), so every rank lands
                        # here together and the job goes on with the serial sharded replay instead of dying.
                        # Any OTHER exception (e.g. the striped runner's one-rank failure) is not agreed: it ends the job
                        print(f"[bench] rank {rank}: frames in flight unavailable ({e}); serial sharded replay", file=sys.stderr)
                        ring = None
                    if ring is not None:

                        def step():  # noqa: F811
                            r_ = ring.step(next_frame())
                            return r_ if r_ is not None else (None, None)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Two rates with the SAME W and K: `serial` = one frame at a time (replay, read the boxes, then submit the next
    # frame: the rate earlier rounds and the reference's loop quote, and the latency of one frame), and the headline `value` =
    # `frames_in_flight` captured copies of the step overlapping on their own streams (throughput mode).
    latency_ms, serial = None, None
    if ring is not None:
        one = (lambda: pipe.replay(next_frame())) if solo else serial_step
        for _ in range(a.warmup):
            one()
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            one()
        fence()
        sdt = time.perf_counter() - t0
        if world > 1:
            t_ = torch.tensor([sdt], device=dev, dtype=torch.float64)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            sdt = float(t_.item())
        latency_ms = sdt / a.steps * 1e3
        serial = {"value": round(a.steps / sdt, 3), "unit": "scenes/s", "ms_per_step": round(latency_ms, 3), "steps": a.steps,
                  "warmup": a.warmup,
                  "what": ("one frame at a time: " + ("hipGraph replay of the whole step" if solo else
                                                     "hipGraph replays of the rank-local stages around the exchange(s)")
                           + ", boxes read back before the next frame is submitted (frames_in_flight = 1)")}
    for _ in range(a.warmup):
        step()
    if ring is not None:
        ring.drain()
    fence()
    if not use_graph:
        ops.TIMING = {}
        ops.SP_TRACE = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = step()
    if ring is not None:          # every one of the K frames is finished and read back inside the timed region
        tail_ = [r_ for r_ in ring.drain() if r_ is not None]
        res = tail_[-1] if tail_ else res
    if res is None:
        res = (None, None)
    fence()
    dt = time.perf_counter() - t0
    if use_graph:
        # per-operator HIP-event timing needs host-side launches: an instrumented eager pass over the
        # same K steps, right after the timed graph replays (events cannot be recorded inside a graph)
        ops.TIMING = {}
        # one stream for this pass: with the modality stems on concurrent streams an operator's event pair would also time
        # the other streams' kernels it shares the chip with (per-operator durations, not the step time, are read from here)
        par_env = os.environ.get("HEAL_PARALLEL_MODALITIES")
        os.environ["HEAL_PARALLEL_MODALITIES"] = "0"
        ops.LAST_CALLS = {}
        for i_ in range(a.steps):
            ops.SP_TRACE = [] if i_ == a.steps - 1 else None   # sparse-layer anatomy of the last instrumented step
            if solo:
                pipe.step(next_frame())
            else:
                eager_step()
        torch.cuda.synchronize()
        if par_env is None:
            os.environ.pop("HEAL_PARALLEL_MODALITIES")
        else:
            os.environ["HEAL_PARALLEL_MODALITIES"] = par_env
    timing = ops.timing_summary()
    work = ops.work_summary()
    sp_trace = ops.SP_TRACE
    ops.TIMING = None
    ops.SP_TRACE = None
    # multi-launch operators (K1, K8): the per-call period of their launch chain inside a captured graph, on the tensors of the last
    # instrumented step (ops.LAST_CALLS) -- what the timed region replays; the eager event pair above mostly times the host
    in_graph = {}
    if rank == 0 and use_graph and ops.LAST_CALLS:
        with torch.no_grad():
            for name_, fn_ in list(ops.LAST_CALLS.items()):
                try:
                    in_graph[name_] = ops.graph_period_ms(fn_)
                except Exception as e:  # noqa: BLE001 - a report, never a reason to lose the bench line
                    print(f"[bench] in-graph timing of {name_} failed: {type(e).__name__}: {e}", file=sys.stderr)
    ops.LAST_CALLS = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        coll_name, striped = "n/a", False
        if not solo:   # name the collective that actually ran (dist._Sharded.collective; VERDICT r3: the label said all-gather)
            lib_ = "RCCL" if backend == "nccl" else backend
            coll_name = {"gather": f"{lib_} gather to rank 0", "all_gather": f"{lib_} all-gather",
                         "p2p": f"peer window in rank 0's HBM (IPC-mapped; rows written by their producers, two 1-element {lib_} "
                                "all-reduces as fences)"}.get(sharded.collective, sharded.collective)
            striped = bool(getattr(sharded, "_striped", False))
            if striped:   # dist.ShardedBaselineStriped: the V2X-ViT tail runs on row stripes of every rank
                coll_name = (f"{lib_} all-to-all of ego-frame stripes + all-gathered split-attention sums + gather of the ego "
                             f"stripe")
        ms_per_step = dt / a.steps * 1e3
        rng_ = hypes["model"]["args"]["lidar_range"]
        nx, ny = int(round((rng_[3] - rng_[0]) / 0.4)), int(round((rng_[4] - rng_[1]) / 0.4))     # 512 x 512, native range: 480 x 240
        with torch.no_grad():
            m_per_agent = []
            for k in sorted(scene.points):
                vs_, pp_ = ([0.1, 0.1, 0.1], 5) if baseline_model else ([0.4, 0.4, 4], 32)
                _, _, nn_ = ops.voxelize(scene.points[k], hypes["model"]["args"]["lidar_range"], vs_, pp_, 70000)
                m_per_agent.append(int(nn_.shape[0]))
        kernels = {k: {"calls": c, "mean_ms": round(ms, 5)} for k, (c, ms) in sorted(timing.items())}
        roof, roof_other = None, []
        try:
            roof, roof_other = roofline_report(a, work, timing, scene, mods, m_per_agent, hypes, solo, world, n_agents, sp_trace,
                                               ny, nx, in_graph=in_graph)
        except Exception as e:  # noqa: BLE001 - the report must never break the bench line
            print(f"[bench] roofline report failed: {type(e).__name__}: {e}", file=sys.stderr)
        line = {
            "metric": "scenes/sec (5-agent OPV2V-H, PointPillars+PyramidFusion)",
            "value": round((world if replicas else 1) * a.steps / dt, 3), "unit": "scenes/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak" if replicas else "strong", "vs_baseline": None,
            "dtype": ("f32" if not os.environ.get("HEAL_ARITH") else
                      f"f32 via {os.environ['HEAL_ARITH']} split (OPT-IN: pointwise convolutions with cin % 32 == 0, cout % 128 == 0 on the "
                      "bf16 matrix cores, fp32 in / out / accumulate; everything else exact-fp32 MFMA)"), "data": "synthetic",
            "config": {"workload": f"{a.workload}: {desc}", "agents": n_agents,
                       "pillars_per_agent": m_per_agent, "modalities": mods,
                       "points_per_agent": [int(scene.points[k].shape[0]) for k in sorted(scene.points)],
                       "parallelism": "1 GPU" if world == 1 else f"{world} independent scene replicas, no collective" if replicas
                       else (f"agent-sharded over {world} ranks, 1 {coll_name}"
                                                                   + (" (fp16 wire)" if os.environ.get("HEAL_WIRE") == "fp16" else "")),
                       "launch": ("eager launches" if not use_graph else "hipGraph replay of the whole step" if solo
                                  else "program of hipGraphs cut at the collectives: local | all-to-all | encoder stripe (cut at each "
                                       "split attention) | gather | heads + decode/NMS" if striped
                                  else f"hipGraph(local stage) -> {coll_name} -> hipGraph(fusion tail + decode/NMS)"),
                       "frames_in_flight": (ring.depth if ring is not None else 1),
                       "frame_latency_ms": (round(latency_ms, 3) if latency_ms is not None else None),
                       "boxes_out": 0 if res[0] is None else int(res[0].shape[0]),
                       "inputs": "sensor frames resident in HBM before the timed region (4 distinct frames cycled); H2D not timed",
                       "job": job, "collective": coll_name,
                       "sharded_equals_single": job.get("sharded_equals_single")},
            "roofline": roof, "roofline_other": roof_other, "op_timing_ms": kernels,
        }
        if serial is not None:
            line["serial"] = serial
        if not a.no_cpu_baseline and world == 1:
            scene_cpu = Scene(n_agents, seed=seed0, device="cpu", modalities=mods)   # the same synthetic frame, host copy
            line["cpu_baseline"] = cpu_baseline(hypes, scene_cpu, cls_shift)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
