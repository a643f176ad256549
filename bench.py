#!/usr/bin/env python
"""bench.py -- scenes/s of HEAL's per-frame perception hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload scene5|pair|single]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one synthetic OPV2V-shaped scene through the whole path with its inputs (device point
clouds, anchors) already resident in HBM: voxelise (K1) -> PFN+scatter (K2) -> BEV backbones ->
pyramid stages -> warp + occupancy-softmax fusion (K5) -> deblocks / shrink / heads -> decode +
rotated NMS (K8) -> host sees the boxes.  N = 1: everything on one GPU.  N > 1: the agents of the
scene are sharded one per rank with a single all-gather of ego-frame maps (heal_amd/dist.py);
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
    "scene8_second_v2xvit": (["m3"] * 8, "8-agent synthetic scene, SECOND (sparse conv) encoders + plain BEV backbone + "
                                         "V2X-ViT fusion, heter_model_baseline (BASELINE config 5)"),
}


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


def cpu_baseline(hypes, scene_points, pairwise, n_agents, cls_shift=0.0):
    """The oracle (CPU port of the reference algorithm) timed on this box's host cores, on a bounded
    sample: ONE scene of the same workload.  Reported, never the thing measured above."""
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
    t0 = time.perf_counter()
    vs, cs, ns = [], [], []
    for b, p in enumerate(scene_points):
        v, c, n = cref.voxelize(p, r, [0.4, 0.4, 4], 32, 70000, batch_idx=b)
        vs.append(v); cs.append(c); ns.append(n)
    out = model_ref.heter_pyramid_collab_m1(sd, args, np.concatenate(vs), np.concatenate(cs), np.concatenate(ns),
                                            n_agents, pairwise)
    anchors = O.generate_anchor_box(r, 0.4, 0.4, int(round((r[3] - r[0]) / 0.4)), int(round((r[4] - r[1]) / 0.4)),
                                    3.9, 1.6, 1.56, [0, 90])
    out["cls_preds"] = out["cls_preds"] + cls_shift  # same calibrated head bias as the GPU pipeline
    O.post_process(out["cls_preds"], out["reg_preds"], out["dir_preds"], anchors, 0.2, 0.7853, 2, 0.15,
                   np.eye(4, dtype=np.float32), r)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "scenes/s", "cores": cores, "kind": "port",
            "sample": f"1 scene of the same workload ({n_agents} agents) through oracle/ "
                      f"(C voxeliser + numpy PFN/warp/fuse + torch-CPU fp32 convs + C rotated NMS), {dt:.1f} s"}


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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

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
        hypes = configs.lidar_pyramid(max_cav=max(5, n_agents))
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
    if solo:
        def step():
            return pipe.step(next_frame())
        if not a.eager:
            try:
                pipe.capture(scene)
                use_graph = True

                def step():  # noqa: F811
                    return pipe.replay(next_frame())   # copies the frame into the graph's static input buffers first
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

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    if not use_graph:
        ops.TIMING = {}
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = step()
    fence()
    dt = time.perf_counter() - t0
    if use_graph:
        # per-operator HIP-event timing needs host-side launches: an instrumented eager pass over the
        # same K steps, right after the timed graph replays (events cannot be recorded inside a graph)
        ops.TIMING = {}
        for _ in range(a.steps):
            if solo:
                pipe.step(next_frame())
            else:
                eager_step()
        torch.cuda.synchronize()
    timing = ops.timing_summary()
    ops.TIMING = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        nx = ny = 512
        # K2 roofline: algorithmic bytes of the operator / its measured duration (HIP events)
        with torch.no_grad():
            m_per_agent = []
            for k in sorted(scene.points):
                vs_, pp_ = ([0.1, 0.1, 0.1], 5) if baseline_model else ([0.4, 0.4, 4], 32)
                _, _, nn_ = ops.voxelize(scene.points[k], hypes["model"]["args"]["lidar_range"], vs_, pp_, 70000)
                m_per_agent.append(int(nn_.shape[0]))
        roof = None
        if "pfn_scatter" in timing:
            calls, mean_ms = timing["pfn_scatter"]
            # one launch of the operator = the collated LiDAR agents this rank encodes (reference: one PillarVFE +
            # PointPillarScatter call per modality batch); algorithmic bytes = sum over those agents
            lidar_ids = [i for i, m in enumerate(mods) if m == "m1"]
            if not solo:
                lidar_ids = [i for i in lidar_ids if i in owned_agents(n_agents, 0, world)]
            order = sorted(scene.points)
            m_launch = [m_per_agent[order.index(i)] for i in lidar_ids if i in order]
            bytes_per_launch = float(sum(k2_algorithmic_bytes(32, m, ny, nx) for m in m_launch))
            achieved = bytes_per_launch / (mean_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_k2_traffic.json")
            if os.path.exists(tpath):  # HBM bytes per launch from the rocprofv3 PMC passes (scripts/profile_round.sh)
                tj = json.load(open(tpath))
                if int(tj.get("agents_per_launch", -1)) == len(m_launch):
                    traffic = float(tj["k2_traffic_bytes_per_launch"])
            roof = {"kernel": f"K2 heal_pfn_scatter, {len(m_launch)} collated LiDAR agents per launch "
                              "(map memset + k_pfn + k_canvas)", "bound": "hbm",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "bytes_per_launch": bytes_per_launch, "launch_ms": round(mean_ms, 5), "launches": calls}
        kernels = {k: {"calls": c, "mean_ms": round(ms, 5)} for k, (c, ms) in sorted(timing.items())}
        # secondary roofline entries (same definition as `roofline`: algorithmic bytes / HIP-event duration) for the other
        # north-star kernels present in this workload; informational, never allowed to break the line
        roof_other = []
        try:
            if "bev_pool" in timing:
                cam_ids = [i for i, m in enumerate(mods) if m in Scene.CAMERA_DIMS and (solo or i in owned_agents(n_agents, 0, world))]
                if cam_ids:
                    per_call = []
                    for i in cam_ids:   # one call per camera agent: 4 cameras, D=48 bins, C=128, 256x256 cells
                        H, W = Scene.CAMERA_DIMS[mods[i]]
                        fhw = (H // 8) * (W // 8)
                        per_call.append(4.0 * (4 * 48 * fhw + 4 * 128 * fhw + 128 * 256 * 256))
                    calls, mean_ms = timing["bev_pool"]
                    b = sum(per_call) / len(per_call)
                    roof_other.append({"kernel": "K4 heal_bev_pool, one camera agent per launch (mean over the scene's camera agents)",
                                       "bound": "hbm", "achieved": round(b / (mean_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "frac": round(b / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "traffic": None, "bytes_per_launch": b, "launch_ms": round(mean_ms, 5), "launches": calls})
        except Exception as e:  # noqa: BLE001
            print(f"[bench] secondary roofline skipped: {type(e).__name__}: {e}", file=sys.stderr)
        line = {
            "metric": "scenes/sec (5-agent OPV2V-H, PointPillars+PyramidFusion)",
            "value": round((world if replicas else 1) * a.steps / dt, 3), "unit": "scenes/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak" if replicas else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.workload}: {desc}", "agents": n_agents,
                       "pillars_per_agent": m_per_agent, "modalities": mods,
                       "points_per_agent": [int(scene.points[k].shape[0]) for k in sorted(scene.points)],
                       "parallelism": "1 GPU" if world == 1 else f"{world} independent scene replicas, no collective" if replicas
                       else (f"agent-sharded over {world} ranks, 1 all-gather"
                                                                   + (" (fp16 wire)" if os.environ.get("HEAL_WIRE") == "fp16" else "")),
                       "launch": ("eager launches" if not use_graph else "hipGraph replay of the whole step" if solo
                                  else "hipGraph(local stage) -> all-gather -> hipGraph(fusion tail + decode/NMS)"),
                       "boxes_out": 0 if res[0] is None else int(res[0].shape[0])},
            "roofline": roof, "roofline_other": roof_other, "op_timing_ms": kernels,
        }
        if not a.no_cpu_baseline and world == 1 and not baseline_model:
            # the CPU port covers the LiDAR (PointPillars) agents; for the heterogeneous workload the
            # sample is the same scene with every agent treated as a LiDAR agent (stated in `sample`)
            lidar_hypes = hypes if lidar_only else configs.lidar_pyramid(max_cav=max(5, n_agents))
            lidar_scene = scene if lidar_only else Scene(n_agents, seed=4, device="cpu")
            line["cpu_baseline"] = cpu_baseline(lidar_hypes,
                                                [lidar_scene.points[k].cpu().numpy() for k in sorted(lidar_scene.points)],
                                                scene.pairwise, n_agents, cls_shift)
            if not lidar_only:
                line["cpu_baseline"]["sample"] += " [all 5 agents as PointPillars LiDAR agents: the CPU port has no camera trunk]"
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
