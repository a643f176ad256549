"""Predicted 1 / 2 / 4 / 8-GPU curve of the agent-sharded step from times MEASURED on one MI355X.

For every world size N the ownership of heal_amd.dist is applied, every rank's `local` stage (encode its agents ... warp into
the ego frame ... pack) and rank 0's `tail` (fuse, deblocks / transformer, heads, decode + NMS) are captured as HIP graphs
exactly as `_Sharded.capture` does and replayed ALONE on this GPU (HIP events), and the one exchange is priced from its bytes:

    exchange(N) = bytes of the fullest sender (slots_per_rank x shard) / link rate      (gather: one xGMI link per sender,
                  senders in parallel into rank 0; 153 GB/s per link x 0.75 sustained)
    period(N)   = max( max_r local_r + exchange,  local_0 + exchange + tail )           (rank 0 runs the tail; the other
                  ranks' next local stage overlaps it, the next exchange waits for rank 0)

Config 5 (HeterModelBaseline + V2X-ViT), N > 1: the tail is STRIPED over the ranks (dist.ShardedBaselineStriped).  Its stages are
timed the same way -- the encoder on one [L, H / N, W, C] stripe with the all-gather replaced by a local copy, the heads + decode
on rank 0 -- and its exchanges priced from their bytes: the all-to-all sends (N - 1) / N of a rank's maps over N - 1 links in
parallel, the three all-gathers of column sums are latency (30 us each assumed), the gather brings 1 / N of one map per link:

    period(N) = max_r local_r + all_to_all + encoder_stripe + 3 x all_gather + gather + heads       (`striped_*` fields)

This is a MODEL, not a measurement: no multi-GPU hardware was available to the builder (SCALE_r01/r02: skipped).  It is what
the first hardware run is to be checked against.

    python scripts/scaling_model.py [--workload scene5|scene8_second_v2xvit] [--json out.json]
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

LINK_GBS = 153.0 * 0.75


def timed_graph(fn, stream, iters=10):
    for _ in range(2):
        out = fn()
    stream.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
        out = fn()
    for _ in range(2):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        g.replay()
    e1.record(stream)
    stream.synchronize()
    return out, e0.elapsed_time(e1) / iters, g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="scene5")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    from bench import WORKLOADS
    from heal_amd import configs, ops
    from heal_amd.dist import make_sharded, owned_agents, slots_per_rank
    from heal_amd.pipeline import Scene, ScenePipeline, StaticInputs
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    mods, desc = WORKLOADS[a.workload]
    n_agents = len(mods)
    baseline = a.workload == "scene8_second_v2xvit"
    if baseline:
        hypes = configs.lidar_baseline("v2xvit", max_cav=n_agents, modality="m3")
    elif all(m == "m1" for m in mods):
        hypes = configs.lidar_pyramid(max_cav=max(5, n_agents))
    else:
        hypes = configs.heal_heter(tuple(sorted(set(mods))), max_cav=max(5, n_agents))
    pipe = ScenePipeline(hypes, dev, seed=0)
    scene = Scene(n_agents, seed=4, device=dev, modalities=mods)
    pipe.calibrate_cls_bias(scene)
    dir_args = pipe.post.params.get("dir_args", {"dir_offset": 0.7853, "num_bins": 2})
    anchors_f32 = pipe.post._anchors_f32(pipe.anchor_box, dev)

    def post_fn(out):
        return ops.decode_nms(out["cls_preds"], out["reg_preds"], out.get("dir_preds"), anchors_f32,
                              pipe.post.params["target_args"]["score_threshold"], dir_args["dir_offset"], dir_args["num_bins"],
                              pipe.post.params["nms_thresh"], np.eye(4, dtype=np.float32), pipe.post.params["gt_range"], sync=False)

    rows = []
    keep = []   # graphs and their outputs stay alive
    with torch.no_grad():
        for N in (1, 2, 4, 8):
            runners = [make_sharded(pipe.model, r, N, collective="gather") for r in range(N)]
            local_ms, bufs, shape = [], [None] * N, None
            for r in range(N):
                mine = owned_agents(n_agents, r, N)
                if not mine:
                    local_ms.append(0.0)
                    continue
                static = StaticInputs(scene, agents=mine)
                static.load(scene)
                li, inp = static.inputs_for(mine), static.scene_meta()
                buf, ms, g = timed_graph(lambda: runners[r].local(inp, n_agents, li), stream)
                keep.append((g, buf, static))
                local_ms.append(ms)
                bufs[r] = buf
                shape = getattr(runners[r], "_shape", None) or shape
            # what dist.prepare()'s all-reduce agrees between the ranks of a real job: the level shapes a rank WITH agents produced
            shapes = next((getattr(rr, "_shapes", None) for rr in runners if getattr(rr, "_shapes", None)), None)
            for r in range(N):   # ranks without agents contribute a zero slot
                if bufs[r] is None:
                    bufs[r] = torch.zeros_like(next(b for b in bufs if b is not None))
                if shape is not None:
                    runners[r]._shape = shape
                if shapes is not None and hasattr(runners[r], "_shapes"):
                    runners[r]._shapes = shapes
            gathered = torch.stack(bufs)
            _, tail_ms, g = timed_graph(lambda: post_fn(runners[0].tail(gathered, n_agents)), stream)
            keep.append((g, gathered))
            shard_bytes = bufs[0].shape[1] * bufs[0].element_size()
            senders = [r for r in range(1, N)]
            exch_ms = (slots_per_rank(n_agents, N) * shard_bytes / (LINK_GBS * 1e9) * 1e3) if senders else 0.0
            period = max(max(local_ms) + exch_ms, local_ms[0] + exch_ms + tail_ms)
            striped = None
            if baseline and N > 1 and getattr(runners[0], "_encoder", None) is not None and shape[1] % (N * 16) == 0:
                from heal_amd.dist import unpack_maps
                C_, H_, W_ = shape
                Hs = H_ // N
                ego = unpack_maps(gathered, shape, n_agents, N)                               # [L, C, H, W]
                xs = ego[:, :, :Hs].permute(0, 2, 3, 1).contiguous()                          # stripe 0, token-major

                class _Local:   # timing stand-in: every rank's column sums are this stripe's
                    world = N

                    def all_gather(self, t):
                        return t.unsqueeze(0).repeat(N, *([1] * t.dim()))
                enc = runners[0]._encoder()
                mods_ = runners[0]._stripe_modules()
                for m_ in mods_:
                    m_._stripe = _Local()
                y, enc_ms, g = timed_graph(lambda: enc(xs)[0].contiguous(), stream)
                for m_ in mods_:
                    m_._stripe = None
                keep.append((g, xs, y))
                fused = y.repeat(N, 1, 1).permute(2, 0, 1).unsqueeze(0).contiguous()
                _, heads_ms, g = timed_graph(lambda: post_fn(dict(zip(("cls_preds", "reg_preds", "dir_preds"),
                                                                     pipe.model.heads(fused)))), stream)
                keep.append((g, fused))
                map_bytes = C_ * H_ * W_ * 4
                a2a_ms = slots_per_rank(n_agents, N) * map_bytes / N / (LINK_GBS * 1e9) * 1e3     # per link: one stripe per slot
                ag_ms, gat_ms = 3 * 0.03, map_bytes / N / (LINK_GBS * 1e9) * 1e3
                sp = max(local_ms) + a2a_ms + enc_ms + ag_ms + gat_ms + heads_ms
                striped = {"encoder_stripe_ms": round(enc_ms, 3), "heads_ms": round(heads_ms, 3), "all_to_all_ms_model": round(a2a_ms, 3),
                           "all_gathers_ms_assumed": ag_ms, "gather_ms_model": round(gat_ms, 3), "period_ms_model": round(sp, 3)}
                period = sp
            # Two frames in flight per rank (dist.ShardedFramesInFlight, what `bench.py --gpus N` runs): every rank's stage(s) captured twice,
            # on two streams with their own static inputs, replayed alternately ALONE on this GPU.  A camera-only rank is a chain of
            # 10-25 us kernels that leave most of the chip idle: two frames overlap (3.1 -> 2.3 ms per frame for the m2 agent).
            inflight = None
            if not baseline and os.environ.get("HEAL_SCALING_INFLIGHT", "1") == "1":
                import time
                stream2 = torch.cuda.Stream(device=dev)
                per_rank = []
                for r in range(N):
                    mine = owned_agents(n_agents, r, N)
                    if not mine and r != 0:
                        per_rank.append(0.0)
                        continue
                    slots = []
                    for st in (stream, stream2):
                        with torch.cuda.stream(st):
                            rr = make_sharded(pipe.model, r, N, collective="gather")
                            if shape is not None:
                                rr._shape = shape
                            if shapes is not None and hasattr(rr, "_shapes"):
                                rr._shapes = shapes
                            graphs = []
                            if mine:
                                static = StaticInputs(scene, agents=mine)
                                static.load(scene)
                                li, inp = static.inputs_for(mine), static.scene_meta()
                                b_, _, g = timed_graph(lambda: rr.local(inp, n_agents, li), st, iters=2)
                                graphs.append(g)
                                keep.append((g, b_, static))
                            if r == 0:
                                o_, _, g = timed_graph(lambda: post_fn(rr.tail(gathered, n_agents)), st, iters=2)
                                graphs.append(g)
                                keep.append((g, o_))
                        slots.append((graphs, st))
                    torch.cuda.synchronize()
                    frames = 20
                    t0 = time.perf_counter()
                    for k in range(frames):
                        graphs, st = slots[k % 2]
                        with torch.cuda.stream(st):
                            for g in graphs:
                                g.replay()
                    torch.cuda.synchronize()
                    per_rank.append((time.perf_counter() - t0) * 1e3 / frames)
                torch.cuda.set_stream(stream)
                p2 = max(per_rank) + (exch_ms if N > 1 else 0.0)
                inflight = {"per_rank_ms_per_frame": [round(v, 3) for v in per_rank], "period_ms_model": round(p2, 3),
                            "scenes_per_s_model": round(1e3 / p2, 1),
                            "what": "every rank's stages (rank 0: local + tail) with two frames in flight, measured alone on one GPU; "
                                    "period = the slowest rank + one exchange (the exchanges of different frames overlap the stages)"}
            rows.append({"n_gpus": N, "owned": [owned_agents(n_agents, r, N) for r in range(N)], "two_frames_in_flight": inflight,
                         "local_ms": [round(v, 3) for v in local_ms], "tail_ms": round(tail_ms, 3),
                         "shard_MB": round(shard_bytes / 1e6, 2), "exchange_ms_model": round(exch_ms, 3),
                         "period_ms_model": round(period, 3), "scenes_per_s_model": round(1e3 / period, 1),
                         "serial_tail_period_ms_model": round(max(max(local_ms) + exch_ms, local_ms[0] + exch_ms + tail_ms), 3),
                         "striped": striped})
            print(json.dumps(rows[-1]), flush=True)
    base = rows[0]["period_ms_model"]
    for r in rows:
        r["speedup_model"] = round(base / r["period_ms_model"], 2)
        r["efficiency_model"] = round(base / r["period_ms_model"] / r["n_gpus"], 3)
    out = {"workload": a.workload, "link_GBs_assumed": LINK_GBS, "note": "model from single-GPU measurements; unmeasured on >1 GPU",
           "rows": rows}
    print(json.dumps({k: v for k, v in out.items() if k != "rows"}))
    for r in rows:
        print(r["n_gpus"], r["period_ms_model"], r["scenes_per_s_model"], r["speedup_model"], r["efficiency_model"],
              (r.get("two_frames_in_flight") or {}).get("scenes_per_s_model"))
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
