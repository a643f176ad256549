"""Per-operator timing of the hand-written kernels at the BASELINE sizes (HIP events, median of `reps`),
with the algorithmic bytes of SURVEY 8d and the resulting fraction of the 8 TB/s HBM roofline.

    python scripts/kernel_bench.py [--reps 20] [--json out.json]
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from heal_amd import configs, ops, synth
from heal_amd.opencood.models.heter_encoders import SECOND, LiftSplatShoot
from heal_amd.pipeline import fill_deterministic

HBM = 8000.0  # GB/s


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))  # microseconds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    R = configs.FULL_RANGE
    rows = []

    def add(name, us, alg_bytes, note=""):
        gbs = alg_bytes / us / 1e3 if alg_bytes else None
        rows.append({"op": name, "us": round(us, 2), "alg_MB": None if not alg_bytes else round(alg_bytes / 1e6, 2),
                     "GBps": None if gbs is None else round(gbs, 1), "frac_hbm": None if gbs is None else round(gbs / HBM, 4),
                     "note": note})
        print(rows[-1], flush=True)

    pts = torch.from_numpy(synth.lidar_frame(4000)).to(dev)
    N = pts.shape[0]
    # K1
    v, c, n = ops.voxelize(pts, R, [0.4, 0.4, 4], 32, 70000)
    M = v.shape[0]
    add("K1 voxelize (PointPillars grid)", timeit(lambda: ops.voxelize(pts, R, [0.4, 0.4, 4], 32, 70000, sync=False), a.reps),
        16 * N + 16 * M * 32 + 20 * M, f"N={N} M={M}")
    v3, c3, n3 = ops.voxelize(pts, R, [0.1, 0.1, 0.1], 5, 70000)
    M3 = v3.shape[0]
    add("K1 voxelize (SECOND grid)", timeit(lambda: ops.voxelize(pts, R, [0.1, 0.1, 0.1], 5, 70000, sync=False), a.reps),
        16 * N + 16 * M3 * 5 + 20 * M3, f"N={N} M={M3}")
    # K2
    g = torch.Generator().manual_seed(0)
    w = torch.randn((64, 10), generator=g).to(dev); sc = torch.rand((64,), generator=g).to(dev) + 0.5
    sh = torch.randn((64,), generator=g).to(dev)
    add("K2 pfn_scatter (1 agent)", timeit(lambda: ops.pfn_scatter(v, c, n, w, sc, sh, [0.4, 0.4, 4], R, 1, 512, 512), a.reps),
        16 * M * 32 + 20 * M + 4 * 64 * 512 * 512, f"M={M}")
    # K4
    for tag, (H, W) in {"m2 384x512": (384, 512), "m4 336x448": (336, 448)}.items():
        fH, fW = H // 8, W // 8
        D, C, Ncam = 48, 128, 4
        args = configs._camera_modality(R, (H, W), "EfficientNet")["encoder_args"]
        fr = LiftSplatShoot.create_frustum.__get__(type("T", (), {"data_aug_conf": args["data_aug_conf"], "downsample": 8,
                                                                  "grid_conf": args["grid_conf"]})())
        from heal_amd.opencood.utils.camera_utils import depth_discretization, gen_dx_bx
        frustum = fr(depth_discretization).to(dev)
        dx, bx, nx = gen_dx_bx(args["grid_conf"]["xbound"], args["grid_conf"]["ybound"], args["grid_conf"]["zbound"])
        rig = synth.camera_rig(0, Ncam, H, W)
        t = {k: torch.from_numpy(val[None]).to(dev) for k, val in rig.items()}
        cam = LiftSplatShoot.camera_matrices(t["rots"], t["trans"], t["intrins"], t["post_rots"], t["post_trans"])
        dl = torch.randn((Ncam, D, fH, fW), generator=g).to(dev); ft = torch.randn((Ncam, C, fH, fW), generator=g).to(dev)
        alg = 4 * (Ncam * D * fH * fW + Ncam * C * fH * fW) + 4 * C * 256 * 256
        add(f"K4 bev_pool ({tag}, 1 agent)", timeit(lambda: ops.bev_pool(dl, ft, frustum, cam, 1, Ncam, dx.tolist(), bx.tolist(),
                                                                       nx.tolist()), a.reps), alg)
    # K5
    for C, HW in ((64, 256), (128, 128), (256, 64)):
        x = torch.randn((5, C, HW, HW), generator=g).to(dev); occ = torch.randn((5, 1, HW, HW), generator=g).to(dev)
        from oracle_free_affine import rows5
        add(f"K5 warp_fuse (5 agents, C={C}, {HW}^2)", timeit(lambda: ops.warp_fuse(x, occ, rows5, True), a.reps),
            4 * HW * HW * (5 * (C + 1) + C))
    # K8
    hy = configs.lidar_pyramid()
    from heal_amd.opencood.data_utils.post_processor.voxel_postprocessor import VoxelPostprocessor
    post = VoxelPostprocessor(hy["postprocess"], False)
    anchors = torch.from_numpy(post.generate_anchor_box()).float().to(dev)
    cls = (torch.randn((1, 2, 256, 256), generator=g) * 1.5 - 4.2).to(dev); reg = (torch.randn((1, 14, 256, 256), generator=g) * 0.15).to(dev)
    dr = torch.randn((1, 4, 256, 256), generator=g).to(dev)
    k = ops.decode_nms(cls, reg, dr, anchors, 0.2, 0.7853, 2, 0.15, np.eye(4, dtype=np.float32), R)
    add("K8 decode_nms (131072 anchors)", timeit(lambda: ops.decode_nms(cls, reg, dr, anchors, 0.2, 0.7853, 2, 0.15,
                                                                        np.eye(4, dtype=np.float32), R, sync=False), a.reps),
        4 * 20 * 256 * 256, f"survivors={0 if k[0] is None else k[0].shape[0]}")
    # K6
    q, kk, vv = (torch.randn((128 * 128, 5, 256), generator=g).to(dev) for _ in range(3))
    add("K6 agent_attention (128^2 px, L=5, 8 heads)", timeit(lambda: ops.agent_attention(q, kk, vv, 8, 32 ** -0.5), a.reps),
        4 * 128 * 128 * 5 * 256 * 4)
    # K3: SECOND encoder per layer
    enc = fill_deterministic(SECOND(configs._second_modality(R)["encoder_args"]), 0).to(dev).eval()
    inp = {"inputs_m3": {"voxel_features": v3, "voxel_coords": c3, "voxel_num_points": n3, "n_agents": 1}}
    with torch.no_grad():
        t_all = timeit(lambda: enc(inp, "m3"), max(5, a.reps // 2))
        ops.TIMING = {}
        enc(inp, "m3"); torch.cuda.synchronize()
        per = ops.timing_summary(); ops.TIMING = None
    add("K3 SECOND encoder (1 agent, whole)", t_all, None, f"M={M3} " + " ".join(f"{k}:{c}x{ms*1e3:.0f}us" for k, (c, ms) in sorted(per.items())))
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
