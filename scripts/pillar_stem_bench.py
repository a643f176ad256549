"""Round 6: the LiDAR backbone's first block from the pillars (heal_pfn_pillars + heal_pillar_stem_block) against the dense path it
replaces (heal_pfn_scatter: PFN + canvas, heal_conv3x3 stride 2, heal_conv1x1 stride 2), 3 collated 64-line agents at 512 x 512.
In-graph periods (ops.graph_period_ms: how the chains run inside the captured step) and kernel-own durations."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from heal_amd import configs, ops, synth

R = configs.FULL_RANGE
n_ag = int(os.environ.get("AGENTS", "3"))
pts = [torch.from_numpy(synth.lidar_frame(4000 + k)).cuda() for k in range(n_ag)]
st = torch.cuda.Stream()
with torch.cuda.stream(st), torch.no_grad():
    v, c, n, off = ops.voxelize_collated(pts, R, [0.4, 0.4, 4], 32, 70000)
    M = int(off[-1].item())
    g = torch.Generator().manual_seed(0)
    w = torch.randn((64, 10), generator=g).cuda(); sc = (torch.rand((64,), generator=g) + 0.5).cuda(); sh = torch.randn((64,), generator=g).cuda()
    w1 = (torch.randn((64, 64, 3, 3), generator=g) / 24).cuda(); wd = (torch.randn((64, 64, 1, 1), generator=g) / 8).cuda()
    b1 = torch.randn((64,), generator=g).cuda(); bd = torch.randn((64,), generator=g).cuda()
    wm, wdf = ops.stem_fragments(w1, wd)
    wm2, wdf2 = ops.pillar_stem_fragments(w1, wd)
    args = (v, c, n, w, sc, sh, [0.4, 0.4, 4], R, n_ag, 512, 512)

    def dense():
        cv = ops.pfn_scatter(*args, n_voxels_dev=off[n_ag:n_ag + 1])
        return ops.conv3x3(cv, w1, b1, None, True, 2), ops.conv1x1(cv, wd, bd, None, 0, stride=2)

    def sparse(layout="lanes"):
        pb = ops.pfn_pillars(*args, n_voxels_dev=off[n_ag:n_ag + 1])
        pb.weight_layout = layout
        return pb.stem_block(*((wm2, b1, wdf2, bd) if layout == "lanes" else (wm, b1, wdf, bd)))

    a_m, a_i = dense(); b_m, b_i = sparse(); c_m, c_i = sparse("tiles")
    st.synchronize()
    err = max(float((a_m - b_m).abs().max() / a_m.abs().max()), float((a_i - b_i).abs().max() / a_i.abs().max()))
    err1 = max(float((a_m - c_m).abs().max() / a_m.abs().max()), float((a_i - c_i).abs().max() / a_i.abs().max()))
    res = {"agents": n_ag, "pillars": M, "max_rel_err_vs_dense_kernels": err, "max_rel_err_v1": err1,
           "dense_chain_us": ops.graph_period_ms(dense) * 1e3, "pillar_chain_us": ops.graph_period_ms(sparse) * 1e3,
           "pillar_chain_v1_us": ops.graph_period_ms(lambda: sparse("tiles")) * 1e3}
    ops.TIMING = {}
    for _ in range(10):
        dense(); sparse()
    st.synchronize()
    res["kernel_own_us"] = {k: round(ms * 1e3, 2) for k, (cnt, ms) in ops.timing_summary().items()}
    ops.TIMING = None
    occ = (ops.pfn_scatter(*args) != 0).any(1).float().mean().item()
    res["cell_occupancy"] = occ
print(json.dumps(res))
