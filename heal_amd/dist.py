"""Agent-sharded execution of one scene over several GPUs.

The reference has no inference-time parallelism.  Everything up to and including the per-agent
pyramid stages and occupancy heads is independent per agent in eval mode, so rank r owns scene
agents {a : a % world == r} (the ego, agent 0, lives on rank 0).  Each rank warps its own agents'
multi-scale features and scores into the ego frame (heal_warp_agent), packs them into one buffer,
and ONE all-gather (RCCL over xGMI on the GPU box, gloo in the CPU tests) brings every agent's
warped maps to every rank.  Rank 0 then runs the fusion tail (heal_fuse_warped, deblocks, shrink
head, detection heads, decode + NMS).
"""
import torch
import torch.distributed as dist


def owned_agents(n_agents, rank, world):
    return [a for a in range(n_agents) if a % world == rank]


def slots_per_rank(n_agents, world):
    return (n_agents + world - 1) // world


def pack_levels(level_feats, level_scores, n_slots):
    """level_feats[l]: [n_local, C_l, H_l, W_l], level_scores[l]: [n_local, 1, H_l, W_l] ->
    one contiguous [n_slots, sum_l (C_l+1) H_l W_l] buffer (unused slots are zero: a zero score is
    masked to -inf by the fusion kernel, so padding slots never contribute)."""
    n_local = level_feats[0].shape[0] if len(level_feats) else 0
    per_slot = sum(f.shape[1] * f.shape[2] * f.shape[3] + s.shape[2] * s.shape[3]
                   for f, s in zip(level_feats, level_scores))
    ref = level_feats[0]
    buf = torch.zeros((n_slots, per_slot), dtype=ref.dtype, device=ref.device)
    off = 0
    for f, s in zip(level_feats, level_scores):
        nf = f.shape[1] * f.shape[2] * f.shape[3]
        ns = s.shape[2] * s.shape[3]
        if n_local:
            buf[:n_local, off:off + nf] = f.reshape(n_local, nf)
            buf[:n_local, off + nf:off + nf + ns] = s.reshape(n_local, ns)
        off += nf + ns
    return buf


def unpack_levels(gathered, shapes, n_agents, world):
    """gathered: [world, n_slots, per_slot]; shapes[l] = (C_l, H_l, W_l).  Returns per level
    (feats [n_agents,C,H,W], scores [n_agents,1,H,W]) in SCENE agent order."""
    # slot s of rank r holds agent r + s*world
    order = []
    for a in range(n_agents):
        order.append((a % world, a // world))
    rows = torch.stack([gathered[r, s] for r, s in order])  # [n_agents, per_slot]
    out = []
    off = 0
    for (C, H, W) in shapes:
        nf, ns = C * H * W, H * W
        feats = rows[:, off:off + nf].reshape(n_agents, C, H, W)
        scores = rows[:, off + nf:off + nf + ns].reshape(n_agents, 1, H, W)
        out.append((feats.contiguous(), scores.contiguous()))
        off += nf + ns
    return out


def all_gather_packed(buf, world):
    """The path's single collective: every rank contributes its [n_slots, per_slot] buffer."""
    if world == 1:
        return buf.unsqueeze(0)
    n_slots, per_slot = buf.shape
    out = torch.empty((world * n_slots, per_slot), dtype=buf.dtype, device=buf.device)
    dist.all_gather_into_tensor(out, buf.contiguous())  # concatenation along dim 0, rank-major
    return out.view(world, n_slots, per_slot)


class ShardedCollab:
    """HeterPyramidCollab forward split at the fusion boundary, one scene, `world` ranks."""

    def __init__(self, model, rank, world):
        self.model = model
        self.rank = rank
        self.world = world

    @torch.no_grad()
    def forward(self, scene_input, n_agents, local_inputs):
        """local_inputs: {'inputs_mX': ...} for the agents this rank owns, in scene order (the
        reference's collated layout, restricted to the local agents).  Returns the model output dict
        on rank 0, None elsewhere."""
        from heal_amd import ops
        from heal_amd.opencood.models.fuse_modules.pyramid_fuse import crop_window
        from heal_amd.opencood.utils.transformation_utils import normalize_pairwise_tfm, pairwise_to_host
        m = self.model
        pairwise, grid_f64 = pairwise_to_host(scene_input["pairwise_t_matrix"])
        affine = normalize_pairwise_tfm(pairwise, m.H, m.W, m.fake_voxel_size)[0]  # [L,L,2,3]
        mine = owned_agents(n_agents, self.rank, self.world)
        mods = scene_input["agent_modality_list"]
        pb = m.pyramid_backbone
        n_slots = slots_per_rank(n_agents, self.world)
        level_feats, level_scores, shapes = [], [], None
        if mine:
            feats = {}
            for mod in m.modality_name_list:
                if f"inputs_{mod}" in local_inputs:
                    feats[mod] = m.encode_modality(local_inputs, mod)
            cursor = {k: 0 for k in feats}
            parts = []
            for a in mine:
                parts.append(feats[mods[a]][cursor[mods[a]]])
                cursor[mods[a]] += 1
            x = torch.stack(parts)
            if m.compress:
                x = m.compressor(x)
            stages = pb.get_multiscale_feature(x)
            for i, f in enumerate(stages):
                occ = getattr(pb, f"single_head_{i}")(f)
                fe_all, se_all = [], []
                for k, a in enumerate(mine):
                    crop = None
                    if mods[a] in m.cam_crop_info:
                        info = m.cam_crop_info[mods[a]]
                        crop = [crop_window(f.shape[2], f.shape[3], info[f"crop_ratio_H_{mods[a]}"],
                                            info[f"crop_ratio_W_{mods[a]}"])]
                    fe, se = ops.warp_agent(f[k], occ[k], affine[0, a], grid_f64, crop)
                    fe_all.append(fe); se_all.append(se)
                level_feats.append(torch.stack(fe_all))
                level_scores.append(torch.stack(se_all))
            shapes = [tuple(f.shape[1:]) for f in level_feats]
        # every rank needs the level shapes to size its (possibly empty) contribution
        shapes = self._level_shapes() if shapes is None else shapes
        if not mine:
            dev = next(m.parameters()).device
            level_feats = [torch.zeros((0,) + s, device=dev) for s in shapes]
            level_scores = [torch.zeros((0, 1) + s[1:], device=dev) for s in shapes]
        buf = pack_levels(level_feats, level_scores, n_slots)
        gathered = all_gather_packed(buf, self.world)
        if self.rank != 0:
            return None
        fused = []
        for feats_ego, scores_ego in unpack_levels(gathered, shapes, n_agents, self.world):
            fused.append(ops.fuse_warped(feats_ego, scores_ego).unsqueeze(0))
        y = pb.decode_multiscale_feature(fused)
        cls_preds, reg_preds, dir_preds = m.heads(y)
        return {"pyramid": "collab", "cls_preds": cls_preds, "reg_preds": reg_preds, "dir_preds": dir_preds}

    def _level_shapes(self):
        m = self.model
        fb = m.args["fusion_backbone"]
        # BEV size entering the pyramid: lidar grid / 2 (all HEAL encoders+backbones end at 0.8 m/px)
        H = int(round(m.H / 0.8))
        W = int(round(m.W / 0.8))
        shapes = []
        for c, s in zip(fb["num_filters"], fb["layer_strides"]):
            H, W = H // s, W // s
            shapes.append((c, H, W))
        return shapes
