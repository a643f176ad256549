"""Per-modality BEV encoders behind the reference's names (opencood/models/heter_encoders.py):
PointPillar (:22-50), SECOND (:52-81), LiftSplatShoot (:83-241), LiftSplatShootVoxel (:244-301).
Classes are discovered by name exactly as the reference does (heter_pyramid_collab.py:41-48).
"""
import os

import numpy as np
import torch
import torch.nn as nn

from heal_amd import ops
from heal_amd.opencood.models.sub_modules.bev_blocks import grad_path
from heal_amd.opencood.models.sub_modules.pillar_vfe import PillarVFE
from heal_amd.opencood.models.sub_modules.point_pillar_scatter import PointPillarScatter


class PointPillar(nn.Module):
    """voxels -> fused PFN + scatter (K2) -> [n, 64, ny, nx].

    Besides the reference input (`voxel_features`, `voxel_coords`, `voxel_num_points`) the encoder
    accepts raw device point clouds -- `inputs_mX = {'points': [tensor [N_k,4] per agent]}` -- and
    voxelises them on the GPU (K1) without a host round trip."""

    def __init__(self, args):
        super().__init__()
        grid_size = (np.array(args["lidar_range"][3:6]) - np.array(args["lidar_range"][0:3])) / \
            np.array(args["voxel_size"])
        grid_size = np.round(grid_size).astype(np.int64)
        args["point_pillar_scatter"]["grid_size"] = grid_size  # the reference mutates args the same way
        self.lidar_range = [float(v) for v in args["lidar_range"]]
        self.voxel_size = [float(v) for v in args["voxel_size"]]
        self.max_points = int(args.get("max_points_per_voxel", 32))
        self.max_voxels = int(args.get("max_voxels", 70000))
        self.pillar_vfe = PillarVFE(args["pillar_vfe"], num_point_features=4, voxel_size=args["voxel_size"],
                                    point_cloud_range=args["lidar_range"])
        self.scatter = PointPillarScatter(args["point_pillar_scatter"])
        self._fold_key = None
        self._fold = None

    # set by the model when this encoder's backbone opens with a block that reads the pillars itself (bev_blocks.BasicBlock.
    # takes_pooled): the inference forward then returns ops.PillarBEV instead of the dense canvas (HEAL_K2_POOLED=0: always dense)
    emit_pooled = False

    def _scatter_op(self, v, c, n, weight, scale, shift, n_agents, n_voxels_dev=None):
        ny, nx = self.scatter.ny, self.scatter.nx
        if self.emit_pooled and os.environ.get("HEAL_K2_POOLED", "1") == "1" and int(weight.shape[0]) == 64:
            return ops.pfn_pillars(v, c, n, weight, scale, shift, self.voxel_size, self.lidar_range, n_agents, ny, nx,
                                   n_voxels_dev=n_voxels_dev)
        return ops.pfn_scatter(v, c, n, weight, scale, shift, self.voxel_size, self.lidar_range, n_agents, ny, nx,
                               n_voxels_dev=n_voxels_dev)

    def _bn(self):
        pfn = self.pillar_vfe.pfn_layers[0]
        tensors = [pfn.linear.weight] + ([pfn.norm.weight, pfn.norm.bias, pfn.norm.running_mean,
                                          pfn.norm.running_var] if pfn.use_norm else [pfn.linear.bias])
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if key != self._fold_key:
            self._fold = pfn.folded_bn()
            self._fold_key = key
        return self._fold

    def encode_points(self, point_list, max_points=None, max_voxels=None):
        """Raw device point clouds -> canvas with NO host round trip: K1 per agent into collated buffers (the running
        row offset stays on the device), then ONE K2 launch over all agents of the modality -- the reference's collated
        PillarVFE + PointPillarScatter call (heter_encoders.py:46-49), 3x67 MB written by one streaming kernel."""
        scale, shift = self._bn()
        weight = self.pillar_vfe.pfn_layers[0].linear.weight.detach()
        v, c, n, offsets = ops.voxelize_collated(point_list, self.lidar_range, self.voxel_size,
                                                 int(max_points or self.max_points), int(max_voxels or self.max_voxels))
        k = len(point_list)
        return self._scatter_op(v, c, n, weight, scale, shift, k, n_voxels_dev=offsets[k:k + 1])

    def forward(self, data_dict, modality_name):
        inp = data_dict[f"inputs_{modality_name}"]
        grad = grad_path(None, self)
        if "points" in inp:  # the caps travel with the clouds when they come from SpVoxelPreprocessor's deferred mode
            if not grad:
                return self.encode_points(inp["points"], inp.get("max_points_per_voxel"), inp.get("max_voxels"))
            # training: K1 still voxelises on the device (no gradient flows into the points); the rows past the voxel
            # count are dropped here, on the host's side of one synchronisation
            v, c, n, offsets = ops.voxelize_collated(inp["points"], self.lidar_range, self.voxel_size,
                                                     int(inp.get("max_points_per_voxel") or self.max_points),
                                                     int(inp.get("max_voxels") or self.max_voxels))
            m = int(offsets[len(inp["points"])].item())
            voxels, coords, num, n_agents = v[:m], c[:m], n[:m], len(inp["points"])
        else:
            voxels, coords, num = inp["voxel_features"], inp["voxel_coords"], inp["voxel_num_points"]
            # point_pillar_scatter.py:45 reads the batch size back from the device the same way
            n_agents = int(inp["n_agents"]) if "n_agents" in inp else int(coords[:, 0].max().item()) + 1
        if grad:   # gradient path: the scatter as a torch index operation; the PFN on its forward / backward kernels on the
            # device (HEAL_K2_BACKWARD=0, the CPU and max_points > 32: Linear / BatchNorm1d / max as torch operators)
            import os
            pfn0 = self.pillar_vfe.pfn_layers[0]
            if (voxels.is_cuda and ops.pfn_train_supported(voxels) and getattr(pfn0, "use_norm", True)
                    and len(self.pillar_vfe.pfn_layers) == 1 and pfn0.linear.bias is None
                    and pfn0.linear.out_features == 64 and pfn0.linear.in_features == 10   # the kernels hard-code Linear(10 -> 64)
                    and os.environ.get("HEAL_K2_BACKWARD", "1") == "1"):
                feats = self.pillar_vfe.pillar_features_kernels(voxels, coords, num)
            else:
                feats = self.pillar_vfe.pillar_features(voxels, coords, num)
            return self.scatter.canvas(feats, coords, n_agents)
        if coords.dtype != torch.int32:
            coords = coords.to(torch.int32)
        if num.dtype != torch.int32:
            num = num.to(torch.int32)
        scale, shift = self._bn()
        weight = self.pillar_vfe.pfn_layers[0].linear.weight.detach()
        return self._scatter_op(voxels, coords, num, weight, scale, shift, n_agents)


class SECOND(nn.Module):
    """voxels -> MeanVFE -> VoxelBackBone8x (sparse conv, K3) -> HeightCompression -> [n,128,256,256].
    Reference: heter_encoders.py:52-81.  Also accepts raw device point clouds ('points')."""

    def __init__(self, args):
        super().__init__()
        from heal_amd.opencood.models.sub_modules.height_compression import HeightCompression
        from heal_amd.opencood.models.sub_modules.mean_vfe import MeanVFE
        from heal_amd.opencood.models.sub_modules.sparse_backbone_3d import VoxelBackBone8x
        lidar_range = np.array(args["lidar_range"])
        grid_size = np.round((lidar_range[3:6] - lidar_range[:3]) / np.array(args["voxel_size"])).astype(np.int64)
        self.lidar_range = [float(v) for v in args["lidar_range"]]
        self.voxel_size = [float(v) for v in args["voxel_size"]]
        self.max_points = int(args.get("max_points_per_voxel", 5))
        self.max_voxels = int(args.get("max_voxels", 70000))
        self.vfe = MeanVFE(args["mean_vfe"], args["mean_vfe"]["num_point_features"])
        self.spconv_block = VoxelBackBone8x(args["spconv"], input_channels=args["spconv"]["num_features_in"],
                                            grid_size=grid_size)
        self.map_to_bev = HeightCompression(args["map2bev"])

    def forward(self, data_dict, modality_name):
        inp = data_dict[f"inputs_{modality_name}"]
        if torch.is_grad_enabled() and self.training:
            # gradient path: K1 still voxelises raw clouds on the device; MeanVFE as torch operators, the sparse backbone with its
            # sparse backward on the device (dense masked evaluation off the device: sparse_backbone_3d.py)
            if "points" in inp:
                v, c, n, offsets = ops.voxelize_collated(inp["points"], self.lidar_range, self.voxel_size,
                                                         int(inp.get("max_points_per_voxel") or self.max_points),
                                                         int(inp.get("max_voxels") or self.max_voxels))
                m = int(offsets[len(inp["points"])].item())
                voxels, coords, num, batch_size = v[:m], c[:m], n[:m], len(inp["points"])
            else:
                voxels, coords, num = inp["voxel_features"], inp["voxel_coords"], inp["voxel_num_points"]
                batch_size = int(inp["n_agents"]) if "n_agents" in inp else int(coords[:, 0].max().item()) + 1
            batch_dict = self.vfe({"voxel_features": voxels, "voxel_coords": coords, "voxel_num_points": num,
                                   "batch_size": batch_size})
            return self.map_to_bev(self.spconv_block.forward_autograd(batch_dict))["spatial_features"]
        n_dev = None
        if "points" in inp:
            # K1 per agent into collated buffers, the voxel count stays on the device: the whole encoder runs without
            # a host round trip (sparse layers take capacity + device count)
            voxels, coords, num, offsets = ops.voxelize_collated(
                inp["points"], self.lidar_range, self.voxel_size, int(inp.get("max_points_per_voxel") or self.max_points),
                int(inp.get("max_voxels") or self.max_voxels))
            batch_size = len(inp["points"])
            n_dev = offsets[batch_size:batch_size + 1]
        else:
            voxels, coords, num = inp["voxel_features"], inp["voxel_coords"], inp["voxel_num_points"]
            batch_size = int(inp["n_agents"]) if "n_agents" in inp else int(coords[:, 0].max().item()) + 1
        batch_dict = {"voxel_features": voxels, "voxel_coords": coords, "voxel_num_points": num,
                      "batch_size": batch_size, "n_voxels_dev": n_dev}
        batch_dict = self.vfe(batch_dict)
        batch_dict = self.spconv_block(batch_dict)
        batch_dict = self.map_to_bev(batch_dict)
        return batch_dict["spatial_features"]


class _LiftPool(torch.autograd.Function):
    """K4 with a hand-written backward: forward = heal_bev_pool (the inference kernels), backward = heal_bev_pool_backward
    (one wave per image pixel gathers the cell gradients of its D depth bins).  Neither direction materialises the
    [B,N,D,fH,fW,C] lifted tensor the reference's autograd keeps (0.3 GB per camera agent at BASELINE size)."""

    @staticmethod
    def forward(ctx, depth_logit, x_img, frustum, cam_mats, B, N, dx, bx, nx):
        depth_logit, x_img = depth_logit.contiguous(), x_img.contiguous()
        ctx.save_for_backward(depth_logit, x_img, frustum, cam_mats)
        ctx.geom = (B, N, dx, bx, nx)
        return ops.bev_pool(depth_logit, x_img, frustum, cam_mats, B, N, dx, bx, nx)

    @staticmethod
    def backward(ctx, grad_out):
        depth_logit, x_img, frustum, cam_mats = ctx.saved_tensors
        B, N, dx, bx, nx = ctx.geom
        g_logit, g_feat = ops.bev_pool_backward(grad_out.contiguous(), depth_logit, x_img, frustum, cam_mats, B, N, dx, bx, nx)
        return g_logit, g_feat, None, None, None, None, None, None, None


class PendingPool:
    """What LiftSplatShoot.forward_head hands back instead of the pooled map: the arguments of ops.bev_pool_pm, so that the lift + splat
    of SEVERAL camera modalities can go into one launch (ops.bev_pool_pm_multi; _heter_common.encode_modalities)."""

    def __init__(self, **args):
        self.args = args

    def finish(self):
        return ops.bev_pool_pm(**self.args)


class LiftSplatShoot(nn.Module):
    """Camera agents: image trunk -> (depth logits, image features) -> fused lift + BEV pool (K4).

    Reference: heter_encoders.py:83-241.  Differences by design: no hard-coded `.to("cuda")` in
    __init__ (the frustum is a buffer-like tensor that follows the module's device lazily), and the
    [BN,C,D,fH,fW] lifted tensor is never materialised."""

    def __init__(self, args):
        super().__init__()
        from heal_amd.opencood.models.sub_modules.lss_submodule import CamEncode, CamEncode_Resnet101
        from heal_amd.opencood.utils.camera_utils import depth_discretization, gen_dx_bx
        self.grid_conf = args["grid_conf"]
        self.data_aug_conf = args["data_aug_conf"]
        dx, bx, nx = gen_dx_bx(self.grid_conf["xbound"], self.grid_conf["ybound"], self.grid_conf["zbound"])
        self.dx_host = [float(v) for v in dx]
        self.bx_host = [float(v) for v in bx]
        self.nx_host = [int(v) for v in nx]
        self.depth_supervision = args["depth_supervision"]
        self.downsample = args["img_downsample"]
        self.camC = args["img_features"]
        self._frustum_cpu = self.create_frustum(depth_discretization)
        self._frustum_dev = {}
        self.D = self._frustum_cpu.shape[0]
        self.camera_encoder_type = args["camera_encoder"]
        enc = {"EfficientNet": CamEncode, "Resnet101": CamEncode_Resnet101}[self.camera_encoder_type]
        self.camencode = enc(self.D, self.camC, self.downsample, self.grid_conf["ddiscr"], self.grid_conf["mode"],
                             args["use_depth_gt"], args["depth_supervision"])
        self.depth_items = None

    def create_frustum(self, depth_discretization):
        """heter_encoders.py:110-123 -> [D, fH, fW, 3] (u, v, depth)."""
        ogfH, ogfW = self.data_aug_conf["final_dim"]
        fH, fW = ogfH // self.downsample, ogfW // self.downsample
        ds = torch.tensor(depth_discretization(*self.grid_conf["ddiscr"], self.grid_conf["mode"]),
                          dtype=torch.float).view(-1, 1, 1).expand(-1, fH, fW)
        D = ds.shape[0]
        xs = torch.linspace(0, ogfW - 1, fW, dtype=torch.float).view(1, 1, fW).expand(D, fH, fW)
        ys = torch.linspace(0, ogfH - 1, fH, dtype=torch.float).view(1, fH, 1).expand(D, fH, fW)
        return torch.stack((xs, ys, ds), -1).contiguous()

    def frustum(self, device):
        key = str(device)
        if key not in self._frustum_dev:
            self._frustum_dev[key] = self._frustum_cpu.to(device)
        return self._frustum_dev[key]

    @staticmethod
    def camera_matrices(rots, trans, intrins, post_rots, post_trans):
        """The 3x3 algebra of get_geometry (heter_encoders.py:137-146) on the device, one launch
        (heal_camera_matrices): [B,N,...] -> [B*N, 27] = rots @ inv(intrins) | inv(post_rots) | post_trans | trans | 0."""
        return ops.camera_matrices(rots, trans, intrins, post_rots, post_trans)

    _pixel_major_pool = True
    # Set by the owning model when what follows is a ResNetBEVBackbone whose first block reads the sparse pixel-major map
    # (bev_blocks.BasicBlock.takes_pooled): forward() then returns ops.PooledBEV instead of the dense [B, C*nz, ny, nx]
    # tensor and the canvas is never written (HEAL_K4_POOLED=0 keeps the dense hand-off for A/B).
    emit_pooled = False

    def pool(self, depth_logit, x_img, cam_mats, B, N):
        return ops.bev_pool(depth_logit, x_img, self.frustum(x_img.device), cam_mats, B, N, self.dx_host,
                            self.bx_host, self.nx_host)

    def pool_pixel_major(self, head, cam_mats, B, N, fH, fW, defer=False):
        pooled = self.emit_pooled and self.nx_host[2] == 1 and os.environ.get("HEAL_K4_POOLED", "1") == "1"
        pend = PendingPool(head=head, C=self.camC, D=self.D, fH=fH, fW=fW, frustum=self.frustum(head.device), cam_mats=cam_mats,
                           n_agents=B, n_cams=N, dx=self.dx_host, bx=self.bx_host, nx=self.nx_host, pooled=pooled)
        return pend if defer else pend.finish()

    def forward_head(self, data_dict, modality_name):
        """forward() up to the lift + splat: -> PendingPool (inference on the fused pixel-major path), else forward()'s result."""
        return self.forward(data_dict, modality_name, defer_pool=True)

    def lift_pool_autograd(self, depth_logit, x_img, inp, B, N):
        """Gradient path of get_geometry + voxel_pooling (heter_encoders.py:125-217) with torch operators: ego coordinates of
        every frustum point, softmax(depth) x features per point, summed into its BEV cell (index_add: differentiable with
        respect to the logits and the features), z folded into the channels.  -> [B, C*nz, ny, nx]"""
        fr = self.frustum(x_img.device)
        D, fH, fW, _ = fr.shape
        C = self.camC
        pts = fr.view(1, 1, D, fH, fW, 3) - inp["post_trans"].view(B, N, 1, 1, 1, 3)
        pts = torch.einsum("bnij,bndhwj->bndhwi", torch.inverse(inp["post_rots"]), pts)
        pts = torch.cat([pts[..., :2] * pts[..., 2:3], pts[..., 2:3]], -1)
        pts = torch.einsum("bnij,bndhwj->bndhwi", inp["rots"].matmul(torch.inverse(inp["intrins"])), pts)
        pts = pts + inp["trans"].view(B, N, 1, 1, 1, 3)
        dx, bx = pts.new_tensor(self.dx_host), pts.new_tensor(self.bx_host)
        cell = ((pts - (bx - dx / 2.0)) / dx).long()                      # truncation toward zero, like the reference's .long()
        nx, ny, nz = self.nx_host
        ok = ((cell[..., 0] >= 0) & (cell[..., 0] < nx) & (cell[..., 1] >= 0) & (cell[..., 1] < ny)
              & (cell[..., 2] >= 0) & (cell[..., 2] < nz))
        batch = torch.arange(B, device=pts.device).view(B, 1, 1, 1, 1).expand_as(ok)
        flat = ((batch * nz + cell[..., 2]) * ny + cell[..., 1]) * nx + cell[..., 0]
        prob = depth_logit.softmax(dim=1).view(B, N, D, fH, fW, 1)
        lifted = prob * x_img.view(B, N, C, fH, fW).permute(0, 1, 3, 4, 2).unsqueeze(2)      # [B,N,D,fH,fW,C]
        out = lifted.new_zeros((B * nz * ny * nx, C)).index_add(0, flat[ok], lifted[ok])
        return out.view(B, nz, ny, nx, C).permute(0, 1, 4, 2, 3).reshape(B, nz * C, ny, nx)

    def forward(self, data_dict, modality_name, defer_pool=False):
        inp = data_dict[f"inputs_{modality_name}"]
        x = inp["imgs"]
        B, N, C, imH, imW = x.shape
        if grad_path(x, self):
            items, depth_logit, x_img = self.camencode(x.view(B * N, C, imH, imW), pixel_major=False)
            if self.depth_supervision:
                self.depth_items = items
            if x.is_cuda and self.D <= 64 and self.camC <= 256 and os.environ.get("HEAL_K4_BACKWARD", "1") == "1":
                with torch.no_grad():
                    cam = self.camera_matrices(inp["rots"], inp["trans"], inp["intrins"], inp["post_rots"], inp["post_trans"])
                out = _LiftPool.apply(depth_logit, x_img, self.frustum(x.device), cam, B, N, self.dx_host, self.bx_host,
                                      self.nx_host)
            else:   # any device: the torch composition (materialises the lifted tensor)
                out = self.lift_pool_autograd(depth_logit, x_img, inp, B, N)
            nz = self.nx_host[2]
            if not self._pixel_major_pool and nz > 1:   # LiftSplatShootVoxel: max over the z bins
                out = out.view(B, nz, self.camC, out.shape[2], out.shape[3]).max(dim=1)[0]
            return out
        res = self.camencode(x.view(B * N, C, imH, imW), pixel_major=self._pixel_major_pool)
        if self.depth_supervision:
            self.depth_items = res[0]
        cam = self.camera_matrices(inp["rots"], inp["trans"], inp["intrins"], inp["post_rots"], inp["post_trans"])
        if len(res) == 2:   # (items, head): pixel-major fused heads -> K4 without a transposition pass
            # feature-map size from the trunk's own output: the trunks round UP (7x7 s2 p3, max-pool p1, TF-same padding), so
            # an image size not divisible by the downsample factor is not imH // downsample
            fH, fW = self.camencode.last_feature_hw
            return self.pool_pixel_major(res[1], cam, B, N, fH, fW, defer=defer_pool)
        return self.pool(res[1].contiguous(), res[2].contiguous(), cam, B, N)


class LiftSplatShootVoxel(LiftSplatShoot):
    """heter_encoders.py:244-301: max over the z bins instead of folding them into channels."""

    _pixel_major_pool = False   # the z-max variant keeps the NCHW entry point

    def pool(self, depth_logit, x_img, cam_mats, B, N):
        out = super().pool(depth_logit, x_img, cam_mats, B, N)
        nz = self.nx_host[2]
        if nz == 1:
            return out
        return out.view(B, nz, self.camC, out.shape[2], out.shape[3]).max(dim=1)[0]
