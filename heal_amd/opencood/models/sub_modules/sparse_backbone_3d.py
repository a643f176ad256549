"""VoxelBackBone8x -- the SECOND sparse 3-D backbone (reference: opencood/models/sub_modules/
sparse_backbone_3d.py:33-152) on the gfx950 gather-GEMM kernel K3 (heal_sp_conv).

Parameter names follow the reference (`conv_input.0.weight`, `conv_input.1.*`, `conv2.0.0.weight`, ...).
Convolution weights are stored in spconv 1.2.1 layout [kz,ky,kx,Cin,Cout] (the layout of the authors'
checkpoints, README "spconv 1.2.1"); spconv 2.x checkpoints ([Cout,kz,ky,kx,Cin]) are permuted on load.

Gradient path (training).  On the device (round 3): SPARSE -- the rulebooks of the inference path, the sparse convolution as an
autograd Function whose forward is heal_sp_conv and whose backward is (i) heal_sp_conv again on the transposed rulebook
(heal_sp_transpose_neighbors) with weight[tap]^T for the gradient of the input features and (ii) per tap a gather of the
paired rows + one matrix product for the weight gradient; BatchNorm1d + ReLU on the active rows and the final densification are
torch operators on [N, C] tensors.  No dense grid: training SECOND at the reference's +-102.4 m / 0.1 m configuration (172 M
cells per agent) fits.  Off the device (and with HEAL_SP_GRAD=dense): spconv's arithmetic as a DENSE conv3d on the densified
grid, masked by the active-site rules -- plain torch, differentiable, only for small grids; it is what pins the sparse path
(tests/test_gpu_train.py) and the CPU oracle comparison.
"""
import os
import torch
import torch.nn as nn
import torch.nn.functional as F


class SparseConvParam(nn.Module):
    """Weight container of one SubMConv3d / SparseConv3d (bias=False)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, subm=False, indice_key=None):
        super().__init__()
        k = (kernel_size,) * 3 if isinstance(kernel_size, int) else tuple(kernel_size)
        s = (stride,) * 3 if isinstance(stride, int) else tuple(stride)
        p = (padding,) * 3 if isinstance(padding, int) else tuple(padding)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = k, s, p
        self.subm = subm
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(*k, in_channels, out_channels))
        nn.init.kaiming_uniform_(self.weight.view(-1, in_channels, out_channels), a=5 ** 0.5)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        key = prefix + "weight"
        w = state_dict.get(key)
        if w is not None and tuple(w.shape) != tuple(self.weight.shape) and w.dim() == 5 and \
                tuple(w.permute(1, 2, 3, 4, 0).shape) == tuple(self.weight.shape):
            state_dict[key] = w.permute(1, 2, 3, 4, 0).contiguous()  # spconv 2.x -> 1.x layout
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def flat_weight(self):
        K = self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        return self.weight.detach().reshape(K, self.in_channels, self.out_channels).contiguous()


class _SparseConvFn(torch.autograd.Function):
    """out[o] = sum_tap W[tap]^T x[nbr[o][tap]] on the gather-GEMM kernel, with a sparse backward."""

    @staticmethod
    def forward(ctx, feats, weight, nbr):
        from heal_amd import ops
        ctx.save_for_backward(feats, weight, nbr)
        return ops.sp_conv_raw(feats.detach().contiguous(), nbr, weight.detach().contiguous())

    @staticmethod
    def backward(ctx, g):
        from heal_amd import ops
        feats, weight, nbr = ctx.saved_tensors
        g = g.contiguous()
        gf = gw = None
        if ctx.needs_input_grad[0]:   # d x[i] = sum_tap W[tap] g[nbr_t[i][tap]]: the same kernel, roles swapped
            nbr_t = ops.sp_transpose_neighbors(nbr, feats.shape[0])
            gf = ops.sp_conv_raw(g, nbr_t, weight.detach().transpose(1, 2).contiguous())
        if ctx.needs_input_grad[1]:   # d W[tap] = X_pairs^T G_pairs
            x = feats.detach().contiguous()
            if ops.sp_wgrad_supported(int(weight.shape[1]), int(weight.shape[2])) and os.environ.get("HEAL_SP_WGRAD", "1") == "1":
                gw = ops.sp_wgrad(x, g, nbr)          # heal_sp_wgrad: pair-compacted gather + MFMA over the pair index
            else:   # gather the paired rows, one matrix product per tap
                gw = torch.zeros_like(weight)
                for t in range(weight.shape[0]):
                    o = (nbr[:, t] >= 0).nonzero(as_tuple=True)[0]
                    if o.numel():
                        gw[t] = x.index_select(0, nbr[o, t].long()).t() @ g.index_select(0, o)
        return gf, gw, None


class _Block(nn.Sequential):
    """SparseSequential(conv, BatchNorm1d(eps 1e-3), ReLU) -- sparse_backbone_3d.py:11-30."""

    def __init__(self, conv, channels):
        super().__init__(conv, nn.BatchNorm1d(channels, eps=1e-3, momentum=0.01), nn.ReLU())
        self._key = None
        self._fold = None

    def bn(self):
        bn = self[1]
        key = tuple((t.data_ptr(), t._version) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
        if key != self._key:
            with torch.no_grad():
                scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                self._fold = (scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous())
            self._key = key
        return self._fold

    def run_dense(self, x, mask):
        """Gradient path of one block on dense tensors: x [B,Cin,D,H,W] (zero at inactive cells), mask [B,1,D,H,W] ->
        (y [B,Cout,D',H',W'], mask')."""
        conv, bn = self[0], self[1]
        w = conv.weight.permute(4, 3, 0, 1, 2)                       # [kz,ky,kx,Cin,Cout] -> [Cout,Cin,kz,ky,kx]
        if conv.subm:
            y = F.conv3d(x, w, None, 1, tuple(k // 2 for k in conv.kernel_size))
            m = mask
        else:
            y = F.conv3d(x, w, None, conv.stride, conv.padding)
            m = (F.max_pool3d(mask, conv.kernel_size, conv.stride, conv.padding) > 0).to(x.dtype)
        site = m[:, 0].nonzero(as_tuple=True)                        # (b, z, y, x) of the active output cells
        rows = F.relu(bn(y.permute(0, 2, 3, 4, 1)[site]))            # BatchNorm1d over the ACTIVE rows only, like spconv
        out = y.new_zeros(y.permute(0, 2, 3, 4, 1).shape).index_put(site, rows)
        return out.permute(0, 4, 1, 2, 3), m

    def run_sparse_autograd(self, x, feats, nbr_cache):
        """Gradient path on the device: x carries the SITES (heal_amd.ops.SparseTensor, no gradient), feats [n, Cin] the
        features with autograd history -> (SparseTensor of the output sites, features [n_out, Cout])."""
        from heal_amd.ops import SparseTensor
        conv, bn = self[0], self[1]
        K = conv.kernel_size[0] * conv.kernel_size[1] * conv.kernel_size[2]
        w = conv.weight.reshape(K, conv.in_channels, conv.out_channels)
        if conv.subm:
            nbr = nbr_cache.get(conv.indice_key)
            if nbr is None:
                nbr = x.neighbors(x.indices, x.spatial_shape, conv.kernel_size, (1, 1, 1), tuple(k // 2 for k in conv.kernel_size))
                nbr_cache[conv.indice_key] = nbr
            y = x
        else:
            out_idx, out_shape, _, rank = x.out_sites_ex(conv.kernel_size, conv.stride, conv.padding)
            nbr = x.neighbors(out_idx, out_shape, conv.kernel_size, conv.stride, conv.padding)
            y = SparseTensor(None, out_idx, out_shape, x.batch_size)
            y._rank = rank
        out = _SparseConvFn.apply(feats, w, nbr)
        return y, F.relu(bn(out))     # BatchNorm1d over the active rows (batch statistics when training), like spconv

    def run(self, x, nbr_cache):
        """x: heal_amd.ops.SparseTensor -> SparseTensor."""
        from heal_amd.ops import SparseTensor
        conv = self[0]
        scale, shift = self.bn()
        if conv.subm:
            # the rulebook of an indice_key comes in the form its layer's kernel reads (ops.SparseTensor.rulebook: pair tiles for the
            # c_in <= 16 layers, the neighbour table otherwise); layers that share a key but not the form each get their own
            key = (conv.indice_key, x.tiles_ok(conv.kernel_size, conv.in_channels, conv.out_channels))
            nbr = nbr_cache.get(key)
            if nbr is None:
                nbr = x.rulebook(x.indices, x.spatial_shape, conv.kernel_size, (1, 1, 1), tuple(k // 2 for k in conv.kernel_size),
                                 conv.in_channels, conv.out_channels, n_out_dev=x.n_dev)
                nbr_cache[key] = nbr
            feats = x.conv(nbr, conv.flat_weight(), scale, shift, relu=True, n_out_dev=x.n_dev)
            y = SparseTensor(feats, x.indices, x.spatial_shape, x.batch_size, x.n_dev, x._checks, x._root_cap)
            y._table, y._rank, y._rank_root = x._table, x._rank, x._rank_root
            return y
        out_idx, out_shape, n_out_dev, rank = x.out_sites_ex(conv.kernel_size, conv.stride, conv.padding)
        nbr = x.rulebook(out_idx, out_shape, conv.kernel_size, conv.stride, conv.padding, conv.in_channels, conv.out_channels,
                         n_out_dev=n_out_dev)
        feats = x.conv(nbr, conv.flat_weight(), scale, shift, relu=True, n_out_dev=n_out_dev)
        y = SparseTensor(feats, out_idx, out_shape, x.batch_size, n_out_dev, x._checks, x._root_cap)
        y._rank = rank   # occupancy bitmap + prefix counts of the new site set: the later layers' neighbour queries
        return y


class _DenseResult:
    """What HeightCompression reads on the gradient path: `.dense()` = [B, C*D, H, W] (channel = c*D + z)."""

    def __init__(self, x):
        self.x = x

    def dense(self):
        B, C, D, H, W = self.x.shape
        return self.x.reshape(B, C * D, H, W)


def _block(cin, cout, k, key, stride=1, padding=0, conv_type="subm"):
    conv = SparseConvParam(cin, cout, k, stride, padding, subm=(conv_type == "subm"), indice_key=key)
    return _Block(conv, cout)


class VoxelBackBone8x(nn.Module):
    DENSE_GRAD_MAX_CELLS = 24_000_000   # dense gradient path: ~1.5 GB for the widest early activation at this size

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        gs = [int(v) for v in grid_size]
        self.sparse_shape = [gs[2] + 1, gs[1], gs[0]]  # grid_size[::-1] + [1, 0, 0]
        self.conv_input = _block(input_channels, 16, 3, "subm1", padding=1)
        self.conv1 = nn.Sequential(_block(16, 16, 3, "subm1", padding=1))
        self.conv2 = nn.Sequential(_block(16, 32, 3, "spconv2", stride=2, padding=1, conv_type="spconv"),
                                   _block(32, 32, 3, "subm2", padding=1), _block(32, 32, 3, "subm2", padding=1))
        self.conv3 = nn.Sequential(_block(32, 64, 3, "spconv3", stride=2, padding=1, conv_type="spconv"),
                                   _block(64, 64, 3, "subm3", padding=1), _block(64, 64, 3, "subm3", padding=1))
        self.conv4 = nn.Sequential(_block(64, 64, 3, "spconv4", stride=2, padding=(0, 1, 1), conv_type="spconv"),
                                   _block(64, 64, 3, "subm4", padding=1), _block(64, 64, 3, "subm4", padding=1))
        self.num_point_features = model_cfg.get("num_features_out", 128)
        self.conv_out = _block(64, self.num_point_features, (3, 1, 1), "spconv_down2", stride=(2, 1, 1), padding=0,
                               conv_type="spconv")
        self.backbone_channels = {"x_conv1": 16, "x_conv2": 32, "x_conv3": 64, "x_conv4": 64}

    def forward_autograd(self, batch_dict):
        """Gradient path: sparse on the device (forward_autograd_sparse), dense masked evaluation otherwise (module docstring)."""
        feats = batch_dict["voxel_features"]
        if feats.is_cuda and os.environ.get("HEAL_SP_GRAD", "sparse") != "dense":
            return self.forward_autograd_sparse(batch_dict)
        coords = batch_dict["voxel_coords"].long()
        B = int(batch_dict["batch_size"])
        D, H, W = self.sparse_shape
        cells = B * D * H * W
        if cells > self.DENSE_GRAD_MAX_CELLS:
            raise NotImplementedError(
                f"VoxelBackBone8x: the gradient path evaluates the sparse encoder as a dense masked conv3d; a grid of "
                f"{B} x {D} x {H} x {W} = {cells / 1e6:.0f} M cells x >= 16 channels does not fit (limit "
                f"{self.DENSE_GRAD_MAX_CELLS / 1e6:.0f} M cells, ~0.1 GB per channel).  Training SECOND at this range needs the "
                "sparse backward of K3, which runs on the device only (tensors on cuda:N and HEAL_SP_GRAD unset); here use a "
                "smaller lidar_range / larger voxel_size, or run inference (no_grad).")
        site = (coords[:, 0], coords[:, 1], coords[:, 2], coords[:, 3])
        x = feats.new_zeros((B, D, H, W, feats.shape[1])).index_put(site, feats).permute(0, 4, 1, 2, 3)
        mask = feats.new_zeros((B, D, H, W)).index_put(site, feats.new_ones(coords.shape[0])).unsqueeze(1)
        x, mask = self.conv_input.run_dense(x, mask)
        for stage in (self.conv1, self.conv2, self.conv3, self.conv4):
            for blk in stage:
                x, mask = blk.run_dense(x, mask)
        x, mask = self.conv_out.run_dense(x, mask)
        batch_dict.update({"encoded_spconv_tensor": _DenseResult(x), "encoded_spconv_tensor_stride": 8})
        return batch_dict

    def forward_autograd_sparse(self, batch_dict):
        """Gradient path on the device: sparse forward AND backward (module docstring)."""
        from heal_amd.ops import SparseTensor
        feats = batch_dict["voxel_features"]
        B = int(batch_dict["batch_size"])
        with torch.no_grad():
            x = SparseTensor.from_unsorted(torch.zeros((feats.shape[0], 1), device=feats.device),
                                           batch_dict["voxel_coords"].int().contiguous(), self.sparse_shape, B)
        f = feats.index_select(0, x._perm.long())       # the sites are kept sorted by linear coordinate
        cache = {}
        x, f = self.conv_input.run_sparse_autograd(x, f, cache)
        for stage in (self.conv1, self.conv2, self.conv3, self.conv4):
            for blk in stage:
                x, f = blk.run_sparse_autograd(x, f, cache)
        x, f = self.conv_out.run_sparse_autograd(x, f, cache)
        D, H, W = x.spatial_shape
        idx = x.indices.long()
        dense = f.new_zeros((B, D, H, W, f.shape[1])).index_put((idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]), f)
        batch_dict.update({"encoded_spconv_tensor": _DenseResult(dense.permute(0, 4, 1, 2, 3)),
                           "encoded_spconv_tensor_stride": 8})
        return batch_dict

    def forward(self, batch_dict):
        from heal_amd.ops import SparseTensor
        x = SparseTensor.from_unsorted(batch_dict["voxel_features"], batch_dict["voxel_coords"].int().contiguous(),
                                       self.sparse_shape, int(batch_dict["batch_size"]),
                                       n_dev=batch_dict.get("n_voxels_dev"))
        cache = {}
        x = self.conv_input.run(x, cache)
        for stage in (self.conv1, self.conv2, self.conv3, self.conv4):
            for blk in stage:
                x = blk.run(x, cache)
        out = self.conv_out.run(x, cache)
        self.last_sparse = out  # plain attribute: lets a caller check SparseTensor.overflow() after the fact
        batch_dict.update({"encoded_spconv_tensor": out, "encoded_spconv_tensor_stride": 8})
        return batch_dict
