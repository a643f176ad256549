"""V2X-ViT fusion transformer (reference: opencood/models/sub_modules/v2xvit_basic.py:13-192,
hmsa.py:7-151, mswin.py:19-122, split_attn.py:6-62, base_transformer.py:7-40).

Module / parameter layout mirrors the reference so that checkpoints load.  Execution differs by design:
  * tensors stay pixel-major [L, H, W, C] (one scene).  Inference (round 3): every Linear is heal_linear -- the token-major
    fp32 MFMA GEMM with PreNorm's LayerNorm applied in its prologue (statistics from heal_ln_stats, gamma / beta folded into
    the weights) and bias / GELU / residual in its epilogue; the three to_qkv projections of the window pyramid are one
    256 -> 2304 GEMM writing three buffers; the three to_out projections, the split-attention weighting of the branches and
    the residual are ONE K = 768 GEMM whose weight rows are scaled per agent (heal_split_attn_weights).  No library GEMM, no
    ATen LayerNorm / GELU / add / mul / mean / cat kernel is left on the path (`_fused_ok`; HEAL_V2XVIT_FUSED=0 restores the
    round-2 library path for A/B);
  * HGTCavAttention: HEAL always passes a zero prior encoding (fusion_in_one.py:346-355), so every agent
    has type 0 and only relation 0 is exercised.  The per-head relation matrices are folded into the q / v
    projections once (cached), and the per-pixel L x L attention runs in the fused kernel K6;
  * padded agents are not materialised: they are masked keys and never reach the ego row, so the
    transformer runs on the real agents only;
  * STTF / ROI mask: with the identity spatial-correction matrices HEAL passes (fusion_in_one.py:367) the
    warp is an identity resample and the ROI mask is all ones -- skipped (non-identity raises).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from heal_amd import ops


def _grad_path(x, module):
    return torch.is_grad_enabled() and (x.requires_grad or module.training)


def _fused_ok(x, module):
    """Inference on the device with shapes heal_linear takes (channels % 128 == 0, H W % 128 == 0): the fused path."""
    if not x.is_cuda or _grad_path(x, module) or os.environ.get("HEAL_V2XVIT_FUSED", "1") == "0":
        return False
    L, H, W, C = x.shape
    return C % 128 == 0 and (H * W) % 128 == 0 and x.dtype == torch.float32


class _Folded:
    """Cache of derived weights, rebuilt when any source tensor changes (data_ptr / version)."""

    def __init__(self):
        self.key, self.val = None, None

    def get(self, tensors, build):
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if key != self.key:
            with torch.no_grad():
                self.val = build()
            self.key = key
        return self.val


def _fold_ln(weight_nk, bias, norm):
    """LayerNorm(x) W^T + b = ((x - mean) rstd) (W * gamma)^T + (b + W beta)."""
    w = (weight_nk * norm.weight[None, :]).contiguous()
    b = weight_nk @ norm.bias
    if bias is not None:
        b = b + bias
    return w, b.contiguous()


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x, **kwargs):
        return self.fn(self.norm(x), **kwargs)

    def residual(self, x):
        """fn(norm(x)) + x -- as one fused sequence when the wrapped module has one (inference on the device)."""
        if hasattr(self.fn, "fused_residual") and _fused_ok(x, self):
            return self.fn.fused_residual(x, self.norm)
        return self.fn(self.norm(x)) + x


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout=0.0):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, dim), nn.Dropout(dropout))
        self._f = _Folded()

    def forward(self, x):
        if _grad_path(x, self):
            return self.net(x)          # training: the reference's Sequential, both Dropouts included (base_transformer.py:19-30)
        return self.net[3](F.gelu(self.net[0](x)))

    def fused_residual(self, x, norm):
        """LayerNorm -> Linear -> GELU -> Linear -> + x as two heal_linear launches."""
        l1, l2 = self.net[0], self.net[3]
        if not (ops.linear_supported(1, l1.in_features, l1.out_features) and ops.linear_supported(1, l2.in_features, l2.out_features)):
            return self.forward(norm(x)) + x
        w1, b1 = self._f.get([l1.weight, l1.bias, norm.weight, norm.bias], lambda: _fold_ln(l1.weight, l1.bias, norm))
        shp = x.shape
        h = ops.linear(x, w1, b1, stats=ops.ln_stats(x, norm.eps), act="gelu")
        return ops.linear(h, l2.weight, l2.bias, residual=x.reshape(-1, shp[-1])).view(shp)


class HGTCavAttention(nn.Module):
    def __init__(self, dim, heads, num_types=2, num_relations=4, dim_head=64, dropout=0.1):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        self.scale = dim_head ** -0.5
        self.num_types = num_types
        self.k_linears, self.q_linears = nn.ModuleList(), nn.ModuleList()
        self.v_linears, self.a_linears = nn.ModuleList(), nn.ModuleList()
        self.norms = nn.ModuleList()
        for _ in range(num_types):
            self.k_linears.append(nn.Linear(dim, inner))
            self.q_linears.append(nn.Linear(dim, inner))
            self.v_linears.append(nn.Linear(dim, inner))
            self.a_linears.append(nn.Linear(inner, dim))
        self.relation_att = nn.Parameter(torch.empty(num_relations, heads, dim_head, dim_head))
        self.relation_msg = nn.Parameter(torch.empty(num_relations, heads, dim_head, dim_head))
        nn.init.xavier_uniform_(self.relation_att)
        nn.init.xavier_uniform_(self.relation_msg)
        self.drop_out = nn.Dropout(dropout)   # hmsa.py:57,149: applied to the projected output (active in training only)
        self._key = None
        self._qkv = None
        self._f = _Folded()

    def _folded_qkv(self):
        """[dim, 3*inner] weight and bias of x -> (q W_att, k, v W_msg) for agent type 0 / relation 0:
        att = (q W_att) . k   (hmsa.py:131-134),  message = v W_msg   (hmsa.py:141-143)."""
        ts = [self.q_linears[0].weight, self.q_linears[0].bias, self.k_linears[0].weight, self.k_linears[0].bias,
              self.v_linears[0].weight, self.v_linears[0].bias, self.relation_att, self.relation_msg]
        key = tuple((t.data_ptr(), t._version) for t in ts)
        if key != self._key:
            with torch.no_grad():
                m, d = self.heads, self.dim_head
                wa = torch.block_diag(*[self.relation_att[0, h] for h in range(m)])   # [inner, inner]
                wm = torch.block_diag(*[self.relation_msg[0, h] for h in range(m)])
                wq = self.q_linears[0].weight.t() @ wa            # x @ Wq^T @ blockdiag(W_att)
                bq = self.q_linears[0].bias @ wa
                wv = self.v_linears[0].weight.t() @ wm
                bv = self.v_linears[0].bias @ wm
                w = torch.cat([wq, self.k_linears[0].weight.t(), wv], dim=1).contiguous()
                b = torch.cat([bq, self.k_linears[0].bias, bv]).contiguous()
                self._qkv = (w, b)
            self._key = key
        return self._qkv

    def forward(self, x, mask=None, prior_encoding=None):
        """x [L,H,W,C] (one scene, real agents only) -> [L,H,W,C]."""
        L, H, W, C = x.shape
        if torch.is_grad_enabled() and (x.requires_grad or self.training):
            # gradient path (hmsa.py:110-151 for agent type 0 / relation 0): per pixel and head
            #   att[i,j] = (q_i W_att) . k_j * scale,  out_i = sum_j softmax_j(att)[i,j] (v_j W_msg)
            m, d = self.heads, self.dim_head
            flat = x.reshape(L, H * W, C)
            q = self.q_linears[0](flat).view(L, H * W, m, d)
            k = self.k_linears[0](flat).view(L, H * W, m, d)
            v = self.v_linears[0](flat).view(L, H * W, m, d)
            qa = torch.einsum("lphd,hde->lphe", q, self.relation_att[0])
            vm = torch.einsum("lphd,hde->lphe", v, self.relation_msg[0])
            if (L <= 8 and m * d == 256 and ops.agent_attention_train_supported(flat, m)
                    and os.environ.get("HEAL_ATTN_GRAD", "kernel") != "torch"):
                # on the device: K6 forward + heal_agent_attention_backward (no [HW, m, L, L] score / [L, HW, m, d] message tensors
                # kept for the backward: q, k, v are saved and the probabilities recomputed per pixel)
                out = ops.AgentAttention.apply(qa.reshape(L, H * W, m * d), k.reshape(L, H * W, m * d),
                                               vm.reshape(L, H * W, m * d), m, self.scale, L, True).view(L, H * W, m, d)
            else:
                att = torch.einsum("iphd,jphd->phij", qa, k) * self.scale        # [HW, m, L, L]
                out = torch.einsum("phij,jphd->iphd", att.softmax(dim=-1), vm)   # [L, HW, m, d]
            return self.drop_out(self.a_linears[0](out.reshape(L, H * W, m * d))).reshape(L, H, W, C)
        w, b = self._folded_qkv()
        qkv = torch.addmm(b, x.reshape(-1, C), w)                       # [L*H*W, 3*inner]
        inner = self.heads * self.dim_head
        qkv = qkv.view(L, H * W, 3, inner).permute(2, 1, 0, 3).contiguous()   # [3, HW, L, inner]
        out = ops.agent_attention(qkv[0], qkv[1], qkv[2], self.heads, self.scale)   # [HW, L, inner]
        out = self.a_linears[0](out)                                     # [HW, L, C]
        return out.permute(1, 0, 2).reshape(L, H, W, C)

    def fused_residual(self, x, norm):
        """LayerNorm -> (q W_att | k | v W_msg) as ONE 256 -> 768 heal_linear writing three [L, HW, inner] buffers -> per-pixel
        agent attention on agent-major tensors -> a_linear + x in the epilogue of the second heal_linear."""
        L, H, W, C = x.shape
        inner = self.heads * self.dim_head
        if inner != 256 or not ops.linear_supported(1, C, 3 * inner):
            return self.forward(norm(x)) + x
        w, b = self._folded_qkv()                                        # [C, 3 inner], [3 inner]
        wq, bq = self._f.get([w, b, norm.weight, norm.bias], lambda: _fold_ln(w.t().contiguous(), b, norm))
        qkv = ops.linear(x, wq, bq, stats=ops.ln_stats(x, norm.eps), parts=3).view(3, L, H * W, inner)
        out = ops.agent_attention(qkv[0], qkv[1], qkv[2], self.heads, self.scale, agent_major=True)   # [L, HW, inner]
        a = self.a_linears[0]
        return ops.linear(out, a.weight, a.bias, residual=x.reshape(-1, C)).view(L, H, W, C)


    def fused_residual_ego(self, x, norm):
        """fused_residual(x, norm)[:1] without computing the other agents' rows: keys / values of every agent (a 256 -> 512
        heal_linear over all tokens), queries of the ego agent only (its H W tokens are the first rows of the agent-major
        tensor), the ego row of the per-pixel agent attention, a_linear + x on the ego tokens.  Used by V2XTEncoder for the last
        block, whose other agents' outputs nobody reads (V2XTransformer returns agent 0).  -> [1,H,W,C], or None when the shape
        is not supported by the fused kernels (the caller then computes every agent)."""
        L, H, W, C = x.shape
        inner = self.heads * self.dim_head
        if inner != 256 or not ops.linear_supported(1, C, 3 * inner):
            return None
        w, b = self._folded_qkv()                                        # [C, 3 inner], [3 inner]: (q W_att | k | v W_msg)
        wq, bq = self._f.get([w, b, norm.weight, norm.bias], lambda: _fold_ln(w.t().contiguous(), b, norm))
        if getattr(self, "_fe", None) is None:
            self._fe = _Folded()
        w_q, b_q, w_kv, b_kv = self._fe.get([wq, bq], lambda: (wq[:inner].contiguous(), bq[:inner].contiguous(),
                                                                wq[inner:].contiguous(), bq[inner:].contiguous()))
        n_pix = H * W
        stats = ops.ln_stats(x, norm.eps)
        kv = ops.linear(x, w_kv, b_kv, stats=stats, parts=2).view(2, L, n_pix, inner)
        q = torch.empty((L, n_pix, inner), dtype=torch.float32, device=x.device)     # only agent 0's rows are written and read
        ops.linear(x[0], w_q, b_q, stats=stats[:n_pix], out=q[0])
        out = ops.agent_attention(q, kv[0], kv[1], self.heads, self.scale, out_rows=1, agent_major=True)   # [1, HW, inner]
        a = self.a_linears[0]
        return ops.linear(out, a.weight, a.bias, residual=x[0].reshape(-1, C)).view(1, H, W, C)


def _relative_indices(ws):
    idx = torch.tensor([[x, y] for x in range(ws) for y in range(ws)])
    return idx[None, :, :] - idx[:, None, :] + ws - 1


class BaseWindowAttention(nn.Module):
    def __init__(self, dim, heads, dim_head, drop_out, window_size, relative_pos_embedding):
        super().__init__()
        inner = dim_head * heads
        self.heads, self.dim_head = heads, dim_head
        self.scale = dim_head ** -0.5
        self.window_size = window_size
        self.relative_pos_embedding = relative_pos_embedding
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        if relative_pos_embedding:
            self.relative_indices = _relative_indices(window_size)
            self.pos_embedding = nn.Parameter(torch.randn(2 * window_size - 1, 2 * window_size - 1))
        else:
            self.pos_embedding = nn.Parameter(torch.randn(window_size ** 2, window_size ** 2))
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(drop_out))

    def forward(self, x):
        """x [L,H,W,C] -> [L,H,W,C]: attention inside ws x ws windows, per agent and head (mswin.py:46-80)."""
        L, H, W, C = x.shape
        ws, m, d = self.window_size, self.heads, self.dim_head
        nh, nw = H // ws, W // ws
        # the reference keeps `relative_indices` as a plain (host) attribute -- not a buffer, so it is not in the state_dict;
        # position_bias() indexes with a cached device copy: a host index tensor costs an H2D copy per call and cannot be
        # captured in a HIP graph
        bias = self.position_bias(x.device)
        qkv = self.to_qkv(x)
        from heal_amd import ops
        if x.is_cuda and ops.window_attention_supported(ws, d, H, W) and not (torch.is_grad_enabled() and (x.requires_grad or self.training)):
            # one kernel per window size: Q K^T, bias, softmax and P V without materialising the window re-layouts or
            # the [windows, T, T] score tensor (heal_window_attention): 16x (ws 4), 4x (ws 8) and 1.7x (ws 16) faster
            # than the library sequence at 8 agents x 128 x 128
            return self.to_out[0](ops.window_attention(qkv, bias, m, d, ws, self.scale))
        if (x.is_cuda and _grad_path(x, self) and os.environ.get("HEAL_WATTN_GRAD", "torch") == "kernel"
                and ops.window_attention_supported(ws, d, H, W) and qkv.dtype == torch.float32):
            # opt-in (unmeasured): K6b forward + heal_window_attention_backward instead of the library composition below
            return self.to_out(ops.WindowAttention.apply(qkv, bias, m, d, ws, self.scale))
        qkv = qkv.view(L, nh, ws, nw, ws, 3, m, d)
        qkv = qkv.permute(5, 0, 6, 1, 3, 2, 4, 7).reshape(3, L * m * nh * nw, ws * ws, d)
        dots = torch.baddbmm(bias.unsqueeze(0).expand(qkv.shape[1], -1, -1), qkv[0], qkv[1].transpose(1, 2),
                             beta=1.0, alpha=self.scale)
        out = torch.bmm(dots.softmax(dim=-1), qkv[2])                    # [L*m*nh*nw, ws*ws, d]
        out = out.view(L, m, nh, nw, ws, ws, d).permute(0, 2, 4, 3, 5, 1, 6).reshape(L, H, W, m * d)
        if _grad_path(x, self):
            return self.to_out(out)     # training: Linear + Dropout (mswin.py:43-44,79)
        return self.to_out[0](out)

    def position_bias(self, device):
        if self.relative_pos_embedding:
            ri = self.__dict__.get("_ri_dev")
            if ri is None or ri[0].device != device:
                r = self.relative_indices.to(device)
                ri = (r[:, :, 0].contiguous(), r[:, :, 1].contiguous())
                self.__dict__["_ri_dev"] = ri
            pe = self.pos_embedding
            if torch.is_grad_enabled() and pe.requires_grad:
                return pe[ri[0], ri[1]]
            # inference: the [T, T] table of a parameter version is looked up once, not per forward (3 index kernels per block)
            key = (pe.data_ptr(), pe._version, str(device))
            hit = self.__dict__.get("_bias_tab")
            if hit is None or hit[0] != key:
                hit = (key, pe.detach()[ri[0], ri[1]].contiguous())
                self.__dict__["_bias_tab"] = hit
            return hit[1]
        return self.pos_embedding


class SplitAttn(nn.Module):
    def __init__(self, input_dim):
        super().__init__()
        self.input_dim = input_dim
        self.fc1 = nn.Linear(input_dim, input_dim, bias=False)
        self.bn1 = nn.LayerNorm(input_dim)
        self.act1 = nn.ReLU()
        self.fc2 = nn.Linear(input_dim, input_dim * 3, bias=False)
        self._stripe = None     # set by heal_amd.dist (row stripes of one scene on several ranks): see PyramidWindowAttention

    def forward(self, window_list):
        """split_attn.py:43-62 on [L,H,W,C] tensors."""
        sw, mw, bw = window_list
        L = sw.shape[0]
        if self._stripe is not None:
            # the global average pool is the one step of the encoder that looks beyond a window: sum of the ranks' stripe sums
            part = (sw + mw + bw).sum((1, 2))                            # [L,C] over this rank's rows
            gap = (self._stripe.all_gather(part).sum(0) / float(self._stripe.world * sw.shape[1] * sw.shape[2]))[:, None, None, :]
        else:
            gap = (sw + mw + bw).mean((1, 2), keepdim=True)              # [L,1,1,C]
        a = self.fc2(F.relu(self.bn1(self.fc1(gap))))                    # [L,1,1,3C]
        a = F.softmax(a.view(L, 1, 3, -1), dim=2).reshape(L, 1, 1, -1)   # radix softmax over the 3 windows
        c = self.input_dim
        return sw * a[..., 0:c] + mw * a[..., c:2 * c] + bw * a[..., 2 * c:]


class PyramidWindowAttention(nn.Module):
    def __init__(self, dim, heads, dim_heads, drop_out, window_size, relative_pos_embedding, fuse_method="naive"):
        super().__init__()
        assert len(dim_heads) == len(heads) == len(window_size)
        self.pwmsa = nn.ModuleList([BaseWindowAttention(dim, h, d, drop_out, w, relative_pos_embedding)
                                    for h, d, w in zip(heads, dim_heads, window_size)])
        self.fuse_mehod = fuse_method
        self._f, self._fo, self._fc = _Folded(), _Folded(), _Folded()
        # Row stripes (heal_amd/dist.py, ShardedBaselineStriped): x is rows [r Hs, (r + 1) Hs) of every agent's map, Hs a multiple
        # of the largest window.  Windows never cross a stripe, so everything here is local EXCEPT split attention's global
        # average: `_stripe.all_gather(t)` -> [world, *t.shape] is the only exchange (the per-chunk column sums, 12 KB per agent).
        self._stripe = None
        if fuse_method.startswith("split_attn"):
            self.split_attn = SplitAttn({"split_attn": 256, "split_attn128": 128, "split_attn64": 64}[fuse_method])

    def forward(self, x):
        outs = [w(x) for w in self.pwmsa]
        if self.fuse_mehod == "naive":
            return sum(outs) / len(outs)
        return self.split_attn(outs)

    def fused_residual(self, x, norm):
        """LayerNorm -> the three to_qkv projections as ONE 256 -> 3 x 768 heal_linear (three output buffers) -> three window
        attentions -> [to_out x 3, split-attention weighting (or the plain mean), + x] as ONE K = 3 x 256 heal_linear whose
        weight rows carry the per-agent branch weights."""
        L, H, W, C = x.shape
        ws = self.pwmsa
        inner = [w.heads * w.dim_head for w in ws]
        ok = len(ws) == 3 and all(i == C for i in inner) and all(
            ops.window_attention_supported(w.window_size, w.dim_head, H, W) for w in ws) and ops.linear_supported(1, C, 9 * C)
        if not ok or (self.fuse_mehod != "naive" and self.split_attn.input_dim != C):
            return self.forward(norm(x)) + x
        srcs = [w.to_qkv.weight for w in ws] + [norm.weight, norm.bias]
        wq, bq = self._f.get(srcs, lambda: _fold_ln(torch.cat([w.to_qkv.weight for w in ws], 0), None, norm))
        T = L * H * W
        qkv = ops.linear(x, wq, bq, stats=ops.ln_stats(x, norm.eps), parts=3)        # [3, T, 3 C]
        branches = torch.empty((3, T, C), dtype=torch.float32, device=x.device)
        for i, w in enumerate(ws):
            ops.window_attention(qkv[i].view(L, H, W, 3 * C), w.position_bias(x.device), w.heads, w.dim_head, w.window_size,
                                 w.scale, out=branches[i])
        wo, bo = self._fo.get([w.to_out[0].weight for w in ws] + [w.to_out[0].bias for w in ws], lambda: (
            torch.stack([w.to_out[0].weight for w in ws]).contiguous(), torch.stack([w.to_out[0].bias for w in ws]).contiguous()))
        wo_cat = self._fc.get([wo], lambda: torch.cat([wo[0], wo[1], wo[2]], 1).contiguous())   # [C, 3 C]: K runs over branches
        if self.fuse_mehod == "naive":
            scale = torch.full((L, 3, C), 1.0 / 3.0, dtype=torch.float32, device=x.device)
            bias = (bo.sum(0) / 3.0).expand(L, C).contiguous()
        else:
            sa = self.split_attn
            if self._stripe is not None:
                sums = self._stripe.all_gather(ops.split_attn_colsum(branches, L, H * W))     # [world, L, 3, chunks, C]
                scale, bias = ops.split_attn_weights_from_colsum(sums, H * W, wo, bo, sa.fc1.weight, sa.bn1.weight, sa.bn1.bias,
                                                                 sa.bn1.eps, sa.fc2.weight)
            else:
                scale, bias = ops.split_attn_weights(branches, L, H * W, wo, bo, sa.fc1.weight, sa.bn1.weight, sa.bn1.bias,
                                                     sa.bn1.eps, sa.fc2.weight)
        return ops.linear(branches, wo_cat, bias, residual=x.reshape(-1, C), colscale=scale, colscale_part=C,
                          group_rows=H * W, bias_per_group=True, x_parts=3).view(L, H, W, C)


class V2XFusionBlock(nn.Module):
    def __init__(self, num_blocks, cav_att_config, pwindow_config):
        super().__init__()
        if not cav_att_config["use_hetero"]:
            raise NotImplementedError("CavAttention (use_hetero: false) is not used by the HEAL configs")
        self.layers = nn.ModuleList([])
        for _ in range(num_blocks):
            att = HGTCavAttention(cav_att_config["dim"], heads=cav_att_config["heads"],
                                  dim_head=cav_att_config["dim_head"], dropout=cav_att_config["dropout"])
            self.layers.append(nn.ModuleList([
                PreNorm(cav_att_config["dim"], att),
                PreNorm(cav_att_config["dim"], PyramidWindowAttention(
                    pwindow_config["dim"], heads=pwindow_config["heads"], dim_heads=pwindow_config["dim_head"],
                    drop_out=pwindow_config["dropout"], window_size=pwindow_config["window_size"],
                    relative_pos_embedding=pwindow_config["relative_pos_embedding"],
                    fuse_method=pwindow_config["fusion_method"]))]))

    def forward(self, x):
        for cav_attn, pwindow_attn in self.layers:
            x = cav_attn.residual(x)
            x = pwindow_attn.residual(x)
        return x


class STTF(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.discrete_ratio = args["voxel_size"][0]
        self.downsample_rate = args["downsample_rate"]


class RelTemporalEncoding(nn.Module):
    def __init__(self, n_hid, RTE_ratio, max_len=100, dropout=0.2):
        super().__init__()
        position = torch.arange(0.0, max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, n_hid, 2) * -(math.log(10000.0) / n_hid))
        emb = nn.Embedding(max_len, n_hid)
        emb.weight.data[:, 0::2] = torch.sin(position * div_term) / math.sqrt(n_hid)
        emb.weight.data[:, 1::2] = torch.cos(position * div_term) / math.sqrt(n_hid)
        self.RTE_ratio = RTE_ratio
        self.emb = emb
        self.lin = nn.Linear(n_hid, n_hid)


class RTE(nn.Module):
    def __init__(self, dim, RTE_ratio=2):
        super().__init__()
        self.RTE_ratio = RTE_ratio
        self.emb = RelTemporalEncoding(dim, RTE_ratio=RTE_ratio)

    def forward(self, x):
        # HEAL's prior encoding is all zeros: dt = 0 for every agent (v2xvit_basic.py:165-168)
        t0 = torch.zeros((), dtype=torch.long, device=x.device)
        return x + self.emb.lin(self.emb.emb(t0 * self.emb.RTE_ratio))


class V2XTEncoder(nn.Module):
    def __init__(self, args):
        super().__init__()
        cav, pw, feed = args["cav_att_config"], args["pwindow_att_config"], args["feed_forward"]
        self.downsample_rate = args["sttf"]["downsample_rate"]
        self.discrete_ratio = args["sttf"]["voxel_size"][0]
        self.use_roi_mask = args["use_roi_mask"]
        self.use_RTE = cav["use_RTE"]
        self.RTE_ratio = cav["RTE_ratio"]
        self.sttf = STTF(args["sttf"])
        self.prior_feed = nn.Linear(cav["dim"] + 3, cav["dim"])  # present in checkpoints, unused in forward
        self.layers = nn.ModuleList([])
        if self.use_RTE:
            self.rte = RTE(cav["dim"], self.RTE_ratio)
        for _ in range(args["depth"]):
            self.layers.append(nn.ModuleList([
                V2XFusionBlock(args["num_blocks"], cav, pw),
                PreNorm(cav["dim"], FeedForward(cav["dim"], feed["mlp_dim"], dropout=feed["dropout"]))]))

    def forward(self, x):
        """x [L,H,W,C] -> [L,H,W,C]; on the fused inference path [1,H,W,C]: the ego agent's rows only (see below)."""
        if self.use_RTE:
            x = self.rte(x)
        for li, (attn, ff) in enumerate(self.layers):
            if li == len(self.layers) - 1 and x.shape[0] > 1 and _fused_ok(x, self) and os.environ.get("HEAL_V2XVIT_EGO_TAIL", "1") == "1":
                ego = self._ego_tail(attn, ff, x)
                if ego is not None:
                    return ego
            x = attn(x)
            x = ff.residual(x)
        return x

    def _ego_tail(self, block, ff, x):
        """The last block, for the rows that are read.  V2XTransformer returns agent 0 of the encoder's output, and after the last
        agent attention nothing mixes agents any more (window attention, split attention and the feed-forward are per agent):
        the other agents' rows of that attention's output, of the window attention and of the feed-forward are never read.  The
        reference computes and discards them (v2xvit_basic.py:183-191); here the last agent attention takes queries from the ego
        agent only (keys / values from everyone) and what follows runs on the ego agent's H x W tokens: the same values for the
        rows that exist (tests/test_gpu_kernels.py::test_v2xvit_ego_tail_equals_full), 1/L of the work."""
        pairs = list(block.layers)
        for cav, pw in pairs[:-1]:
            x = cav.residual(x)
            x = pw.residual(x)
        cav, pw = pairs[-1]
        if not hasattr(cav.fn, "fused_residual_ego"):
            return None
        x0 = cav.fn.fused_residual_ego(x, cav.norm)
        if x0 is None:
            return None
        x0 = pw.residual(x0)
        return ff.residual(x0)


class V2XTransformer(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.encoder = V2XTEncoder(args["encoder"])

    def forward(self, x):
        """x [L,H,W,C] (ego first) -> fused ego map [H,W,C]."""
        return self.encoder(x)[0]
