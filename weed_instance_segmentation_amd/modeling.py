"""Drop-in `Mask2FormerForUniversalSegmentation` for the reference's train / eval glue.

Boundary: the reference calls
    model(pixel_values=..., mask_labels=[...], class_labels=[...]).loss      train.py:28-33, :196-198
    model(pixel_values=...) -> .class_queries_logits, .masks_queries_logits   metrics.py:56-63, inference.py:25-30
    .to(device) .train() .eval() .parameters() .config.id2label                train.py:173-178, model_utils.py:16
    .save_pretrained(dir) / .from_pretrained(local_dir, id2label=..., label2id=..., ignore_mismatched_sizes=True)
on the class of the same name in `transformers`.  This module keeps that call contract and that
class's parameter names (so checkpoints round-trip), and routes the four hot operations to the
hand-written HIP kernels in libwm2f.so through `ops`:
    K1 ops.ms_deform_attn[_fused]   (HF:798-837, :983-1002)
    K2 ops.masked_xattn             (HF:1644-1650, :1912-1914)
    K3 ops.mask_einsum + ops.attn_mask_build (HF:2046-2054)
    K4 ops.matcher_cost             (HF:444-472)            -- in loss.py
There is no CPU path: tensors must live on an MI355X.  Everything else (backbone convolutions,
1x1 convs, GroupNorm, LayerNorm, Linear) is stock PyTorch-ROCm.

Internal layout is batch-first (B, tokens, C) everywhere; the dependency's sequence-first layout
(HF:2099-2101) is not reproduced because nothing at the boundary exposes it.
"""
from __future__ import annotations

import json
import math
import os
from typing import Sequence

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .backbone_resnet import build_backbone
from .configuration import Mask2FormerConfig
from .loss import Mask2FormerLoss


class Mask2FormerForUniversalSegmentationOutput:
    """Attribute- and key-addressable result, same field names as HF:196-241."""

    _fields = ("loss", "class_queries_logits", "masks_queries_logits", "auxiliary_logits", "encoder_last_hidden_state",
               "pixel_decoder_last_hidden_state", "transformer_decoder_last_hidden_state", "encoder_hidden_states",
               "pixel_decoder_hidden_states", "transformer_decoder_hidden_states", "attentions")

    def __init__(self, **kw):
        for f in self._fields:
            setattr(self, f, kw.get(f))
        self.loss_dict = kw.get("loss_dict")
        self.matched_indices = kw.get("matched_indices")

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return self.to_tuple()[k]

    def keys(self):
        return [f for f in self._fields if getattr(self, f) is not None]

    def to_tuple(self):
        return tuple(getattr(self, f) for f in self.keys())


_POS_CACHE: dict = {}
# K1 inference layout.  The merged offsets | logits projection of MSDeformAttn runs on the hand-written
# token GEMM (csrc/token_gemm.hip), whose epilogue writes the rows HEAD-major, (heads, B, S, 36), and K1 walks its tiles in
# SLAB order (heads outermost: an XCD's workgroups share one (image, head) slab of `value` in its L2).  Together: HBM
# traffic of a K1 launch 1.29 x -> 1.005 x its algorithmic bytes, 156 -> 138 us in the model.  Either one alone gains
# nothing (head-major rows under the old order: round 2; slab order on token-major rows re-fetches every 1152-byte row
# once per head).  Plain module attributes -- tools/ flip them for A/B runs; the package reads no environment.
SDPA_SELF_ATTENTION = True  # inference: the 100 x 100 self-attention through the stock fused attention op (-0.28 ms per forward; tools/probes/sdpa_self_attention_probe.py)
HEAD_MAJOR_ROWS = True
HEAD_MAJOR_VALUE = False  # value (heads, B, S, 32) from the token GEMM too: A/B only (tools/k1_slab_inmodel.py)


def sine_position_embedding(H: int, W: int, num_pos_feats: int, device, dtype=torch.float32, temperature=10000):
    """HF:864-904 (normalize=True, scale=2*pi, no mask) for ONE image: (num_pos_feats*2, H, W).
    Identical for every image of a batch, so it is built once per (shape, device) and cached."""
    key = (H, W, num_pos_feats, str(device), dtype)
    if key not in _POS_CACHE:
        scale, eps = 2 * math.pi, 1e-6
        y = torch.arange(1, H + 1, dtype=dtype, device=device)[:, None].expand(H, W)
        x = torch.arange(1, W + 1, dtype=dtype, device=device)[None, :].expand(H, W)
        y = y / (y[-1:, :] + eps) * scale
        x = x / (x[:, -1:] + eps) * scale
        dim_t = torch.arange(num_pos_feats, dtype=torch.int64, device=device).to(dtype)
        dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
        px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
        px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=3).flatten(2)
        py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=3).flatten(2)
        _POS_CACHE[key] = torch.cat((py, px), dim=2).permute(2, 0, 1).contiguous()
    return _POS_CACHE[key]


# ------------------------------------------------------------------------------ pixel decoder
class MSDeformAttn(nn.Module):
    """Mask2FormerPixelDecoderEncoderMultiscaleDeformableAttention, HF:919-1014."""

    def __init__(self, embed_dim: int, num_heads: int, n_levels: int = 3, n_points: int = 4):
        super().__init__()
        if embed_dim % num_heads:
            raise ValueError(f"embed_dim {embed_dim} not divisible by num_heads {num_heads}")
        self.d_model, self.n_heads, self.n_levels, self.n_points = embed_dim, num_heads, n_levels, n_points
        self.sampling_offsets = nn.Linear(embed_dim, num_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(embed_dim, num_heads * n_levels * n_points)
        self.value_proj = nn.Linear(embed_dim, embed_dim)
        self.output_proj = nn.Linear(embed_dim, embed_dim)
        self._cat: dict = {}

    def _offsets_logits_weight(self, lanes: bool = False):
        """[sampling_offsets ; attention_weights] as ONE (288, 256) projection for the inference path:
        two skinny GEMMs (N = 192 and N = 96, ~15 % of the fp32 matrix peak each) become one.
        lanes=True: the same rows in the kernel's record order (ops.k1_lane_order, wm2f_msdeform_fused_lanes_fwd) -- a row
        permutation of the weight, nothing else."""
        so, aw = self.sampling_offsets, self.attention_weights
        key = (so.weight._version, so.bias._version, aw.weight._version, aw.bias._version, so.weight.device)
        if self._cat.get("key") != key:
            with torch.no_grad():
                w, b = torch.cat([so.weight, aw.weight], 0).contiguous(), torch.cat([so.bias, aw.bias], 0).contiguous()
                self._cat = dict(key=key, w=w, b=b)
                H, L, P = self.n_heads, self.n_levels, self.n_points
                if L == 3 and P == 4:
                    idx = ops.k1_lane_order(H).to(w.device)
                    self._cat.update(w_lanes=w[idx].contiguous(), b_lanes=b[idx].contiguous())
        if lanes:
            return self._cat["w_lanes"], self._cat["b_lanes"]
        return self._cat["w"], self._cat["b"]

    def forward(self, hidden, pos, ref, level_hw, hp=None, hidden_lp=None):
        """hidden (B,S,C); pos (S,C) shared by the batch; ref (S,L,2); hp = hidden + pos if already known (training: possibly in
        bf16, as the fused LayerNorm of the previous layer wrote it); hidden_lp = hidden in bf16 if already known."""
        B, S, C = hidden.shape
        H, L, P = self.n_heads, self.n_levels, self.n_points
        if hp is None:
            hp = hidden + pos[None]
        if torch.is_grad_enabled() and (hidden.requires_grad or self.value_proj.weight.requires_grad):
            # (ops.linear_tokens = F.linear; under bf16 autocast its weight gradient runs on wm2f_token_wgrad_bf16)
            lin = ops.linear_tokens
            value = lin(hidden if hidden_lp is None else hidden_lp, self.value_proj.weight, self.value_proj.bias).view(B, S, H, C // H)
            # sampling_offsets and attention_weights as ONE projection of hp (one cast, one GEMM, one weight-gradient pass
            # instead of two); the concatenation is tracked by autograd, so each Linear's parameters get their rows of dW
            so, aw_ = self.sampling_offsets, self.attention_weights
            ol = lin(hp, torch.cat([so.weight, aw_.weight], 0), torch.cat([so.bias, aw_.bias], 0))
            if ops.k1_rows_applies(value, ol, level_hw, H, P):
                # the encoder's own shape: K1 on the rows themselves -- softmax, location arithmetic and their backward inside the
                # kernels (ops.ms_deform_attn_rows; `ref` is the pixel-centre grid of reference_points(), which they rebuild)
                out = ops.ms_deform_attn_rows(value, level_hw, ol, H)
                return lin(out, self.output_proj.weight, self.output_proj.bias)
            n_off = H * L * P * 2
            off = ol[..., :n_off].view(B, S, H, L, P, 2)
            logits = ol[..., n_off:].view(B, S, H, L * P)
            norm = torch.tensor([[w, h] for h, w in level_hw], dtype=hidden.dtype, device=hidden.device)
            loc = ref[None, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
            aw = torch.softmax(logits, -1).view(B, S, H, L, P)
            out = ops.ms_deform_attn(value, level_hw, loc, aw)
            return lin(out, self.output_proj.weight, self.output_proj.bias)
        elif ops.k1_lanes_applies(level_hw, S, C // H, P, B, H) and hp.dtype == torch.float32:
            # inference, the encoder's own shape: one merged projection with its rows in the kernel's lane order
            w, b = self._offsets_logits_weight(lanes=True)
            # (under autocast the projection stays a bf16 library GEMM, as the dependency's Linear is there)
            hm_rows = HEAD_MAJOR_ROWS and not torch.is_autocast_enabled("cuda") and ops.token_linear_applies(hp, w)
            hm_value = HEAD_MAJOR_VALUE and hm_rows and ops.token_linear_applies(hidden, self.value_proj.weight)
            if hm_value:
                value = ops.token_linear(hidden, self.value_proj.weight, self.value_proj.bias, out_group=C // H)
            else:
                value = self.value_proj(hidden).view(B, S, H, C // H)
            rows = ops.token_linear(hp, w, b, out_group=36) if hm_rows else F.linear(hp, w, b)
            out = ops.ms_deform_attn_fused_lanes(value, level_hw, rows, H, head_major=hm_rows, value_head_major=hm_value,
                                                 slab_order=hm_rows)
        else:  # inference: one merged projection; softmax + location arithmetic fused into the kernel
            value = self.value_proj(hidden).view(B, S, H, C // H)
            w, b = self._offsets_logits_weight()
            ol = F.linear(hp, w, b)  # (B, S, 288): [offsets (H*L*P*2) | logits (H*L*P)] per token
            out = ops.ms_deform_attn_fused_packed(value, level_hw, ol, ref, H, L, P)
        return self.output_proj(out)


class PixelDecoderEncoderLayer(nn.Module):
    """HF:1017-1103 (post-norm; dropout is 0.0 in every published configuration)."""

    def __init__(self, config: Mask2FormerConfig):
        super().__init__()
        d = config.feature_size
        self.self_attn = MSDeformAttn(d, config.num_attention_heads, 3, 4)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.dropout = config.dropout
        self.fc1 = nn.Linear(d, config.encoder_feedforward_dim)
        self.fc2 = nn.Linear(config.encoder_feedforward_dim, d)
        self.final_layer_norm = nn.LayerNorm(d)

    def forward(self, hidden, pos, ref, level_hw, hp=None, emit_next=True):
        """Returns (hidden, next) where next = hidden + pos (inference), (hidden + pos, hidden in bf16 or None) (training, fused
        LayerNorm route) or None.  pos is (S, C), shared by the batch; emit_next=False for the last layer."""
        hidden_lp = None
        if isinstance(hp, tuple):
            hp, hidden_lp = hp
        if (not torch.is_grad_enabled() and self.dropout == 0.0 and hidden.shape[-1] == 256
                and hidden.dtype == torch.float32 and not torch.is_autocast_enabled("cuda")):
            # inference: residual + LayerNorm fused (and the next layer's hidden + pos with the second one);
            # bias + ReLU in the fc1 GEMM epilogue
            B_, S_, C_ = hidden.shape
            ln1, ln2 = self.self_attn_layer_norm, self.final_layer_norm
            a = self.self_attn(hidden, pos, ref, level_hw, hp)
            hidden = ops.add_layernorm(a, hidden, ln1.weight, ln1.bias, ln1.eps)
            f = torch._addmm_activation(self.fc1.bias, hidden.reshape(B_ * S_, C_), self.fc1.weight.t(), use_gelu=False)
            f = self.fc2(f).view(B_, S_, C_)
            return ops.add_layernorm(f, hidden, ln2.weight, ln2.bias, ln2.eps, pos=pos)
        amp = torch.is_autocast_enabled("cuda")
        bf16 = amp and torch.get_autocast_dtype("cuda") == torch.bfloat16
        if (torch.is_grad_enabled() and self.dropout == 0.0 and hidden.is_cuda and hidden.shape[-1] == 256
                and hidden.dtype == torch.float32 and (bf16 or not amp)):
            # training: residual add + LayerNorm as ONE pass that also writes what the next Linears read (the bf16 copy under
            # autocast, and the next layer's hidden + pos), with ONE backward pass (ops.add_layernorm_train)
            ln1, ln2 = self.self_attn_layer_norm, self.final_layer_norm
            a = self.self_attn(hidden, pos, ref, level_hw, hp, hidden_lp)
            hidden, h_lp, _ = ops.add_layernorm_train(a, hidden, ln1.weight, ln1.bias, ln1.eps, want_bf16=bf16)
            f = F.relu(ops.linear_tokens(hidden if h_lp is None else h_lp, self.fc1.weight, self.fc1.bias))
            f = ops.linear_tokens(f, self.fc2.weight, self.fc2.bias)
            # HF:1090-1093 clamps the layer's output to finfo.max - 1000 `if` it holds an inf / NaN -- a host synchronisation per
            # layer.  On finite values that clamp is the identity and a NaN passes through it, so applying it ALWAYS (inside the
            # kernel's store) is the same function without the wait.
            cv = (torch.finfo(torch.float32).max - 1000) if self.training else 0.0
            hidden, h_lp, hp_next = ops.add_layernorm_train(f, hidden, ln2.weight, ln2.bias, ln2.eps, pos=pos if emit_next else None,
                                                             want_bf16=bf16 and emit_next, pos_bf16=bf16, clamp=cv)
            return hidden, ((hp_next, h_lp) if emit_next else None)
        else:
            a = F.dropout(self.self_attn(hidden, pos, ref, level_hw, hp, hidden_lp), self.dropout, self.training)
            hidden = self.self_attn_layer_norm(hidden + a)
            f = F.dropout(F.relu(ops.linear_tokens(hidden, self.fc1.weight, self.fc1.bias)), self.dropout, self.training)
            f = F.dropout(ops.linear_tokens(f, self.fc2.weight, self.fc2.bias), self.dropout, self.training)
            hidden = self.final_layer_norm(hidden + f)
            nxt = None
        if self.training and not torch.isfinite(hidden).all():  # HF:1090-1093
            cv = torch.finfo(hidden.dtype).max - 1000
            hidden = torch.clamp(hidden, min=-cv, max=cv)
            nxt = None  # (the by-products were written from the unclamped values)
        return hidden, nxt


class PixelDecoderEncoderOnly(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layers = nn.ModuleList([PixelDecoderEncoderLayer(config) for _ in range(config.encoder_layers)])

    @staticmethod
    def reference_points(level_hw, device, dtype=torch.float32):
        """HF:1127-1156 with valid ratios of 1 (HF:1343-1345 builds all-False padding masks): (S, L, 2)."""
        pts = []
        for h, w in level_hw:
            ry, rx = torch.meshgrid(torch.linspace(0.5, h - 0.5, h, dtype=dtype, device=device),
                                    torch.linspace(0.5, w - 0.5, w, dtype=dtype, device=device), indexing="ij")
            pts.append(torch.stack((rx.reshape(-1) / w, ry.reshape(-1) / h), -1))
        ref = torch.cat(pts, 0)
        return ref[:, None, :].expand(-1, len(level_hw), -1).contiguous()

    def forward(self, hidden, pos, level_hw):
        ref = self.reference_points(level_hw, hidden.device, hidden.dtype)
        hp = None
        for i, layer in enumerate(self.layers):
            hidden, hp = layer(hidden, pos, ref, level_hw, hp, emit_next=i + 1 < len(self.layers))
        return hidden


class Mask2FormerPixelDecoder(nn.Module):
    """HF:1236-1419."""

    def __init__(self, config: Mask2FormerConfig, feature_channels: Sequence[int]):
        super().__init__()
        self.config = config
        d, md = config.feature_size, config.mask_feature_size
        self.num_feature_levels = 3
        tin = list(feature_channels[-3:])
        self.level_embed = nn.Parameter(torch.zeros(3, d))
        self.input_projections = nn.ModuleList(
            [nn.Sequential(nn.Conv2d(c, d, kernel_size=1), nn.GroupNorm(32, d)) for c in tin[::-1]])
        self.encoder = PixelDecoderEncoderOnly(config)
        self.mask_projection = nn.Conv2d(d, md, kernel_size=1)
        stride = min(config.feature_strides[-3:])
        self.num_fpn_levels = int(np.log2(stride) - np.log2(config.common_stride))
        for idx, c in enumerate(feature_channels[: self.num_fpn_levels]):
            self.add_module(f"adapter_{idx + 1}", nn.Sequential(nn.Conv2d(c, d, kernel_size=1, bias=False), nn.GroupNorm(32, d)))
            self.add_module(f"layer_{idx + 1}", nn.Sequential(nn.Conv2d(d, d, kernel_size=3, padding=1, bias=False),
                                                              nn.GroupNorm(32, d), nn.ReLU()))

    def forward(self, features: Sequence[torch.Tensor]):
        d = self.config.feature_size
        levels = features[::-1][:3]
        level_hw = [(int(x.shape[2]), int(x.shape[3])) for x in levels]
        B = levels[0].shape[0]
        fast = (not torch.is_grad_enabled() and levels[0].is_cuda and all(x.dtype == torch.float32 for x in levels)
                and not torch.is_autocast_enabled("cuda"))
        poss = []
        for lvl, x in enumerate(levels):
            pe = sine_position_embedding(x.shape[2], x.shape[3], d // 2, x.device, x.dtype)
            poss.append(pe.flatten(1).transpose(0, 1) + self.level_embed[lvl][None, :])  # (HW, C)
        if fast and all((h * w) % 4 == 0 for h, w in level_hw):
            # 1x1 convolution, then bias + GroupNorm + transpose written straight into the token buffer (one pass for
            # the statistics, one for the tokens) instead of bias add, GroupNorm, strided flatten and concatenation
            hidden = torch.empty(B, sum(h * w for h, w in level_hw), d, device=levels[0].device, dtype=torch.float32)
            start = 0
            for lvl, x in enumerate(levels):
                conv, gn = self.input_projections[lvl]
                raw = F.conv2d(x, conv.weight, None, conv.stride, conv.padding)
                ops.group_norm_tokens_(raw, conv.bias, gn.num_groups, gn.weight, gn.bias, gn.eps, hidden, start)
                start += level_hw[lvl][0] * level_hw[lvl][1]
        else:
            embeds = [self.input_projections[lvl](x) for lvl, x in enumerate(levels)]
            hidden = torch.cat([e.flatten(2).transpose(1, 2) for e in embeds], 1)
        pos = torch.cat(poss, 0).contiguous()  # (S, C): identical for every image of the batch
        hidden = self.encoder(hidden, pos, level_hw)
        outs, tokens, start = [], [], 0
        for h, w in level_hw:
            tokens.append(hidden[:, start:start + h * w])  # (B, hw, C) views: what the transformer decoder consumes
            if fast:  # tiled transpose instead of a generic strided copy (0.9 ms -> <0.1 ms for the finest level)
                outs.append(ops.tokens_to_nchw(hidden, start, h, w))
            else:
                outs.append(hidden[:, start:start + h * w].transpose(1, 2).reshape(B, d, h, w))
            start += h * w
        self._last_tokens = tokens
        n = self.num_fpn_levels
        for idx, feat in enumerate(features[:n][::-1]):  # HF:1395-1405
            k = n - idx
            adapter, layer = getattr(self, f"adapter_{k}"), getattr(self, f"layer_{k}")
            if fast and feat.shape[-1] % 4 == 0 and feat.dtype == torch.float32:
                # GroupNorm + upsample-add and GroupNorm + ReLU as one statistics pass and one fused pass each
                gn_a, gn_l = adapter[1], layer[1]
                out = ops.group_norm_act_(adapter[0](feat), gn_a.num_groups, gn_a.weight, gn_a.bias, gn_a.eps,
                                          up=outs[-1].contiguous())
                outs.append(ops.group_norm_act_(layer[0](out), gn_l.num_groups, gn_l.weight, gn_l.bias, gn_l.eps, relu=True))
                continue
            lat = adapter(feat)
            out = lat + F.interpolate(outs[-1], size=lat.shape[-2:], mode="bilinear", align_corners=False)
            outs.append(layer(out))
        return self.mask_projection(outs[-1]), outs[:3]


class Mask2FormerPixelLevelModule(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.encoder = build_backbone(config.backbone_config)
        self.decoder = Mask2FormerPixelDecoder(config, self.encoder.channels)

    def forward(self, pixel_values):
        if self.training and torch.is_grad_enabled() and pixel_values.is_cuda and pixel_values.dim() == 4:
            # Training: hand the backbone a channels_last batch.  MIOpen's implicit-GEMM convolutions (forward, data- and
            # weight-gradient) are NHWC kernels; with NCHW activations every one of them is wrapped in batched transposes
            # (190 launches, 9.5 ms of a config-2 step).  Only the ACTIVATIONS change format -- parameters, optimiser state and
            # the state_dict stay as they are; every stock op of the backbone (BatchNorm, ReLU, max-pool, add) keeps the
            # format it is given.  Measured: 214.7 -> 200.1 ms per config-2 step (tools/probes/train_channels_last.py).
            # Inference stays NCHW: its fused bias / ReLU / GroupNorm passes are NCHW kernels.
            pixel_values = pixel_values.contiguous(memory_format=torch.channels_last)
        feats = self.encoder(pixel_values)
        mask_features, multi_scale = self.decoder(feats)
        return feats, mask_features, multi_scale

    def last_tokens(self):
        """Token-layout (B, hw, C) views of the three multi-scale maps of the last forward."""
        return self.decoder._last_tokens


# ------------------------------------------------------------------------------ transformer decoder
class SelfAttention(nn.Module):
    """Mask2FormerAttention, HF:1451-1584 (100 x 100 dense attention; stock ops, out of kernel scope)."""

    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.scaling = self.head_dim ** -0.5
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)

    def _qk_weight(self):
        """Inference: [q_proj * scaling ; k_proj] as ONE (2E, E) projection of h + qpos (two GEMM launches and the scaling pass
        become one GEMM); cached per parameter version."""
        qp, kp = self.q_proj, self.k_proj
        key = (qp.weight._version, qp.bias._version, kp.weight._version, kp.bias._version, qp.weight.device, qp.weight.dtype)
        c = self.__dict__.setdefault("_qk", {})
        if c.get("key") != key:
            with torch.no_grad():
                c.update(key=key, w=torch.cat([qp.weight * self.scaling, kp.weight], 0).contiguous(),
                         b=torch.cat([qp.bias * self.scaling, kp.bias], 0).contiguous())
        return c["w"], c["b"]

    def forward(self, h, qpos):
        B, Q, E = h.shape
        hq = h + qpos
        sh = lambda t: t.view(B, Q, self.num_heads, self.head_dim).transpose(1, 2)
        if not torch.is_grad_enabled() and not torch.is_autocast_enabled("cuda"):
            w, bqk = self._qk_weight()
            qk = F.linear(hq, w, bqk)
            q, k, v = sh(qk[..., :E]), sh(qk[..., E:]), sh(self.v_proj(h))
            if SDPA_SELF_ATTENTION:  # (A/B switch for tools/: the fused attention op of the stock library)
                a = F.scaled_dot_product_attention(q, k, v, scale=1.0)
                return self.out_proj(a.transpose(1, 2).reshape(B, Q, E))
        else:
            q, k, v = sh(self.q_proj(hq) * self.scaling), sh(self.k_proj(hq)), sh(self.v_proj(h))
        a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)), -1)
        return self.out_proj(torch.matmul(a, v).transpose(1, 2).reshape(B, Q, E))


class _OutProj(nn.Linear):
    pass


class MaskedCrossAttention(nn.Module):
    """nn.MultiheadAttention as used at HF:1618, :1644-1650, with the same parameter names
    (in_proj_weight, in_proj_bias, out_proj.*); the attention itself is kernel K2."""

    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = _OutProj(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)

    def project_kv(self, key_in, value_in):
        E = self.embed_dim
        # (up to B x 16384 tokens per level: in training the weight gradients of these two run on wm2f_token_wgrad_*)
        k = ops.linear_tokens(key_in, self.in_proj_weight[E:2 * E], self.in_proj_bias[E:2 * E])
        v = ops.linear_tokens(value_in, self.in_proj_weight[2 * E:], self.in_proj_bias[2 * E:])
        return k, v

    def _q_weight(self):
        """Inference: the query projection with 1 / sqrt(head_dim) folded in (one launch less per layer); cached per version."""
        E = self.embed_dim
        key = (self.in_proj_weight._version, self.in_proj_bias._version, self.in_proj_weight.device, self.in_proj_weight.dtype)
        c = self.__dict__.setdefault("_qw", {})
        if c.get("key") != key:
            with torch.no_grad():
                sc = 1.0 / math.sqrt(self.head_dim)
                c.update(key=key, w=(self.in_proj_weight[:E] * sc).contiguous(), b=(self.in_proj_bias[:E] * sc).contiguous())
        return c["w"], c["b"]

    def forward(self, query_in, k, v, mask, row_open):
        E = self.embed_dim
        if not torch.is_grad_enabled() and not torch.is_autocast_enabled("cuda"):
            q = F.linear(query_in, *self._q_weight())
        else:
            q = F.linear(query_in, self.in_proj_weight[:E], self.in_proj_bias[:E]) * (1.0 / math.sqrt(self.head_dim))
        ctx = ops.masked_xattn(q, k, v, mask, row_open, self.num_heads)
        return self.out_proj(ctx)


class MaskedAttentionDecoderLayer(nn.Module):
    """HF:1587-1797."""

    def __init__(self, config):
        super().__init__()
        d = config.hidden_dim
        self.pre_norm, self.dropout = config.pre_norm, config.dropout
        if config.activation_function != "relu":
            raise NotImplementedError("activation_function other than relu")
        self.self_attn = SelfAttention(d, config.num_attention_heads)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.cross_attn = MaskedCrossAttention(d, config.num_attention_heads)
        self.cross_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, config.dim_feedforward)
        self.fc2 = nn.Linear(config.dim_feedforward, d)
        self.final_layer_norm = nn.LayerNorm(d)

    def forward(self, h, qpos, k, v, mask, row_open):
        drop = lambda t: F.dropout(t, self.dropout, self.training)
        if (not self.pre_norm and not torch.is_grad_enabled() and h.is_cuda and h.dtype == torch.float32 and h.shape[-1] == 256
                and not torch.is_autocast_enabled("cuda") and (not self.training or self.dropout == 0.0)):
            # inference, forward_post (HF:1636-1690): the three residual + LayerNorm pairs as ONE pass each and the ReLU in fc1's
            # GEMM epilogue -- on (B, 100, 256) tensors every stock op is a launch-latency-sized kernel (36 fewer per forward)
            ln1, ln2, ln3 = self.cross_attn_layer_norm, self.self_attn_layer_norm, self.final_layer_norm
            h = ops.add_layernorm(self.cross_attn(h + qpos, k, v, mask, row_open), h, ln1.weight, ln1.bias, ln1.eps)
            h = ops.add_layernorm(self.self_attn(h, qpos), h, ln2.weight, ln2.bias, ln2.eps)
            B_, Q_, C_ = h.shape
            f = torch._addmm_activation(self.fc1.bias, h.reshape(B_ * Q_, C_), self.fc1.weight.t(), use_gelu=False)
            return ops.add_layernorm(self.fc2(f).view(B_, Q_, C_), h, ln3.weight, ln3.bias, ln3.eps)
        if not self.pre_norm:  # forward_post, HF:1636-1690
            h = self.cross_attn_layer_norm(h + drop(self.cross_attn(h + qpos, k, v, mask, row_open)))
            h = self.self_attn_layer_norm(h + drop(self.self_attn(h, qpos)))
            f = drop(self.fc2(drop(F.relu(self.fc1(h)))))
            return self.final_layer_norm(h + f)
        x = self.cross_attn_layer_norm(h)  # forward_pre, HF:1692-1750
        h = h + drop(self.cross_attn(x + qpos, k, v, mask, row_open))
        h = h + drop(self.self_attn(self.self_attn_layer_norm(h), qpos))
        x = self.final_layer_norm(h)
        return h + drop(self.fc2(drop(F.relu(self.fc1(x)))))


def _mlp_head(din, dh, dout, n=3):
    """Mask2FormerMLPPredictionHead (HF:1979-2015): names '<i>.0.weight'."""
    dims = [din] + [dh] * (n - 1) + [dout]
    return nn.Sequential(*[nn.Sequential(nn.Linear(a, b), nn.ReLU() if i < n - 1 else nn.Identity())
                           for i, (a, b) in enumerate(zip(dims[:-1], dims[1:]))])


class MaskPredictor(nn.Module):
    """HF:2018-2056: MLP, K3 einsum, attention-mask build (un-replicated bytes + row flags)."""

    def __init__(self, hidden, heads, mask_feature_size):
        super().__init__()
        self.mask_embedder = _mlp_head(hidden, hidden, mask_feature_size)

    def forward(self, h_norm, mask_features, next_size, pix_t=None):
        """pix_t: the pixel-major bf16 copy of mask_features (bf16 autocast) -> K3 on the bf16 matrix cores.
        next_size None: the last prediction -- no layer is left to mask (the dependency builds that mask and drops it)."""
        if pix_t is not None:
            logits = ops.mask_einsum_bf16(self.mask_embedder(h_norm), mask_features, pix_t)
        else:
            logits = ops.mask_einsum(self.mask_embedder(h_norm), mask_features)
        if next_size is None:
            return logits, None, None
        mask, row_open = ops.attn_mask_build(logits, next_size)
        return logits, mask, row_open

    def attention_mask_only(self, h_norm, pix_level, size):
        """The attention mask of the next layer WITHOUT the full-resolution logits (inference, when no caller sees the
        intermediate predictions).  HF:2046-2054 computes einsum(E, P) at the mask-feature resolution and resizes it
        bilinearly to the level's size; both are linear, so resize(einsum(E, P)) == einsum(E, resize(P)): `pix_level` is
        the mask-feature map resized ONCE per forward to this level's size, the einsum runs on 1/64 ... 1/4 of the
        pixels, and its epilogue thresholds the accumulators straight into the mask bytes and the row flags -- the
        logits are never written.  The values differ from the full-resolution route by fp32 rounding only."""
        return ops.mask_einsum_attn_mask(self.mask_embedder(h_norm), pix_level, tag=f"hw{size[0] * size[1]}")


class MaskedAttentionDecoder(nn.Module):
    """HF:1801-1960."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layerdrop = config.dropout
        self.layers = nn.ModuleList([MaskedAttentionDecoderLayer(config) for _ in range(config.decoder_layers - 1)])
        self.layernorm = nn.LayerNorm(config.hidden_dim)
        self.mask_predictor = MaskPredictor(config.hidden_dim, config.num_attention_heads, config.mask_feature_size)
        # switch for A/B tests of the low-resolution attention-mask route (WM2F_LOW_RES_MASKS=0 turns it off at construction)
        self.low_res_masks = True  # False: every intermediate prediction at full resolution (A/B runs and tests)
        # tests: set to a list to receive every layer's attention-mask bytes (B, Q, HW_l) in order
        self.record_attention_masks: list | None = None

    def forward(self, h, qpos, feats, poss, mask_features, sizes, need_all_logits=True):
        """need_all_logits=False (inference without auxiliary outputs): only the LAST prediction is computed at the
        mask-feature resolution; the earlier ones only feed the next layer's attention mask and run at that level's
        resolution (MaskPredictor.attention_mask_only).  `all_logits[:-1]` is then None."""
        pix_t = None
        if (mask_features.dtype == torch.bfloat16 and mask_features.is_cuda and mask_features.shape[1] % 32 == 0
                and mask_features.shape[1] <= 512):
            # bf16 autocast: K3 runs on the bf16 matrix cores from a pixel-major copy made once for its 10 calls
            pix_t = ops.nchw_to_pixel_major_bf16(mask_features.contiguous())
        elif mask_features.dtype != torch.float32:
            mask_features = mask_features.float()  # once, not once per mask-predictor call
        low = (not need_all_logits and self.low_res_masks and not torch.is_grad_enabled() and pix_t is None
               and mask_features.is_cuda and mask_features.dtype == torch.float32 and not torch.is_autocast_enabled("cuda")
               and all(ww % 4 == 0 for hh, ww in sizes))
        pix_level = None
        if low:
            mf = mask_features.contiguous()
            Hm, Wm = int(mf.shape[2]), int(mf.shape[3])
            if Hm % 8 == 0 and Wm % 8 == 0 and list(sizes) == [(Hm >> 3, Wm >> 3), (Hm >> 2, Wm >> 2), (Hm >> 1, Wm >> 1)]:
                pix_level = ops.resize_pyramid(mf)[::-1]  # the usual strides 32 / 16 / 8 against 4: one pass for all three
            else:
                pix_level = [ops.resize_bilinear(mf, s) for s in sizes]
        inter = [self.layernorm(h)]
        if low:
            logits = None
            mask, row_open = self.mask_predictor.attention_mask_only(inter[0], pix_level[0], sizes[0])
        else:
            logits, mask, row_open = self.mask_predictor(inter[0], mask_features, sizes[0], pix_t)
        if self.record_attention_masks is not None:
            self.record_attention_masks.append(mask)
        all_logits = [logits]
        keys_in = [None, None, None]  # feats[lvl] + poss[lvl]: the same for the three layers that attend to a level
        for idx, layer in enumerate(self.layers):
            if self.training and self.layerdrop > 0 and float(torch.rand([])) < self.layerdrop:  # HF:1905-1908
                continue
            lvl = idx % 3
            if keys_in[lvl] is None:
                f, pe = feats[lvl], poss[lvl]
                if (not torch.is_grad_enabled() and f.is_cuda and f.dtype == torch.float32 and pe.dtype == torch.float32
                        and pe.shape[0] == 1 and pe.shape[1:] == f.shape[1:] and (f[0].numel() % 4) == 0 and f.is_contiguous()):
                    keys_in[lvl] = ops.add_broadcast(f, pe.contiguous())  # (the stock broadcast add runs at 1.1 TB/s)
                else:
                    keys_in[lvl] = f + pe
            k, v = layer.cross_attn.project_kv(keys_in[lvl], feats[lvl])
            h = layer(h, qpos, k, v, mask, row_open)
            inter.append(self.layernorm(h))
            nxt = (idx + 1) % 3
            if low and idx + 1 < len(self.layers):
                logits = None
                mask, row_open = self.mask_predictor.attention_mask_only(inter[-1], pix_level[nxt], sizes[nxt])
            elif low:  # the prediction that is returned: full resolution, and no layer left to mask
                logits = ops.mask_einsum(self.mask_predictor.mask_embedder(inter[-1]), mask_features)
            else:
                logits, mask, row_open = self.mask_predictor(inter[-1], mask_features,
                                                             sizes[nxt] if idx + 1 < len(self.layers) else None, pix_t)
            if self.record_attention_masks is not None and idx + 1 < len(self.layers):
                self.record_attention_masks.append(mask)
            all_logits.append(logits)
        return h, inter, all_logits


class Mask2FormerTransformerModule(nn.Module):
    """HF:2059-2129."""

    def __init__(self, in_features, config):
        super().__init__()
        d = config.hidden_dim
        self.config = config
        self.queries_embedder = nn.Embedding(config.num_queries, d)
        self.queries_features = nn.Embedding(config.num_queries, d)
        # The dependency keeps these projections in a plain Python list (HF:2073-2079): they are NOT
        # registered, never saved and never trained.  With in_features == hidden_dim they are identity.
        if in_features != d or config.enforce_input_projection:
            raise NotImplementedError("feature_size != hidden_dim / enforce_input_projection: the dependency's "
                                      "unregistered input projections are not reproduced")
        self.decoder = MaskedAttentionDecoder(config)
        self.level_embed = nn.Embedding(3, d)

    def forward(self, multi_scale, mask_features, tokens=None, need_all_logits=True):
        """tokens: optional (B, hw, C) token-layout views of `multi_scale` (saves re-flattening the NCHW maps)."""
        d = self.config.hidden_dim
        B = mask_features.shape[0]
        feats, poss, sizes = [], [], []
        for i in range(3):
            f = multi_scale[i]
            sizes.append((int(f.shape[2]), int(f.shape[3])))
            pe = sine_position_embedding(f.shape[2], f.shape[3], d // 2, f.device, f.dtype)
            poss.append(pe.flatten(1).transpose(0, 1)[None])  # (1, HW, C)
            tok = tokens[i] if tokens is not None else f.flatten(2).transpose(1, 2)
            feats.append(tok + self.level_embed.weight[i][None, None, :])  # (B, HW, C)
        qpos = self.queries_embedder.weight[None].expand(B, -1, -1)
        h = self.queries_features.weight[None].expand(B, -1, -1)
        return self.decoder(h, qpos, feats, poss, mask_features, sizes, need_all_logits=need_all_logits)


class Mask2FormerModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.pixel_level_module = Mask2FormerPixelLevelModule(config)
        self.transformer_module = Mask2FormerTransformerModule(config.feature_size, config)


# ------------------------------------------------------------------------------ top level
_LEGACY_SWIN = (  # transformers 4.x checkpoint names -> 5.x (conversion_mapping.py:411-426 of 5.15.0), in order
    ("attention.self.query", "attention.q_proj"), ("attention.self.key", "attention.k_proj"),
    ("attention.self.value", "attention.v_proj"),
    ("attention.self.relative_position_bias_table", "attention.relative_position_bias.relative_position_bias_table"),
    ("attention.output.dense", "attention.o_proj"), ("intermediate.dense", "mlp.fc1"), ("output.dense", "mlp.fc2"))


def _remap_legacy_backbone_keys(sd: dict, config) -> dict:
    """Hub checkpoints written before transformers 5 (the reference's `facebook/mask2former-swin-large-coco-
    instance`, config.py:4) name the Swin tensors differently; rename them the way the dependency does at load."""
    bc = config.backbone_config if isinstance(config.backbone_config, dict) else {}
    if bc.get("model_type") != "swin":
        return sd
    pre, out = "model.pixel_level_module.encoder.", {}
    for k, v in sd.items():
        if k.startswith(pre):
            t = k[len(pre):]
            if t.endswith("relative_position_index"):
                continue  # recomputed buffer
            for a, b in _LEGACY_SWIN:
                t = t.replace(a, b)
            if t.startswith("encoder.") or t.startswith("embeddings."):
                t = "swin." + t
            elif t.startswith("layernorm."):
                t = "swin." + t
            k = pre + t
        out[k] = v
    return out


class Mask2FormerForUniversalSegmentation(nn.Module):
    """HF:2278-2530.  See the module docstring for the call contract."""

    main_input_name = "pixel_values"
    config_class = Mask2FormerConfig

    def __init__(self, config: Mask2FormerConfig):
        super().__init__()
        self.config = config
        self.model = Mask2FormerModel(config)
        self.weight_dict = {"loss_cross_entropy": config.class_weight, "loss_mask": config.mask_weight,
                            "loss_dice": config.dice_weight}
        self.class_predictor = nn.Linear(config.hidden_dim, config.num_labels + 1)
        self.criterion = Mask2FormerLoss(config, self.weight_dict)
        self.apply(self._init_weights)

    # -- initialisation, HF:2139-2199 ------------------------------------------------------
    def _init_weights(self, m):
        cfg = self.config
        if isinstance(m, MSDeformAttn):
            nn.init.constant_(m.sampling_offsets.weight, 0.0)
            thetas = torch.arange(m.n_heads, dtype=torch.int64).float() * (2.0 * math.pi / m.n_heads)
            grid = torch.stack([thetas.cos(), thetas.sin()], -1)
            grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(m.n_heads, 1, 1, 2).repeat(1, m.n_levels, m.n_points, 1)
            for i in range(m.n_points):
                grid[:, :, i, :] *= i + 1
            with torch.no_grad():
                m.sampling_offsets.bias.copy_(grid.view(-1))
            nn.init.constant_(m.attention_weights.weight, 0.0)
            nn.init.constant_(m.attention_weights.bias, 0.0)
            nn.init.xavier_uniform_(m.value_proj.weight)
            nn.init.constant_(m.value_proj.bias, 0.0)
            nn.init.xavier_uniform_(m.output_proj.weight)
            nn.init.constant_(m.output_proj.bias, 0.0)
        elif isinstance(m, MaskedAttentionDecoderLayer):
            for p in m.parameters():
                if p.dim() > 1:
                    nn.init.xavier_uniform_(p, gain=cfg.init_xavier_std)
            nn.init.zeros_(m.cross_attn.in_proj_bias)
        elif isinstance(m, Mask2FormerPixelDecoder):
            nn.init.zeros_(m.level_embed)
        elif isinstance(m, (nn.Linear, nn.Conv2d)) and not isinstance(m, _OutProj):
            pass  # torch defaults (kaiming-uniform), as the dependency leaves them unless listed above
        elif isinstance(m, nn.Embedding):
            nn.init.normal_(m.weight, mean=0.0, std=1.0)

    # -- forward, HF:2332-2530 -------------------------------------------------------------
    def forward(self, pixel_values: torch.Tensor, mask_labels=None, class_labels=None, pixel_mask=None,
                output_hidden_states=None, output_auxiliary_logits=None, output_attentions=None, return_dict=None,
                point_provider=None, **kwargs):
        if output_attentions:
            raise NotImplementedError("output_attentions: attention maps are never materialised by the kernels")
        want_aux = self.config.output_auxiliary_logits if output_auxiliary_logits is None else output_auxiliary_logits
        with_loss = mask_labels is not None and class_labels is not None
        feats, mask_features, multi_scale = self.model.pixel_level_module(pixel_values)
        h, inter, all_logits = self.model.transformer_module(multi_scale, mask_features,
                                                             tokens=self.model.pixel_level_module.last_tokens(),
                                                             need_all_logits=bool(want_aux or with_loss))
        all_classes = [self.class_predictor(s) for s in inter]
        aux = ([{"masks_queries_logits": m, "class_queries_logits": c} for m, c in zip(all_logits[:-1], all_classes[:-1])]
               if want_aux else None)
        loss = loss_dict = indices = None
        if mask_labels is not None and class_labels is not None:
            use_aux = self.config.use_auxiliary_loss
            embedder = self.model.transformer_module.decoder.mask_predictor.mask_embedder
            loss_dict, indices = self.criterion(all_logits if use_aux else all_logits[-1:],
                                                all_classes if use_aux else all_classes[-1:], mask_labels, class_labels,
                                                point_provider=point_provider,
                                                matched_rows=(inter if use_aux else inter[-1:], embedder, mask_features))
            loss = sum(loss_dict.values())
        out = Mask2FormerForUniversalSegmentationOutput(
            loss=loss, class_queries_logits=all_classes[-1], masks_queries_logits=all_logits[-1],
            auxiliary_logits=aux, encoder_last_hidden_state=feats[-1],
            pixel_decoder_last_hidden_state=mask_features, transformer_decoder_last_hidden_state=h,
            encoder_hidden_states=tuple(feats) if output_hidden_states else None,
            pixel_decoder_hidden_states=tuple(multi_scale) if output_hidden_states else None,
            transformer_decoder_hidden_states=tuple(inter) if output_hidden_states else None,
            loss_dict=loss_dict, matched_indices=indices)
        if return_dict is False:
            t = out.to_tuple()
            return t
        return out

    # -- persistence: the dependency's on-disk format (config.json + model.safetensors) -----
    def save_pretrained(self, directory: str, **_):
        from safetensors.torch import save_file
        os.makedirs(directory, exist_ok=True)
        self.config.save_pretrained(directory)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        save_file(sd, os.path.join(directory, "model.safetensors"), metadata={"format": "pt"})

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, id2label=None, label2id=None,
                        ignore_mismatched_sizes: bool = False, **config_overrides):
        d = pretrained_model_name_or_path
        if not os.path.isdir(d):
            raise FileNotFoundError(
                f"{d!r} is not a local directory. This build never downloads: pass a directory written by "
                "save_pretrained (config.json + model.safetensors).")
        over = dict(config_overrides)
        if id2label is not None:
            over["id2label"] = id2label
            over["label2id"] = label2id
        config = Mask2FormerConfig.from_pretrained(d, **over)
        model = cls(config)
        st_path, bin_path = os.path.join(d, "model.safetensors"), os.path.join(d, "pytorch_model.bin")
        if os.path.isfile(st_path):
            from safetensors.torch import load_file
            sd = load_file(st_path)
        elif os.path.isfile(bin_path):
            sd = torch.load(bin_path, map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin in {d}")
        sd = _remap_legacy_backbone_keys(sd, config)
        own = model.state_dict()
        mismatched = [k for k, v in sd.items() if k in own and tuple(own[k].shape) != tuple(v.shape)]
        if mismatched and not ignore_mismatched_sizes:
            raise RuntimeError(f"size mismatch for {mismatched}; pass ignore_mismatched_sizes=True to re-initialise them")
        for k in mismatched:
            sd.pop(k)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        # the final Swin layernorm is unused by the backbone call and absent from pre-5.x checkpoints
        missing = [k for k in missing if k not in mismatched and ".encoder.swin.layernorm." not in k]
        if missing or unexpected:
            raise RuntimeError(f"checkpoint does not match the module: missing={missing[:8]} unexpected={unexpected[:8]}")
        model.eval()
        return model
