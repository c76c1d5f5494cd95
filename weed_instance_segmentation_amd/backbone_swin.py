"""Swin Transformer backbone (stock PyTorch-ROCm ops; out of the hot-path scope).

Needed because the reference's own checkpoint is Swin-Large (config.py:4) and BASELINE.json's configs 4 / 5
are Swin-T / Swin-B.  Restates transformers' SwinBackbone (models/swin/modeling_swin.py:1070-1150 of
5.15.0) with that package's parameter names, so `save_pretrained` / `from_pretrained` round-trip:
  swin.embeddings.patch_embeddings.projection, swin.embeddings.norm,
  swin.encoder.layers.S.blocks.J.{attention.{q,k,v,o}_proj, attention.relative_position_bias.
      relative_position_bias_table, layernorm_before, layernorm_after, mlp.fc1, mlp.fc2},
  swin.encoder.layers.S.downsample.{reduction, norm}, swin.layernorm, hidden_states_norms.<stage>
As in the dependency's backbone call, windows are ALWAYS partitioned at the configured size
(always_partition=True, :1131) and the feature maps are taken before each stage's down-sampling (:1132).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn


def _window_partition(x, ws):  # (B,H,W,C) -> (B*nW, ws, ws, C)            modeling_swin.py:486-495
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.transpose(2, 3).contiguous().view(-1, ws, ws, C)


def _window_reverse(w, ws, H, W):  # inverse                                  modeling_swin.py:498-505
    C = w.shape[-1]
    w = w.view(-1, H // ws, W // ws, ws, ws, C)
    return w.transpose(2, 3).contiguous().view(-1, H, W, C)


class RelativePositionBias(nn.Module):  # modeling_swin.py:329-370
    def __init__(self, num_heads, ws):
        super().__init__()
        self.ws = ws
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws - 1
        rel[:, :, 1] += ws - 1
        rel[:, :, 0] *= 2 * ws - 1
        self.register_buffer("relative_position_index", rel.sum(-1).view(-1), persistent=False)

    def forward(self):
        a = self.ws * self.ws
        return self.relative_position_bias_table[self.relative_position_index].view(a, a, -1).permute(2, 0, 1).contiguous()[None]


class Attention(nn.Module):  # modeling_swin.py:401-468
    def __init__(self, dim, heads, ws, qkv_bias):
        super().__init__()
        self.heads, self.head_dim = heads, dim // heads
        self.q_proj = nn.Linear(dim, dim, bias=qkv_bias)
        self.k_proj = nn.Linear(dim, dim, bias=qkv_bias)
        self.v_proj = nn.Linear(dim, dim, bias=qkv_bias)
        self.o_proj = nn.Linear(dim, dim)
        self.relative_position_bias = RelativePositionBias(heads, ws)

    def forward(self, x, mask):
        nB, L, C = x.shape
        sh = lambda t: t.view(nB, L, self.heads, self.head_dim).transpose(1, 2)
        q, k, v = sh(self.q_proj(x)), sh(self.k_proj(x)), sh(self.v_proj(x))
        bias = self.relative_position_bias()
        if mask is not None:
            nW = mask.shape[0]
            bias = bias + mask[None, :, None].expand(nB // nW, -1, -1, -1, -1).reshape(-1, 1, L, L)
        a = torch.matmul(q, k.transpose(2, 3)) * self.head_dim ** -0.5 + bias
        a = F.softmax(a, dim=-1, dtype=torch.float32).to(q.dtype)
        return self.o_proj(torch.matmul(a, v).transpose(1, 2).reshape(nB, L, C))


class MLP(nn.Module):
    def __init__(self, dim, ratio):
        super().__init__()
        self.fc1 = nn.Linear(dim, int(ratio * dim))
        self.fc2 = nn.Linear(int(ratio * dim), dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


def _drop_path(x, p, training):  # modeling_swin.py:42-60 (per-sample stochastic depth)
    if p == 0.0 or not training:
        return x
    keep = 1 - p
    r = keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)
    return x.div(keep) * r.floor_()


class Layer(nn.Module):  # modeling_swin.py:508-626
    def __init__(self, cfg, dim, heads, drop_path, shift):
        super().__init__()
        self.ws, self.shift, self.drop_path = cfg["window_size"], shift, drop_path
        self.attention = Attention(dim, heads, self.ws, cfg.get("qkv_bias", True))
        eps = cfg.get("layer_norm_eps", 1e-5)
        self.layernorm_before = nn.LayerNorm(dim, eps=eps)
        self.layernorm_after = nn.LayerNorm(dim, eps=eps)
        self.mlp = MLP(dim, cfg.get("mlp_ratio", 4.0))

    def _mask(self, H, W, dtype, device):
        if self.shift <= 0:
            return None
        hi, wi = torch.arange(H, device=device), torch.arange(W, device=device)
        hr = (hi >= H - self.ws).long() + (hi >= H - self.shift).long()
        wr = (wi >= W - self.ws).long() + (wi >= W - self.shift).long()
        img = (hr[None, :, None, None] * 3 + wr[None, None, :, None]).to(dtype)
        mw = _window_partition(img, self.ws).view(-1, self.ws * self.ws)
        m = mw.unsqueeze(1) - mw.unsqueeze(2)
        return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)

    def forward(self, x, dims):
        H, W = dims
        B, _, C = x.shape
        ws = self.ws
        h = self.layernorm_before(x).view(B, H, W, C)
        pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws
        h = F.pad(h, (0, 0, 0, pr, 0, pb))
        Hp, Wp = H + pb, W + pr
        if self.shift > 0:
            h = torch.roll(h, shifts=(-self.shift, -self.shift), dims=(1, 2))
        a = self.attention(_window_partition(h, ws).view(-1, ws * ws, C), self._mask(Hp, Wp, h.dtype, h.device))
        a = _window_reverse(a.view(-1, ws, ws, C), ws, Hp, Wp)
        if self.shift > 0:
            a = torch.roll(a, shifts=(self.shift, self.shift), dims=(1, 2))
        if pr > 0 or pb > 0:
            a = a[:, :H, :W, :].contiguous()
        x = x + _drop_path(a.view(B, H * W, C), self.drop_path, self.training)
        return x + self.mlp(self.layernorm_after(x))


class PatchMerging(nn.Module):  # modeling_swin.py:289-326
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x, dims):
        H, W = dims
        B, _, C = x.shape
        x = x.view(B, H, W, C)
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = torch.cat([x[:, r::2, c::2, :] for c in range(2) for r in range(2)], dim=-1)
        return self.reduction(self.norm(x.view(B, -1, 4 * C)))


class Stage(nn.Module):
    def __init__(self, cfg, dim, depth, heads, dpr, downsample):
        super().__init__()
        self.blocks = nn.ModuleList([Layer(cfg, dim, heads, dpr[i], 0 if i % 2 == 0 else cfg["window_size"] // 2)
                                     for i in range(depth)])
        self.downsample = PatchMerging(dim) if downsample else None


class _PatchEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.p = cfg.get("patch_size", 4)
        self.projection = nn.Conv2d(cfg.get("num_channels", 3), cfg["embed_dim"], kernel_size=self.p, stride=self.p)

    def forward(self, x):
        H, W = x.shape[-2:]
        if W % self.p:
            x = F.pad(x, (0, self.p - W % self.p))
        if H % self.p:
            x = F.pad(x, (0, 0, 0, self.p - H % self.p))
        e = self.projection(x)
        return e.flatten(2).transpose(1, 2), (e.shape[2], e.shape[3])


class _Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        if cfg.get("use_absolute_embeddings", False):
            raise NotImplementedError("Swin use_absolute_embeddings")
        self.patch_embeddings = _PatchEmbeddings(cfg)
        self.norm = nn.LayerNorm(cfg["embed_dim"])

    def forward(self, x):
        e, dims = self.patch_embeddings(x)
        return self.norm(e), dims


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        depths, n = cfg["depths"], len(cfg["depths"])
        rate = cfg.get("drop_path_rate", 0.1)
        dpr = [rate * i / max(sum(depths) - 1, 1) for i in range(sum(depths))]
        self.layers = nn.ModuleList([
            Stage(cfg, int(cfg["embed_dim"] * 2 ** i), depths[i], cfg["num_heads"][i],
                  dpr[sum(depths[:i]):sum(depths[:i + 1])], i < n - 1) for i in range(n)])


class _SwinModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _Embeddings(cfg)
        self.encoder = _Encoder(cfg)
        self.layernorm = nn.LayerNorm(int(cfg["embed_dim"] * 2 ** (len(cfg["depths"]) - 1)), eps=cfg.get("layer_norm_eps", 1e-5))


class SwinBackbone(nn.Module):
    def __init__(self, cfg: dict):
        super().__init__()
        if cfg.get("hidden_act", "gelu") != "gelu":
            raise NotImplementedError("Swin hidden_act other than gelu")
        self.swin = _SwinModel(cfg)
        n = len(cfg["depths"])
        names = ["stem"] + [f"stage{i + 1}" for i in range(n)]
        feats = [cfg["embed_dim"]] + [int(cfg["embed_dim"] * 2 ** i) for i in range(n)]
        self.out_features = list(cfg.get("out_features") or [names[-1]])
        self.out_indices = [names.index(s) for s in self.out_features]
        self.channels = [feats[i] for i in self.out_indices]
        self.hidden_states_norms = nn.ModuleDict({s: nn.LayerNorm(c) for s, c in zip(self.out_features, self.channels)})

    def forward(self, pixel_values):
        x, dims = self.swin.embeddings(pixel_values)
        B = x.shape[0]
        maps = {"stem": (x, dims)}
        for i, st in enumerate(self.swin.encoder.layers):
            for blk in st.blocks:
                x = blk(x, dims)
            maps[f"stage{i + 1}"] = (x, dims)  # before down-sampling
            if st.downsample is not None:
                x = st.downsample(x, dims)
                dims = ((dims[0] + 1) // 2, (dims[1] + 1) // 2)
        out = []
        for s in self.out_features:
            h, (H, W) = maps[s]
            h = self.hidden_states_norms[s](h)
            out.append(h.view(B, H, W, -1).permute(0, 3, 1, 2).contiguous())
        return out
