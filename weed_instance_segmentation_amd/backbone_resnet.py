"""ResNet backbone (stock PyTorch-ROCm ops: MIOpen convolutions, BatchNorm).

Out of the hot-path scope: kept as plain torch.nn so the drop-in module is
self-contained.  Sub-module and parameter names follow transformers' ResNetBackbone so that the
reference's checkpoints (`save_pretrained` / `from_pretrained`, train.py:224, :245) round-trip:
  embedder.embedder.{convolution,normalization}
  encoder.stages.S.layers.J.{shortcut,layer.K}.{convolution,normalization}
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops


def _fused_ok(x: torch.Tensor) -> bool:
    """Inference on the GPU with a float4-friendly map: use the fused bias(+residual)+ReLU pass."""
    return ((not torch.is_grad_enabled()) and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled("cuda")
            and (x.shape[-1] * x.shape[-2]) % 4 == 0)


def _folded(conv: nn.Conv2d, bn: nn.BatchNorm2d, cache: dict):
    """Inference-time BatchNorm folding: conv(x, w) * g/sqrt(v+eps) + (b - m*g/sqrt(v+eps)) == conv(x, w', b').
    Cached per (weight / statistics version), so the fold is recomputed only after the parameters change."""
    key = (conv.weight._version, bn.weight._version, bn.bias._version, bn.running_mean._version,
           bn.running_var._version, conv.weight.device, conv.weight.dtype)
    if cache.get("key") != key:
        with torch.no_grad():
            scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
            cache["w"] = (conv.weight * scale[:, None, None, None]).contiguous()
            cache["b"] = (bn.bias - bn.running_mean * scale).contiguous()
            cache["key"] = key
    return cache["w"], cache["b"]


class ConvLayer(nn.Module):
    def __init__(self, cin, cout, kernel_size=3, stride=1, activation=True):
        super().__init__()
        self.convolution = nn.Conv2d(cin, cout, kernel_size, stride, kernel_size // 2, bias=False)
        self.normalization = nn.BatchNorm2d(cout)
        self.activation = nn.ReLU() if activation else nn.Identity()
        self._fold: dict = {}

    def forward(self, x, residual=None, extra_bias=None, force_relu=False):
        """`residual` / `extra_bias` / `force_relu` let a bottleneck fold its shortcut add and final ReLU
        into this layer's epilogue (inference only)."""
        c = self.convolution
        relu = force_relu or isinstance(self.activation, nn.ReLU)
        if not self.training and not torch.is_grad_enabled():
            # inference: convolution with folded BatchNorm statistics; bias (+ residual) + ReLU in ONE pass
            w, b = _folded(c, self.normalization, self._fold)
            if extra_bias is not None:
                b = b + extra_bias
            if _fused_ok(x):
                y = torch.nn.functional.conv2d(x, w, None, c.stride, c.padding)
                if (y.shape[-1] * y.shape[-2]) % 4 == 0:
                    return ops.bias_act_(y, b, residual, relu)
                y = y + b[None, :, None, None]
            else:
                y = torch.nn.functional.conv2d(x, w, b, c.stride, c.padding)
            if residual is not None:
                y = y + residual
            return torch.relu(y) if relu else y
        y = self.normalization(c(x))
        if residual is not None:
            y = y + residual
        return torch.relu(y) if relu else y


class ShortCut(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.convolution = nn.Conv2d(cin, cout, 1, stride, bias=False)
        self.normalization = nn.BatchNorm2d(cout)
        self._fold: dict = {}

    def forward(self, x):
        return self.normalization(self.convolution(x))

    def folded_raw(self, x):
        """Inference: the shortcut convolution WITHOUT its folded bias, and that bias (the bottleneck adds it
        in its fused epilogue)."""
        w, b = _folded(self.convolution, self.normalization, self._fold)
        return torch.nn.functional.conv2d(x, w, None, self.convolution.stride), b


class BottleNeckLayer(nn.Module):
    def __init__(self, cin, cout, stride, reduction=4, downsample_in_bottleneck=False):
        super().__init__()
        red = cout // reduction
        self.shortcut = ShortCut(cin, cout, stride) if (cin != cout or stride != 1) else nn.Identity()
        self.layer = nn.Sequential(
            ConvLayer(cin, red, 1, stride if downsample_in_bottleneck else 1),
            ConvLayer(red, red, 3, 1 if downsample_in_bottleneck else stride),
            ConvLayer(red, cout, 1, activation=False),
        )
        self.activation = nn.ReLU()

    def forward(self, x):
        if not self.training and not torch.is_grad_enabled():
            h = self.layer[1](self.layer[0](x))
            if isinstance(self.shortcut, ShortCut):
                sc, sc_bias = self.shortcut.folded_raw(x)
                return self.layer[2](h, residual=sc, extra_bias=sc_bias, force_relu=True)
            return self.layer[2](h, residual=x, force_relu=True)
        return self.activation(self.layer(x) + self.shortcut(x))


class Stage(nn.Module):
    def __init__(self, cin, cout, stride, depth, dib):
        super().__init__()
        self.layers = nn.Sequential(BottleNeckLayer(cin, cout, stride, downsample_in_bottleneck=dib),
                                    *[BottleNeckLayer(cout, cout, 1, downsample_in_bottleneck=dib) for _ in range(depth - 1)])

    def forward(self, x):
        return self.layers(x)


class _Embedder(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.embedder = ConvLayer(cin, cout, 7, 2)
        self.pooler = nn.MaxPool2d(3, 2, 1)

    def forward(self, x):
        e, pool = self.embedder, self.pooler
        if (not e.training and _fused_ok(x) and isinstance(e.activation, nn.ReLU)
                and (pool.kernel_size, pool.stride, pool.padding, pool.dilation, pool.ceil_mode) == (3, 2, 1, 1, False)):
            # inference: bias + ReLU + 3x3 / 2 max pool in one pass over the raw convolution output
            c = e.convolution
            w, b = _folded(c, e.normalization, e._fold)
            y = torch.nn.functional.conv2d(x, w, None, c.stride, c.padding)
            if y.shape[-2] % 2 == 0 and y.shape[-1] % 8 == 0:
                return ops.bias_relu_maxpool(y, b)
            return pool(torch.relu_(y.add_(b[None, :, None, None])))
        return pool(e(x))


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        hs, depths = cfg["hidden_sizes"], cfg["depths"]
        dib = bool(cfg.get("downsample_in_bottleneck", False))
        first_stride = 2 if cfg.get("downsample_in_first_stage", False) else 1
        stages = [Stage(cfg["embedding_size"], hs[0], first_stride, depths[0], dib)]
        for cin, cout, d in zip(hs[:-1], hs[1:], depths[1:]):
            stages.append(Stage(cin, cout, 2, d, dib))
        self.stages = nn.ModuleList(stages)


class ResNetBackbone(nn.Module):
    def __init__(self, cfg: dict):
        super().__init__()
        if cfg.get("layer_type", "bottleneck") != "bottleneck":
            raise NotImplementedError("ResNetBackbone: only layer_type='bottleneck' is built")
        self.embedder = _Embedder(cfg.get("num_channels", 3), cfg["embedding_size"])
        self.encoder = _Encoder(cfg)
        names = ["stem"] + [f"stage{i + 1}" for i in range(len(cfg["depths"]))]
        out = cfg.get("out_features") or [names[-1]]
        self.out_indices = [names.index(n) for n in out]
        chans = [cfg["embedding_size"]] + list(cfg["hidden_sizes"])
        self.channels = [chans[i] for i in self.out_indices]

    def forward(self, pixel_values: torch.Tensor) -> list[torch.Tensor]:
        x = self.embedder(pixel_values)
        feats = [x]
        for st in self.encoder.stages:
            x = st(x)
            feats.append(x)
        return [feats[i] for i in self.out_indices]


def build_backbone(cfg: dict) -> nn.Module:
    mt = cfg.get("model_type")
    if mt == "resnet":
        return ResNetBackbone(cfg)
    if mt == "swin":
        from .backbone_swin import SwinBackbone
        return SwinBackbone(cfg)
    raise NotImplementedError(f"backbone model_type={mt!r} is not built (resnet and swin are)")
