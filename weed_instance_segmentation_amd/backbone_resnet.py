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


class ConvLayer(nn.Module):
    def __init__(self, cin, cout, kernel_size=3, stride=1, activation=True):
        super().__init__()
        self.convolution = nn.Conv2d(cin, cout, kernel_size, stride, kernel_size // 2, bias=False)
        self.normalization = nn.BatchNorm2d(cout)
        self.activation = nn.ReLU() if activation else nn.Identity()

    def forward(self, x):
        return self.activation(self.normalization(self.convolution(x)))


class ShortCut(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.convolution = nn.Conv2d(cin, cout, 1, stride, bias=False)
        self.normalization = nn.BatchNorm2d(cout)

    def forward(self, x):
        return self.normalization(self.convolution(x))


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
        return self.pooler(self.embedder(x))


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
    raise NotImplementedError(f"backbone model_type={mt!r} is not built yet (resnet only; Swin is planned)")
