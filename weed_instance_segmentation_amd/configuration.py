"""Configuration object of the drop-in module.

Mirrors the fields of `transformers.Mask2FormerConfig` that the hot path reads
(configuration_mask2former.py:81-129) and round-trips the same `config.json`, so a directory
written by the reference's `save_pretrained` (models/mask2former/train.py:224, :234) loads here and
vice versa.  Unknown keys are preserved verbatim.
"""
from __future__ import annotations

import copy
import json
import os
from typing import Any

_DEFAULTS: dict[str, Any] = dict(
    feature_size=256, mask_feature_size=256, hidden_dim=256, encoder_feedforward_dim=1024,
    activation_function="relu", encoder_layers=6, decoder_layers=10, num_attention_heads=8, dropout=0.0,
    dim_feedforward=2048, pre_norm=False, enforce_input_projection=False, common_stride=4, ignore_value=255,
    num_queries=100, no_object_weight=0.1, class_weight=2.0, mask_weight=5.0, dice_weight=5.0,
    train_num_points=12544, oversample_ratio=3.0, importance_sample_ratio=0.75, init_std=0.02,
    init_xavier_std=1.0, use_auxiliary_loss=True, feature_strides=[4, 8, 16, 32], output_auxiliary_logits=None,
    model_type="mask2former",
)

_RESNET50 = dict(model_type="resnet", num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048],
                 depths=[3, 4, 6, 3], layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False,
                 downsample_in_bottleneck=False, out_features=["stage1", "stage2", "stage3", "stage4"])


class Mask2FormerConfig:
    model_type = "mask2former"

    def __init__(self, backbone_config: dict | None = None, num_labels: int | None = None,
                 id2label: dict | None = None, label2id: dict | None = None, **kwargs):
        cfg = copy.deepcopy(_DEFAULTS)
        self._extra: dict[str, Any] = {}
        for k, v in kwargs.items():
            if k in cfg:
                cfg[k] = v
            else:
                self._extra[k] = v
        self.__dict__.update(cfg)
        if backbone_config is None:
            backbone_config = copy.deepcopy(_RESNET50)
        elif not isinstance(backbone_config, dict):
            backbone_config = backbone_config.to_dict()  # a transformers config object
        self.backbone_config = copy.deepcopy(backbone_config)
        if id2label is None:
            n = 2 if num_labels is None else int(num_labels)
            id2label = {i: f"LABEL_{i}" for i in range(n)}
        self.id2label = {int(k): v for k, v in id2label.items()}
        if num_labels is not None and int(num_labels) != len(self.id2label):
            raise ValueError(f"num_labels={num_labels} disagrees with id2label of size {len(self.id2label)}")
        self.label2id = dict(label2id) if label2id is not None else {v: k for k, v in self.id2label.items()}

    @property
    def num_labels(self) -> int:
        return len(self.id2label)

    def to_dict(self) -> dict:
        d = {k: copy.deepcopy(getattr(self, k)) for k in _DEFAULTS}
        d.update(copy.deepcopy(self._extra))
        d["backbone_config"] = copy.deepcopy(self.backbone_config)
        d["id2label"] = {str(k): v for k, v in self.id2label.items()}
        d["label2id"] = dict(self.label2id)
        return d

    @classmethod
    def from_dict(cls, d: dict) -> "Mask2FormerConfig":
        d = dict(d)
        if d.get("id2label") is not None:
            d.pop("num_labels", None)  # id2label is authoritative (config.json stores no num_labels)
        return cls(**d)

    def save_pretrained(self, directory: str) -> None:
        os.makedirs(directory, exist_ok=True)
        with open(os.path.join(directory, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True, default=str)

    @classmethod
    def from_pretrained(cls, directory: str, **overrides) -> "Mask2FormerConfig":
        path = os.path.join(directory, "config.json")
        if not os.path.isfile(path):
            raise FileNotFoundError(
                f"{path} not found. Only LOCAL directories are supported: this build never contacts a model hub.")
        with open(path) as f:
            d = json.load(f)
        d.update(overrides)
        return cls.from_dict(d)
