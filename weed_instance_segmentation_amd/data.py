"""Sample files and label expansion: the step directly before the hot path.

The reference pre-processes every image once and stores a dictionary per image as a `.pt` file
(`datasets/dataset_utils.py:56-70`, read back by `PreprocessedDataset` `:7-29` and batched by `collate_fn` `:32-53`).
Each dictionary holds `mask_labels` as a float (T, H, W) stack -- what `Mask2FormerImageProcessor` makes of the
instance map through `convert_segmentation_map_to_binary_masks` -- i.e. 64 MB per 1024 x 1024 image with 16 plants,
which then crosses PCIe every step.

This module keeps the reference's file layout, key names and `collate_fn` contract, and adds a COMPACT sample: the
(H, W) instance-id map plus the id -> class dictionary (`instance_map`, `id_to_semantic`; both already in the
reference's dictionary under `original_map` / `id_to_semantic` semantics) instead of the mask stack.  The stack is
then produced on the device by `expand_labels` (HIP: `wm2f_labelmap_to_masks`) as uint8, which the loss and the
matcher read directly.  Full samples written by the reference keep loading: `tests/golden/ref_samples/*.pt` were written
by the reference's own `process_and_save` (`tests/golden/make_reference_samples.py`) and `tests/test_data.py` holds this
module to what the reference's own loader and `collate_fn` make of them.

Files are read with `torch.load(..., weights_only=True)`: nothing in a file is executed.  The reference's samples carry
`original_map` as a NUMPY array (`datasets/pheno_bench/dataset.py:85`, `:132`), which the restricted unpickler refuses
by default; exactly the three globals a plain numeric array pickles to (`numpy.ndarray`, `numpy.dtype`, numpy's
`_reconstruct`) plus numpy's numeric dtype classes are allow-listed for the duration of the load.  Object arrays stay
refused (they would need arbitrary globals), and so does everything else a pickle can name.
"""
from __future__ import annotations

import glob
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from . import ops

IGNORE_INDEX = 255  # the reference's choice: datasets/pheno_bench/dataset.py:85, :121


def _numpy_array_globals() -> list:
    """What a numeric numpy array inside a pickled sample names: the array type, the dtype type, the reconstruct helper
    (numpy 1.x pickles call it numpy.core.multiarray._reconstruct, numpy 2.x numpy._core.multiarray._reconstruct: the
    installed numpy resolves both to the same function) and, from numpy 1.25 on, the per-dtype classes."""
    try:
        from numpy._core.multiarray import _reconstruct  # numpy >= 2
    except ImportError:  # pragma: no cover
        from numpy.core.multiarray import _reconstruct
    allowed = [np.ndarray, np.dtype, _reconstruct, (_reconstruct, "numpy.core.multiarray._reconstruct"),
               (_reconstruct, "numpy._core.multiarray._reconstruct")]
    for name in ("bool_", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float16", "float32",
                 "float64"):
        allowed.append(type(np.dtype(getattr(np, name))))
    return allowed


def load_sample(path: str) -> dict:
    """One sample dictionary from a `.pt` file, executing nothing from the file (see the module docstring)."""
    with torch.serialization.safe_globals(_numpy_array_globals()):
        return torch.load(path, weights_only=True)


class PreprocessedDataset(Dataset):
    """Same behaviour as the reference class of this name (`datasets/dataset_utils.py:7-29`): the sorted `*.pt`
    files of a directory, one dictionary each."""

    def __init__(self, processed_dir: str):
        self.processed_dir = processed_dir
        self.files = sorted(glob.glob(os.path.join(processed_dir, "*.pt")))
        if not self.files:
            print(f'WARNING: No .pt files found in "{processed_dir}"')

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx):
        return load_sample(self.files[idx])


def collate_fn(batch) -> dict:
    """`datasets/dataset_utils.py:32-53`: stacked `pixel_values`, everything else as lists -- plus, for compact samples,
    `instance_maps` (list of (H, W) tensors).  `mask_labels` / `class_labels` are None for compact samples until
    `expand_labels` fills them."""
    out = {
        "pixel_values": torch.stack([item["pixel_values"] for item in batch]),
        "mask_labels": [item.get("mask_labels") for item in batch],
        "class_labels": [item.get("class_labels") for item in batch],
        "target_sizes": [item["target_size"] for item in batch],
        "original_maps": [item.get("original_map") for item in batch],
        "id_mappings": [item["id_to_semantic"] for item in batch],
        "file_names": [item["file_name"] for item in batch],
    }
    if any("instance_map" in item for item in batch):
        out["instance_maps"] = [item.get("instance_map") for item in batch]
    return out


def compact_sample(item: dict, instance_map: torch.Tensor) -> dict:
    """A reference-style sample dictionary without the float mask stack: keeps every other key, adds the id map the
    stack was made from (uint8 when the ids fit, else int32)."""
    out = {k: v for k, v in item.items() if k not in ("mask_labels", "class_labels")}
    im = torch.as_tensor(instance_map)
    out["instance_map"] = im.to(torch.uint8) if int(im.max()) <= 255 and int(im.min()) >= 0 else im.to(torch.int32)
    return out


def process_and_save(dataset, output_dir: str, compact: bool = False) -> None:
    """`datasets/dataset_utils.py:56-70`; with compact=True items must carry `instance_map` and are stored without the
    mask stack."""
    os.makedirs(output_dir, exist_ok=True)
    for i in range(len(dataset)):
        item = dataset[i]
        if compact:
            item = compact_sample(item, item["instance_map"])
        torch.save(item, os.path.join(output_dir, os.path.splitext(item["file_name"])[0] + ".pt"))


def segmentation_map_to_binary_masks(instance_map: torch.Tensor, instance_id_to_semantic_id: dict | None = None,
                                     ignore_index: int | None = None):
    """Device version of `convert_segmentation_map_to_binary_masks` (image_processing_mask2former.py:227-259; the
    reference never sets do_reduce_labels): ascending unique ids without `ignore_index` -> ((T, H, W) uint8 masks,
    (T,) int64 class ids).  `instance_map` must be on the GPU; the masks are uint8 instead of float -- the loss and
    the matcher sample them as such (`tgt_dtype = 1`)."""
    m = instance_map.to(torch.int32).contiguous()
    ids = torch.unique(m)
    if ignore_index is not None:
        ids = ids[ids != ignore_index]
    masks = ops.labelmap_to_masks(m, ids.to(torch.int32).contiguous())
    if instance_id_to_semantic_id is not None:
        lut = {int(k): int(v) for k, v in instance_id_to_semantic_id.items()}
        labels = torch.tensor([lut[int(i)] for i in ids.tolist()], dtype=torch.int64, device=m.device)
    else:
        labels = ids.to(torch.int64)
    return masks, labels


def expand_labels(batch: dict, device) -> dict:
    """Moves a collated batch to the device; compact samples get their `mask_labels` (uint8) / `class_labels` there."""
    out = dict(batch)
    out["pixel_values"] = batch["pixel_values"].to(device, non_blocking=True)
    ml, cl = [], []
    for i, (m, c) in enumerate(zip(batch["mask_labels"], batch["class_labels"])):
        if m is None:
            masks, labels = segmentation_map_to_binary_masks(batch["instance_maps"][i].to(device), batch["id_mappings"][i],
                                                             IGNORE_INDEX)
            ml.append(masks)
            cl.append(labels)
        else:
            ml.append(m.to(device, non_blocking=True))
            cl.append(c.to(device, non_blocking=True))
    out["mask_labels"], out["class_labels"] = ml, cl
    return out
