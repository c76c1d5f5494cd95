#!/usr/bin/env python3
"""Sample files exactly as the REFERENCE writes them, plus what the reference's own loader makes of them.

Runs ONLY in the build container: it imports the reference's `datasets.dataset_utils` from /root/reference
(`process_and_save`, `PreprocessedDataset`, `collate_fn` -- datasets/dataset_utils.py:7-70; they import here,
SURVEY section 8c) and the dependency's PIL-backed image processor (the torchvision-backed one does not import here).

An item is what `PhenoBenchDataset.__getitem__` returns (datasets/pheno_bench/dataset.py:85-135): the processor's
`pixel_values` / `mask_labels` / `class_labels` of ONE image, `target_size` as a tuple, `original_map` as a NUMPY int32
instance map with 255 = background / ignore, `id_to_semantic` as an int -> int dict, `file_name`.  The instance maps
are synthetic (cv2 and the PhenoBench files are absent here); everything after them is the reference's own code path.

Writes
  tests/golden/ref_samples/<name>.pt     the reference's torch.save of each item (pickled dict with a numpy array)
  tests/golden/ref_samples_batch.npz     the reference's collate_fn over the reference's PreprocessedDataset of that
                                         directory: what tests/test_data.py holds data.PreprocessedDataset/collate_fn to

Usage:  HF_HUB_OFFLINE=1 TRANSFORMERS_OFFLINE=1 python tests/golden/make_reference_samples.py
"""
import json
import os
import shutil
import sys

os.environ.setdefault("HF_HUB_OFFLINE", "1")
os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")
sys.path.insert(0, "/root/reference")  # before anything imports the unrelated PyPI package called `datasets`
import datasets.dataset_utils as ref_du  # noqa: E402  (the reference's module)

assert ref_du.__file__.startswith("/root/reference/"), ref_du.__file__

import numpy as np  # noqa: E402
import torch  # noqa: E402
import transformers  # noqa: E402
from PIL import Image  # noqa: E402
from transformers.models.mask2former.image_processing_pil_mask2former import Mask2FormerImageProcessorPil  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "ref_samples")


class SyntheticPhenoBench:
    """Stands in for PhenoBenchDataset up to the instance map; from there on the same calls (dataset.py:118-135)."""

    def __init__(self, n, processor, seed=11, hw=(48, 80)):
        self.n, self.processor, self.seed, self.hw = n, processor, seed, hw

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        rng = np.random.default_rng(self.seed + idx)
        H, W = self.hw
        image = Image.fromarray(rng.integers(0, 255, (H, W, 3), dtype=np.uint8))
        instance_map = np.full((H, W), 255, dtype=np.int32)
        id_to_semantic = {}
        for iid in range(1, 3 + idx):  # 2, 3, ... instances; image 2 holds none (an empty target list)
            if idx == 2:
                break
            hh, ww = rng.integers(6, 20, 2)
            y0, x0 = rng.integers(0, H - hh), rng.integers(0, W - ww)
            instance_map[y0:y0 + hh, x0:x0 + ww] = iid
            id_to_semantic[iid] = int(rng.integers(1, 5))
        id_to_semantic = {k: v for k, v in id_to_semantic.items() if (instance_map == k).any()}
        inputs = self.processor(images=[image], segmentation_maps=[instance_map], instance_id_to_semantic_id=id_to_semantic,
                                return_tensors="pt", ignore_index=255)
        return {"pixel_values": inputs["pixel_values"][0], "mask_labels": inputs["mask_labels"][0],
                "class_labels": inputs["class_labels"][0], "target_size": (H, W), "original_map": instance_map,
                "id_to_semantic": id_to_semantic, "file_name": f"phenobench_like_{idx:02d}.png"}


def main():
    proc = Mask2FormerImageProcessorPil(size={"height": 64, "width": 64}, ignore_index=255)
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    ref_du.process_and_save(SyntheticPhenoBench(3, proc), OUT)  # the reference's writer
    ds = ref_du.PreprocessedDataset(OUT)                          # the reference's reader (weights_only=False: our own files)
    batch = ref_du.collate_fn([ds[i] for i in range(len(ds))])   # the reference's collate
    arrays = {"pixel_values": batch["pixel_values"].numpy(), "n": np.asarray(len(ds)),
              "hf_version": np.asarray(transformers.__version__), "torch_version": np.asarray(torch.__version__),
              "numpy_version": np.asarray(np.__version__)}
    meta = {"target_sizes": [list(t) for t in batch["target_sizes"]], "file_names": batch["file_names"],
            "id_mappings": [{str(k): int(v) for k, v in m.items()} for m in batch["id_mappings"]],
            "files": [os.path.basename(f) for f in ds.files]}
    for i in range(len(ds)):
        arrays[f"mask_labels_{i}"] = batch["mask_labels"][i].numpy()
        arrays[f"class_labels_{i}"] = batch["class_labels"][i].numpy()
        arrays[f"original_map_{i}"] = np.asarray(batch["original_maps"][i])
        assert isinstance(batch["original_maps"][i], np.ndarray) and isinstance(batch["target_sizes"][i], tuple)
    arrays["meta_json"] = np.asarray(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "ref_samples_batch.npz"), **arrays)
    for f in ds.files:
        print(os.path.basename(f), os.path.getsize(f) // 1024, "KiB", torch.serialization.get_unsafe_globals_in_checkpoint(f))


if __name__ == "__main__":
    main()
