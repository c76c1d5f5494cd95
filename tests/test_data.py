"""Sample files and label expansion."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import m2f_oracle as O

T = torch.from_numpy


def _fixture():
    g = load_golden("labelmap_masks.npz")
    return g, {int(k): v for k, v in json.loads(str(g["id2sem_json"])).items()}


def test_oracle_label_expansion_matches_dependency():
    g, id2sem = _fixture()
    masks, labels = O.convert_segmentation_map_to_binary_masks(T(g["instance_map"]), id2sem, ignore_index=255)
    assert torch.equal(masks.to(torch.uint8), T(g["masks"])) and torch.equal(labels, T(g["labels"]))


def _sample(i, compact):
    g, id2sem = _fixture()
    item = {"pixel_values": torch.full((3, 48, 64), float(i)), "target_size": (48, 64), "original_map": T(g["instance_map"]),
            "id_to_semantic": id2sem, "file_name": f"img_{i}.png"}
    if compact:
        item["instance_map"] = T(g["instance_map"])
    else:
        m, c = O.convert_segmentation_map_to_binary_masks(T(g["instance_map"]), id2sem, 255)
        item["mask_labels"], item["class_labels"] = m, c
    return item


def test_sample_files_round_trip_and_collate_contract(tmp_path):
    """Reference-format (full) and compact samples: written, read back with weights_only=True, collated with the
    reference's keys (datasets/dataset_utils.py:32-53)."""
    from weed_instance_segmentation_amd import data
    full = [_sample(i, False) for i in range(2)]
    data.process_and_save(full, str(tmp_path / "full"))
    data.process_and_save([_sample(i, True) for i in range(2)], str(tmp_path / "compact"), compact=True)
    ds_full, ds_c = data.PreprocessedDataset(str(tmp_path / "full")), data.PreprocessedDataset(str(tmp_path / "compact"))
    assert len(ds_full) == len(ds_c) == 2
    b = data.collate_fn([ds_full[0], ds_full[1]])
    assert set(b) == {"pixel_values", "mask_labels", "class_labels", "target_sizes", "original_maps", "id_mappings", "file_names"}
    assert b["pixel_values"].shape == (2, 3, 48, 64) and torch.equal(b["mask_labels"][1], full[1]["mask_labels"])
    c = data.collate_fn([ds_c[0], ds_c[1]])
    assert c["mask_labels"] == [None, None] and c["instance_maps"][0].dtype == torch.int32  # ids above 255 -> int32
    size_full = os.path.getsize(ds_full.files[0])
    size_c = os.path.getsize(ds_c.files[0])
    assert size_c < size_full / 2  # toy sizes: pixel_values dominate; at 1024 x 1024 with 16 instances it is 64 MB vs 4 MB


def test_reference_written_samples_load_and_collate_like_the_reference():
    """tests/golden/ref_samples/*.pt were written by the REFERENCE's process_and_save (datasets/dataset_utils.py:56-70)
    from items shaped like PhenoBenchDataset.__getitem__'s (datasets/pheno_bench/dataset.py:127-135: numpy int32
    `original_map`, int-keyed `id_to_semantic`, tuple `target_size`); ref_samples_batch.npz is what the reference's own
    PreprocessedDataset + collate_fn (:7-53) made of them (tests/golden/make_reference_samples.py).  This module's loader
    -- which executes nothing from the files -- must give the same batch, type for type."""
    from weed_instance_segmentation_amd import data
    g = load_golden("ref_samples_batch.npz")
    meta = json.loads(str(g["meta_json"]))
    ds = data.PreprocessedDataset(os.path.join(os.path.dirname(__file__), "golden", "ref_samples"))
    assert [os.path.basename(f) for f in ds.files] == meta["files"] and len(ds) == int(g["n"]) == 3
    b = data.collate_fn([ds[i] for i in range(len(ds))])
    assert set(b) == {"pixel_values", "mask_labels", "class_labels", "target_sizes", "original_maps", "id_mappings", "file_names"}
    assert torch.equal(b["pixel_values"], T(g["pixel_values"])) and b["file_names"] == meta["file_names"]
    for i in range(3):
        assert torch.equal(b["mask_labels"][i], T(g[f"mask_labels_{i}"])) and b["mask_labels"][i].dtype == torch.float32
        assert torch.equal(b["class_labels"][i], T(g[f"class_labels_{i}"])) and b["class_labels"][i].dtype == torch.int64
        assert isinstance(b["original_maps"][i], np.ndarray) and b["original_maps"][i].dtype == np.int32
        assert np.array_equal(b["original_maps"][i], g[f"original_map_{i}"])
        assert isinstance(b["target_sizes"][i], tuple) and list(b["target_sizes"][i]) == meta["target_sizes"][i]
        assert b["id_mappings"][i] == {int(k): v for k, v in meta["id_mappings"][i].items()}
        assert all(isinstance(k, int) for k in b["id_mappings"][i])
    assert b["mask_labels"][2].shape[0] == 0  # an image without instances keeps an empty (0, H, W) stack
    # what models/metrics.py:27-52 does with a batch: numpy ops on original_maps, `uid in mapping` with int keys
    uids = [int(u) for u in np.unique(b["original_maps"][1]) if u != 255 and u in b["id_mappings"][1]]
    assert uids == sorted(b["id_mappings"][1])


def test_sample_loader_refuses_everything_but_plain_arrays(tmp_path):
    """The allow-list admits numeric numpy arrays only: a pickle that names any other global (here os.system), or an
    object array, is refused -- nothing from a sample file is ever executed."""
    import pickle
    from weed_instance_segmentation_amd import data

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))

    torch.save({"x": Evil()}, str(tmp_path / "evil.pt"))
    torch.save({"x": np.array([{"a": 1}], dtype=object)}, str(tmp_path / "obj.pt"))
    for name in ("evil.pt", "obj.pt"):
        with pytest.raises(pickle.UnpicklingError):
            data.load_sample(str(tmp_path / name))


@pytest.mark.gpu
def test_expand_labels_on_device_matches_dependency(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from weed_instance_segmentation_amd import data
    g, id2sem = _fixture()
    masks, labels = data.segmentation_map_to_binary_masks(T(g["instance_map"]).cuda(), id2sem, ignore_index=255)
    assert masks.dtype == torch.uint8 and torch.equal(masks.cpu(), T(g["masks"])) and torch.equal(labels.cpu(), T(g["labels"]))
    batch = data.expand_labels(data.collate_fn([_sample(0, True), _sample(1, False)]), "cuda")
    assert torch.equal(batch["mask_labels"][0].cpu(), T(g["masks"]))
    assert torch.equal(batch["mask_labels"][1].cpu().to(torch.uint8), T(g["masks"]))
    assert torch.equal(batch["class_labels"][0].cpu(), batch["class_labels"][1].cpu())
    # a larger random map, odd ids, against the oracle
    rng = np.random.default_rng(3)
    big = torch.from_numpy(rng.integers(0, 40, (256, 380)).astype(np.int32) * 7)
    m, l = data.segmentation_map_to_binary_masks(big.cuda(), None, ignore_index=0)
    rm, rl = O.convert_segmentation_map_to_binary_masks(big, None, ignore_index=0)
    assert torch.equal(m.cpu(), rm.to(torch.uint8)) and torch.equal(l.cpu(), rl)
