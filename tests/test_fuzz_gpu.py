"""Random-shape fuzz of K1 / K2 / K3 (fp32, bf16) / the attention-mask build against the oracle, several launches back to
back with guard allocations behind the outputs (an out-of-bounds write of one launch shows up in the next check).
The full version is tools/probes/fuzz_kernels.py."""
import importlib.util
import os

import pytest
import torch


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4])
def test_kernel_fuzz(seed):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "probes", "fuzz_kernels.py")
    spec = importlib.util.spec_from_file_location("fuzz_kernels", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(seed, 10) == 0


@pytest.mark.gpu
def test_backward_fuzz():
    """Backward kernels (K1, K2, point samplers, mask-loss rows) and K4 on random shapes against the oracle's autograd."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "probes", "fuzz_backward.py")
    spec = importlib.util.spec_from_file_location("fuzz_backward", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(5, 6) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,seed,cases", [("fuzz_model", 7, 3), ("fuzz_postprocess", 7, 8)])
def test_model_and_postprocess_fuzz(name, seed, cases):
    """Small random-init models (query counts, odd sizes, ragged / empty label sets) and the device post-processing on
    random shapes against the oracle."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "probes", name + ".py")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(seed, cases) == 0
