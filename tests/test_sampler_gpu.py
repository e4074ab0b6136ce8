"""The sampler mirror (framepack_amd.sampler) with its tensors on the GPU against the golden trajectories made from
the reference's own uni_pc_fm.py (tests/golden/sampler.npz): the loop the product runs - latents, model output and the
AXPYs of every update on the device, every UniPC coefficient on the host from ONE copy of the sigma schedule."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from framepack_amd import sampler as S  # noqa: E402

DEV = "cuda:0"


def _toy(x, t, **extra):
    tt = t[(...,) + (None,) * (x.ndim - 1)]
    return x * (1 - tt) * 0.7 + 0.1 * torch.sin(3.0 * x) + 0.05 * tt


@pytest.mark.parametrize("n", [1, 2, 3, 5, 25])
def test_unipc_on_the_gpu_matches_reference_trajectories(golden_dir, n):
    G = np.load(os.path.join(golden_dir, "sampler.npz"))
    x = torch.randn(2, 4, 6, 5, 8, generator=torch.Generator("cpu").manual_seed(100 + n)).to(DEV)
    sig = torch.from_numpy(G[f"unipc_{n}_sigmas"]).to(DEV)
    xs = []
    with torch.inference_mode():
        res = S.sample_unipc(_toy, x.clone(), sig, extra_args={"movement_scale": 0.1},
                             callback=lambda d: xs.append(d["x"].clone()), disable=True)
    # fp32 on another device (its own sin, fused multiply-adds): a few ulp per step over up to 25 steps
    torch.testing.assert_close(torch.stack(xs).cpu(), torch.from_numpy(G[f"unipc_{n}_xs"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(res.cpu(), torch.from_numpy(G[f"unipc_{n}_final"]), rtol=1e-4, atol=1e-5)


def test_unipc_loop_makes_one_device_to_host_copy(golden_dir):
    """The reference reads several scalars back per step (rks, torch.tensor(b), two linalg.solve on the device); the
    mirror copies the sigma schedule once and computes every coefficient on the host."""
    G = np.load(os.path.join(golden_dir, "sampler.npz"))
    sig = torch.from_numpy(G["unipc_25_sigmas"]).to(DEV)
    x = torch.randn(1, 4, 3, 4, 4, device=DEV)
    copies = {"n": 0}
    real = torch.Tensor.cpu

    def counting(self, *a, **k):
        if self.is_cuda:
            copies["n"] += 1
        return real(self, *a, **k)
    torch.Tensor.cpu = counting
    try:
        S.sample_unipc(_toy, x, sig, extra_args={"movement_scale": 0.1}, disable=True)
    finally:
        torch.Tensor.cpu = real
    assert copies["n"] == 1
