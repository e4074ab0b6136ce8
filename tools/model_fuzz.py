"""Randomised check of the whole tiny packed DiT (2 + 2 blocks) against the CPU oracle: random buckets (odd post-patch
sizes included: replicate padding of the 2x / 4x streams), frame counts, text lengths and mask shapes, latent paddings,
missing 2x / 4x streams, 4-D and 5-D inputs.  Not part of the test suite; run on the GPU box:
python tools/model_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import synthetic  # noqa: E402
from framepack_amd.model import HunyuanVideoTransformer3DModelPacked  # noqa: E402
from oracle import dit_oracle as O  # noqa: E402  (the checker; tools are not product code)

dev, BF = "cuda:0", torch.bfloat16
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.set_num_threads(16)
m = HunyuanVideoTransformer3DModelPacked(**synthetic.CONFIG_TINY).to(BF)
synthetic.init_synthetic_(dict(m.state_dict()), seed=0)
P32 = {k: v.float().clone() for k, v in m.state_dict().items()}
m = m.to(dev).eval()
cfg = O.DiTConfig(**synthetic.CONFIG_TINY)
worst = 0.0
for case in range(cases):
    h, w = 16 * rnd.randint(2, 14), 16 * rnd.randint(2, 14)
    frames = rnd.choice([1, 2, 3, 5])
    n_text = rnd.choice([1, 2, 77, 256, rnd.randint(1, 511)])
    with_2x, with_4x = rnd.random() < 0.8, rnd.random() < 0.8
    kw = synthetic.make_inputs(seed=case, frames_latent=frames, height=h, width=w, n_text=n_text, dtype=BF,
                               latent_padding=rnd.choice([0, 0, 1, 3]), sigma=rnd.choice([1.0, 0.6, 0.05]),
                               with_2x=with_2x, with_4x=with_4x)
    if rnd.random() < 0.3:                                  # a mask that is not a prefix (the reference only counts it)
        mask = kw["encoder_attention_mask"]
        perm = torch.randperm(mask.shape[1], generator=torch.Generator().manual_seed(case))
        kw["encoder_attention_mask"] = mask[:, perm]
    ref = O.dit_forward(P32, cfg, **{k: (v.float() if torch.is_tensor(v) and v.dtype == BF else v) for k, v in kw.items()})
    with torch.inference_mode():
        got = m(return_dict=False, **{k: (v.to(dev) if torch.is_tensor(v) and v.dtype in (BF, torch.bool) else v)
                                      for k, v in kw.items()})[0]
    err = ((got.double().cpu() - ref.double()).norm() / ref.double().norm()).item()
    worst = max(worst, err)
    if tuple(got.shape) != tuple(ref.shape) or not err < 2e-2:
        print(f"FAIL case {case}: {h}x{w} frames {frames} text {n_text} 2x {with_2x} 4x {with_4x}: rel-L2 {err:.3e}")
        sys.exit(1)
print(f"{cases} cases passed; worst rel-L2 {worst:.2e}")
