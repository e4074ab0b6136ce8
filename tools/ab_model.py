"""Same-process A/B of kernel generations inside the 13B forward (ms per forward, alternating).
usage: ab_model.py [gemm|attn] [fp8]      (fp8: the opt-in e4m3 mode, linears and attention)
AB_VARIANTS=3,1 (env): the two values handed to the fp_debug_*_variant switch instead of 0, 1
(gemm: 3 = static tile schedule, 1 = tiles from the counters; attn: 18 = one block per part, 2 = parts from the counter)
AB_ATTR=group_text_gemms (env): alternate a boolean attribute of the model (False, True) instead of a library switch"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from framepack_amd import lib, synthetic  # noqa: E402
from framepack_amd.model import HunyuanVideoTransformer3DModelPacked  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
with torch.device(dev):
    model = HunyuanVideoTransformer3DModelPacked(**synthetic.CONFIG_13B).to(bf)
synthetic.init_synthetic_(dict(model.state_dict()), seed=0, fast_device_rng=True)
model.eval()
if "fp8" in sys.argv[2:]:
    model.enable_fp8_compute()
    model.enable_fp8_attention()
kw = synthetic.make_inputs(seed=31337, frames_latent=9, height=640, width=640, n_text=256, dtype=bf)
kw = {k: (v.to(dev) if torch.is_tensor(v) and v.dtype in (bf, torch.bool) else v) for k, v in kw.items()}
L = lib.load()
with torch.inference_mode():
    for _ in range(2):
        model(return_dict=False, **kw)
    attr = os.environ.get("AB_ATTR")
    for rep in range(3):
        for v in ([0, 1] if attr else [int(x) for x in os.environ.get("AB_VARIANTS", "0,1").split(",")]):
            if attr:
                assert isinstance(getattr(model, attr), bool)
                setattr(model, attr, bool(v))
            else:
                (L.fp_debug_attn_variant if (len(sys.argv) > 1 and sys.argv[1] == "attn") else L.fp_debug_gemm_variant)(v)
            model(return_dict=False, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                out = model(return_dict=False, **kw)[0]
            torch.cuda.synchronize()
            print(f"rep {rep} {attr or (sys.argv[1] if len(sys.argv) > 1 else 'gemm')} variant {v}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per forward", flush=True)
