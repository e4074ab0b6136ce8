"""Generate the golden vectors under tests/golden/ from the reference's OWN modules.

Run in the build container only (needs /root/reference; nothing here travels to the GPU box
except the .npz outputs):   python tests/golden/make_golden.py

What is pinned by what:
  * rope.npz, dit_tiny.npz   <- diffusers_helper/models/hunyuan_video_packed.py + dit_common.py,
                                 imported with tests/golden/diffusers_standin.py in place of the
                                 absent diffusers package (so the diffusers containers themselves
                                 stay "parity unpinned").
  * sampler.npz              <- diffusers_helper/k_diffusion/{uni_pc_fm,wrapper}.py (import as is).
  * fp8.npz                  <- utils/fp8_optimization_utils.py (imports as is).
  * lora.npz                 <- utils/lora_utils.py (imports as is): the three LoRA layouts merged into seeded
                                 state dicts (inputs rebuilt in the test by tests/lora_cases.py).
  * buckets.npz              <- diffusers_helper/bucket_tools.py (imports as is): find_nearest_bucket over a size sweep.
  * hunyuan.npz              <- diffusers_helper/hunyuan.py: vae_decode_fake and the vae_decode / vae_encode wrappers on
                                 a stub VAE (the module's diffusers import, a prompt template, is met by an empty dict).
  * pipeline.npz             <- diffusers_helper/pipelines/k_diffusion_hunyuan.py: sample_hunyuan end to end with the stub
                                 transformer of tests/pipeline_cases.py (imports utils.py: see utils.npz).
  * utils.npz                <- diffusers_helper/utils.py: soft_append_bcthw, crop_or_pad_yield_mask,
                                 repeat_to_batch_size.  The module's top-level `import cv2` / `import torchvision` name
                                 packages this image lacks; two EMPTY module objects stand in for them during the import
                                 (none of the three functions touches either; the functions that do - mp4 / image I/O -
                                 are out of scope and are not called).  Inputs rebuilt in the test from the same seeds.
Inputs are regenerated in the tests from the same seeds (framepack_amd/synthetic.py,
oracle.dit_oracle.synth_params); only outputs (sub-sampled where large) are stored.
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import diffusers_standin  # noqa: E402

diffusers_standin.install()
with contextlib.redirect_stdout(io.StringIO()):
    import diffusers_helper.dit_common  # noqa: E402,F401  (monkey-patches the norm forwards)
    from diffusers_helper.models import hunyuan_video_packed as ref  # noqa: E402
from diffusers_helper.k_diffusion import uni_pc_fm, wrapper  # noqa: E402
from utils import fp8_optimization_utils as ref_fp8  # noqa: E402

from framepack_amd import synthetic  # noqa: E402
from oracle import dit_oracle as O  # noqa: E402

torch.set_grad_enabled(False)


def sub(t, tok=97, ch=5):
    """Sub-sample a [1,S,C] tensor for storage."""
    return t[0, ::tok, ::ch].float().numpy()


# ----------------------------------------------------------------------------- rope / packing
def gen_rope():
    out = {}
    rope = ref.HunyuanVideoRotaryPosEmbed((16, 56, 56), 256.0)
    cases = {
        "noisy": (torch.arange(1, 10).unsqueeze(0), 6, 8),
        "clean": (torch.tensor([[0, 10]]), 6, 8),
        "x2": (torch.tensor([[11, 12]]), 6, 8),
        "x4": (torch.arange(13, 29).unsqueeze(0), 6, 8),
        "x2_odd": (torch.tensor([[11, 12]]), 5, 9),       # non-divisible -> replicate pad
        "x4_odd": (torch.arange(13, 29).unsqueeze(0), 5, 9),
        "x4_pad": (torch.arange(22, 38).unsqueeze(0), 10, 19),   # latent_padding > 0, 40x76-like
    }
    for name, (idx, h, w) in cases.items():
        t = rope(frame_indices=idx, height=h, width=w, device=torch.device("cpu"))
        if name.startswith("x2"):
            t = ref.center_down_sample_3d(ref.pad_for_3d_conv(t, (2, 2, 2)), (2, 2, 2))
        if name.startswith("x4"):
            t = ref.center_down_sample_3d(ref.pad_for_3d_conv(t, (4, 4, 4)), (4, 4, 4))
        out[name] = t.flatten(2).transpose(1, 2).numpy()
        out[name + "_idx"] = idx.numpy()
        out[name + "_hw"] = np.array([h, w])
    # apply_rotary_emb_transposed on seeded bf16 input (ref :80-86)
    g = torch.Generator("cpu").manual_seed(7)
    x = torch.randn(1, 48 * 9, 3, 128, generator=g).to(torch.bfloat16)
    fc = torch.from_numpy(out["noisy"])
    y = ref.apply_rotary_emb_transposed(x, fc)
    out["apply_in_seed"] = np.array([7])
    out["apply_out"] = y.float().numpy()
    # pad_for_3d_conv on a ramp (ref :45-51)
    r = torch.arange(2 * 3 * 5 * 7, dtype=torch.float32).view(1, 2, 3, 5, 7)
    out["pad_248"] = ref.pad_for_3d_conv(r, (2, 4, 8)).numpy()
    np.savez_compressed(os.path.join(HERE, "rope.npz"), **out)
    print("rope.npz", {k: v.shape for k, v in out.items()})


# ----------------------------------------------------------------------------- tiny DiT
def build_ref_model(cfg_kw, P):
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref.HunyuanVideoTransformer3DModelPacked(**cfg_kw)
    m.eval()
    m.load_state_dict(P, strict=True)
    return m


def gen_dit_tiny():
    cfg_kw = dict(synthetic.CONFIG_TINY)
    cfg = O.DiTConfig(**cfg_kw)
    P = O.synth_params(cfg, seed=0)
    m = build_ref_model(cfg_kw, P)
    out = {}

    # (A) config 1: 256x256, 3 latent frames, all clean streams, n_text = 37
    kw = synthetic.make_inputs(seed=31337, frames_latent=3, height=256, width=256, n_text=37, dtype=torch.float32,
                               sigma=0.9375)
    taps = {}
    hooks = []
    for kind, blocks in (("double", m.transformer_blocks), ("single", m.single_transformer_blocks)):
        for i, blk in enumerate(blocks):
            hooks.append(blk.register_forward_hook(
                lambda mod, a, o, key=f"{kind}.{i}": taps.__setitem__(key, o)))
    y = m(return_dict=False, **kw)[0]
    for h in hooks:
        h.remove()
    out["A_out"] = y.numpy()
    for k, (h, e) in taps.items():
        out[f"A_{k}_h"] = sub(h)
        out[f"A_{k}_e"] = sub(e, tok=29)
        out[f"A_{k}_norms"] = np.array([h.norm().item(), e.norm().item()])
    hp, rp = m.process_input_hidden_states(
        kw["hidden_states"], kw["latent_indices"], kw["clean_latents"], kw["clean_latent_indices"],
        kw["clean_latents_2x"], kw["clean_latent_2x_indices"], kw["clean_latents_4x"], kw["clean_latent_4x_indices"])
    out["A_packed_h"] = sub(hp)
    out["A_packed_rope"] = sub(rp, tok=31, ch=3)
    out["A_packed_shape"] = np.array(hp.shape)

    # (B) odd bucket 160x304 (latent 20x38 -> post-patch 10x19: 2x and 4x need replicate pad),
    #     latent_padding=1, fp32 output head, 4-D... no: 5-D input
    kwb = synthetic.make_inputs(seed=5, frames_latent=2, height=160, width=304, n_text=5, dtype=torch.float32,
                                sigma=0.31, latent_padding=1)
    m.high_quality_fp32_output_for_inference = True
    out["B_out"] = m(return_dict=False, **kwb)[0].numpy()
    m.high_quality_fp32_output_for_inference = False

    # (C) one-frame mode: no 2x/4x streams (ref worker.py:474-477)
    kwc = synthetic.make_inputs(seed=6, frames_latent=1, height=128, width=128, n_text=256, dtype=torch.float32,
                                sigma=0.77, with_2x=False, with_4x=False)
    out["C_out"] = m(return_dict=False, **kwc)[0].numpy()

    # (D) TeaCache over a 6-step schedule (ref :944-1022): outputs + decisions
    with contextlib.redirect_stdout(io.StringIO()):
        m.initialize_teacache(enable_teacache=True, num_steps=6, rel_l1_thresh=0.15)
    sig = [1.0, 0.98, 0.95, 0.9, 0.6, 0.2]
    outs, hits = [], []
    for s in sig:
        kwd = dict(kw)
        kwd["timestep"] = torch.tensor([s * 1000.0])
        y = m(return_dict=False, **kwd)[0]
        outs.append(y[0, :, :, ::4, ::4].numpy())
        hits.append([m.cache_hits, m.cache_misses, m.cache_queries])
    out["D_sigmas"] = np.array(sig)
    out["D_outs"] = np.stack(outs)
    out["D_counts"] = np.array(hits)
    m.enable_teacache = False

    # (E) bf16 run of case A (the reference's own rounding points), final output only
    Pb = {k: v.to(torch.bfloat16) for k, v in P.items()}
    mb = build_ref_model(cfg_kw, P).to(torch.bfloat16)
    kwe = synthetic.make_inputs(seed=31337, frames_latent=3, height=256, width=256, n_text=37, dtype=torch.bfloat16,
                                sigma=0.9375)
    out["E_out_bf16"] = mb(return_dict=False, **kwe)[0].float().numpy()
    del Pb

    np.savez_compressed(os.path.join(HERE, "dit_tiny.npz"), **out)
    print("dit_tiny.npz", {k: v.shape for k, v in out.items() if not k.startswith("A_d") and not k.startswith("A_s")})


# ----------------------------------------------------------------------------- sampler
def gen_sampler():
    out = {}
    g = torch.Generator("cpu").manual_seed(11)

    def toy_model(x, t, **extra):                      # analytic "x0 prediction"
        tt = t[(...,) + (None,) * (x.ndim - 1)]
        return x * (1 - tt) * 0.7 + 0.1 * torch.sin(3.0 * x) + 0.05 * tt

    for n in (1, 2, 3, 5, 25):
        x = torch.randn(2, 4, 6, 5, 8, generator=torch.Generator("cpu").manual_seed(100 + n))
        sig = torch.linspace(1, 0, n + 1)
        mu = 1.2
        import math
        sig = math.exp(mu) / (math.exp(mu) + (1 / sig - 1) ** 1.0)
        xs = []
        with contextlib.redirect_stdout(io.StringIO()):
            res = uni_pc_fm.sample_unipc(toy_model, x.clone(), sig, extra_args={"movement_scale": 0.1},
                                         callback=lambda d: xs.append(d["x"].clone().numpy()), disable=True)
        out[f"unipc_{n}_final"] = res.numpy()
        out[f"unipc_{n}_xs"] = np.stack(xs)
        out[f"unipc_{n}_sigmas"] = sig.numpy()

    # fm_wrapper.k_model with a stub transformer (ref wrapper.py:19-71)
    def stub(hidden_states, timestep, return_dict, bias=0.0, **kw):
        return (torch.tanh(hidden_states.float() * 0.5 + timestep.float().view(-1, 1, 1, 1, 1) * 1e-3 + bias)
                .to(hidden_states.dtype),)

    x = torch.randn(2, 4, 3, 5, 6, generator=g)
    sigma = torch.tensor([0.7, 0.7])
    km = wrapper.fm_wrapper(stub)
    for name, kwargs in {
        "pos_only": dict(cfg_scale=1.0, cfg_rescale=0.0, has_negative_prompt=False),
        "cfg": dict(cfg_scale=3.0, cfg_rescale=0.0, has_negative_prompt=True),
        "cfg_rescale": dict(cfg_scale=3.0, cfg_rescale=0.7, has_negative_prompt=True),
    }.items():
        for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            y = km(x, sigma, dtype=dt, concat_latent=None, positive=dict(bias=0.0), negative=dict(bias=-0.3), **kwargs)
            out[f"kmodel_{name}_{dt_name}"] = y.numpy()
    out["kmodel_x"] = x.numpy()
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **out)
    print("sampler.npz", len(out), "arrays")


# ----------------------------------------------------------------------------- fp8
def gen_fp8():
    out = {}
    g = torch.Generator("cpu").manual_seed(3)
    w = (torch.randn(48, 64, generator=g) * 0.05)
    w[0, 0] = 0.0
    w[1, 1] = 0.73
    for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        sd = {"transformer_blocks.0.attn.to_q.weight": w.to(dt), "transformer_blocks.0.attn.to_q.bias": torch.zeros(48, dtype=dt),
              "transformer_blocks.0.attn.norm_q.weight": torch.ones(8, dtype=dt),
              "transformer_blocks.0.norm1.linear.weight": w.to(dt).clone(),
              "proj_out.weight": w.to(dt).clone()}
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            q = ref_fp8.optimize_state_dict_with_fp8(dict(sd), None, ["transformer_blocks", "single_transformer_blocks"],
                                                     ["norm"])
        out[f"{dt_name}_keys"] = np.array(sorted(q.keys()))
        k = "transformer_blocks.0.attn.to_q"
        out[f"{dt_name}_w8_bytes"] = q[k + ".weight"].view(torch.uint8).numpy()
        out[f"{dt_name}_scale"] = q[k + ".scale_weight"].float().numpy()
        lin = torch.nn.Linear(64, 48).to(dt)
        lin.weight = torch.nn.Parameter(q[k + ".weight"], requires_grad=False)
        lin.bias = torch.nn.Parameter(sd[k + ".bias"], requires_grad=False)
        lin.register_buffer("scale_weight", q[k + ".scale_weight"])
        x = torch.randn(1, 5, 64, generator=torch.Generator("cpu").manual_seed(4)).to(dt)
        y = ref_fp8.fp8_linear_forward_patch(lin, x, False, None)
        out[f"{dt_name}_linear_out"] = y.float().numpy()
    out["w"] = w.numpy()
    out["maxval_e4m3"] = np.array([ref_fp8.calculate_fp8_maxval(4, 3)])
    out["maxval_e5m2"] = np.array([ref_fp8.calculate_fp8_maxval(5, 2)])
    np.savez_compressed(os.path.join(HERE, "fp8.npz"), **out)
    print("fp8.npz", len(out), "arrays")


# ----------------------------------------------------------------------------- lora
def gen_lora():
    import tempfile
    from safetensors.torch import save_file
    from utils import lora_utils as ref_lora
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lora_cases
    out = {}
    for name in lora_cases.CASES:
        sd, lora, mult = lora_cases.build(name)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, name + ".safetensors")
            save_file({k: v.contiguous() for k, v in lora.items()}, path)
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                merged = ref_lora.merge_lora_to_state_dict(dict(sd), path, mult, torch.device("cpu"))
        for k, v in merged.items():
            changed = not (v.dtype == sd[k].dtype and torch.equal(v, sd[k]))
            out[f"{name}/{k}/changed"] = np.array([int(changed)])
            out[f"{name}/{k}/dtype"] = np.array([str(v.dtype)])
            if changed:
                out[f"{name}/{k}/sample"] = lora_cases.sample(v).numpy()
                out[f"{name}/{k}/sum"] = np.array([v.double().sum().item(), v.double().abs().sum().item()])
    np.savez_compressed(os.path.join(HERE, "lora.npz"), **out)
    print("lora.npz", len(out), "arrays")


# ----------------------------------------------------------------------------- tensor helpers of the section loop
def utils_cases():
    """Seeded inputs shared with tests/test_sections_cpu.py (which imports this function's twin from
    tests/utils_cases.py - kept there so the test does not import this generator)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import utils_cases as UC
    return UC


def gen_utils():
    import types
    for name in ("cv2", "torchvision"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    from diffusers_helper import utils as ref_utils
    UC = utils_cases()
    out = {}
    for name, (h, c, ov) in UC.soft_append_cases().items():
        out[f"soft_append/{name}"] = ref_utils.soft_append_bcthw(h, c, ov).float().numpy()
    for name, (x, length) in UC.crop_or_pad_cases().items():
        y, m = ref_utils.crop_or_pad_yield_mask(x, length)
        out[f"crop_or_pad/{name}/y"] = y.float().numpy()
        out[f"crop_or_pad/{name}/mask"] = m.numpy()
    for name, (x, b) in UC.repeat_cases().items():
        try:
            out[f"repeat/{name}"] = ref_utils.repeat_to_batch_size(x, b).numpy()
        except ValueError as e:
            out[f"repeat/{name}/error"] = np.array([str(e)])
    np.savez_compressed(os.path.join(HERE, "utils.npz"), **out)
    print("utils.npz", len(out), "arrays")


# ----------------------------------------------------------------------------- the sampling pipeline
def gen_pipeline():
    """sample_hunyuan of the reference (noise draw, sigma schedule, CFG packing, UniPC) on tests/pipeline_cases.py.  The
    module imports diffusers_helper/utils.py: same two empty stand-in modules as gen_utils."""
    import types
    for name in ("cv2", "torchvision"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    from diffusers_helper.pipelines import k_diffusion_hunyuan as ref_pipe
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pipeline_cases as PC
    out = {}
    for i, (name, kw) in enumerate(PC.cases().items()):
        steps = []
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            res = ref_pipe.sample_hunyuan(transformer=PC.StubTransformer(), generator=torch.Generator().manual_seed(40 + i),
                                          callback=lambda d: steps.append(d["denoised"].float().clone().numpy()), **kw)
        out[f"{name}/result"] = res.float().numpy()
        out[f"{name}/denoised"] = np.stack(steps)
    for n in (1, 1024, 14400, 100000):
        out[f"mu/{n}"] = np.array([ref_pipe.calculate_flux_mu(n, exp_max=7.0)])
    out["sigmas/25_mu1.2"] = ref_pipe.get_flux_sigmas_from_mu(25, 1.2).numpy()
    np.savez_compressed(os.path.join(HERE, "pipeline.npz"), **out)
    print("pipeline.npz", len(out), "arrays")


# ----------------------------------------------------------------------------- VAE entry points of the worker
def gen_hunyuan():
    """diffusers_helper/hunyuan.py: vae_decode_fake (the preview projection) and the vae_decode / vae_encode wrappers
    (scaling factor, image_mode's per-frame decode) on the stub VAE of tests/utils_cases.py.  The module's one diffusers
    import (a prompt template the prompt encoder uses; out of scope, never called) is met by an empty dict."""
    import types
    name = "diffusers.pipelines.hunyuan_video.pipeline_hunyuan_video"
    parts = name.split(".")
    for i in range(2, len(parts) + 1):
        sys.modules.setdefault(".".join(parts[:i]), types.ModuleType(".".join(parts[:i])))
    sys.modules[name].DEFAULT_PROMPT_TEMPLATE = {}
    from diffusers_helper import hunyuan as ref_h
    UC = utils_cases()
    out = {}
    for cname, z in UC.preview_cases().items():
        out[f"fake/{cname}"] = ref_h.vae_decode_fake(z).float().numpy()
    vae = UC.StubVAE()
    z = UC.preview_cases()["f32"]
    out["decode/video"] = ref_h.vae_decode(z, vae).float().numpy()
    out["decode/image_mode"] = ref_h.vae_decode(z, vae, image_mode=True).float().numpy()
    out["encode"] = ref_h.vae_encode(UC.stub_image(), vae).float().numpy()
    np.savez_compressed(os.path.join(HERE, "hunyuan.npz"), **out)
    print("hunyuan.npz", len(out), "arrays")


# ----------------------------------------------------------------------------- resolution buckets
def gen_buckets():
    """diffusers_helper/bucket_tools.py (imports as is): find_nearest_bucket over a sweep of sizes and resolutions."""
    from diffusers_helper import bucket_tools as ref_b
    UC = utils_cases()
    q = UC.bucket_queries()
    with contextlib.redirect_stdout(io.StringIO()):
        got = np.array([ref_b.find_nearest_bucket(h, w, resolution=r) for h, w, r in q], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "buckets.npz"), queries=np.array(q, dtype=np.int64), buckets=got)
    print("buckets.npz", got.shape)


if __name__ == "__main__":
    gen_buckets()
    gen_hunyuan()
    gen_pipeline()
    gen_utils()
    gen_rope()
    gen_sampler()
    gen_fp8()
    gen_lora()
    gen_dit_tiny()
