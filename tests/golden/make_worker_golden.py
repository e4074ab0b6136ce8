"""Golden vectors of the section loop, from the REFERENCE's own framepack/worker.py.

Build container only (needs /root/reference):   python tests/golden/make_worker_golden.py  -> tests/golden/worker.npz

`worker()` is one 800-line function: its bookkeeping (latent-padding schedule, index split, clean latents from the
history, end-frame latent, per-section start images, start-latent prepend, which latent frames are decoded and how they
are cross-faded) cannot be called piecewise.  So the whole function is run, with every collaborator that is NOT
bookkeeping replaced in the module's namespace by a small deterministic function that records what it is handed:
  sample_hunyuan   -> records frames / width / height, the index tensors and the clean latents of every section and
                      returns latents drawn from the generator the worker passes (so the generator's state is pinned too)
  vae_encode / vae_decode / encode_prompt_conds / hf_clip_vision_encode -> fixed functions of their inputs
  save_bcthw_as_mp4 -> records the pixel history instead of writing a file
  models, stream, performance_tracker -> attribute bags (the tracker reads CUDA memory statistics)
Everything else the worker touches runs as it is (bucket selection, PIL resize, crop_or_pad_yield_mask,
soft_append_bcthw, its own helpers).  Import-time needs of modules that are out of scope: `diffusers_helper.memory` asks
for the current CUDA device when imported (answered with 0 for the duration of the import: only a torch.device object is
made of it); cv2 / torchvision / diffusers are absent (empty module objects / tests/golden/diffusers_standin.py, as in
make_golden.py).  tests/test_sections_cpu.py replays the recorded cases through framepack_amd.sections.
"""
import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")

import diffusers_standin  # noqa: E402

diffusers_standin.install()
for _name in ("cv2", "torchvision"):
    sys.modules.setdefault(_name, types.ModuleType(_name))
_tmpl = "diffusers.pipelines.hunyuan_video.pipeline_hunyuan_video"
_parts = _tmpl.split(".")
for _i in range(2, len(_parts) + 1):
    sys.modules.setdefault(".".join(_parts[:_i]), types.ModuleType(".".join(_parts[:_i])))
sys.modules[_tmpl].DEFAULT_PROMPT_TEMPLATE = {}

_cur = torch.cuda.current_device
torch.cuda.current_device = lambda: 0
try:
    with contextlib.redirect_stdout(io.StringIO()):
        import framepack.worker as W  # noqa: E402
finally:
    torch.cuda.current_device = _cur

import worker_cases as WC  # noqa: E402


def run_case(case):
    rec = dict(samples=[], decodes=[], saves=[], encodes=[])

    def sample_hunyuan(**kw):
        return WC.stub_sample(rec, **kw)

    def vae_encode(image, vae):
        rec["encodes"].append(tuple(image.shape))
        return WC.stub_encode_hw(len(rec["encodes"]) - 1, image.shape[-2] // 8, image.shape[-1] // 8)

    def vae_decode(latents, vae, image_mode=False):
        rec["decodes"].append(latents.clone())
        return WC.stub_decode(latents)

    def save(x, output_filename, fps=10, crf=0):
        rec["saves"].append(x.clone())
        return x

    W.sample_hunyuan, W.vae_encode, W.vae_decode, W.save_bcthw_as_mp4 = sample_hunyuan, vae_encode, vae_decode, save
    W.encode_prompt_conds = lambda prompt, *a, **k: WC.stub_prompt(prompt)
    W.hf_clip_vision_encode = lambda img, fe, enc: types.SimpleNamespace(last_hidden_state=torch.full((1, 5, 16), 0.25))
    W.aggressive_memory_cleanup = lambda *a, **k: None

    class Tracker:                                   # performance_tracker reads CUDA memory statistics: timings only
        def track_memory(self, *a, **k):
            return dict(current_gb=0.0, free_gb=0.0, peak_gb=0.0, total_gb=0.0)

        def __getattr__(self, name):
            return lambda *a, **k: 0.0
    W.performance_tracker = Tracker()
    W.time.sleep = lambda s: None
    tr = types.SimpleNamespace(dtype=torch.float32, to=lambda *a, **k: None)
    models = types.SimpleNamespace(high_vram=True, transformer=tr, vae=None, text_encoder=None, text_encoder_2=None,
                                   tokenizer=None, tokenizer_2=None, image_encoder=None, feature_extractor=None,
                                   prepare_for_inference=lambda *a, **k: None)
    q = types.SimpleNamespace(top=lambda: None, pop=lambda: None, push=lambda item: None)
    stream = types.SimpleNamespace(input_queue=q, output_queue=q)
    with tempfile.TemporaryDirectory() as td, contextlib.redirect_stdout(io.StringIO()), \
            contextlib.redirect_stderr(io.StringIO()):
        W.worker(input_image=WC.image(case["image_hw"], 1), end_frame=None,
                 prompt="a prompt", n_prompt=case.get("n_prompt", ""), seed=case["seed"],
                 total_latent_sections=case["sections"], latent_window_size=case["window"], steps=3, cfg=1.0, gs=10.0,
                 rs=0.0, gpu_memory_preservation=6, use_teacache=False, teacache_thresh=0.15,
                 resolution_scale=case.get("resolution_scale", "Half (0.5x)"), mp4_crf=16, keep_section_videos=True,
                 end_frame_strength=case.get("end_frame_strength", 1.0), movement_scale=0.1,
                 section_settings=WC.section_settings(case), one_frame_inference=case.get("one_frame", False),
                 models=models, stream=stream, outputs_folder=td)
    return rec


def main():
    out = {}
    for name, case in WC.cases().items():
        rec = run_case(case)
        assert rec["samples"] and rec["saves"], name
        out[f"{name}/n_sections"] = np.array([len(rec["samples"])])
        for i, s in enumerate(rec["samples"]):
            for k, v in s.items():
                out[f"{name}/sample{i}/{k}"] = v
        for i, d in enumerate(rec["decodes"]):
            out[f"{name}/decode{i}"] = WC.sub(d)
        out[f"{name}/n_decodes"] = np.array([len(rec["decodes"])])
        out[f"{name}/pixels"] = WC.sub(rec["saves"][-1])
        out[f"{name}/pixel_frames"] = np.array([int(x.shape[2]) for x in rec["saves"]])
        out[f"{name}/encodes"] = np.array([len(rec["encodes"])])
    np.savez_compressed(os.path.join(HERE, "worker.npz"), **out)
    print("worker.npz", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "worker.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
