"""Section driver (framepack_amd/sections.py) against the oracle's restatement and the values the reference
documents: padding schedule, index layout, history update, decode windows and cross-fade.  The sampler and the
VAE are stand-ins with the right contracts (T latent frames -> 4(T-1)+1 frames), so only the bookkeeping is
under test here; the real kernels are covered by the GPU tests."""
import numpy as np
import pytest
import torch

from framepack_amd import sections as S
from oracle import sections_oracle as SO


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 9])
def test_latent_padding_schedule(n):
    assert S.latent_paddings(n) == SO.paddings(n)
    assert S.latent_paddings(n)[-1] == 0
    assert S.latent_paddings(n, one_frame_inference=True) == [0]
    assert S.latent_paddings(5) == [3, 2, 2, 1, 0] and S.latent_paddings(3) == [2, 1, 0] and S.latent_paddings(0) == [0]


@pytest.mark.parametrize("pad,window", [(0, 9), (1, 9), (3, 9), (2, 5), (0, 33)])
def test_section_indices_layout(pad, window):
    got, want = S.section_indices(pad, window), SO.indices(pad, window)
    assert got["latent_indices"][0].tolist() == want["latent"].tolist()
    assert got["clean_latent_indices"][0].tolist() == want["clean"].tolist()
    assert got["clean_latent_2x_indices"][0].tolist() == want["two"].tolist()
    assert got["clean_latent_4x_indices"][0].tolist() == want["four"].tolist()
    assert got["latent_indices"].dtype == torch.int64
    if (pad, window) == (0, 9):        # the last section at window 9 (from ref worker.py:463-465)
        assert got["latent_indices"][0].tolist() == list(range(1, 10))
        assert got["clean_latent_indices"][0].tolist() == [0, 10]
        assert got["clean_latent_2x_indices"][0].tolist() == [11, 12]
        assert got["clean_latent_4x_indices"][0].tolist() == list(range(13, 29))
    one = S.section_indices(pad, window, one_frame_inference=True)
    assert one["latent_indices"][0].tolist() == [want["latent"][-1]] and one["clean_latent_2x_indices"] is None


@pytest.mark.parametrize("overlap", [0, 1, 5, 33])
def test_soft_append_matches_oracle(overlap):
    g = torch.Generator().manual_seed(overlap)
    h, c = torch.randn(1, 3, 40, 4, 4, generator=g), torch.randn(1, 3, 37, 4, 4, generator=g)
    got = S.soft_append_bcthw(h, c, overlap)
    want = SO.crossfade(h[0].permute(1, 0, 2, 3).numpy(), c[0].permute(1, 0, 2, 3).numpy(), overlap)
    np.testing.assert_allclose(got[0].permute(1, 0, 2, 3).numpy(), want, rtol=1e-6, atol=1e-6)
    assert got.shape[2] == 40 + 37 - overlap
    with pytest.raises(AssertionError):
        S.soft_append_bcthw(h[:, :, :2], c, 5)


def _stubs(calls):
    def sample_fn(width, height, frames, generator, latent_indices, clean_latents, clean_latent_indices,
                  clean_latents_2x, clean_latent_2x_indices, clean_latents_4x, clean_latent_4x_indices, **kw):
        T = (frames + 3) // 4
        calls.append(dict(T=T, latent_indices=latent_indices[0].tolist(), clean=clean_latents.clone(),
                          idx_2x=clean_latent_2x_indices, idx_4x=clean_latent_4x_indices,
                          two=None if clean_latents_2x is None else clean_latents_2x.clone(),
                          four=None if clean_latents_4x is None else clean_latents_4x.clone()))
        return torch.full((1, 16, T, height // 8, width // 8), float(len(calls)))

    def decode_fn(lat):
        T = lat.shape[2]
        px = lat[:, :3].repeat_interleave(4, dim=2)[:, :, 3:]               # 4(T-1)+1 frames, value = latent value
        return px.repeat_interleave(8, dim=3).repeat_interleave(8, dim=4)
    return sample_fn, decode_fn


@pytest.mark.parametrize("n_sections,window", [(1, 9), (3, 9), (5, 4)])
def test_generate_sections_history_and_decode_windows(n_sections, window):
    calls = []
    sample_fn, decode_fn = _stubs(calls)
    start = torch.full((1, 16, 1, 4, 4), -1.0)
    end = torch.full((1, 16, 1, 4, 4), 7.0)
    res = S.generate_sections(sample_fn, decode_fn, start, 32, 32, n_sections, latent_window_size=window,
                              end_latent=end, end_frame_strength=0.5)
    counts = SO.expected_frame_counts(n_sections, window)
    assert [(s["total_latent_frames"], s["pixel_frames"]) for s in res.sections] == counts
    assert res.total_generated_latent_frames == counts[-1][0]
    assert tuple(res.history_latents.shape) == (1, 16, counts[-1][0] + 19, 4, 4)
    # first section: the end frame (scaled) is the post-context latent; later sections see the newest history
    assert torch.all(calls[0]["clean"][:, :, 0] == -1.0) and torch.all(calls[0]["clean"][:, :, 1] == 3.5)
    for i in range(1, len(calls)):
        assert torch.all(calls[i]["clean"][:, :, 1] == float(i))              # first frame of the previous section
        assert torch.all(calls[i]["two"] == float(i)) and calls[i]["four"].shape[2] == 16
    # newest section first in the history; the last section is preceded by the start latent
    assert torch.all(res.history_latents[:, :, 0] == -1.0)
    assert torch.all(res.history_latents[:, :, 1:1 + window] == float(n_sections))
    assert res.history_pixels.shape[2] == counts[-1][1] and res.history_pixels.shape[-1] == 32


def test_one_frame_inference_path():
    calls = []
    sample_fn, decode_fn = _stubs(calls)
    res = S.generate_sections(sample_fn, decode_fn, torch.zeros(1, 16, 1, 4, 4), 32, 32, 7, one_frame_inference=True)
    # one frame: no 2x / 4x INDICES (the latents themselves are still handed over, as the reference does; the
    # transformer needs both to use them)
    assert len(calls) == 1 and calls[0]["T"] == 1 and calls[0]["latent_indices"] == [9]
    assert calls[0]["idx_2x"] is None and calls[0]["idx_4x"] is None and calls[0]["two"] is not None
    assert res.total_generated_latent_frames == 1 and res.history_pixels.shape[2] == 1


def test_section_maps_are_keyed_by_ui_section_and_take_the_nearest_earlier_entry():
    """ref worker.py:354-367, 404-418: maps are keyed by UI section (counted from the start of the video); processing
    step i of n is UI section n - i - 1 and uses the largest key <= that.  Sparse map {0, 2} over 4 sections:
    steps 0, 1 (UI sections 3, 2) take key 2, steps 2, 3 (UI sections 1, 0) take key 0."""
    assert [S.pick_ui_section(i, {0, 2}, 4) for i in range(4)] == [2, 2, 0, 0]
    assert [S.pick_ui_section(i, {1}, 3) for i in range(3)] == [1, 1, None]
    assert S.pick_ui_section(0, [], 3) is None
    calls = []
    seen_kw = []

    def sample_fn(width, height, frames, generator, clean_latents, tag=None, **kw):
        calls.append(float(clean_latents[0, 0, 0, 0, 0]))
        seen_kw.append(tag)
        return torch.zeros(1, 16, (frames + 3) // 4, height // 8, width // 8)

    start = torch.full((1, 16, 1, 4, 4), -1.0)
    lat = {0: torch.full((1, 16, 1, 4, 4), 10.0), 2: torch.full((1, 16, 1, 4, 4), 12.0)}
    res = S.generate_sections(sample_fn, None, start, 32, 32, 4, section_latents=lat, sample_kwargs=dict(tag="default"),
                              section_sample_kwargs={2: dict(tag="late")})
    assert calls == [12.0, 12.0, 10.0, 10.0]
    assert seen_kw == ["late", "late", "default", "default"]
    assert [s["ui_section_latent"] for s in res.sections] == [2, 2, 0, 0]
    # a map with no key at or before a UI section leaves the start latent in place
    calls.clear()
    S.generate_sections(sample_fn, None, start, 32, 32, 3, section_latents={1: lat[2]})
    assert calls == [12.0, 12.0, -1.0]


# ---- the tensor helpers either side of the section loop, pinned by the reference's own functions: tests/golden/utils.npz
#      holds what diffusers_helper/utils.py returned for the seeded inputs of tests/utils_cases.py (make_golden.gen_utils)
def _utils_golden():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "utils.npz"))


def test_soft_append_bit_identical_to_the_reference():
    import utils_cases as UC
    G = _utils_golden()
    for name, (h, c, ov) in UC.soft_append_cases().items():
        got = S.soft_append_bcthw(h, c, ov)
        assert got.dtype == h.dtype, name
        assert np.array_equal(got.float().numpy(), G[f"soft_append/{name}"]), name
    with pytest.raises(AssertionError, match="History length"):
        S.soft_append_bcthw(torch.zeros(1, 3, 2, 2, 2), torch.zeros(1, 3, 5, 2, 2), 3)
    with pytest.raises(AssertionError, match="Current length"):
        S.soft_append_bcthw(torch.zeros(1, 3, 5, 2, 2), torch.zeros(1, 3, 2, 2, 2), 3)


def test_crop_or_pad_yield_mask_bit_identical_to_the_reference():
    import utils_cases as UC
    G = _utils_golden()
    for name, (x, length) in UC.crop_or_pad_cases().items():
        y, m = S.crop_or_pad_yield_mask(x, length)
        assert y.dtype == x.dtype and m.dtype == torch.bool, name
        assert np.array_equal(y.float().numpy(), G[f"crop_or_pad/{name}/y"]), name
        assert np.array_equal(m.numpy(), G[f"crop_or_pad/{name}/mask"]), name


def test_repeat_to_batch_size_bit_identical_to_the_reference():
    import utils_cases as UC
    from framepack_amd.sampler import repeat_to_batch_size
    G = _utils_golden()
    for name, (x, b) in UC.repeat_cases().items():
        if f"repeat/{name}/error" in G.files:
            with pytest.raises(ValueError) as e:
                repeat_to_batch_size(x, b)
            assert str(e.value) == str(G[f"repeat/{name}/error"][0]), name
        else:
            assert np.array_equal(repeat_to_batch_size(x, b).numpy(), G[f"repeat/{name}"]), name
    assert repeat_to_batch_size(None, 3) is None


def test_vae_entry_points_bit_identical_to_the_reference(golden_dir):
    """tests/golden/hunyuan.npz: the reference's diffusers_helper/hunyuan.py - vae_decode_fake (latent -> RGB preview)
    and the vae_decode / vae_encode wrappers (scaling factor, image_mode) on the stub VAE of tests/utils_cases.py."""
    import os
    import utils_cases as UC
    from framepack_amd.vae import vae_decode, vae_decode_fake, vae_encode
    G = np.load(os.path.join(golden_dir, "hunyuan.npz"))
    for name, z in UC.preview_cases().items():
        got = vae_decode_fake(z)
        assert got.dtype == z.dtype and np.array_equal(got.float().numpy(), G[f"fake/{name}"]), name
    vae, z = UC.StubVAE(), UC.preview_cases()["f32"]
    assert np.array_equal(vae_decode(z, vae).float().numpy(), G["decode/video"])
    assert np.array_equal(vae_decode(z, vae, image_mode=True).float().numpy(), G["decode/image_mode"])
    assert np.array_equal(vae_encode(UC.stub_image(), vae).float().numpy(), G["encode"])


def test_find_nearest_bucket_matches_the_reference(golden_dir):
    """tests/golden/buckets.npz: the reference's bucket_tools.find_nearest_bucket over 363 (h, w, resolution) queries
    (ties, both orientations, the precomputed 640 / 320 tables and scaled ones)."""
    import os
    import utils_cases as UC
    G = np.load(os.path.join(golden_dir, "buckets.npz"))
    q = UC.bucket_queries()
    assert np.array_equal(np.array(q, dtype=np.int64), G["queries"])
    got = np.array([S.find_nearest_bucket(h, w, resolution=r) for h, w, r in q], dtype=np.int64)
    assert np.array_equal(got, G["buckets"])
    assert S.find_nearest_bucket(640, 640) == (640, 608)        # no square bucket: the later of the two nearest wins


# ---- the section loop against the REFERENCE's own worker: tests/golden/worker.npz holds what framepack/worker.py handed
#      to its sampler / decoder / video writer when run on the cases of tests/worker_cases.py with recording stand-ins for
#      those collaborators (tests/golden/make_worker_golden.py).  The same stand-ins are driven here by generate_sections.
def _replay(case):
    import worker_cases as WC
    H, W = case["image_hw"]
    h, w = S.find_nearest_bucket(H, W, resolution=640 if case.get("resolution_scale") == "Full (1x)" else 320)
    encodes = [WC.stub_encode_hw(0, h // 8, w // 8)]

    def prompt_kwargs(text):
        vec, pool = WC.stub_prompt(text)
        vec, mask = S.crop_or_pad_yield_mask(vec, length=512)
        return dict(prompt_embeds=vec, prompt_embeds_mask=mask, prompt_poolers=pool)

    section_latents, section_kwargs = {}, {}
    for num, has_img, text in case.get("ui", []):
        if has_img:
            encodes.append(WC.stub_encode_hw(len(encodes), h // 8, w // 8))
            section_latents[num] = encodes[-1]
        # a UI row without a prompt falls back to the default prompt (worker.py:372-373), it does not look further back
        section_kwargs[num] = prompt_kwargs(text) if text.strip() else {}
    rec = dict(samples=[], decodes=[])

    def decode_fn(lat):
        rec["decodes"].append(lat.clone())
        return WC.stub_decode(lat)

    res = S.generate_sections(lambda **kw: WC.stub_sample(rec, **kw), decode_fn, encodes[0], h, w, case["sections"],
                              latent_window_size=case["window"], seed=case["seed"],
                              one_frame_inference=case.get("one_frame", False), section_latents=section_latents or None,
                              sample_kwargs=prompt_kwargs("a prompt"), section_sample_kwargs=section_kwargs or None)
    return rec, res, len(encodes)


def test_section_loop_hands_over_what_the_reference_worker_does(golden_dir):
    import os
    import worker_cases as WC
    G = np.load(os.path.join(golden_dir, "worker.npz"))
    for name, case in WC.cases().items():
        rec, res, n_enc = _replay(case)
        assert len(rec["samples"]) == int(G[f"{name}/n_sections"][0]), name
        assert len(rec["decodes"]) == int(G[f"{name}/n_decodes"][0]) and n_enc == int(G[f"{name}/encodes"][0]), name
        for i, s in enumerate(rec["samples"]):
            for k, v in s.items():
                want = G[f"{name}/sample{i}/{k}"]
                assert v.shape == want.shape and np.array_equal(v, want), (name, i, k)
        for i, d in enumerate(rec["decodes"]):                 # which latent frames are decoded, bit for bit
            assert np.array_equal(WC.sub(d), G[f"{name}/decode{i}"]), (name, i)
        assert [s["pixel_frames"] for s in res.sections] == G[f"{name}/pixel_frames"].tolist(), name
        assert np.array_equal(WC.sub(res.history_pixels), G[f"{name}/pixels"]), name


def test_sections_oracle_is_pinned_by_the_reference_worker_too(golden_dir):
    """oracle/sections_oracle.py (the independent restatement the GPU composition test checks against): its padding
    schedule and index layout against what the reference's worker handed to its sampler (tests/golden/worker.npz)."""
    import os
    import worker_cases as WC
    G = np.load(os.path.join(golden_dir, "worker.npz"))
    for name, case in WC.cases().items():
        if case.get("one_frame"):
            continue                                   # (the oracle restates the video mode)
        pads = SO.paddings(case["sections"])
        assert len(pads) == int(G[f"{name}/n_sections"][0]), name
        for i, pad in enumerate(pads):
            ix = SO.indices(pad, case["window"])
            for mine, theirs in (("latent", "latent_indices"), ("clean", "clean_latent_indices"),
                                 ("two", "clean_latent_2x_indices"), ("four", "clean_latent_4x_indices")):
                assert ix[mine].tolist() == G[f"{name}/sample{i}/{theirs}"][0].tolist(), (name, i, mine)
        # frame counts of the pixel history after every section: 4 (latent frames) - 3
        frames = np.cumsum([case["window"] + (1 if p == 0 else 0) for p in pads]) * 4 - 3
        assert frames.tolist() == G[f"{name}/pixel_frames"].tolist(), name
