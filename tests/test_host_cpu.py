"""CPU-side checks of the host mirror: state-dict names/shapes, bit-exact packing tables, the C-ABI library
exporting every declared symbol, loud failure without a GPU.  No GPU compute happens here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from framepack_amd import lib, packing, synthetic
from framepack_amd.model import HunyuanVideoTransformer3DModelPacked
from oracle import dit_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tiny_model():
    m = HunyuanVideoTransformer3DModelPacked(**synthetic.CONFIG_TINY)
    synthetic.init_synthetic_(dict(m.state_dict()), seed=0)
    return m


def test_state_dict_names_and_shapes_match_reference_layout(tiny_model):
    shapes = O.param_shapes(O.DiTConfig(**synthetic.CONFIG_TINY))
    sd = tiny_model.state_dict()
    assert set(sd) == set(shapes)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k


def test_13b_layout_has_the_documented_dimensions():
    shapes = O.param_shapes(O.DiTConfig(**synthetic.CONFIG_13B))
    # ref utils/lora_utils.py:195-206: hidden 3072, fused single-block width 21504 = 3*3072 + 12288
    assert shapes["single_transformer_blocks.39.proj_out.weight"] == (3072, 15360)
    assert shapes["transformer_blocks.19.norm1.linear.weight"] == (18432, 3072)
    n = sum(int(np.prod(s)) for s in shapes.values())
    assert 12.5e9 < n < 13.5e9


def test_synthetic_init_equals_oracle_params(tiny_model):
    P = O.synth_params(O.DiTConfig(**synthetic.CONFIG_TINY), seed=0)
    sd = tiny_model.state_dict()
    for k in P:
        assert torch.equal(sd[k], P[k]), k


def test_fp8_targets_are_linear_modules(tiny_model):
    """ref utils/fp8_optimization_utils.py:258-272 patches nn.Linear instances found by these names."""
    mods = dict(tiny_model.named_modules())
    for k in tiny_model.state_dict():
        if ("transformer_blocks" in k) and k.endswith(".weight") and "norm" not in k:
            assert isinstance(mods[k[:-len(".weight")]], torch.nn.Linear), k


@pytest.mark.parametrize("name,pool", [("noisy", 1), ("clean", 1), ("x2", 2), ("x4", 4), ("x2_odd", 2), ("x4_odd", 4),
                                       ("x4_pad", 4)])
def test_rope_tables_bit_exact(golden_dir, name, pool):
    G = np.load(os.path.join(golden_dir, "rope.npz"))
    h, w = (int(v) for v in G[name + "_hw"])
    t = packing.rope_tokens(torch.from_numpy(G[name + "_idx"]), h, w, pool)
    assert np.array_equal(t.numpy(), G[name])


def test_rope_cache_concatenates_streams_in_packed_order(golden_dir):
    G = np.load(os.path.join(golden_dir, "rope.npz"))
    c = packing.RopeCache((16, 56, 56), 256.0)
    streams = [(torch.from_numpy(G["x4_idx"]), 4), (torch.from_numpy(G["x2_idx"]), 2),
               (torch.from_numpy(G["clean_idx"]), 1), (torch.from_numpy(G["noisy_idx"]), 1)]
    t = c.get(streams, 6, 8, "cpu")
    want = np.concatenate([G["x4"], G["x2"], G["clean"], G["noisy"]], axis=1)
    assert np.array_equal(t.numpy(), want)
    assert c.get(streams, 6, 8, "cpu") is t                       # cached


def test_replicate_pad_and_patch_rows_match_conv3d(golden_dir):
    G = np.load(os.path.join(golden_dir, "rope.npz"))
    r = torch.arange(2 * 3 * 5 * 7, dtype=torch.float32).view(1, 2, 3, 5, 7)
    assert np.array_equal(packing.pad_to_multiple(r, (2, 4, 8)).numpy(), G["pad_248"])
    g = torch.Generator("cpu").manual_seed(0)
    x = torch.randn(1, 16, 4, 8, 16, generator=g)
    w = torch.randn(32, 16, 2, 4, 4, generator=g)
    want = torch.nn.functional.conv3d(x, w, stride=(2, 4, 4)).flatten(2).transpose(1, 2)[0]
    got = packing.patch_rows(x, (2, 4, 4)) @ w.flatten(1).T
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)


def test_unpatchify_matches_einops_pattern():
    import einops
    x = torch.arange(1 * 24 * 64, dtype=torch.float32).view(1, 24, 64)
    want = einops.rearrange(x, "b (t h w) (c pt ph pw) -> b c (t pt) (h ph) (w pw)", t=2, h=3, w=4, pt=1, ph=2, pw=2)
    assert torch.equal(packing.unpatchify(x[0], 1, 2, 3, 4, 16, 1, 2), want)


def test_cu_seqlens_layout():
    mask = torch.zeros(2, 8, dtype=torch.bool)
    mask[0, :3] = True
    mask[1, :] = True
    assert packing.get_cu_seqlens(mask, 10).tolist() == O.get_cu_seqlens(mask, 10).tolist() == [0, 13, 18, 36, 36]


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "framepack_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t)\s+(fp_\w+)\s*\(", header, flags=re.M))
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    so = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(so, name), name
    assert lib.load().fp_abi_version() == 5


def test_forward_refuses_cpu_tensors(tiny_model):
    kw = synthetic.make_inputs(frames_latent=1, height=64, width=64, n_text=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tiny_model.to(torch.bfloat16)(return_dict=False, **kw)


def test_bad_rank_raises_value_error(tiny_model):
    with pytest.raises(ValueError, match="Unexpected hidden_states shape"):
        tiny_model(torch.zeros(2, 2, 2), None, None, None, None, None)


# ---------------------------------------------------------------- VAE decode boundary (host logic, no GPU)
def test_vae_state_dict_names_and_plan_match_oracle_layout():
    from framepack_amd.vae import AutoencoderKLHunyuanVideo, _down_block_plan, _up_block_plan
    from oracle import vae_oracle as VO
    cfg = VO.VAEConfig()
    vae = AutoencoderKLHunyuanVideo()
    small_cfg = VO.VAEConfig(block_out_channels=(32, 32, 64, 64))
    want = VO.random_state_dict(small_cfg, with_encoder=True)                             # names only (small)
    small = AutoencoderKLHunyuanVideo(block_out_channels=(32, 32, 64, 64))
    assert set(small.state_dict().keys()) == set(want.keys())
    assert {k: tuple(v.shape) for k, v in small.state_dict().items()} == {k: tuple(v.shape) for k, v in want.items()}
    assert _up_block_plan(cfg.block_out_channels, 8, 4) == VO.up_block_plan(cfg) == [
        (512, 512, (1, 2, 2)), (512, 512, (2, 2, 2)), (512, 256, (2, 2, 2)), (256, 128, None)]
    assert _down_block_plan(cfg.block_out_channels, 8, 4) == VO.down_block_plan(cfg) == [
        (128, 128, (1, 2, 2)), (128, 256, (2, 2, 2)), (256, 512, (2, 2, 2)), (512, 512, None)]
    assert vae.config.scaling_factor == 0.476986 and vae.use_framewise_decoding
    # a full checkpoint loads strictly into both halves; a decoder-only one strictly into the decode half
    small.load_state_dict(want, strict=True)
    assert small._encoder_loaded
    dec_only = VO.random_state_dict(small_cfg)
    assert not any(k.startswith("encoder.") or k.startswith("quant_conv.") for k in dec_only)
    small.load_state_dict(dec_only, strict=True)
    assert not small._encoder_loaded
    with pytest.raises(RuntimeError):
        small.encode(torch.zeros(1, 3, 1, 64, 64))                    # CPU tensor (and no encoder weights)
    for broken in (dict(dec_only), dict(want)):
        del broken["decoder.conv_out.conv.bias"]
        with pytest.raises(RuntimeError):
            small.load_state_dict(broken, strict=True)
    half = dict(want)
    del half["encoder.conv_out.conv.bias"]                            # an encoder half must be complete too
    with pytest.raises(RuntimeError):
        small.load_state_dict(half, strict=True)


def test_vae_decode_refuses_cpu_and_preview_matches_projection():
    from framepack_amd.vae import AutoencoderKLHunyuanVideo, vae_decode, vae_decode_fake, _LATENT_RGB_FACTORS, _LATENT_RGB_BIAS
    vae = AutoencoderKLHunyuanVideo(block_out_channels=(32, 32, 64, 64))
    with pytest.raises(RuntimeError):
        vae_decode(torch.zeros(1, 16, 1, 2, 2), vae)
    z = torch.randn(1, 16, 2, 3, 3, generator=torch.Generator().manual_seed(0))
    want = (torch.einsum("bcthw,cr->brthw", z, torch.tensor(_LATENT_RGB_FACTORS)) +
            torch.tensor(_LATENT_RGB_BIAS)[None, :, None, None, None]).clamp(0, 1)
    torch.testing.assert_close(vae_decode_fake(z), want, rtol=1e-5, atol=1e-6)


def test_vae_oracle_framewise_equals_whole_decode_when_short():
    from oracle import vae_oracle as VO
    cfg = VO.VAEConfig(block_out_channels=(32, 32, 64, 64))
    P = VO.random_state_dict(cfg)
    z = torch.randn(1, 16, 4, 2, 2, generator=torch.Generator().manual_seed(1))
    assert torch.equal(VO.decode(P, cfg, z, framewise=True), VO.decode(P, cfg, z, framewise=False))   # T <= 4: one window
    z = torch.randn(1, 16, 7, 2, 2, generator=torch.Generator().manual_seed(2))
    out = VO.decode(P, cfg, z, framewise=True)
    assert tuple(out.shape) == (1, 3, 25, 16, 16)
    mask = VO.frame_causal_mask(2, 3, torch.float32)
    assert mask[:3, 3:].eq(float("-inf")).all() and mask[3:, :].eq(0).all() and mask[:3, :3].eq(0).all()


def test_w4_gemm_keeps_compiler_out_of_the_agprs(tmp_path):
    """gemm_w4.h addresses its 256 accumulator registers (AGPRs) by number from inline asm; the compiler does not
    know they are live, so it must never use an AGPR itself (it would, as spill space, if a kernel needed more than
    256 VGPRs).  Disassemble the built code objects and check: inside gemm_w4_kernel the only AGPR traffic is the
    MFMAs and the epilogue's v_accvgpr_read."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    so = os.path.join(ROOT, "framepack_amd", "libframepack_hip.so")
    if not (os.path.exists(objdump) and os.path.exists(so)):
        pytest.skip("needs the ROCm llvm-objdump and the built library")
    local = shutil.copy(so, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", local], check=True, capture_output=True, cwd=tmp_path)
    seen = 0
    for name in os.listdir(tmp_path):
        if "gfx950" not in name:
            continue
        dis = subprocess.run([objdump, "-d", str(tmp_path / name)], check=True, capture_output=True, text=True).stdout
        cur = None
        for line in dis.splitlines():
            if line.endswith(">:"):
                cur = line
                seen += "gemm_w4_kernel" in line
            elif cur and "gemm_w4_kernel" in cur:
                assert "v_accvgpr_write" not in line and "v_accvgpr_mov" not in line, (cur, line)
                assert "scratch_" not in line, (cur, line)
    assert seen >= 9          # bf16 x 5 epilogues, f16 dense x 2, f16 conv x 2


def test_attention_kernels_keep_compiler_out_of_the_agprs(tmp_path):
    """attn_fwd.hip keeps O^T and Q' in AGPRs addressed by number from inline asm; the compiler does not know they are
    live.  A kernel that needs more than 256 VGPRs gets the excess parked in AGPRs by the register allocator (seen once:
    spilled pointers, zeroed by the accumulator initialisation, faulted the GPU).  Compile the file to assembly with the
    Makefile's flags and check, per attention kernel: every v_accvgpr_* instruction sits inside an inline-asm block,
    nothing goes to scratch memory, and the VGPR count leaves the AGPR file alone."""
    import re
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "framepack_amd", "csrc", "attn_fwd.hip")
    if not os.path.exists(hipcc):
        pytest.skip("needs hipcc")
    mk = open(os.path.join(ROOT, "framepack_amd", "csrc", "Makefile")).read()
    flags = re.search(r"^FLAGS_attn_fwd = (.*)$", mk, re.M).group(1).split()
    out = tmp_path / "attn.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", *flags, "-S", "--cuda-device-only", src, "-o",
                    str(out)], check=True, capture_output=True)
    text = out.read_text()
    seen = 0
    for m in re.finditer(r"^(_ZN\S*attn_(?:w64|v3)_kernel\S*):[^\n]*\n(.*?)^\s*\.amdhsa_kernel \1", text, re.S | re.M):
        seen += 1
        name, body = m.group(1), m.group(2)
        in_asm = False
        for line in body.splitlines():
            if "#ASMSTART" in line:
                in_asm = True
            elif "#ASMEND" in line:
                in_asm = False
            elif not in_asm:
                assert "v_accvgpr" not in line, (name, line)
                assert "scratch_" not in line, (name, line)
        assert int(re.search(re.escape(name) + r"\.num_vgpr, (\d+)", text).group(1)) <= 256
        assert int(re.search(re.escape(name) + r"\.private_seg_size, (\d+)", text).group(1)) == 0
    assert seen == 2


def test_gemm_fragment_reads_all_feed_an_mfma(tmp_path):
    """The 16-bit GEMM / convolution main loop reads its MFMA fragments with inline asm (the compiler's wait counts do
    not see the loop's own s_waitcnt, gemm_w4.h).  The compiler takes an asm output for written at once: a fragment that
    no MFMA reads is dead to it, its register is handed out again, and the LDS data that lands later overwrites the new
    owner (seen once: the narrow-tile kernels read W fragments they never multiply, addresses were overwritten, the GPU
    faulted).  Compile to assembly and check, per 4-wave kernel: every inline-asm ds_read_b128 destination is read by
    an MFMA before anything writes that register again."""
    import re
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("needs hipcc")
    seen = 0
    files = ("gemm_bf16.hip", "vae_kernels.hip")
    procs = []                                # (the two compilations run side by side: about a minute each)
    for fname in files:
        src = os.path.join(ROOT, "framepack_amd", "csrc", fname)
        procs.append(subprocess.Popen([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", src,
                                       "-o", str(tmp_path / (fname + ".s"))], stdout=subprocess.DEVNULL,
                                      stderr=subprocess.DEVNULL))
    for fname, pr in zip(files, procs):
        assert pr.wait() == 0, fname
        text = (tmp_path / (fname + ".s")).read_text()
        for m in re.finditer(r"^(_ZN7fp_gemm14gemm_w4_kernel\S*):[^\n]*\n(.*?)^\s*\.amdhsa_kernel \1", text, re.S | re.M):
            name, body = m.group(1), m.group(2)
            seen += 1
            pending = {}                      # first register of an asm fragment read -> not yet used by an MFMA
            in_asm = False
            for line in body.splitlines():
                t = line.strip()
                if "#ASMSTART" in t:
                    in_asm = True
                    continue
                if "#ASMEND" in t:
                    in_asm = False
                    continue
                if t.startswith("v_mfma"):
                    for r in re.findall(r"v\[(\d+):\d+\]", t):
                        pending.pop(int(r), None)
                    continue
                rd = re.match(r"ds_read_b128 v\[(\d+):(\d+)\]", t) if in_asm else None
                if rd:
                    lo = int(rd.group(1))
                    assert lo not in pending, (name, "a fragment read is overwritten before any MFMA used it", t)
                    pending[lo] = t
                elif pending and not in_asm:
                    # any compiler instruction that WRITES a register of a pending fragment
                    w = re.match(r"(?:v_|ds_read|global_load|buffer_load)\S*\s+v(?:\[(\d+):(\d+)\]|(\d+))", t)
                    if w and not t.startswith(("v_cmp", "v_mfma")):
                        lo = int(w.group(1) or w.group(3))
                        hi = int(w.group(2) or w.group(3))
                        for p in list(pending):
                            assert not (lo <= p + 3 and hi >= p), (name, "register of an in-flight fragment read reused", t)
    assert seen >= 12
