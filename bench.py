"""bench.py - seconds per 25-step section of the 13B packed DiT (BASELINE.json metric) on N MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one denoising step of the section: one forward of HunyuanVideoTransformer3DModelPacked (TeaCache
off, no negative prompt: ref wrapper.py:53-58) plus the flow-match UniPC update that drives it
(framepack_amd/sampler.py).  Workload = BASELINE.json configs[1]: 13B model, 640x640 bucket, 33-frame section
(9 latent frames), packed sequence S = 18 400 image + 729 image-projection + 256 text tokens = 19 385,
random-init bf16 weights and synthetic inputs resident in HBM.  W warm-up steps, then exactly K timed steps
bracketed by barrier + torch.cuda.synchronize(); the reported time is the max over ranks.
value = seconds per 25-step section = 25 * (timed seconds / K).

N > 1: Ulysses sequence parallelism (image tokens sharded; per attention the heads go through RCCL all-to-alls in
chunks, a chunk's exchange running under the other chunk's attention where that keeps the chip full -
framepack_amd/ulysses.py); the whole section is ONE job spread over N GPUs, so scaling is "strong".

Also printed on the same JSON line:
  roofline     - the dominant kernel (K1 flash attention): algorithmic FLOPs per launch (4 * S^2 * heads * 128,
                 per rank) / its average duration measured live with events on its own stream over the timed
                 region, against the 2.5 PFLOP/s dense bf16 MFMA peak.
  short_prompt - (N == 1, informational) ms per step of the same section with a 64-token prompt (S = 19 193).
  vae_decode   - (N == 1, informational; not part of `value`) the causal-3D VAE decode of the section's 9 latent
                 frames to 33 frames at 640x640 (framepack_amd/vae.py, random-init fp16 weights), ms per decode.
  cpu_baseline - the CPU oracle (kind "port") timed on this box's host cores on a bounded sample
                 (rank 0, N == 1 only): one double-stream + one single-stream block at full width and full S,
                 extrapolated x20 / x40 / x25 to a section (labelled as such).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_PEAK = 2.5e15


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="13b", choices=["13b", "tiny"])
    ap.add_argument("--n-text", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp8", action="store_true",
                    help="block linears and attention on the fp8 MFMA (e4m3 operands: NOT the bf16 headline configuration)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N > 1 ranks that all use cuda:0 and exchange over gloo (RCCL refuses two ranks on one device): "
                         "runs the multi-rank code path on a one-GPU box; the number it prints is NOT a measurement")
    ap.add_argument("--force-sp", action="store_true",
                    help="N == 1 only (diagnostics): a one-rank nccl group and the sequence-parallel code path on it - every "
                         "exchange, gather and buffer of the multi-GPU path at full size, with the data movement an "
                         "identity; the number it prints is NOT the single-GPU measurement")
    return ap.parse_args()


def cpu_baseline(S_img, S_txt, threads):
    """Oracle (CPU port of the reference's SDPA path) on one double + one single block, full width, full S."""
    from framepack_amd import synthetic
    from oracle import dit_oracle as O
    if threads > 0:
        torch.set_num_threads(threads)
    threads = torch.get_num_threads()
    cfg_kw = dict(synthetic.CONFIG_13B, num_layers=1, num_single_layers=1)
    cfg = O.DiTConfig(**cfg_kw)
    g = torch.Generator("cpu").manual_seed(0)
    P = {}
    for name, shape in O.param_shapes(cfg).items():
        if name.startswith("transformer_blocks.0") or name.startswith("single_transformer_blocks.0"):
            P[name] = torch.randn(shape, generator=g) * (0.02 if len(shape) > 1 else 1.0)
    d = cfg.inner_dim
    h = torch.randn(1, S_img, d, generator=g)
    e = torch.randn(1, S_txt, d, generator=g)
    temb = torch.randn(1, d, generator=g)
    ang = torch.rand(1, S_img, 128, generator=g) * 6.28
    rope = torch.cat([ang.cos(), ang.sin()], dim=-1)
    with torch.no_grad():
        t0 = time.time()
        h, e = O.double_block(P, cfg, 0, h, e, temb, rope)
        t1 = time.time()
        O.single_block(P, cfg, 0, h, e, temb, rope)
        t2 = time.time()
    td, ts = t1 - t0, t2 - t1
    sec_section = (20 * td + 40 * ts) * 25
    return {"value": round(sec_section, 1), "unit": "s per 25-step section (extrapolated)", "cores": threads,
            "kind": "port",
            "sample": f"1 double block ({td:.1f} s) + 1 single block ({ts:.1f} s) of the fp32 CPU oracle at full width "
                      f"(d=3072, 24 heads) and full S={S_img + S_txt}; section = (20*double + 40*single)*25, extrapolated"}


def vae_decode_ms(dev, t_lat, h_lat, w_lat):
    """One section's VAE decode (ref: diffusers_helper/hunyuan.py:93-103) on the HIP decoder, events on the stream."""
    from framepack_amd.vae import AutoencoderKLHunyuanVideo, vae_decode
    torch.manual_seed(0)
    vae = AutoencoderKLHunyuanVideo()
    for p in vae.parameters():
        if p.dim() > 1:
            torch.nn.init.normal_(p, std=p[0].numel() ** -0.5)
    vae = vae.to(dev, torch.float16).eval()
    z = torch.randn(1, 16, t_lat, h_lat, w_lat, device=dev)
    out = vae_decode(z, vae)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(2):
        out = vae_decode(z, vae)
    b.record()
    torch.cuda.synchronize()
    return {"ms": round(a.elapsed_time(b) / 2, 1), "latent": [1, 16, t_lat, h_lat, w_lat], "frames": list(out.shape),
            "dtype": "f16", "note": "framewise windows of 4+1 latent frames, as diffusers' default"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    elif args.force_sp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    from framepack_amd import ops, sampler, synthetic
    from framepack_amd.model import HunyuanVideoTransformer3DModelPacked

    bf = torch.bfloat16
    if args.config == "13b":
        cfg_kw, height, width, frames_latent = synthetic.CONFIG_13B, 640, 640, 9
        workload = "13B packed DiT, 640x640 bucket, 33-frame section (9 latent frames), 25 UniPC steps"
    else:
        cfg_kw, height, width, frames_latent = synthetic.CONFIG_TINY, 256, 256, 3
        workload = "tiny packed DiT (2+2 blocks, hidden 256), 256x256x9f"

    with torch.device(dev):
        model = HunyuanVideoTransformer3DModelPacked(**cfg_kw).to(bf)
    synthetic.init_synthetic_(dict(model.state_dict()), seed=0, fast_device_rng=True)
    model.eval()
    model.high_quality_fp32_output_for_inference = True        # the product's setting (ref framepack/models.py:241)
    if args.fp8:
        model.enable_fp8_compute()
        model.enable_fp8_attention()
    if world > 1:
        model.enable_sequence_parallel()
    elif args.force_sp:
        model.enable_sequence_parallel(force=True)

    kw = synthetic.make_inputs(seed=31337, frames_latent=frames_latent, height=height, width=width, n_text=args.n_text,
                               dtype=bf)
    latents0 = kw.pop("hidden_states").float().to(dev)
    kw.pop("timestep")
    guidance = kw.pop("guidance").to(dev)
    positive = {k: (v.to(dev) if torch.is_tensor(v) and v.dtype in (bf, torch.bool) else v) for k, v in kw.items()}
    positive["guidance"] = guidance
    Si, St, S = synthetic.packed_lengths(frames_latent, height, width, args.n_text)

    extra = dict(dtype=bf, cfg_scale=1.0, cfg_rescale=0.0, concat_latent=None, has_negative_prompt=False,
                 movement_scale=0.1, positive=positive, negative=None)
    k_model = sampler.fm_wrapper(model)
    mu = sampler.calculate_flux_mu(frames_latent * (height // 8) * (width // 8) // 4, exp_max=7.0)

    def run_steps(n, extra=extra):
        # the first n+1 entries of the real 25-step schedule: n model calls (UniPC never reads the last sigma)
        out = None
        while n > 0:
            m = min(n, 25)
            sig = sampler.get_flux_sigmas_from_mu(25, mu)[: m + 1].clone()
            out = sampler.sample_unipc(k_model, latents0.clone(), sig.to(dev), extra_args=extra, disable=True)
            n -= m
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.inference_mode():
        if args.warmup > 0:
            run_steps(args.warmup)
        fence()
        ops.ATTN_EVENTS = []
        t0 = time.perf_counter()
        out = run_steps(args.steps)
        fence()
        elapsed = time.perf_counter() - t0
        events, ops.ATTN_EVENTS = ops.ATTN_EVENTS, None
        if not bool(torch.isfinite(out).all()):
            raise SystemExit("bench: the timed steps produced non-finite latents - the number would be meaningless")

    # informational ("worst case S = 19 385; also report n = 64"): the same section with a 64-token prompt
    short_prompt = None
    if world == 1 and args.config == "13b" and args.n_text != 64:
        kw64 = synthetic.make_inputs(seed=31337, frames_latent=frames_latent, height=height, width=width, n_text=64, dtype=bf)
        for k in ("hidden_states", "timestep"):
            kw64.pop(k)
        pos64 = {k: (v.to(dev) if torch.is_tensor(v) and v.dtype in (bf, torch.bool) else v) for k, v in kw64.items()}
        pos64["guidance"] = guidance
        extra64 = dict(extra, positive=pos64)
        with torch.inference_mode():
            run_steps(1, extra64)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(5, extra64)
            torch.cuda.synchronize()
        short_prompt = {"n_text": 64, "packed_tokens": synthetic.packed_lengths(frames_latent, height, width, 64)[2],
                        "ms_per_step": round((time.perf_counter() - t1) / 5 * 1e3, 3), "steps": 5}

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()

    # dominant kernel, measured live on its own stream
    # (the token refiner's two small launches per step go through the same op: the block launches are the largest)
    rows, heads = max((e[2], e[3]) for e in events)
    att_ms = [a.elapsed_time(b) for a, b, r, h in events if (r, h) == (rows, heads)]
    att_avg_s = sum(att_ms) / len(att_ms) / 1e3
    flops_per_launch = 4.0 * rows * rows * heads * 128
    achieved = flops_per_launch / att_avg_s
    # HBM-side bytes per launch of the same kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes; tools/profile_round.sh) - valid for the workload shape only
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r3_h_attn_hbm_traffic.json")
    if args.config == "13b" and world == 1 and rows == 19385 and os.path.exists(tpath):
        with open(tpath) as f:
            traffic = round(json.load(f)["traffic_bytes_per_launch"])
    peak = MFMA_BF16_PEAK * (2 if args.fp8 else 1)          # the block-scaled fp8 MFMA runs at twice the bf16 rate
    if args.fp8:
        traffic = None                                      # the PMC passes were made on the bf16 kernel
    roofline = {"kernel": "attn_fp8_kernel<8> + its quantisation kernels (K1 fp8 variant, MFMA 32x32x64 f8f6f4)" if args.fp8
                else "attn_v3_kernel<60> (K1 flash attention, bf16 MFMA 32x32x16, one wave per SIMD, S^T - m accumulated in VGPRs)", "bound": "mfma",
                "achieved": round(achieved / 1e12, 1), "peak": peak / 1e12, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                "traffic_unit": "bytes per launch (PMC, profiles/r3_h_attn_hbm_traffic.json; algorithmic 476 MB)",
                "launches": len(att_ms), "avg_launch_ms": round(att_avg_s * 1e3, 4),
                "flops_per_launch": flops_per_launch,
                "share_of_step_time": round(sum(att_ms) / 1e3 / elapsed, 4)}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            "metric": "sec per 25-step section (13B packed DiT, 640x640x33f)" if args.config == "13b"
            else "sec per 25-step section (tiny packed DiT)",
            "value": round(ms_per_step * 25 / 1e3, 4), "unit": "s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": False,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "fp8-e4m3 block linears and attention (opt-in, outside the 1e-2 parity bar)" if args.fp8 else "bf16",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo, not a measurement)" if args.rehearse_one_gpu else "")
            + (" (DIAGNOSTIC: sequence-parallel path on a one-rank nccl group, not the single-GPU measurement)" if args.force_sp else ""),
            "config": {"workload": workload, "packed_tokens": S, "image_tokens": Si, "text_tokens": St,
                       "teacache": False, "parallelism": "single GPU" if world == 1 else f"ulysses-sp{world}"},
            "model_tflops_per_step": round((24 * 3072 ** 2 * S + 4 * S * S * 3072) * 60 / 1e12, 1) if args.config == "13b" else None,
            "roofline": roofline,
        }
        if args.config == "13b":
            fl = (24 * 3072 ** 2 * S + 4 * S * S * 3072) * 60
            line["model_mfma_frac"] = round(fl / (ms_per_step / 1e3) / (MFMA_BF16_PEAK * world), 4)
        if short_prompt is not None:
            line["short_prompt"] = short_prompt
        if world == 1 and args.config == "13b":
            line["vae_decode"] = vae_decode_ms(dev, frames_latent, height // 8, width // 8)
        if world == 1 and not args.no_cpu_baseline and args.config == "13b":
            line["cpu_baseline"] = cpu_baseline(Si, St, args.cpu_threads)
        print(json.dumps(line), flush=True)
    if world > 1 or args.force_sp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
