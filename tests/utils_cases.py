"""Seeded inputs for the tensor helpers of the section loop (soft_append_bcthw, crop_or_pad_yield_mask,
repeat_to_batch_size).  Shared by tests/golden/make_golden.py (which feeds them to the reference's own
diffusers_helper/utils.py and stores the outputs in tests/golden/utils.npz) and tests/test_sections_cpu.py."""
import torch


def _randn(seed, shape, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g).to(dtype)


def soft_append_cases():
    """name -> (history, current, overlap); pixel clips [B, C, T, H, W] as the worker appends them (worker.py:713)."""
    return {
        "no_overlap": (_randn(1, (1, 3, 5, 4, 6)), _randn(2, (1, 3, 4, 4, 6)), 0),
        "negative_overlap": (_randn(3, (1, 3, 2, 4, 6)), _randn(4, (1, 3, 3, 4, 6)), -2),
        "overlap_3": (_randn(5, (1, 3, 9, 4, 6)), _randn(6, (1, 3, 7, 4, 6)), 3),
        "overlap_1": (_randn(7, (2, 3, 4, 2, 2)), _randn(8, (2, 3, 4, 2, 2)), 1),
        "overlap_all_of_history": (_randn(9, (1, 3, 5, 4, 6)), _randn(10, (1, 3, 8, 4, 6)), 5),
        "overlap_all_of_current": (_randn(11, (1, 3, 8, 4, 6)), _randn(12, (1, 3, 5, 4, 6)), 5),
        # the product's own sizes in miniature: window 9 -> 33 new frames, overlap 4*9-3 = 33
        "section_overlap_33": (_randn(13, (1, 3, 36, 2, 2)), _randn(14, (1, 3, 69, 2, 2)), 33),
        "fp16": (_randn(15, (1, 3, 6, 4, 6), torch.float16), _randn(16, (1, 3, 6, 4, 6), torch.float16), 4),
        "bf16_history_fp32_current": (_randn(17, (1, 3, 6, 2, 2), torch.bfloat16), _randn(18, (1, 3, 6, 2, 2)), 2),
    }


def crop_or_pad_cases():
    """name -> (x [B, F, C], length)."""
    return {
        "pad": (_randn(21, (1, 37, 16)), 64),
        "pad_batch_fp16": (_randn(22, (2, 5, 8), torch.float16), 12),
        "equal": (_randn(23, (1, 64, 16)), 64),
        "crop": (_randn(24, (1, 90, 16)), 64),
        "empty": (_randn(25, (1, 0, 16)), 8),
    }


def repeat_cases():
    """name -> (tensor, batch_size)."""
    return {
        "same": (_randn(31, (2, 3, 4)), 2),
        "times_3": (_randn(32, (1, 3, 4)), 3),
        "times_2_of_2": (_randn(33, (2, 5)), 4),
        "uneven": (_randn(34, (2, 5)), 3),
    }


def preview_cases():
    """name -> latents [B, 16, T, H, W] for vae_decode_fake / the vae_decode wrappers."""
    return {"f32": _randn(41, (1, 16, 3, 5, 4)) * 2.0, "bf16": (_randn(42, (2, 16, 1, 2, 3)) * 2.0).to(torch.bfloat16),
            "f16": (_randn(43, (1, 16, 2, 2, 2)) * 3.0).to(torch.float16)}


def stub_image():
    return _randn(44, (1, 3, 1, 8, 8))


class StubVAE:
    """Duck-typed stand-in for AutoencoderKLHunyuanVideo in the vae_decode / vae_encode wrappers: `decode` turns a
    latent [B,16,T,H,W] into 'frames' [B,3,4(T-1)+1,H,W] by a fixed formula that mixes time (so image_mode's per-frame
    decode differs from the video decode), `encode` returns a distribution whose sample is a fixed function."""
    device = torch.device("cpu")
    dtype = torch.float32

    class config:
        scaling_factor = 0.476986

    def decode(self, z):
        t = z.shape[2]
        frames = z[:, :3].repeat_interleave(4, dim=2)[:, :, 3:] if t > 1 else z[:, :3]
        frames = frames + 0.1 * z[:, 3:6].mean(dim=2, keepdim=True)

        class Out:
            sample = frames * 0.5 + 0.25
        return Out()

    def encode(self, x):
        class Dist:
            @staticmethod
            def sample():
                return torch.tanh(x.mean(dim=1, keepdim=True)).expand(-1, 16, -1, -1, -1) * 1.5

        class Out:
            latent_dist = Dist()
        return Out()


def bucket_queries():
    """(h, w, resolution) sweep for find_nearest_bucket: squares, both orientations, extreme aspects, ties, the
    precomputed resolutions (640, 320) and scaled ones."""
    import random
    rnd = random.Random(5)
    q = [(h, w, r) for r in (640, 320, 480, 512, 416, 720, 1024)
         for h, w in ((1, 1), (640, 640), (1080, 1920), (1920, 1080), (100, 1000), (1000, 100), (607, 641), (3, 4), (768, 512))]
    q += [(rnd.randint(1, 4000), rnd.randint(1, 4000), rnd.choice((640, 320, 352, 544, 608))) for _ in range(300)]
    return q
