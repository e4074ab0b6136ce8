"""CPU oracle for the FramePack packed-DiT hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``framepack_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker, never as the thing that
is measured or shipped.

What it is: a functional (state-dict in, tensors out) PyTorch-CPU restatement
of the reference's SDPA code path, written from the reference's behaviour with
each function citing the reference file:line it follows.

Pinning status (see DESIGN.md "Oracle"):
  * code that lives in /root/reference (hunyuan_video_packed.py, dit_common.py,
    wrapper.py, uni_pc_fm.py, fp8_optimization_utils.py): PINNED by the golden
    vectors under tests/golden/, generated in the build container by
    tests/golden/make_golden.py from the reference's own modules.
  * arithmetic that lives in diffusers==0.33.1 (Attention, FeedForward,
    Timesteps, TimestepEmbedding, PixArtAlphaTextProjection and the whole VAE):
    the package is absent from the image and the reference holds no tests or
    fixtures for it, so those pieces are restated from the published
    algorithm: PARITY UNPINNED for them.
"""
