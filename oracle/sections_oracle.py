"""CPU restatement of the reference's section bookkeeping (TEST INFRASTRUCTURE, see oracle/__init__.py).

Written independently of framepack_amd/sections.py, from ref: framepack/worker.py:337-350 (padding schedule),
:463-465 (index layout), :655-664 / :703-720 (history and decode windows), diffusers_helper/utils.py:64-75
(cross-fade).  Pinned by the values SURVEY.md 8 quotes from the reference (indices of the last section at
window 9) and by closed forms (frame counts); the reference module itself cannot be imported (gradio, cv2)."""
from __future__ import annotations

import numpy as np


def paddings(n_sections: int):
    n = max(int(n_sections), 1)
    return [3] + [2] * (n - 3) + [1, 0] if n > 4 else list(range(n - 1, -1, -1))


def indices(padding: int, window: int):
    p = padding * window
    pos = np.arange(1 + p + window + 1 + 2 + 16)
    cuts = np.cumsum([1, p, window, 1, 2, 16])
    pre, _, lat, post, two, four = np.split(pos, cuts[:-1])
    return dict(latent=lat, clean=np.concatenate([pre, post]), two=two, four=four)


def crossfade(history: np.ndarray, current: np.ndarray, overlap: int) -> np.ndarray:
    """history, current: [frames, ...]; `history` keeps its head, `current` its tail."""
    if overlap <= 0:
        return np.concatenate([history, current])
    w = np.linspace(1, 0, overlap, dtype=history.dtype).reshape(-1, *([1] * (history.ndim - 1)))
    mid = w * history[-overlap:] + (1 - w) * current[:overlap]
    return np.concatenate([history[:-overlap], mid, current[overlap:]])


def expected_frame_counts(n_sections: int, window: int):
    """(latent frames, pixel frames) after every section: each section adds `window` latent frames, the last one
    also the start frame; pixels = 4 * latents - 3."""
    out, lat = [], 0
    pads = paddings(n_sections)
    for i, p in enumerate(pads):
        lat += window + (1 if p == 0 else 0)
        out.append((lat, 4 * lat - 3))
    return out
