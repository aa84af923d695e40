#!/usr/bin/env python3
"""Golden vectors for the overlap-add helper: outputs of the REFERENCE's `_linear_overlap_add`
(encoder/utils.py:17-56), imported here (torchaudio placeholder as in _ref_import.py), on seeded inputs.
Writes tests/golden/overlap_add.npz and asserts oracle/audio_ref.py reproduces the reference bit for bit."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import _stub_torchaudio, REFERENCE_ROOT  # noqa: E402

_stub_torchaudio()
sys.path.insert(0, REFERENCE_ROOT)
from encoder.utils import _linear_overlap_add  # noqa: E402
from oracle.audio_ref import linear_overlap_add  # noqa: E402

CASES = [  # (n_frames, frame_len, last_len, stride, batch)
    (1, 100, 100, 50, 1), (2, 100, 100, 50, 2), (3, 64, 40, 48, 3), (5, 1000, 333, 990, 2), (4, 240, 240, 80, 1),
    (3, 24000, 12345, 23760, 1), (2, 5, 1, 4, 4),
]
out = {}
for ci, (nf, fl, ll, st, B) in enumerate(CASES):
    g = torch.Generator().manual_seed(100 + ci)
    frames = [torch.randn(B, 1, fl if i + 1 < nf else ll, generator=g) for i in range(nf)]
    ref = _linear_overlap_add(frames, st).numpy()
    mine = linear_overlap_add([f.numpy() for f in frames], st)
    assert ref.shape == mine.shape and np.array_equal(ref, mine), (ci, np.abs(ref - mine).max())
    out[f"case{ci}_meta"] = np.array([nf, fl, ll, st, B], np.int64)
    for i, f in enumerate(frames):
        out[f"case{ci}_frame{i}"] = f.numpy()
    out[f"case{ci}_out"] = ref
np.savez_compressed(os.path.join(HERE, "overlap_add.npz"), **out)
print("wrote overlap_add.npz:", len(CASES), "cases, oracle bit-identical to the reference")
