#!/usr/bin/env python3
"""Fixture for ISTFT padding="center" (decoder/spectral_ops.py:43-45: torch.istft(center=True)) from the REAL reference.

No reference YAML selects "center"; the reference model is built from the hop-600 YAML as in make_golden.py and its
ISTFT module switched to padding = "center" (the attribute ISTFT.forward tests).  The script asserts that
oracle/cpu_ref.py with ArchConfig(padding="center") is bit-identical, then writes inputs + the reference's outputs:

    python tests/golden/make_golden_center.py        # build container only (needs /root/reference)
"""
import dataclasses
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")

from wavtokenizer_amd import synth  # noqa: E402
from wavtokenizer_amd.config import ARCH_HOP600  # noqa: E402
from oracle.cpu_ref import OracleWavTokenizer  # noqa: E402
from _ref_import import build_reference  # noqa: E402
from make_golden import YAMLS, WEIGHT_SEED  # noqa: E402


def main():
    arch = dataclasses.replace(ARCH_HOP600, padding="center")
    sd = synth.make_state_dict(ARCH_HOP600, seed=WEIGHT_SEED)
    ref = build_reference(YAMLS["hop600"], sd)
    ref.head.istft.padding = "center"
    orc = OracleWavTokenizer(arch, sd)
    bw = torch.tensor([0])
    out = {}
    for tag, B, T in (("b2_t24000", 2, 24000), ("b1_t1300", 1, 1300)):       # 40 frames; 3 frames (ragged last hop)
        wav = torch.from_numpy(synth.make_clips(B, T, seed=5000 + T))
        with torch.inference_mode():
            feats, codes = ref.encode_infer(wav, bandwidth_id=bw)
            want = ref.decode(feats, bandwidth_id=bw)
            got = orc.decode(feats, bw)
        L = arch.frames(T)
        assert want.shape == (B, (L - 1) * arch.hop_length), want.shape
        assert torch.equal(got, want), "oracle differs from the reference in center mode"
        out[f"{tag}/wav_in"] = wav.numpy()
        out[f"{tag}/features"] = feats.numpy()
        out[f"{tag}/wav_out"] = want.numpy()
        print(tag, "frames", L, "->", tuple(want.shape), "oracle bit-identical")
    np.savez_compressed(os.path.join(HERE, "hop600_center.npz"), **out)


if __name__ == "__main__":
    main()
