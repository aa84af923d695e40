#!/usr/bin/env python3
"""Fixture for codes_to_features with K > 1 codebooks (decoder/pretrained.py:209-239): run in the build container only.

    python tests/golden/make_golden_codebooks.py

Every YAML of the reference has num_quantizers: 1, so a copy of the hop-600 YAML with num_quantizers: 3 is written to a
temporary directory, the reference class is built from it, the three codebooks of synth.make_state_dict are loaded, and
WavTokenizer.codes_to_features is run on random codes with K = 1, 2, 3 (layouts (K, B, L) and (K, L)).  The oracle must
reproduce every output bit for bit; stored: codes and features.  Data only."""
import dataclasses
import os
import re
import sys
import tempfile
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
from make_golden import YAMLS, check_identical  # noqa: E402


def main():
    arch = dataclasses.replace(ARCH_HOP600, num_quantizers=3)
    sd = synth.make_state_dict(arch, seed=0)
    text = open(YAMLS["hop600"]).read()
    text2, n = re.subn(r"num_quantizers:\s*1", "num_quantizers: 3", text)
    assert n == 1, "expected exactly one num_quantizers key in the YAML"
    with tempfile.TemporaryDirectory() as td:
        y = os.path.join(td, "nq3.yaml")
        open(y, "w").write(text2)
        ref = build_reference(y, sd)
    orc = OracleWavTokenizer(arch, sd)
    gen = torch.Generator().manual_seed(77)
    out = {}
    for tag, shape in (("k1", (1, 2, 9)), ("k2", (2, 2, 9)), ("k3", (3, 2, 9)), ("k2_2d", (2, 13))):
        codes = torch.randint(0, arch.vq_bins, shape, generator=gen)
        with torch.inference_mode():
            fr = ref.codes_to_features(codes)
            fo = orc.codes_to_features(codes)
        check_identical(tag, fo, fr)
        out[f"{tag}/codes"] = codes.numpy()
        out[f"{tag}/features"] = fr.numpy()
        print(tag, tuple(fr.shape), float(fr.abs().max()))
    np.savez_compressed(os.path.join(HERE, "hop600_codebooks3.npz"), weight_seed=np.int64(0), **out)


if __name__ == "__main__":
    main()
