#!/usr/bin/env python3
"""Fixtures for inputs OUTSIDE the synth family: run in the build container only.

    python tests/golden/make_golden_inputs.py

Every other fixture feeds the path AM/FM tones + 0.05 noise with |x| <= 0.5 (wavtokenizer_amd/synth.py).  Here one batch of
eight 1 s clips per architecture holds what a tokenizer meets in the wild and what the split-f16 (S32) operand form is
most sensitive to: digital silence, the base clip at -70 dBFS and at -110 dBFS (f16 hi halves go subnormal), a full-scale
220 Hz square wave, a unit impulse, DC 0.9, the base clip x 30 (far beyond full scale) and the base clip hard-clipped to
+-1.  The batch goes through the imported reference (decoder/feature_extractors.py:131-142 -> decoder/pretrained.py:192-207)
with the fixtures' synthetic weights; oracle/cpu_ref.py must reproduce features, codes and waveform bit for bit; stored are
the inputs, codes, argmin margins, the waveform and the float64 run of the oracle's codes (conditioning check).  Data only."""
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
from wavtokenizer_amd.config import ARCH_HOP600, ARCH_HOP320  # noqa: E402
from oracle.cpu_ref import OracleWavTokenizer  # noqa: E402
from _ref_import import build_reference  # noqa: E402
from make_golden import YAMLS, check_identical  # noqa: E402

NAMES = ["silence", "base_x3e-4", "base_x3e-6", "square_220Hz_fullscale", "unit_impulse", "dc_0.9", "base_x30", "base_clipped_pm1"]
T = 24000


def make_inputs() -> np.ndarray:
    base = synth.make_clips(1, T, seed=31337)[0]
    t = np.arange(T, dtype=np.float64) / 24000.0
    x = np.zeros((8, T), dtype=np.float32)
    x[1] = base * np.float32(3e-4)
    x[2] = base * np.float32(3e-6)
    x[3] = np.where(np.sin(2 * np.pi * 220.0 * t) >= 0, 1.0, -1.0).astype(np.float32)
    x[4, T // 2] = 1.0
    x[5] = 0.9
    x[6] = base * np.float32(30.0)
    x[7] = np.clip(base * np.float32(4.0), -1.0, 1.0)
    return x


def main():
    import json
    with open(os.path.join(HERE, "manifest.json")) as f:
        weight_seed = json.load(f)["weight_seed"]
    wav = make_inputs()
    for name, arch in (("hop600", ARCH_HOP600), ("hop320", ARCH_HOP320)):
        sd = synth.make_state_dict(arch, seed=weight_seed)
        ref = build_reference(YAMLS[name], sd)
        orc = OracleWavTokenizer(arch, sd)
        bw = torch.tensor([0])
        with torch.inference_mode():
            fr, cr = ref.encode_infer(torch.from_numpy(wav), bandwidth_id=bw)
            wr = ref.decode(fr, bandwidth_id=bw)
            taps = {}
            fo, co = orc.encode_infer(torch.from_numpy(wav), bw, taps)
            wo = orc.decode(fo, bw)
        check_identical("features", fo, fr)
        check_identical("codes", co, cr)
        check_identical("waveform", wo, wr)
        margin = taps["vq.margin"].numpy().reshape(8, -1)
        # conditioning: the same op sequence in float64 must pick the same codes (then exact codes are a fair demand of any
        # fp32-class implementation on these inputs)
        orc64 = OracleWavTokenizer(arch, {k: torch.from_numpy(v).double() for k, v in sd.items()})
        with torch.inference_mode():
            _f64, c64 = orc64.encode_infer(torch.from_numpy(wav.astype(np.float64)), bw)
        same64 = (c64.numpy() == cr.numpy()).reshape(8, -1).all(axis=1)
        for i, nm in enumerate(NAMES):
            print(f"{name} {nm:26s} min margin {margin[i].min():8.4f}  codes==float64 {bool(same64[i])}  |wav_out| max {np.abs(wr[i].numpy()).max():.3e}  "
                  f"rms {wr[i].numpy().std():.3e}")
        np.savez_compressed(os.path.join(HERE, f"{name}_inputs.npz"), wav_in=wav, codes=cr.numpy(), wav_out=wr.numpy(),
                            margin=margin.astype(np.float32), codes_equal_float64=same64, names=np.array(NAMES),
                            bandwidth_id=np.int64(0), weight_seed=np.int64(weight_seed))


if __name__ == "__main__":
    main()
