#!/usr/bin/env python3
"""Fixture for weights with trained-like statistics (VERDICT r02 next #4b): run in the build container only.

    python tests/golden/make_golden_trained_like.py

Loads synth.make_trained_like_state_dict (log-normal weight_g, heavy-tailed matrices, layer scale up to 10, spread norm
scales) into the imported reference, runs encode_infer / decode on seeded clips, asserts oracle/cpu_ref.py reproduces
the outputs bit for bit (the oracle stays pinned on this weight class too), and writes inputs + expected outputs +
argmin margins to tests/golden/<arch>_trained_like.npz.  Data only."""
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
from make_golden import YAMLS, check_identical, sha  # noqa: E402

SEED = 7


def main():
    for name, arch in (("hop600", ARCH_HOP600), ("hop320", ARCH_HOP320)):
        sd = synth.make_trained_like_state_dict(arch, seed=SEED)
        ref = build_reference(YAMLS[name], sd)
        orc = OracleWavTokenizer(arch, sd)
        wav = synth.make_clips(2, 24000 + 37, seed=4242)        # not a multiple of the hop
        bw = torch.tensor([1])
        with torch.inference_mode():
            fr, cr = ref.encode_infer(torch.from_numpy(wav), bandwidth_id=bw)
            wr = ref.decode(fr, bandwidth_id=bw)
            taps = {}
            fo, co = orc.encode_infer(torch.from_numpy(wav), bw, taps)
            wo = orc.decode(fo, bw)
        check_identical("features", fo, fr)
        check_identical("codes", co, cr)
        check_identical("waveform", wo, wr)
        margin = taps["vq.margin"].numpy()
        emb = taps["enc.emb"] if "enc.emb" in taps else None
        print(f"{name}: codes {tuple(cr.shape)} distinct {len(np.unique(cr.numpy()))}, min margin {margin.min():.4g}, "
              f"frames with margin < 0.02: {(margin < 0.02).sum()}, |wav| max {np.abs(wr.numpy()).max():.3g}, "
              f"feature rms {fr.numpy().std():.3g}")
        np.savez_compressed(os.path.join(HERE, f"{name}_trained_like.npz"), wav_in=wav, codes=cr.numpy(), wav_out=wr.numpy(),
                            margin=margin, bandwidth_id=np.int64(1), weight_seed=np.int64(SEED),
                            weights_sha=np.array(sha(np.concatenate([v.reshape(-1).view(np.uint8) if v.dtype != np.uint8 else v.reshape(-1)
                                                                    for _k, v in sorted(sd.items())]))))


if __name__ == "__main__":
    main()
