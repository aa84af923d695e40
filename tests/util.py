"""Shared helpers for the parity tests."""
import functools
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# A code mismatch is a "near tie" (not a defect) only when the reference's own top-2 distance
# margin for that frame is below this; distances are O(400) here, so this is ~5e-5 relative,
# i.e. the size of fp32 re-association noise accumulated through the 16-layer encoder.
NEAR_TIE_MARGIN = 0.02
# north_star: reconstructed waveform within 1e-4 relative of the reference CPU path
WAV_REL_TOL = 1e-4


@functools.lru_cache(maxsize=None)
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


@functools.lru_cache(maxsize=4)
def synth_state_dict(arch_name: str):
    """Regenerate the synthetic weights and check them against the fixture manifest."""
    from wavtokenizer_amd import synth
    from wavtokenizer_amd.config import NAMED_ARCHS
    m = manifest()
    sd = synth.make_state_dict(NAMED_ARCHS[arch_name], seed=m["weight_seed"])
    got = synth.weights_manifest(sd)
    want = m["archs"][arch_name]["weights"]
    assert got.keys() == want.keys()
    bad = [k for k in got if got[k]["sha256"] != want[k]["sha256"]]
    assert not bad, f"synthetic weights differ from the fixtures' manifest: {bad[:3]}"
    return sd


def load_case(arch_name: str, case: str):
    return np.load(os.path.join(GOLDEN, f"{arch_name}_{case}.npz"))


def rel_l2(a, b) -> float:
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def check_codes(got, want, margin, what=""):
    """Codes must be bit-exact; a differing frame is tolerated only if the reference's own
    top-2 margin there is a near tie. Returns the number of near-tie flips."""
    got = np.asarray(got).reshape(-1)
    want = np.asarray(want).reshape(-1)
    margin = np.asarray(margin).reshape(-1)
    assert got.shape == want.shape, (got.shape, want.shape)
    bad = np.nonzero(got != want)[0]
    hard = [int(i) for i in bad if margin[i] >= NEAR_TIE_MARGIN]
    assert not hard, (f"{what}: {len(hard)} code mismatches at frames with healthy margin, e.g. frame {hard[0]}: "
                      f"got {got[hard[0]]} want {want[hard[0]]} margin {margin[hard[0]]:.4g}")
    return len(bad)
