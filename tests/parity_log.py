"""Parity numbers of a `-m gpu` session, kept as an artifact: every test that measures something records it here;
tests/conftest.py writes gpu_parity_summary.json and prints one summary line at session end (so the tail of the
driver's GPUTEST log carries the numbers, not only dots)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = {}


def record(test: str, **numbers):
    RECORDS.setdefault(test, {}).update({k: (float(v) if isinstance(v, float) else v) for k, v in numbers.items()})


def dump():
    if not RECORDS:
        return None
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, os.environ.get("WAVTOK_PARITY_SUMMARY", "gpu_parity_summary.json"))
    with open(path, "w") as f:
        json.dump(RECORDS, f, indent=1, sort_keys=True, default=str)
    flips = sum(int(v.get("code_flips", 0)) for v in RECORDS.values())
    frames = sum(int(v.get("frames", 0)) for v in RECORDS.values())
    wl2 = [float(v["wav_rel_l2"]) for v in RECORDS.values() if "wav_rel_l2" in v]
    return (f"[gpu parity] {len(RECORDS)} records -> {os.path.relpath(path, ROOT)}: code flips {flips} / {frames} frames compared; "
            f"waveform rel-L2 max {max(wl2) if wl2 else float('nan'):.3g} over {len(wl2)} comparisons (bar 1e-4)")
