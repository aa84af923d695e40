#!/usr/bin/env python3
"""Head-room of every split-f16 (S32) operand below the f16 limit 65504 (VERDICT r03 #3a): model.range_report on the
synthetic fixture weights and on the trained-like weights, fixture inputs and the loud inputs of tests/golden/*_inputs.npz.

    python tools/range_report.py [out.md]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS, synth


def model_of(arch, sd):
    m = WavTokenizer.from_arch(arch)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return m.eval().to("cuda")


def main(out):
    lines = ["# Range report: largest magnitude of every S32 (split-f16) buffer behind every step, and its head-room below 65504",
             "", "`model.range_report(wav)` (WT_PLAN_FLAG_RANGE_REPORT plans: the shipped kernels, an amax pass behind every step); "
             "head-room = log2(65504 / amax) bits.  tools/range_report.py", ""]
    golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    for name in ("hop600", "hop320"):
        arch = NAMED_ARCHS[name]
        cases = [("synthetic weights (fixtures, seed 0), 2 synth clips x 3 s", synth.make_state_dict(arch, seed=0), synth.make_clips(2, 72000, seed=2000)),
                 ("synthetic weights, the 8 inputs outside the synth family (silence ... base x 30)", synth.make_state_dict(arch, seed=0),
                  np.load(os.path.join(golden, f"{name}_inputs.npz"))["wav_in"]),
                 ("trained-like weights (heavy tails, layer scales up to 10), 2 synth clips x 1 s", synth.make_trained_like_state_dict(arch, seed=7),
                  synth.make_clips(2, 24037, seed=4242))]
        for what, sd, wav in cases:
            m = model_of(arch, sd)
            rep = m.range_report(torch.from_numpy(wav).cuda(), bandwidth_id=torch.tensor([0]))
            m.check_status()
            worst = sorted(rep, key=lambda r: r["headroom_bits"])
            lines += [f"## {name}: {what}", "", f"{len(rep)} (step, buffer) entries; least head-room {worst[0]['headroom_bits']:.2f} bits.  The ten tightest:", "",
                      "| plan | step | buffer | amax | head-room (bits) |", "|---|---|---|---|---|"]
            for r in worst[:10]:
                lines.append(f"| {r['plan']} | `{r['step']}` | `{r['buffer']}` | {r['amax']:.4g} | {r['headroom_bits']:.2f} |")
            # per site: the tightest operand of the encoder and of each decoder block
            by_step = {}
            for r in rep:
                key = (r["plan"], r["step"])
                if key not in by_step or r["headroom_bits"] < by_step[key]["headroom_bits"]:
                    by_step[key] = r
            lines += ["", "Tightest operand per step name: " + "; ".join(f"`{k[1]}` {v['headroom_bits']:.1f}" for k, v in sorted(by_step.items(), key=lambda kv: kv[1]['headroom_bits'])[:24]), ""]
            del m
            torch.cuda.empty_cache()
    text = "\n".join(lines) + "\n"
    print(text[:3000])
    if out:
        open(out, "w").write(text)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
