"""Per-step GPU time of the encode and decode plans at the benchmark shape (wt_plan_set_timing: HIP events around every
launch whose step name contains the filter), one filter at a time, 10 round trips each.

    python tools/step_times.py [out.md] [clips] [samples]
"""
import collections
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(out, B=64, T=72000):
    import torch
    from wavtokenizer_amd import WavTokenizer, ARCH_HOP600, synth, _capi
    lib = _capi.lib
    sd = synth.make_state_dict(ARCH_HOP600, seed=0)
    m = WavTokenizer.from_arch(ARCH_HOP600)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m = m.eval().to("cuda")
    wav = torch.from_numpy(synth.make_clips(B, T, seed=2000)).cuda()
    m.set_graph_max_clips(0)                               # time the launches themselves, not a graph replay
    bw = torch.tensor([0])

    def trip():
        f, c = m.encode_infer(wav, bandwidth_id=bw)
        return m.decode(f, bandwidth_id=bw)

    for _ in range(3):
        trip()
    torch.cuda.synchronize()
    rows = []
    for key, (plan, _ws) in list(m._engine.plans.items()):       # (the LRU reorders the dict on every call)
        names = []
        for i in range(lib.wt_plan_num_steps(plan)):
            p = ctypes.c_char_p()
            assert lib.wt_plan_step_name(plan, i, ctypes.byref(p)) == 0
            names.append(p.value.decode())
        count = collections.Counter(names)
        per_call = {}

        def timed(flt):
            _capi.check(lib.wt_plan_set_timing(plan, flt.encode()), "set_timing")
            for _ in range(10):
                trip()
            tot, n = ctypes.c_double(), ctypes.c_int64()
            _capi.check(lib.wt_plan_read_timing(plan, ctypes.byref(tot), ctypes.byref(n), 1), "read_timing")
            lib.wt_plan_set_timing(plan, b"")
            return tot.value / 10                          # ms per call, all matching launches

        # the filter is a substring match: names that contain no other name first, then the rest by subtraction
        order = sorted(count, key=lambda nm: sum(1 for o in count if o != nm and nm in o))
        for name in order:
            ms = timed(name)
            for other in count:
                if other != name and name in other:
                    ms -= per_call[other]
            per_call[name] = ms
        for name in sorted(count, key=names.index):
            rows.append((("encode" if key[0] == 0 else "decode"), name, count[name], 1e3 * per_call[name] / count[name], per_call[name]))
    lines = ["# Per-step GPU time, hop-600, %d x %.1f s (HIP events around each launch; tools/step_times.py)" % (B, T / 24000), "",
             "| plan | step | launches per call | us per launch | ms per call |", "|---|---|---|---|---|"]
    for plan, name, cnt, us, ms in rows:
        lines.append(f"| {plan} | `{name}` | {cnt} | {us:.1f} | {ms:.3f} |")
    tot = collections.defaultdict(float)
    for plan, _n, _c, _u, ms in rows:
        tot[plan] += ms
    lines += ["", "sum of the timed steps: " + ", ".join(f"{k} {v:.2f} ms" for k, v in tot.items())]
    text = "\n".join(lines) + "\n"
    print(text)
    if out:
        open(out, "w").write(text)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None, *(int(a) for a in sys.argv[2:4]))
