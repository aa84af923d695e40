"""head.out / head.istft / head.ola GPU time of the head-only plan for inputs of different scale (is the time data dependent?).

    python tools/head_time.py
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(B=64, L=120):
    import torch
    from wavtokenizer_amd import WavTokenizer, ARCH_HOP600, synth, _capi
    lib = _capi.lib
    sd = synth.make_state_dict(ARCH_HOP600, seed=0)
    m = WavTokenizer.from_arch(ARCH_HOP600)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m = m.eval().to("cuda")
    m.set_graph_max_clips(0)
    w = sd["head.out.weight"]
    print("head.out.weight std %.4f  bias std %.4f" % (w.std(), sd["head.out.bias"].std()))
    g = torch.Generator(device="cuda").manual_seed(1)
    for scale in (0.0, 0.1, 1.0, 4.0):
        x = scale * torch.randn((B, L, 768), device="cuda", generator=g)
        for _ in range(3):
            m._run_head(x)
        y = torch.nn.functional.linear(x.reshape(-1, 768), torch.from_numpy(w).cuda(), torch.from_numpy(sd["head.out.bias"]).cuda())
        print("input scale %.1f: log-mag max %.1f  phase absmax %.1f" % (scale, y[:, :641].max().item(), y[:, 641:].abs().max().item()))
        plan = [p for k, (p, _ws) in m._engine.plans.items() if k[0] == _capi.WT_PLAN_HEAD][0]
        for name in (b"head.out", b"head.istft", b"head.ola"):
            _capi.check(lib.wt_plan_set_timing(plan, name), "set_timing")
            for _ in range(10):
                m._run_head(x)
            tot, n = ctypes.c_double(), ctypes.c_int64()
            _capi.check(lib.wt_plan_read_timing(plan, ctypes.byref(tot), ctypes.byref(n), 1), "read_timing")
            lib.wt_plan_set_timing(plan, b"")
            print("   %-11s %.1f us" % (name.decode(), 1e3 * tot.value / max(1, n.value)))
    m.check_status()


if __name__ == "__main__":
    main()
