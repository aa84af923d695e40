"""Time the stage-1 kernel (first conv + resblock + down conv, resblock16.hip) alone at the benchmark shape, optionally
under the WT_RB16_DBG ablation masks (1 no tile fill, 2 no resblock MFMAs, 4 no stores, 8 no down-conv taps, 16 no ELU).

    python tools/rb16_bench.py [r] [dbg masks ...]        r = 4 / 2: stage 1 (hop-600 / hop-320); r = 0: the stage-2 resblock
"""
# the WT_* switches these measurements flip exist in the LAB build only (the product library reads no environment variable)
import os as _os
_os.environ.setdefault("WAVTOK_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tools", "lib", "libwavtok_hip_lab.so"))
import os
import subprocess
import sys


def run_stage2(iters=20):
    """the 64-channel resblock of encoder stage 2 (wt_resblock: 18000 frames x 64 clips, S32(elu) out)"""
    import torch
    from wavtokenizer_amd._capi import lib, check
    B, T, C = 64, 18000, 64
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.1).cuda()
    x = rnd(B, T, C)
    ws = [rnd(C // 2, 3, C), rnd(C // 2), rnd(C, C // 2), rnd(C), rnd(C, C), rnd(C)]
    y = torch.empty(B, T, C, device="cuda")
    p = lambda t: t.data_ptr()
    call = lambda: check(lib.wt_resblock(p(x), None, None, None, *[p(w) for w in ws], p(y), B, T, C, 1, 1, 0, None), "wt_resblock")
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(r, iters=20):
    if r == 0:
        return run_stage2(iters)
    import torch
    from wavtokenizer_amd._capi import lib, check
    B, T = 64, 72000
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.1).cuda()
    wav = rnd(B, T)
    ws = [rnd(7, 32), rnd(32), rnd(16, 3, 32), rnd(16), rnd(32, 16), rnd(32), rnd(32, 32), rnd(32), rnd(64, 2 * r, 32), rnd(64)]
    y = torch.empty(B, T // r, 64, device="cuda")
    p = lambda t: t.data_ptr()
    call = lambda: check(lib.wt_resblock_down(p(wav), *[p(w) for w in ws], p(y), B, T, r, None), "wt_resblock_down")
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        print("%.1f" % run(int(sys.argv[2])))
        sys.exit(0)
    r = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    masks = [int(m) for m in sys.argv[2:]] or [0]
    prev = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "lib", "libwavtok_hip_prev.so")
    for m in masks:                      # the mask is read once per process: one child per mask
        env = dict(os.environ, WT_RB16_DBG=str(m))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(r)], env=env, capture_output=True, text=True)
        print("r=%d dbg=%2d: %s us" % (r, m, out.stdout.strip() or out.stderr.strip()[-300:]), flush=True)
        if m == 0 and os.path.exists(prev):          # the same call on tools/build_prev.sh's library, same box
            env["WAVTOK_HIP_LIB"] = prev
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(r)], env=env, capture_output=True, text=True)
            print("r=%d prev  : %s us" % (r, out.stdout.strip() or out.stderr.strip()[-300:]), flush=True)
