#!/usr/bin/env python3
"""gemm16s (S32 split-f16, LDS-DMA) vs gemm16 (in-loop split) vs the fp32 MFMA chain: accuracy against float64
and per-kernel time.  Run under `rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/gemm16s_bench.py
<manifest.json>`; then `python3 tools/gemm16s_bench.py --summarize <manifest.json> <dir>` pairs the trace's GEMM
dispatches with the calls listed in the manifest (the call order is the dispatch order)."""
# the WT_* switches these measurements flip exist in the LAB build only (the product library reads no environment variable)
import os as _os
_os.environ.setdefault("WAVTOK_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tools", "lib", "libwavtok_hip_lab.so"))
import csv, ctypes, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summarize(manifest, trace_dir):
    calls = json.load(open(manifest))
    f = sorted(glob.glob(os.path.join(trace_dir, "**", "*_kernel_trace.csv"), recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"] and "split" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    i = 0
    for c in calls:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i:i + c["n"]]]
        name = rows[i]["Kernel_Name"] if i < len(rows) else "?"
        i += c["n"]
        d = sorted(d[c["warm"]:])
        us = d[len(d) // 2]
        print(f'{c["label"]:34s} {us:8.1f} us  {c["flop"] / us / 1e6:7.1f} TF(fp32-equiv)  err {c["err"]:.2e}   {name[:70]}')
    if i != len(rows):
        print(f"warning: {len(rows)} GEMM dispatches in the trace, manifest accounts for {i}")


def main():
    import torch
    from wavtokenizer_amd._capi import lib, check
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    calls = []
    WARM, REPS = 2, 8

    def decode_s32(y, M, N):
        h = y.view(torch.float16).view(M, N // 32, 2, 32).float()
        return (h[:, :, 0, :] + h[:, :, 1, :] / 2048.0).reshape(M, N)

    def linear(label, M, N, K, mode, tile=None):
        g = torch.Generator().manual_seed(M + N + K)
        x = (torch.randn(M, K, generator=g) * torch.exp(0.5 * torch.randn(M, K, generator=g))).cuda()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        b = torch.randn(N, generator=g).cuda()
        y = torch.zeros(M, N, device="cuda")
        ws = torch.zeros(4 * (N + M) * K + 8192, dtype=torch.uint8, device="cuda")
        if tile is not None:
            os.environ["WT_GEMM16S_TILE"] = str(tile)
        else:
            os.environ.pop("WT_GEMM16S_TILE", None)
        for _ in range(WARM + REPS):
            check(lib.wt_linear(ptr(x), ptr(w), ptr(b), ptr(y), M, N, K, mode, ptr(ws), None), "wt_linear")
        torch.cuda.synchronize()
        if os.environ.get("WT_GEMM16S_DBG"):      # in-kernel clock of the timing-experiment builds: d(s_memtime) / d(s_memrealtime) x 100 MHz
            st = ws[4 * (N + M) * K + 256: 4 * (N + M) * K + 256 + 4096].view(torch.int64).view(256, 2).cpu().double()
            ok = st[:, 1] > 0
            if ok.any():
                ghz = (st[ok, 0] / st[ok, 1] * 0.1).median().item()
                print(f"   in-kernel clock (median over {int(ok.sum())} workgroups): {ghz:.3f} GHz", flush=True)
                label = f"{label} [{ghz:.2f} GHz]"
        out = decode_s32(y, M, N) if mode >= 3 else y
        act = (lambda v: torch.nn.functional.gelu(v)) if mode == 4 else (lambda v: v)
        ref = act(x[:1024].double() @ w.double().t() + b.double())
        err = ((out[:1024].double() - ref).norm() / ref.norm()).item()
        ref2 = act(x[-256:].double() @ w.double().t() + b.double())
        err = max(err, ((out[-256:].double() - ref2).norm() / ref2.norm()).item())
        calls.append({"label": label, "n": WARM + REPS, "warm": WARM, "flop": 2.0 * M * N * K, "err": err})
        print(label, "err", err, flush=True)

    def conv(label, B, T, Cin, Cout, k, stride, zero_same, tile=None):
        g = torch.Generator().manual_seed(B + T + Cin + k)
        x = torch.randn(B, T, Cin, generator=g).cuda()
        w = (torch.randn(Cout, k, Cin, generator=g) / (k * Cin) ** 0.5).cuda()
        b = torch.randn(Cout, generator=g).cuda()
        Tout = T if zero_same else -(-T // stride)
        y = torch.zeros(B, Tout, Cout, device="cuda")
        ws = torch.empty(4 * (B * T * Cin + Cout * k * Cin) + 1024, dtype=torch.uint8, device="cuda")
        if tile is not None:
            os.environ["WT_GEMM16S_TILE"] = str(tile)
        else:
            os.environ.pop("WT_GEMM16S_TILE", None)
        for _ in range(WARM + REPS):
            check(lib.wt_conv1d_s32(ptr(x), ptr(w), ptr(b), ptr(y), B, T, Cin, Cout, k, stride, zero_same, ptr(ws), None), "wt_conv1d_s32")
        torch.cuda.synchronize()
        xd = x[:4].double().transpose(1, 2)
        wd = w.double().permute(0, 2, 1)
        if zero_same:
            ref = torch.nn.functional.conv1d(xd, wd, b.double(), padding=(k - 1) // 2)
        else:
            pt = k - stride
            extra = (Tout - 1) * stride + k - pt - T
            pr = pt // 2
            ref = torch.nn.functional.conv1d(torch.nn.functional.pad(xd, (pt - pr, pr + extra), mode="reflect"), wd, b.double(), stride=stride)
        err = ((y[:4].double() - ref.transpose(1, 2)).norm() / ref.norm()).item()
        calls.append({"label": label, "n": WARM + REPS, "warm": WARM, "flop": 2.0 * B * Tout * Cout * k * Cin, "err": err})
        print(label, "err", err, flush=True)

    if len(sys.argv) > 2 and sys.argv[2] == "pmc":
        WARM, REPS = 1, 3
        linear("pwconv1 s32 128x192x3", 7680, 2304, 768, 2, 2)
        os.environ["WT_GEMM16S_DBG"] = "5"
        linear("pwconv1 s32 128x192x3 no DMA, no epilogue", 7680, 2304, 768, 2, 2)
        os.environ.pop("WT_GEMM16S_DBG")
        json.dump(calls, open(sys.argv[1], "w"))
        return
    if len(sys.argv) > 2 and sys.argv[2] == "dbg":
        WARM, REPS = 30, 40             # long enough for the chip to settle at the clock it holds under this load
        for mode, mn in ((2, "bias -> fp32"),):
            linear(f"pwconv1 {mn} shipped build", 7680, 2304, 768, mode, 2)
            for dbg, dn in ((1024, "full (stamped)"), (4, "no epilogue"), (5, "no epilogue, no DMA"), (13, "no epilogue, no DMA, no barrier"),
                            (45, "MFMA + LDS fragment reads"), (61, "MFMA only"), (21, "no epilogue, no DMA, no LDS reads"), (64, "full, younger half at priority 1")):
                os.environ["WT_GEMM16S_DBG"] = str(dbg)
                linear(f"pwconv1 {mn}: {dn}", 7680, 2304, 768, mode, 2)
                os.environ.pop("WT_GEMM16S_DBG")
        json.dump(calls, open(sys.argv[1], "w"))
        return
    if len(sys.argv) > 2 and sys.argv[2] == "tune":
        for name, M, N, K in [("pwconv1", 7680, 2304, 768), ("pwconv2", 7680, 768, 2304)]:
            for tile, tn in ((2, "128x192x3"), (3, "128x128x3")):
                linear(f"{name} s32 {tn}", M, N, K, 2, tile)
            os.environ["WT_GEMM16S_NONPERSISTENT"] = "1"
            linear(f"{name} s32 128x192x3 non-persistent", M, N, K, 2, 2)
            os.environ.pop("WT_GEMM16S_NONPERSISTENT")

        json.dump(calls, open(sys.argv[1], "w"))
        return
    shapes = [("pwconv1", 7680, 2304, 768), ("pwconv2", 7680, 768, 2304), ("head", 7680, 2432, 768),
              ("hop320 pw1", 14400, 2304, 768), ("ragged", 7000, 800, 96)]
    for name, M, N, K in shapes:
        linear(f"{name} fp32", M, N, K, 0)
        for tile, tn in ((2, "128x192"), (3, "128x128")):
            linear(f"{name} s32 {tn}", M, N, K, 2, tile)
        if N % 32 == 0:
            linear(f"{name} s32 auto ->S32", M, N, K, 3)
    for tile, tn in ((2, "128x192"), (3, "128x128")):
        conv(f"res k3 768 {tn}", 64, 120, 768, 768, 3, 1, 1, tile)
        conv(f"embed k7 512 {tn}", 64, 120, 512, 768, 7, 1, 1, tile)
    conv("down k10 s5 reflect", 8, 3600, 128, 256, 10, 5, 0)
    conv("k7 reflect short", 3, 5, 512, 512, 7, 1, 0)
    conv("k3 zero T=1", 2, 1, 768, 768, 3, 1, 1)
    json.dump(calls, open(sys.argv[1], "w"))


if __name__ == "__main__":
    if sys.argv[1] == "--summarize":
        summarize(sys.argv[2], sys.argv[3])
    else:
        main()
