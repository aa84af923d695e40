#!/usr/bin/env python3
"""wt_linear: fp32-MFMA chain vs the fp32-equivalent split-f16 kernel — accuracy against float64 and speed."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavtokenizer_amd._capi import lib, check

def ptr(t): return ctypes.c_void_p(t.data_ptr())

def run(M, N, K, mode, reps=20):
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K, generator=g))).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    y = torch.empty(M, N, device="cuda")
    ws = torch.empty(4 * N * K, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        check(lib.wt_linear(ptr(x), ptr(w), ptr(b), ptr(y), M, N, K, mode, ptr(ws), None), "wt_linear")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        lib.wt_linear(ptr(x), ptr(w), ptr(b), ptr(y), M, N, K, mode, ptr(ws), None)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ref = x[:512].double() @ w.double().t() + b.double()
    err = ((y[:512].double() - ref).norm() / ref.norm()).item()
    return ms, 2.0 * M * N * K / ms / 1e9, err

if __name__ == "__main__":
    for name, M, N, K in [("pwconv1", 7680, 2304, 768), ("pwconv2", 7680, 768, 2304), ("hop320 pw1", 14400, 2304, 768),
                          ("big", 16384, 4096, 1024)]:
        for mode in (0, 1):
            ms, tf, err = run(M, N, K, mode)
            print(f"{name:12s} {'f16x3' if mode else 'fp32 '}  {ms*1e3:8.1f} us  {tf:7.1f} TFLOP/s(fp32-equiv)  rel err vs fp64 {err:.2e}"
                  + ("  (includes the weight split pass)" if mode else ""), flush=True)
