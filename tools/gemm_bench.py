#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM kernel through wt_sconv1d (k=1 -> plain GEMM).
Usage (GPU box): WT_GEMM_TILE=<n> python tools/gemm_bench.py"""
# the WT_* switches these measurements flip exist in the LAB build only (the product library reads no environment variable)
import os as _os
_os.environ.setdefault("WAVTOK_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tools", "lib", "libwavtok_hip_lab.so"))
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavtokenizer_amd._capi import lib, check

def ptr(t): return ctypes.c_void_p(t.data_ptr())

def bench(B, T, Cin, Cout, k, stride=1, reps=30):
    x = torch.randn(B, T, Cin, device="cuda")
    w = torch.randn(Cout, k, Cin, device="cuda") / (Cin * k) ** 0.5
    b = torch.randn(Cout, device="cuda")
    Tout = -(-T // stride)
    y = torch.empty(B, Tout, Cout, device="cuda")
    for _ in range(3):
        check(lib.wt_sconv1d(ptr(x), ptr(w), ptr(b), ptr(y), B, T, Cin, Cout, k, stride, 1, 0, None), "sconv")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        lib.wt_sconv1d(ptr(x), ptr(w), ptr(b), ptr(y), B, T, Cin, Cout, k, stride, 1, 0, None)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * Tout * Cout * Cin * k
    return ms, fl / ms / 1e9

if __name__ == "__main__":
    tile = os.environ.get("WT_GEMM_TILE", "default")
    shapes = [("pwconv1 7680x2304x768", 64, 120, 768, 2304, 1), ("pwconv2 7680x768x2304", 64, 120, 2304, 768, 1),
              ("res k3 7680x768x2304", 64, 120, 768, 768, 3), ("hop320 pw1 14400x2304x768", 64, 225, 768, 2304, 1),
              ("hop320 pw2 14400x768x2304", 64, 225, 2304, 768, 1), ("big 16384x4096x1024", 1, 16384, 1024, 4096, 1)]
    for name, B, T, Cin, Cout, k in shapes:
        ms, tf = bench(B, T, Cin, Cout, k)
        print(f"tile={tile:8s} {name:32s} {ms*1e3:8.1f} us  {tf:7.1f} TFLOP/s", flush=True)
