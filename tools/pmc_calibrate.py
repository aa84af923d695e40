#!/usr/bin/env python3
"""Known-byte-count launches for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950
(MI355X_MICROARCH.md, HBM section: 'calibrate on a known byte count in your own access pattern').

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_calibrate.py

1. D2D copy of 1 GiB (wide coalesced stream): reads 1 GiB, writes 1 GiB.
2. The implicit-GEMM kernel in its own access pattern (8 lanes x 16 B per 128-B row piece):
   A = 2^20 rows x 256 floats = 1 GiB read exactly once (one column tile, Cout = 96), W 96 KiB,
   C = 2^20 x 96 floats = 384 MiB written once.  Both exceed the 256 MiB Infinity Cache.
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavtokenizer_amd._capi import lib, check

def ptr(t): return ctypes.c_void_p(t.data_ptr())

src = torch.randn(1 << 28, device="cuda")            # 1 GiB
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()
Bc, T, Cin, Cout = 8, 1 << 17, 256, 96        # 8 clips x 128 MiB (one plan window must stay < 1 GiB)
x = src.view(Bc, T, Cin)
w = torch.randn(Cout, 1, Cin, device="cuda") * 0.05
b = torch.zeros(Cout, device="cuda")
y = torch.empty(Bc, T, Cout, device="cuda")
for _ in range(3):
    check(lib.wt_sconv1d(ptr(x), ptr(w), ptr(b), ptr(y), Bc, T, Cin, Cout, 1, 1, 1, 0, None), "sconv")
torch.cuda.synchronize()
print("done: copy 1 GiB x3, gemm A=1 GiB C=384 MiB x3")
