#!/usr/bin/env python3
"""How fast is the host -> device leg of the host pipeline (sharding.HostPipeline), alone and beside the codec's kernels?
18.4 MB of pinned fp32 (64 x 3 s) per copy; events on the copy stream.  Also: the encoder reading the waveform straight from
pinned (device-mapped) host memory instead of a copy (zero-copy input), timed as the whole encode_infer call.
    python tools/h2d_probe.py"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS, synth, _capi

dev = torch.device("cuda", 0)
arch = NAMED_ARCHS["hop600"]
sd = synth.make_state_dict(arch, seed=0)
m = WavTokenizer.from_arch(arch)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
m = m.eval().to(dev)
B, T = 64, 72000
h = torch.from_numpy(synth.make_clips(B, T, seed=5)).pin_memory()
d = torch.empty((B, T), dtype=torch.float32, device=dev)
bw = torch.tensor([0])
s_copy = torch.cuda.Stream(device=dev)


def copy_ms(n=8):
    out = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s_copy):
            e0.record()
            d.copy_(h, non_blocking=True)
            e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1))
    return sorted(out)[len(out) // 2]


f, c = m.encode_infer(d, bandwidth_id=bw)
m.decode(f, bandwidth_id=bw)
torch.cuda.synchronize()
print("H2D alone: %.3f ms = %.1f GB/s" % (copy_ms(), B * T * 4 / copy_ms() / 1e6))
# beside compute: keep the default stream busy with round trips while the copies run
t_end = time.time() + 3.0
vals = []
while time.time() < t_end:
    for _ in range(4):
        f, c = m.encode_infer(d, bandwidth_id=bw)
        m.decode(f, bandwidth_id=bw)
    vals.append(copy_ms(1))
    torch.cuda.synchronize()
vals.sort()
print("H2D beside the codec's kernels: median %.3f ms (min %.3f, max %.3f) = %.1f GB/s" % (vals[len(vals) // 2], vals[0], vals[-1], B * T * 4 / vals[len(vals) // 2] / 1e6))

# zero-copy: the stage-1 kernel reads the pinned host buffer itself (device-mapped host memory, the same pointer)
L = arch.frames(T)
flags = m._graph_flags(B)
plan, ws = m._engine.plan(_capi.WT_PLAN_ENCODE, B, T, flags, dev)
feats = torch.empty((B, 512, L), dtype=torch.float32, device=dev)
codes = torch.empty((1, B, L), dtype=torch.int64, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def enc_ms(ptr, n=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        _capi.check(_capi.lib.wt_encode(plan, ctypes.c_void_p(ptr), ctypes.c_void_p(feats.data_ptr()), ctypes.c_void_p(codes.data_ptr()), None,
                                        ctypes.c_void_p(ws.data_ptr()), st), "wt_encode")
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


enc_ms(d.data_ptr(), 3)
a = enc_ms(d.data_ptr())
c_dev = codes.clone()
try:
    b = enc_ms(h.data_ptr())
    same = bool(torch.equal(codes, c_dev))
    print("encode_infer, input in HBM: %.3f ms; input read from pinned host memory by the stage-1 kernel: %.3f ms (codes equal: %s)" % (a, b, same))
except Exception as e:
    print("zero-copy input failed:", repr(e)[:200])
m.check_status()
