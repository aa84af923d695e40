#!/usr/bin/env python3
"""Schedules for K round trips of 64 x 3 s, interleaved blocks in one process: one stream; two alternating lanes
(StepRunner(lanes=2), the headline); an encode stream + a decode stream (step i's decode waits for its encode's event, so
encode(i+1) always runs beside decode(i) and only one encode / one decode is ever in flight).   python tools/ed_pipeline.py"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS, synth
from wavtokenizer_amd.sharding import StepRunner

arch = NAMED_ARCHS["hop600"]
m = WavTokenizer.from_arch(arch)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(arch, seed=0).items()}, strict=False)
m = m.eval().to("cuda")
m._engine.max_streams = 8
bw = torch.tensor([0])
wav = torch.from_numpy(synth.make_clips(64, 72000, seed=2000)).cuda()
dev = wav.device
f0, c0 = m.encode_infer(wav, bandwidth_id=bw)
w0 = m.decode(f0, bandwidth_id=bw)
torch.cuda.synchronize()


class EDRunner:
    def __init__(self, depth=2):
        self.se, self.sd = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        self.last = None
        self.depth = depth
        self.pending = []

    def step(self):
        # at most `depth` encodes ahead of their decodes (the feature tensors live until their decode has run)
        if len(self.pending) >= self.depth:
            self.se.wait_event(self.pending.pop(0))
        with torch.cuda.stream(self.se):
            feats, codes = m.encode_infer(wav, bandwidth_id=bw)
            ev = torch.cuda.Event()
            ev.record()
        with torch.cuda.stream(self.sd):
            self.sd.wait_event(ev)
            feats.record_stream(self.sd)
            out = m.decode(feats, bandwidth_id=bw)
            done = torch.cuda.Event()
            done.record()
        self.pending.append(done)
        self.last = (codes, out)
        return codes, out, None

    def drain(self):
        self.se.synchronize()
        self.sd.synchronize()


def block(r, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        res = r.step()
    r.drain()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, res


runners = {"one stream": StepRunner(m, wav, bw, None, 1, 0, False, "nccl", lanes=1),
           "two alternating lanes": StepRunner(m, wav, bw, None, 1, 0, False, "nccl", lanes=2),
           "encode stream + decode stream, depth 2": EDRunner(2), "encode stream + decode stream, depth 3": EDRunner(3)}
for r in runners.values():
    for _ in range(4):
        r.step()
    r.drain()
res = {k: [] for k in runners}
for rep in range(6):
    for k, r in runners.items():
        ms, out = block(r)
        res[k].append(ms)
        assert torch.equal(out[0], c0) and torch.equal(out[1], w0), k
for k, v in res.items():
    print(f"{k:44s} median {statistics.median(v):.3f} ms/step  (min {min(v):.3f}, max {max(v):.3f})")
m.check_status()
