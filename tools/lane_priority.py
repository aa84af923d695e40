#!/usr/bin/env python3
"""Two lanes with equal HIP stream priorities against one high- and one normal-priority lane, interleaved blocks in one
process (StepRunner(lanes=2); 64 x 3 s).      python tools/lane_priority.py"""
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


def runner(prios):
    r = StepRunner(m, wav, bw, None, 1, 0, False, "nccl", lanes=len(prios))
    r.streams = [torch.cuda.Stream(device=dev, priority=p) for p in prios]
    for _ in range(2 * len(prios)):
        r.step()
    r.drain()
    torch.cuda.synchronize()
    return r


def block(r, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r.step()
    r.drain()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


variants = {"one lane": None, "two lanes, equal priority": (0, 0), "two lanes, high + normal": (-1, 0), "two lanes, both high": (-1, -1)}
runners = {}
for k, p in variants.items():
    runners[k] = StepRunner(m, wav, bw, None, 1, 0, False, "nccl", lanes=1) if p is None else runner(p)
res = {k: [] for k in variants}
for rep in range(6):
    for k, r in runners.items():
        res[k].append(block(r))
for k, v in res.items():
    print(f"{k:32s} median {statistics.median(v):.3f} ms/step  (min {min(v):.3f}, max {max(v):.3f})")
m.check_status()
