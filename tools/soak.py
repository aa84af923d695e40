#!/usr/bin/env python3
"""Soak of the headline schedule: N round trips of 64 x 3 s issued alternately on two HIP streams (StepRunner(lanes=2)),
every 25th step's codes and waveform compared bit for bit with the one-stream result, check_status() at the end (a lost
LSTM co-residency or a range report anywhere in the run raises), then the host pipeline for N / 4 steps with its PCM output
checked the same way.      python tools/soak.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS, synth, audio
from wavtokenizer_amd.sharding import StepRunner, HostPipeline


def main(n):
    arch = NAMED_ARCHS["hop600"]
    m = WavTokenizer.from_arch(arch)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(arch, seed=0).items()}, strict=False)
    m = m.eval().to("cuda")
    bw = torch.tensor([0])
    clips = synth.make_clips(64, 72000, seed=2000)
    wav = torch.from_numpy(clips).cuda()
    f0, c0 = m.encode_infer(wav, bandwidth_id=bw)
    w0 = m.decode(f0, bandwidth_id=bw)
    torch.cuda.synchronize()
    r = StepRunner(m, wav, bw, None, 1, 0, False, "nccl", lanes=2)
    t0 = time.perf_counter()
    bad = 0
    keep = []
    for i in range(n):
        codes, out, _ = r.step()
        if i % 25 == 0:
            keep.append((i, codes, out))
        if len(keep) >= 8:
            r.drain()
            torch.cuda.synchronize()
            for j, c, o in keep:
                if not (torch.equal(c, c0) and torch.equal(o, w0)):
                    bad += 1
                    print("MISMATCH at step", j)
            keep = []
    r.drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for j, c, o in keep:
        if not (torch.equal(c, c0) and torch.equal(o, w0)):
            bad += 1
    m.check_status()
    print(f"two lanes: {n} steps in {dt:.2f} s = {1e3 * dt / n:.3f} ms/step, {bad} mismatching samples, persistent LSTM still on: {m.persistent_lstm}")
    hp = HostPipeline(m, 64, 72000, bw, lanes=2)
    for k in range(2):
        hp.h_in[k].copy_(torch.from_numpy(clips))
    want = audio.to_pcm16(w0).cpu()
    t0 = time.perf_counter()
    nh = max(8, n // 4)
    badh = 0
    for i in range(nh):
        k = hp.step()
        if i % 50 == 49:
            hp.drain()
            badh += int(not torch.equal(hp.h_out[0], want)) + int(not torch.equal(hp.h_out[1], want))
    hp.drain()
    dt = time.perf_counter() - t0
    badh += int(not torch.equal(hp.h_out[0], want)) + int(not torch.equal(hp.h_out[1], want))
    m.check_status()
    print(f"host pipeline: {nh} steps in {dt:.2f} s = {1e3 * dt / nh:.3f} ms/step, {badh} mismatching buffers")
    assert bad == 0 and badh == 0


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3000)
