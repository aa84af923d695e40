"""Round-trip latency at small batch sizes (BASELINE configs[0] is B = 1): launch-bound, ~3.2 ms at B = 1 on MI355X."""
import os
import sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS, synth
arch = NAMED_ARCHS["hop600"]
sd = synth.make_state_dict(arch, seed=0)
m = WavTokenizer.from_arch(arch)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
m = m.eval().cuda()
bw = torch.tensor([0])
for B in [int(b) for b in os.environ.get("WT_SB_BATCHES", "1,4,8,16,32").split(",")]:
    wav = torch.from_numpy(synth.make_clips(B, 72000, seed=5)).cuda()
    for _ in range(3):
        f, c = m.encode_infer(wav, bandwidth_id=bw); o = m.decode(f, bandwidth_id=bw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        f, c = m.encode_infer(wav, bandwidth_id=bw); o = m.decode(f, bandwidth_id=bw)
    th = (time.perf_counter() - t0) / 20            # host time to enqueue one round trip
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    # one call at a time (what a latency-sensitive caller sees)
    lat = []
    for _ in range(10):
        torch.cuda.synchronize(); t1 = time.perf_counter()
        f, c = m.encode_infer(wav, bandwidth_id=bw); o = m.decode(f, bandwidth_id=bw)
        torch.cuda.synchronize(); lat.append(time.perf_counter() - t1)
    lat.sort()
    print(f"B={B}: {dt*1e3:.2f} ms per round trip back to back = {B*3/dt:.0f} audio-s/s; host enqueue {th*1e3:.2f} ms; "
          f"single-call latency p50 {lat[len(lat)//2]*1e3:.2f} ms", flush=True)
