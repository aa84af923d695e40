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
for B in (1, 4, 16):
    wav = torch.from_numpy(synth.make_clips(B, 72000, seed=5)).cuda()
    for _ in range(3):
        f, c = m.encode_infer(wav, bandwidth_id=bw); o = m.decode(f, bandwidth_id=bw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        f, c = m.encode_infer(wav, bandwidth_id=bw); o = m.decode(f, bandwidth_id=bw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"B={B}: {dt*1e3:.2f} ms per round trip = {B*3/dt:.0f} audio-s/s", flush=True)
