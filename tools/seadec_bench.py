#!/usr/bin/env python3
"""Time the secondary path feature_extractor.encodec.decoder(z) (SEANetDecoder) per plan step."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavtokenizer_amd import WavTokenizer, ARCH_HOP600, synth, _capi
B, L = 64, 120
sd = synth.make_state_dict(ARCH_HOP600, seed=0, with_seanet_decoder=True)
m = WavTokenizer.from_arch(ARCH_HOP600)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
m = m.eval().cuda()
z = torch.randn(B, 512, L, device="cuda") * 0.6
dec = m.feature_extractor.encodec.decoder
for _ in range(2):
    dec(z)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    dec(z)
e1.record(); torch.cuda.synchronize()
print(f"seanet_decoder B={B} L={L}: {e0.elapsed_time(e1)/5:.3f} ms per call = {B*3/(e0.elapsed_time(e1)/5e3):.0f} audio-s/s")
plan = m._engine.plans[(_capi.WT_PLAN_SEANET_DECODER, B, L, 0)][0]
n = _capi.lib.wt_plan_num_steps(plan)
names = []
for i in range(n):
    p = ctypes.c_char_p(); _capi.lib.wt_plan_step_name(plan, i, ctypes.byref(p)); names.append(p.value.decode())
for filt in sorted(set(names)):
    _capi.lib.wt_plan_set_timing(plan, filt.encode())
    for _ in range(3):
        dec(z)
    ms, cnt = ctypes.c_double(), ctypes.c_int64()
    _capi.lib.wt_plan_read_timing(plan, ctypes.byref(ms), ctypes.byref(cnt), 1)
    print(f"  step {filt:24s} {ms.value/3:8.3f} ms ({cnt.value//3} launches)")
