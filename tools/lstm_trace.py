# the WT_* switches these measurements flip exist in the LAB build only (the product library reads no environment variable)
import os as _os
_os.environ.setdefault("WAVTOK_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tools", "lib", "libwavtok_hip_lab.so"))
import ctypes, os, sys
os.environ["WT_LSTM_TRACE"] = "1"
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS, synth, _capi
arch = NAMED_ARCHS["hop600"]
sd = synth.make_state_dict(arch, seed=0)
m = WavTokenizer.from_arch(arch)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
m = m.eval().cuda()
m.set_debug_keep_stages(True)
B = int(os.environ.get("B", "64"))
wav = torch.from_numpy(synth.make_clips(B, 72000, seed=5)).cuda()
bw = torch.tensor([0])
for _ in range(3):
    m.encode_infer(wav, bandwidth_id=bw)
torch.cuda.synchronize()
key = [k for k in m._engine.plans if k[0] == _capi.WT_PLAN_ENCODE][0]
plan, ws = m._engine.plans[key]
i = 0
name = None
while True:
    p = ctypes.c_char_p()
    if _capi.lib.wt_plan_buffer_name(plan, i, ctypes.byref(p)) != 0: break
    if p.value.decode().endswith(".hx"): name = p.value
    i += 1
off, n = ctypes.c_size_t(), ctypes.c_size_t()
_capi.check(_capi.lib.wt_plan_find_buffer(plan, name, ctypes.byref(off), ctypes.byref(n)), "find")
buf = ws[off.value: off.value + 4 * n.value].view(torch.int32).cpu().numpy().view(np.uint32)
hxn = 8 * 3 * 2 * 16 * 2048 // 4
ctl = buf[hxn:]
t = np.zeros((8, 6))
for s in range(8):
    for ph in range(6):
        lo, hi = int(ctl[520 + (s * 6 + ph) * 2]), int(ctl[521 + (s * 6 + ph) * 2])
        t[s, ph] = ((hi << 32) | lo) * 10.0   # ns at 100 MHz
names = ["start->polled+staged", "barrier1", "mfma+gbuf", "barrier2", "cell+stores"]
d = np.diff(t, axis=1)
print("per-phase ns (steps 64..71):")
for s in range(8): print(" ".join(f"{x:7.0f}" for x in d[s]), " | step total", (t[s+1,0]-t[s,0]) if s < 7 else "")
print("mean", " ".join(f"{x:7.0f}" for x in d.mean(0)), names)
