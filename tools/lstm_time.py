"""Time the encoder's SLSTM alone (WT_PLAN_UNIT_LSTM: input projection + recurrence) at B x L; env WT_LSTM_GROUPS=1 = one clip group.
    python tools/lstm_time.py [B] [L]"""
# the WT_* switches these measurements flip exist in the LAB build only (the product library reads no environment variable)
import os as _os
_os.environ.setdefault("WAVTOK_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tools", "lib", "libwavtok_hip_lab.so"))
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavtokenizer_amd import WavTokenizer, ARCH_HOP600, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 120
sd = synth.make_state_dict(ARCH_HOP600, seed=0)
m = WavTokenizer.from_arch(ARCH_HOP600)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
m = m.eval().to("cuda")
x = torch.randn(B, L, 512, generator=torch.Generator().manual_seed(1)).cuda()
for _ in range(3):
    y = m._run_unit_lstm(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    y = m._run_unit_lstm(x)
e1.record()
torch.cuda.synchronize()
if not os.environ.get('WAVTOK_HIP_LIB'):
    m.check_status()
print("B=%d L=%d lib=%s: %.1f us per call (input projection + recurrence + copies), checksum %.6f" %
      (B, L, os.environ.get("WAVTOK_HIP_LIB", "current")[-12:], e0.elapsed_time(e1) / 20 * 1e3, float(y.double().sum())))
