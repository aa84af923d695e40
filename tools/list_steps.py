"""Print the step (launch) names of the encode and decode plans at the benchmark shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes
from wavtokenizer_amd import WavTokenizer, ARCH_HOP600, synth, _capi
sd = synth.make_state_dict(ARCH_HOP600, seed=0)
m = WavTokenizer.from_arch(ARCH_HOP600)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
m = m.eval().to("cuda")
wav = torch.from_numpy(synth.make_clips(64, 72000, seed=1)).cuda()
bw = torch.tensor([0])
f, c = m.encode_infer(wav, bandwidth_id=bw); y = m.decode(f, bandwidth_id=bw)
lib = _capi.lib
for key, (plan, ws) in m._engine.plans.items():
    names = []
    for i in range(lib.wt_plan_num_steps(plan)):
        p = ctypes.c_char_p()
        assert lib.wt_plan_step_name(plan, i, ctypes.byref(p)) == 0
        names.append(p.value.decode())
    print(key, len(names), names)
