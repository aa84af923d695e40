import sys, torch
sys.path.insert(0, "/root/repo")
from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS, synth
arch = NAMED_ARCHS["hop600"]
sd = synth.make_state_dict(arch, seed=0)
m = WavTokenizer.from_arch(arch)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
m = m.eval().cuda()
bw = torch.tensor([0])
for B, T in ((4, 720000), (33, 240000), (70, 48000)):
    wav = torch.from_numpy(synth.make_clips(B, T, seed=77 + B)).cuda()
    f1, c1 = m.encode_infer(wav, bandwidth_id=bw)
    m.set_lstm_mode("step")
    f2, c2 = m.encode_infer(wav, bandwidth_id=bw)
    m.set_lstm_mode("persistent")
    torch.cuda.synchronize()
    print(B, T, c1.shape, "codes equal:", bool(torch.equal(c1, c2)), "mismatches:", int((c1 != c2).sum()))
