"""Experiment: does running step i+1's encode_infer beside step i's decode (two HIP streams, two engines) raise throughput?
Encodes are chained by events so that two persistent LSTM launches never share the GPU.

    python tools/two_stream.py [clips] [samples] [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(B=64, T=72000, steps=40, lanes=3):
    import torch
    from wavtokenizer_amd import WavTokenizer, ARCH_HOP600, synth
    sd = synth.make_state_dict(ARCH_HOP600, seed=0)
    ms = []
    for _ in range(lanes):
        m = WavTokenizer.from_arch(ARCH_HOP600)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        ms.append(m.eval().to("cuda"))
    wav = torch.from_numpy(synth.make_clips(B, T, seed=2000)).cuda()
    bw = torch.tensor([0])
    streams = [torch.cuda.Stream() for _ in range(lanes)]
    torch.cuda.synchronize()

    def run(n, two):
        enc_done = None
        for i in range(n):
            k = i % two
            st = streams[k]
            with torch.cuda.stream(st):
                if enc_done is not None and two > 1:
                    st.wait_event(enc_done)
                f, c = ms[k].encode_infer(wav, bandwidth_id=bw)
                enc_done = torch.cuda.Event()
                enc_done.record(st)
                ms[k].decode(f, bandwidth_id=bw)
        torch.cuda.synchronize()

    for two in (1, 2, 3, 1, 2, 3)[:2 * lanes]:
        if two > lanes:
            continue
        run(6, two)
        t0 = time.perf_counter()
        run(steps, two)
        dt = time.perf_counter() - t0
        print("%d lane(s): %.3f ms/step  %.0f audio-s/s" % (two, 1e3 * dt / steps, B * T / 24000 * steps / dt), flush=True)
    for m in ms:
        m.check_status()


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:5]))
