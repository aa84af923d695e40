"""Helpers around the hot path: oracle vs the reference's own outputs (overlap-add), oracle properties
(resampler: torchaudio is absent, parity unpinned), and the HIP kernels vs the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import audio_ref

HERE = os.path.dirname(os.path.abspath(__file__))


def _ola_cases():
    z = np.load(os.path.join(HERE, "golden", "overlap_add.npz"))
    ci = 0
    while f"case{ci}_meta" in z:
        nf, fl, ll, st, B = (int(v) for v in z[f"case{ci}_meta"])
        yield ci, [z[f"case{ci}_frame{i}"] for i in range(nf)], st, z[f"case{ci}_out"]
        ci += 1


def test_overlap_add_oracle_matches_the_reference_outputs():
    n = 0
    for _ci, frames, stride, want in _ola_cases():
        got = audio_ref.linear_overlap_add(frames, stride)
        assert got.dtype == want.dtype and np.array_equal(got, want)
        n += 1
    assert n >= 6


def test_resample_oracle_properties():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 2, 4410)).astype(np.float32)
    same = audio_ref.convert_audio(x, 24000, 24000, 1)
    assert same.shape == (2, 1, 4410) and np.array_equal(same[:, 0], x.mean(1, dtype=np.float32))
    for sr, n in ((48000, 9600), (44100, 8820), (16000, 3200), (22050, 4410)):
        t = np.arange(n) / sr
        tone = (0.5 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)[None, None]
        y = audio_ref.convert_audio(tone, sr, 24000, 1)
        L = int(np.ceil(24000 * n / sr))
        assert y.shape == (1, 1, L)                                   # torchaudio's length rule
        want = 0.5 * np.sin(2 * np.pi * 440.0 * np.arange(L) / 24000)
        mid = slice(200, L - 200)                                     # away from the zero-padded edges
        assert np.abs(y[0, 0, mid] - want[mid]).max() < 2e-3          # a 440 Hz tone is far inside the pass band
    a = rng.standard_normal((1, 1, 3000)).astype(np.float32)
    b = rng.standard_normal((1, 1, 3000)).astype(np.float32)
    ya, yb, yab = (audio_ref.convert_audio(v, 44100, 24000, 1) for v in (a, b, a + b))
    assert np.abs(yab - (ya + yb)).max() < 1e-5                       # linear


def test_pcm16_oracle():
    x = np.array([0.0, 0.5, -0.5, 1.5, -1.5, 0.99, 1e-5], np.float32)
    got = audio_ref.to_pcm16(x)
    assert got.dtype == np.int16 and list(got[:3]) == [0, 16384, -16384]
    assert got[3] == got[5] == int(np.rint(np.float32(0.99) * 32768)) and got[4] == -got[3]
    assert np.abs(audio_ref.to_pcm16(x * 4, rescale=True)).max() == int(np.rint(np.float32(0.99) * 32768))


# ----------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_overlap_add_kernel_bit_exact():
    from wavtokenizer_amd import audio
    for _ci, frames, stride, want in _ola_cases():
        got = audio.linear_overlap_add([torch.from_numpy(f).cuda() for f in frames], stride)
        assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("sr,C,T", [(48000, 2, 96000), (44100, 1, 50000), (16000, 2, 16001), (24000, 2, 1000), (22050, 1, 7)])
def test_convert_audio_kernel(sr, C, T):
    from wavtokenizer_amd import audio
    rng = np.random.default_rng(sr + T)
    x = rng.standard_normal((3, C, T)).astype(np.float32)
    want = audio_ref.convert_audio(x, sr, 24000, 1)
    got = audio.convert_audio(torch.from_numpy(x).cuda(), sr, 24000, 1).cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())


@pytest.mark.gpu
def test_pcm16_kernel():
    from wavtokenizer_amd import audio
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((2, 50000)) * 0.6).astype(np.float32)
    got = audio.to_pcm16(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(got, audio_ref.to_pcm16(x))
    got_r = audio.to_pcm16(torch.from_numpy(x).cuda(), rescale=True).cpu().numpy().astype(np.int32)
    assert np.abs(got_r - audio_ref.to_pcm16(x, rescale=True).astype(np.int32)).max() <= 1    # fp32 vs fp64 scale factor


@pytest.mark.gpu
def test_segmented_round_trip():
    from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS, audio, synth
    from tests.util import synth_state_dict
    arch = NAMED_ARCHS["hop600"]
    m = WavTokenizer.from_arch(arch)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict("hop600").items()}, strict=False)
    m = m.eval().to("cuda")
    wav = torch.from_numpy(synth.make_clips(2, 48000, seed=9)).cuda()
    bw = torch.tensor([0])
    whole = m.decode(m.encode_infer(wav, bandwidth_id=bw)[0], bandwidth_id=bw)[..., :48000]
    one = audio.segmented_round_trip(m, wav, 48000, 48000)            # a single segment: weights cancel exactly
    assert one.shape == wav.shape and torch.allclose(one, whole, rtol=0, atol=1e-6)
    # overlapping segments against the restated reference loop (encoder/model.py:122-190): 3 segments of 1 s at a 0.75 s
    # stride (the last one ragged), and 4 segments whose length is not a multiple of the hop
    from oracle import audio_ref
    from oracle.cpu_ref import OracleWavTokenizer
    from tests.util import rel_l2, WAV_REL_TOL
    from tests import parity_log
    orc = OracleWavTokenizer(arch, synth_state_dict("hop600"))
    for seg_len, stride in ((24000, 18000), (17777, 11111)):
        got = audio.segmented_round_trip(m, wav, seg_len, stride)
        want = audio_ref.segmented_round_trip(orc, wav.cpu(), seg_len, stride)
        assert got.shape == wav.shape == want.shape
        err = rel_l2(got.cpu().numpy(), want)
        parity_log.record(f"segmented_round_trip[{seg_len},{stride}]", wav_rel_l2=err)
        assert err < WAV_REL_TOL, (seg_len, stride, err)


@pytest.mark.gpu
def test_host_pipeline_matches_the_resident_round_trip():
    """sharding.HostPipeline (infer.py:44-70 for batches: pinned host waveforms -> H2D -> encode_infer + decode + PCM16 -> D2H,
    two lanes): the int16 samples that arrive in host memory equal to_pcm16(decode(encode_infer(wav))) computed the plain way,
    for every step of a run that reuses each lane's buffers several times."""
    import numpy as np
    import torch
    from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS, synth, audio
    from wavtokenizer_amd.sharding import HostPipeline
    from tests.util import synth_state_dict
    arch = NAMED_ARCHS["hop600"]
    m = WavTokenizer.from_arch(arch)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict("hop600").items()}, strict=False)
    m = m.eval().to("cuda")
    bw = torch.tensor([0])
    B, T = 4, 24000
    hp = HostPipeline(m, B, T, bw, lanes=2)
    got, want = [], []
    for i in range(6):
        wav = synth.make_clips(B, T, seed=800 + i)
        k = i % hp.lanes
        hp.s_lane[k].synchronize()                   # the lane's previous step has left its host buffers
        if i >= hp.lanes:
            got.append(hp.h_out[k].numpy().copy())
        hp.h_in[k].copy_(torch.from_numpy(wav))
        hp.step()
        f, _c = m.encode_infer(torch.from_numpy(wav).cuda(), bandwidth_id=bw)
        want.append(audio.to_pcm16(m.decode(f, bandwidth_id=bw)).cpu().numpy())
    hp.drain()
    got.append(hp.h_out[0].numpy().copy())
    got.append(hp.h_out[1].numpy().copy())
    assert len(got) == len(want) == 6
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.dtype == np.int16 and np.array_equal(g, w), i
    h2d, d2h = hp.copy_times_ms(2)
    assert h2d > 0 and d2h > 0
    wdev = torch.from_numpy(synth.make_clips(B, T, seed=1)).cuda()
    h2d, d2h = hp.copy_times_ms(2, lambda: m.decode(m.encode_infer(wdev, bandwidth_id=bw)[0], bandwidth_id=bw))
    assert h2d > 0 and d2h > 0
    m.check_status()
