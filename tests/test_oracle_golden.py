"""CPU: the oracle (oracle/cpu_ref.py) against the fixtures written from the real reference.

In the container that wrote the fixtures the oracle is bit-identical to them (make_golden.py
asserts it against the imported reference).  On another host CPU oneDNN may pick different
kernels, so floats are compared at 2e-5 relative and codes with the near-tie rule.
"""
import numpy as np
import pytest
import torch

from oracle.cpu_ref import OracleWavTokenizer, reflect_index_map, pad1d_reflect, get_extra_padding_for_conv1d
from wavtokenizer_amd.config import NAMED_ARCHS
from tests.util import load_case, synth_state_dict, rel_l2, check_codes, manifest

FLOAT_TOL = 2e-5
BW = torch.tensor([0])


@pytest.fixture(scope="module", params=["hop600", "hop320"])
def oracle(request):
    torch.set_num_threads(8)
    name = request.param
    return name, OracleWavTokenizer(NAMED_ARCHS[name], synth_state_dict(name))


def test_b2_full(oracle):
    name, orc = oracle
    g = load_case(name, "b2_t72000")
    taps = {}
    with torch.inference_mode():
        feats, codes = orc.encode_infer(torch.from_numpy(g["wav_in"]), BW, taps)
        wav = orc.decode(feats, BW, taps)
    assert codes.shape == g["codes"].shape and codes.dtype == torch.int64
    assert check_codes(codes.numpy(), g["codes"], g["margin"], name) == 0
    assert rel_l2(wav.numpy(), g["wav_out"]) < FLOAT_TOL
    assert rel_l2(taps["bb.out"].numpy(), g["bb_out"]) < FLOAT_TOL
    # stage checkpoints: both ends of clip 0 + global L2
    for key in [k for k in g.files if k.startswith("tap/") and k.endswith("/l2")]:
        stage = key.split("/")[1]
        a = taps[stage].numpy()
        assert tuple(g[f"tap/{stage}/shape"]) == a.shape
        assert abs(np.sqrt((a.astype(np.float64) ** 2).sum()) - float(g[key])) <= FLOAT_TOL * float(g[key])
        head = a[0, :, :16] if a.ndim == 3 else a[:16]
        assert rel_l2(head, g[f"tap/{stage}/head"]) < 5 * FLOAT_TOL, stage


def test_non_multiple_of_hop(oracle):
    name, orc = oracle
    g = load_case(name, "b1_t61920")
    with torch.inference_mode():
        feats, codes = orc.encode_infer(torch.from_numpy(g["wav_in"]), BW)
        wav = orc.decode(orc.codes_to_features(codes), BW)
    L = NAMED_ARCHS[name].frames(61920)
    assert codes.shape == (1, 1, L) and wav.shape == (1, L * NAMED_ARCHS[name].hop_length)
    check_codes(codes.numpy(), g["codes"], g["margin"], name)
    assert rel_l2(wav.numpy(), g["wav_out"]) < FLOAT_TOL


def test_inputs_outside_the_synth_family(oracle):
    """Silence, -70 / -110 dBFS, full-scale square, impulse, DC, x30, hard-clipped: the fixture written from the reference
    (tests/golden/make_golden_inputs.py, which also asserted that the float64 run picks the same codes)."""
    name, orc = oracle
    g = load_case(name, "inputs")
    assert g["codes_equal_float64"].all() and float(g["margin"].min()) >= 0.02
    with torch.inference_mode():
        feats, codes = orc.encode_infer(torch.from_numpy(g["wav_in"]), BW)
        wav = orc.decode(feats, BW).numpy()
    assert check_codes(codes.numpy(), g["codes"], g["margin"], name) == 0
    for i, nm in enumerate(g["names"]):
        assert rel_l2(wav[i], g["wav_out"][i]) < FLOAT_TOL, str(nm)


def test_edge_lengths(oracle):
    name, orc = oracle
    g = load_case(name, "edge")
    for T in manifest()["archs"][name]["cases"]["edge"]["T"]:
        with torch.inference_mode():
            feats, codes = orc.encode_infer(torch.from_numpy(g[f"T{T}/wav_in"]), BW)
            wav = orc.decode(feats, BW)
        n_flip = check_codes(codes.numpy(), g[f"T{T}/codes"], g[f"T{T}/margin"], f"{name} T={T}")
        if n_flip == 0:
            assert rel_l2(wav.numpy(), g[f"T{T}/wav_out"]) < FLOAT_TOL, T


# ---- weight-independent known-answer tests -------------------------------------------------
@pytest.mark.parametrize("T", [1, 2, 5, 7, 599, 600, 601])
@pytest.mark.parametrize("pl,pr", [(3, 3), (2, 2), (3, 2), (1, 1), (4, 4), (6, 11)])
def test_reflect_index_map_matches_pad1d(T, pl, pr):
    x = torch.arange(1, T + 1, dtype=torch.float32).view(1, 1, T)
    want = pad1d_reflect(x, (pl, pr)).view(-1).tolist()
    idx = reflect_index_map(T, pl, pr)
    got = [float(i + 1) if i >= 0 else 0.0 for i in idx]
    assert got == want


def test_extra_padding_examples():
    # conv.py:54-61: k=8,s=4,pad_total=4 on 61920 -> no extra; on 61921 -> 3 extra samples
    assert get_extra_padding_for_conv1d(61920, 8, 4, 4) == 0
    assert get_extra_padding_for_conv1d(61921, 8, 4, 4) == 3
    assert get_extra_padding_for_conv1d(5, 7, 1, 6) == 0


def test_argmax_tie_goes_to_lowest_index():
    d = torch.tensor([[1.0, 3.0, 3.0, 2.0]])
    assert int(d.max(dim=-1).indices) == 1


def test_hann_envelope_same_padding():
    # 'same' ISTFT envelope: 1.5 in the interior, 0.75-ish ramp region at the clip edges
    n_fft, hop, T = 2400, 600, 8
    win = torch.hann_window(n_fft)
    wsq = win.square().expand(1, T, -1).transpose(1, 2)
    out = (T - 1) * hop + n_fft
    pad = (n_fft - hop) // 2
    env = torch.nn.functional.fold(wsq, (1, out), (1, n_fft), stride=(1, hop)).squeeze()[pad:-pad]
    assert env.shape[0] == T * hop
    assert abs(float(env[env.shape[0] // 2]) - 1.5) < 1e-5
    assert float(env.min()) > 0.7


def test_seanet_decoder_fixture():
    """Secondary path: oracle.seanet_decoder against the reference's
    feature_extractor.encodec.decoder output captured by make_golden.py."""
    from wavtokenizer_amd import synth
    m = manifest()
    sd = synth.make_state_dict(NAMED_ARCHS["hop600"], seed=m["weight_seed"], with_seanet_decoder=True)
    want_hash = m["archs"]["hop600"]["weights_seanet_decoder"]
    got_hash = synth.weights_manifest(sd)
    assert all(got_hash[k]["sha256"] == v["sha256"] for k, v in want_hash.items())
    g = load_case("hop600", "seanet_decoder")
    orc = OracleWavTokenizer(NAMED_ARCHS["hop600"], sd)
    with torch.inference_mode():
        out = orc.seanet_decoder(torch.from_numpy(g["z"]))
    assert tuple(out.shape) == g["wav_out"].shape == (2, 1, 20 * 600)
    assert rel_l2(out.numpy(), g["wav_out"]) < FLOAT_TOL


def test_istft_center_padding_fixture():
    """ISTFT padding="center" (spectral_ops.py:43-45, torch.istft(center=True)): the oracle against outputs captured from the
    reference with its ISTFT switched to that mode (tests/golden/make_golden_center.py); (L - 1) * hop samples per clip."""
    import dataclasses
    import os
    from tests.util import GOLDEN
    arch = dataclasses.replace(NAMED_ARCHS["hop600"], padding="center")
    orc = OracleWavTokenizer(arch, synth_state_dict("hop600"))
    g = np.load(os.path.join(GOLDEN, "hop600_center.npz"))
    for tag in ("b2_t24000", "b1_t1300"):
        feats = torch.from_numpy(g[f"{tag}/features"])
        want = g[f"{tag}/wav_out"]
        with torch.inference_mode():
            got = orc.decode(feats, torch.tensor([0])).numpy()
        assert got.shape == want.shape == (feats.shape[0], (feats.shape[2] - 1) * arch.hop_length)
        assert np.array_equal(got, want), tag


def test_codes_to_features_with_several_codebooks():
    """pretrained.py:230-237 with K codebooks: offsets k * bins into the concatenated table, summed over k."""
    import dataclasses
    from wavtokenizer_amd import synth
    arch = dataclasses.replace(NAMED_ARCHS["hop600"], num_quantizers=2)
    sd = synth.make_state_dict(arch, seed=0)
    orc = OracleWavTokenizer(arch, sd)
    codes = torch.randint(0, arch.vq_bins, (2, 3, 11), generator=torch.Generator().manual_seed(1))
    e0 = torch.from_numpy(sd["feature_extractor.encodec.quantizer.vq.layers.0._codebook.embed"])
    e1 = torch.from_numpy(sd["feature_extractor.encodec.quantizer.vq.layers.1._codebook.embed"])
    assert torch.equal(orc.codes_to_features(codes), (e0[codes[0]] + e1[codes[1]]).transpose(1, 2))
    assert torch.equal(orc.codes_to_features(codes[:1]), e0[codes[0]].transpose(1, 2))


def test_codes_to_features_fixture_three_codebooks():
    """The reference's own codes_to_features on a num_quantizers = 3 model (tests/golden/make_golden_codebooks.py)."""
    import dataclasses
    import os
    from wavtokenizer_amd import synth
    from tests.util import GOLDEN
    arch = dataclasses.replace(NAMED_ARCHS["hop600"], num_quantizers=3)
    g = np.load(os.path.join(GOLDEN, "hop600_codebooks3.npz"))
    orc = OracleWavTokenizer(arch, synth.make_state_dict(arch, seed=int(g["weight_seed"])))
    for tag in ("k1", "k2", "k3", "k2_2d"):
        assert np.array_equal(orc.codes_to_features(torch.from_numpy(g[f"{tag}/codes"])).numpy(), g[f"{tag}/features"]), tag
