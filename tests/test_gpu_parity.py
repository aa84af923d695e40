"""GPU parity: the HIP path (through the C-ABI, via the drop-in WavTokenizer class) against
 (a) the golden fixtures written from the real reference and (b) the oracle run on this host.

Bars (north_star): codes bit-exact; waveform within 1e-4 relative.  Stage checkpoints are held
to 3e-5 relative L2 (fp32 re-association only).  A code mismatch is tolerated only at frames
whose reference top-2 margin is a near tie (tests/util.py NEAR_TIE_MARGIN) and is reported.
"""
import numpy as np
import pytest
import torch

from tests.util import (WAV_REL_TOL, check_codes, load_case, manifest, rel_l2, synth_state_dict)

pytestmark = pytest.mark.gpu

STAGE_TOL = 3e-5
BW = torch.tensor([0])


def _model(arch_name, with_seanet_decoder=False):
    from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS, synth
    arch = NAMED_ARCHS[arch_name]
    if with_seanet_decoder:
        sd = synth.make_state_dict(arch, seed=manifest()["weight_seed"], with_seanet_decoder=True)
    else:
        sd = synth_state_dict(arch_name)
    m = WavTokenizer.from_arch(arch)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return m.eval().to("cuda"), sd


@pytest.fixture(scope="module", params=["hop600", "hop320"])
def gpu_model(request):
    m, sd = _model(request.param)
    return request.param, m, sd


def _oracle(arch_name, sd):
    from oracle.cpu_ref import OracleWavTokenizer
    from wavtokenizer_amd import NAMED_ARCHS
    return OracleWavTokenizer(NAMED_ARCHS[arch_name], sd)


def test_native_library_loaded():
    from wavtokenizer_amd import _capi
    assert b"gfx950" in _capi.lib.wt_version()
    with open("/proc/self/maps") as f:
        assert "libwavtok_hip.so" in f.read()


def _plan_step_names(plan):
    import ctypes
    from wavtokenizer_amd import _capi
    names = []
    for i in range(_capi.lib.wt_plan_num_steps(plan)):
        p = ctypes.c_char_p()
        assert _capi.lib.wt_plan_step_name(plan, i, ctypes.byref(p)) == 0
        names.append(p.value.decode())
    return names


# taps of the reference that never exist in HBM on the shipped path: the first conv and the stage-1 resblock run inside
# the kernel that also computes the stage's down conv (checked through enc.3, and alone against the oracle in
# test_resblock16_down_kernel_against_oracle), pos_net.5 + AdaLayerNorm are one row pass (checked through bb.norm), the
# head's Linear output becomes the spectrum in the GEMM epilogue (checked through the waveform)
FUSED_AWAY = {"enc.0", "enc.1", "bb.pos_net.5", "head.out"}


@pytest.mark.parametrize("variant", ["shipped", "unfused"])
def test_b2_stage_checkpoints(gpu_model, variant):
    """Every stage of encode and decode against the reference's activations.

    shipped: the DEFAULT plan's kernels (gemm16s / resblock16 / lstm_persist ...) with WT_PLAN_FLAG_KEEP_STAGES only
    un-aliasing their buffers; a tap may be S32-encoded and / or hold elu(x) (wt_plan_buffer_info), so the full
    tensors are compared with the oracle's taps (which tests/test_oracle_golden.py pins bit-for-bit to the reference's
    captured ones) and the raw ones also with the fixture's heads.
    unfused: the debug twin (raw fp32 tensors between unfused stages) against the fixture, as in round 1."""
    import torch.nn.functional as F
    from wavtokenizer_amd import _capi
    from tests import parity_log
    name, m, sd = gpu_model
    g = load_case(name, "b2_t72000")
    wav_np = g["wav_in"]
    wav = torch.from_numpy(wav_np).cuda()
    B, T = wav.shape
    taps = {}
    if variant == "shipped":
        orc = _oracle(name, sd)
        with torch.inference_mode():
            fo, co = orc.encode_infer(torch.from_numpy(wav_np), BW, taps)
            orc.decode(fo, BW, taps)
        assert torch.equal(co, torch.from_numpy(g["codes"]))       # the oracle is the fixture's source
    m.set_debug_keep_stages(True, unfused=(variant == "unfused"))
    try:
        feats, codes = m.encode_infer(wav, bandwidth_id=BW)
        out = m.decode(feats, bandwidth_id=BW)
        torch.cuda.synchronize()
        L = codes.shape[-1]
        report, skipped = [], set()
        stages = [k.split("/")[1] for k in g.files if k.startswith("tap/") and k.endswith("/l2")]
        for st in stages:
            shape = tuple(int(v) for v in g[f"tap/{st}/shape"])
            kind, length = (_capi.WT_PLAN_ENCODE, T) if st.startswith("enc.") else (_capi.WT_PLAN_DECODE, L)
            try:
                buf, fmt = m.debug_stage(kind, B, length, st)
            except Exception:
                skipped.add(st)
                continue
            if st == "bb.out":
                mine = buf.view(shape).cpu()                              # (B, L, C) in both
                my_head = mine[0][:, :16].numpy()
            else:
                Bc, C, Tt = shape                                         # reference (B, C, T); ours (B, T, C)
                mine = buf.view(Bc, Tt, C).permute(0, 2, 1).cpu()
                my_head = mine[0, :, :16].numpy()
            ref_head = torch.from_numpy(g[f"tap/{st}/head"])
            if fmt & _capi.BUF_ELU:
                ref_head = F.elu(ref_head)
            e_head = rel_l2(my_head, ref_head.numpy())
            if variant == "shipped":
                ref = taps[st]
                ref = F.elu(ref) if fmt & _capi.BUF_ELU else ref
                e_full = rel_l2(mine.numpy(), ref.numpy())
            else:
                assert fmt == 0, (st, fmt)                                # the debug twin keeps raw fp32 tensors
                l2 = float(np.sqrt((mine.numpy().astype(np.float64) ** 2).sum()))
                e_full = abs(l2 - float(g[f"tap/{st}/l2"])) / float(g[f"tap/{st}/l2"])
            report.append((st, fmt, e_head, e_full))
        parity_log.record(f"stage_taps[{name},{variant}]", worst_tap=max(report, key=lambda r: r[3])[0],
                          worst_rel_l2=max(r[3] for r in report), taps=len(report))
        bad = [r for r in report if r[2] > STAGE_TOL or r[3] > STAGE_TOL]
        assert not bad, "first diverging stage: %s (fmt %d, head rel-L2 %.3g, full %.3g); all: %s" % (*bad[0], report)
        if variant == "shipped":
            assert skipped == FUSED_AWAY, skipped
            # the plan that kept its stages launches exactly the default plan's steps (plus the residual snapshots)
            flags = m._plan_flags
            for kind, length in ((_capi.WT_PLAN_ENCODE, T), (_capi.WT_PLAN_DECODE, L)):
                keep_names = _plan_step_names(m._engine.plans[(kind, B, length, flags)][0])
                m.set_debug_keep_stages(False)
                m.set_graph_max_clips(0)
                try:
                    if kind == _capi.WT_PLAN_ENCODE:
                        m.encode_infer(wav, bandwidth_id=BW)
                    else:
                        m.decode(feats, bandwidth_id=BW)
                    default_names = _plan_step_names(m._engine.plans[(kind, B, length, m._plan_flags)][0])
                finally:
                    m.set_graph_max_clips(16)
                    m.set_debug_keep_stages(True)
                snap = [n for n in keep_names if n.startswith(("bb.embed", "bb.pos_net.", "bb.convnext."))]
                core = [n for n in keep_names if n not in snap]
                assert len(snap) == 9 or kind == _capi.WT_PLAN_ENCODE, snap
                # keep: row pass -> "bb.x2" then a copy named "bb.norm"; default: the row pass itself is "bb.norm"
                core = [n for i, n in enumerate(core) if not (n == "bb.norm" and i and core[i - 1] == "bb.x2")]
                core = ["bb.norm" if n == "bb.x2" else n for n in core]
                assert core == default_names, (core, default_names)
        else:
            assert skipped == {"bb.pos_net.5", "head.out"}, skipped       # fused into their consumer in every plan
    finally:
        m.set_debug_keep_stages(False)
    assert check_codes(codes.cpu().numpy(), g["codes"], g["margin"], name) == 0
    assert codes.dtype == torch.int64 and tuple(codes.shape) == g["codes"].shape
    assert rel_l2(out.cpu().numpy(), g["wav_out"]) < WAV_REL_TOL


def test_b2_outputs(gpu_model):
    name, m, _sd = gpu_model
    g = load_case(name, "b2_t72000")
    wav = torch.from_numpy(g["wav_in"]).cuda()
    feats, codes = m.encode_infer(wav, bandwidth_id=BW)
    emb = m.feature_extractor.encodec.encoder(wav.unsqueeze(1))
    bb = m.backbone(feats, bandwidth_id=BW)
    out = m.decode(feats, bandwidth_id=BW)
    assert check_codes(codes.cpu().numpy(), g["codes"], g["margin"], name) == 0
    assert rel_l2(emb.cpu().numpy(), g["emb"]) < STAGE_TOL
    assert rel_l2(bb.cpu().numpy(), g["bb_out"]) < STAGE_TOL
    err = rel_l2(out.cpu().numpy(), g["wav_out"])
    from tests import parity_log
    parity_log.record(f"b2_outputs[{name}]", code_flips=0, frames=int(codes.numel()), wav_rel_l2=err,
                      emb_rel_l2=rel_l2(emb.cpu().numpy(), g["emb"]), bb_rel_l2=rel_l2(bb.cpu().numpy(), g["bb_out"]))
    assert err < WAV_REL_TOL, err
    # features are exactly the codebook rows of the codes (core_vq.py:188-190)
    sd_embed = torch.from_numpy(synth_state_dict(name)["feature_extractor.encodec.quantizer.vq.layers.0._codebook.embed"])
    want = sd_embed[torch.from_numpy(g["codes"])[0]].permute(0, 2, 1)
    assert torch.equal(feats.cpu(), want)
    # codes_to_features == dequantise; decode(codes_to_features(codes)) == decode(features) bit for bit
    f2 = m.codes_to_features(codes)
    assert torch.equal(f2, feats)
    assert torch.equal(m.codes_to_features(codes[:, 0]), feats[:1]) or codes.shape[1] != 1
    assert torch.equal(m.decode(f2, bandwidth_id=BW), out)
    # forward() = copy-synthesis
    assert torch.equal(m(wav, bandwidth_id=BW), out)
    # another bandwidth row changes the output (AdaLayerNorm conditioning is live)
    assert not torch.equal(m.decode(feats, bandwidth_id=torch.tensor([2])), out)


@pytest.mark.parametrize("chain", ["shipped", "fp32"])
def test_inputs_outside_the_synth_family(gpu_model, chain):
    """every other GPU parity input is one signal family (AM/FM tone + noise, |x| <= 0.5).  This batch holds
    digital silence, the base clip at -70 and -110 dBFS (the f16 hi halves of the S32 form go subnormal), a full-scale
    square wave, a unit impulse, DC 0.9, the base clip x 30 and a hard-clipped clip; expected codes / waveforms were
    captured from the reference (tests/golden/make_golden_inputs.py; the float64 run picks the same codes, smallest margin
    0.03).  Codes exact, every clip's waveform within 1e-4 of the reference, no status bit - on the shipped path and on
    the plain fp32 chain."""
    from tests import parity_log
    name, m, _sd = gpu_model
    g = load_case(name, "inputs")
    wav = torch.from_numpy(g["wav_in"]).cuda()
    m.set_gemm_precision("f32" if chain == "fp32" else "f16x3")
    try:
        feats, codes = m.encode_infer(wav, bandwidth_id=BW)
        out = m.decode(feats, bandwidth_id=BW).cpu().numpy()
        m.check_status()                                        # no range / LSTM report on any of them
    finally:
        m.set_gemm_precision("f16x3")
    flips = check_codes(codes.cpu().numpy(), g["codes"], g["margin"], f"{name} inputs[{chain}]")
    assert flips == 0                                           # every margin >= 0.02 here: exact
    errs = {}
    for i, nm in enumerate(g["names"]):
        assert np.isfinite(out[i]).all(), str(nm)
        errs[str(nm)] = rel_l2(out[i], g["wav_out"][i])
        assert errs[str(nm)] < WAV_REL_TOL, (str(nm), errs[str(nm)])
    parity_log.record(f"inputs[{name},{chain}]", code_flips=flips, frames=int(g["codes"].size), wav_rel_l2=max(errs.values()),
                      **{f"wav_rel_l2.{k}": v for k, v in errs.items()})
    if chain == "shipped":
        rep = m.range_report(wav, bandwidth_id=BW)              # where the x30 clip sits against the f16 limit
        worst = min(rep, key=lambda r: r["headroom_bits"])
        parity_log.record(f"inputs_headroom[{name}]", least_headroom_bits=worst["headroom_bits"], at=worst["step"] + ":" + worst["buffer"])
        assert worst["headroom_bits"] > 1.0, worst


def test_length_not_multiple_of_hop(gpu_model):
    name, m, _sd = gpu_model
    g = load_case(name, "b1_t61920")
    feats, codes = m.encode_infer(torch.from_numpy(g["wav_in"]).cuda(), bandwidth_id=BW)
    out = m.decode(feats, bandwidth_id=BW)
    assert tuple(codes.shape) == g["codes"].shape and tuple(out.shape) == g["wav_out"].shape
    assert check_codes(codes.cpu().numpy(), g["codes"], g["margin"], name) == 0
    assert rel_l2(out.cpu().numpy(), g["wav_out"]) < WAV_REL_TOL


def test_edge_lengths(gpu_model):
    """T = 1, 5, 599, 600, 601, 1927: short-input reflect padding and ragged last frames."""
    name, m, _sd = gpu_model
    g = load_case(name, "edge")
    for T in manifest()["archs"][name]["cases"]["edge"]["T"]:
        feats, codes = m.encode_infer(torch.from_numpy(g[f"T{T}/wav_in"]).cuda(), bandwidth_id=BW)
        out = m.decode(feats, bandwidth_id=BW)
        assert tuple(codes.shape) == g[f"T{T}/codes"].shape, T
        flips = check_codes(codes.cpu().numpy(), g[f"T{T}/codes"], g[f"T{T}/margin"], f"{name} T={T}")
        if flips == 0:
            assert rel_l2(out.cpu().numpy(), g[f"T{T}/wav_out"]) < WAV_REL_TOL, T


def test_30s_clip():
    """BASELINE config 5 shape: one 30 s clip (L = 1200), hop-600."""
    m, _sd = _model("hop600")
    g = load_case("hop600", "b1_t720000")
    feats, codes = m.encode_infer(torch.from_numpy(g["wav_in"]).cuda(), bandwidth_id=BW)
    out = m.decode(feats, bandwidth_id=BW).cpu().numpy()
    assert check_codes(codes.cpu().numpy(), g["codes"], g["margin"], "30s") == 0
    assert rel_l2(out[:, :4096], g["wav_out_head"]) < WAV_REL_TOL
    assert rel_l2(out[:, -4096:], g["wav_out_tail"]) < WAV_REL_TOL
    assert abs(np.sqrt((out.astype(np.float64) ** 2).sum()) - float(g["wav_out_l2"])) < WAV_REL_TOL * float(g["wav_out_l2"])


def test_full_batch_against_oracle_and_invariants(gpu_model):
    """BASELINE config 2/3 size (B = 64 x 3 s): GPU vs the oracle on this host, plus
    size-independent properties: batch invariance (a clip's result does not depend on its
    neighbours) and the encode -> codes -> decode round trip."""
    from wavtokenizer_amd import synth
    from tests import parity_log
    from tests.util import NEAR_TIE_MARGIN
    name, m, sd = gpu_model
    B = 64
    wav_np = synth.make_clips(B, 72000, seed=7000)
    wav = torch.from_numpy(wav_np).cuda()
    feats, codes = m.encode_infer(wav, bandwidth_id=BW)
    out = m.decode(feats, bandwidth_id=BW)
    # --- batch invariance, bit for bit
    f3, c3 = m.encode_infer(wav[5:8].contiguous(), bandwidth_id=BW)
    assert torch.equal(c3, codes[:, 5:8])
    assert torch.equal(m.decode(f3, bandwidth_id=BW), out[5:8])
    # --- round trip through codes
    assert torch.equal(m.decode(m.codes_to_features(codes), bandwidth_id=BW), out)
    # --- against the oracle on the host cores
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    orc = _oracle(name, sd)
    taps = {}
    with torch.inference_mode():
        fo, co = orc.encode_infer(torch.from_numpy(wav_np), BW, taps)
    flips = check_codes(codes.cpu().numpy(), co.numpy(), taps["vq.margin"].numpy(), f"{name} B=64")
    # decode the ORACLE's features on both sides so near-tie flips cannot leak into the waveform check
    with torch.inference_mode():
        wo = orc.decode(fo, BW)
    wg = m.decode(fo.cuda(), bandwidth_id=BW)
    err = rel_l2(wg.cpu().numpy(), wo.numpy())
    per_clip = max(rel_l2(wg[i].cpu().numpy(), wo[i].numpy()) for i in range(B))
    margin = taps["vq.margin"].numpy().reshape(-1)
    near_ties = int((margin < NEAR_TIE_MARGIN).sum())
    parity_log.record(f"full_batch[{name}]", code_flips=flips, frames=int(codes.numel()), near_tie_frames=near_ties,
                      min_margin=float(margin.min()), wav_rel_l2=err, worst_clip_rel_l2=per_clip)
    print(f"[{name}] B=64: near-tie code flips {flips}/{codes.numel()} ({near_ties} frames have a near-tie margin), "
          f"waveform rel-L2 {err:.3g} (worst clip {per_clip:.3g})")
    assert err < WAV_REL_TOL and per_clip < WAV_REL_TOL
    # check_codes has already failed on any flip at a healthy margin: what is left can only sit on near-tie frames
    assert flips <= near_ties


def test_random_ragged_lengths_against_oracle(gpu_model):
    """Seeded random clip lengths (not multiples of the hop, odd batch sizes): GPU vs the oracle on this host."""
    from wavtokenizer_amd import synth
    from tests import parity_log
    from tests.util import NEAR_TIE_MARGIN
    name, m, sd = gpu_model
    orc = _oracle(name, sd)
    rng = np.random.default_rng(20240 + len(name))
    cases = [(int(rng.integers(1, 6)), int(rng.integers(700, 30000))) for _ in range(5)] + [(3, 2), (2, 7), (1, 1201)]
    for B, T in cases:
        wav_np = synth.make_clips(B, T, seed=9000 + T)
        feats, codes = m.encode_infer(torch.from_numpy(wav_np).cuda(), bandwidth_id=BW)
        taps = {}
        with torch.inference_mode():
            fo, co = orc.encode_infer(torch.from_numpy(wav_np), BW, taps)
            wo = orc.decode(fo, BW)
        assert tuple(codes.shape) == tuple(co.shape), (B, T)
        flips = check_codes(codes.cpu().numpy(), co.numpy(), taps["vq.margin"].numpy(), f"{name} B={B} T={T}")
        margin = taps["vq.margin"].numpy().reshape(-1)
        assert flips <= int((margin < NEAR_TIE_MARGIN).sum()), (B, T, flips)
        wg = m.decode(fo.cuda(), bandwidth_id=BW)
        assert tuple(wg.shape) == tuple(wo.shape), (B, T)
        err = rel_l2(wg.cpu().numpy(), wo.numpy())
        parity_log.record(f"ragged[{name},B={B},T={T}]", code_flips=flips, frames=int(codes.numel()), wav_rel_l2=err)
        assert err < WAV_REL_TOL, (B, T)


def test_b8_fixture_codes(gpu_model):
    from wavtokenizer_amd import synth
    from tests.util import NEAR_TIE_MARGIN
    import hashlib
    name, m, _sd = gpu_model
    case = manifest()["archs"][name]["cases"]["b8_t72000"]
    wav_np = synth.make_clips(8, 72000, case["clip_seed"])
    if hashlib.sha256(wav_np.tobytes()).hexdigest() != case["wav_in_sha256"]:
        pytest.skip("host libm renders the synthetic clips differently from the fixture container")
    g = load_case(name, "b8_t72000")
    feats, codes = m.encode_infer(torch.from_numpy(wav_np).cuda(), bandwidth_id=BW)
    flips = check_codes(codes.cpu().numpy(), g["codes"], g["margin"], name)
    assert flips <= int((g["margin"] < NEAR_TIE_MARGIN).sum())
    if flips == 0:
        out = m.decode(feats, bandwidth_id=BW).cpu().numpy().astype(np.float64)
        assert np.allclose(np.sqrt((out ** 2).sum(axis=1)), g["wav_out_l2"], rtol=WAV_REL_TOL)
        assert rel_l2(out[:, :256], g["wav_out_head"]) < 5 * WAV_REL_TOL


def test_istft_center_padding():
    """ISTFTHead with padding="center" (spectral_ops.py:43-45: torch.istft(center=True); no YAML selects it): decode of the
    reference's features against the reference's captured waveform, (L - 1) * hop samples per clip; one frame is refused."""
    import dataclasses
    import os
    from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS
    from tests.util import GOLDEN
    from tests import parity_log
    arch = dataclasses.replace(NAMED_ARCHS["hop600"], padding="center")
    m = WavTokenizer.from_arch(arch)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict("hop600").items()}, strict=False)
    m = m.eval().to("cuda")
    g = np.load(os.path.join(GOLDEN, "hop600_center.npz"))
    for tag in ("b2_t24000", "b1_t1300"):
        feats = torch.from_numpy(g[f"{tag}/features"]).cuda()
        want = g[f"{tag}/wav_out"]
        got = m.decode(feats, bandwidth_id=BW).cpu().numpy()
        assert got.shape == want.shape, (got.shape, want.shape)
        err = rel_l2(got, want)
        parity_log.record(f"istft_center[{tag}]", wav_rel_l2=err)
        assert err < WAV_REL_TOL, (tag, err)
    # round trip through the class: the same length rule
    wav = torch.from_numpy(g["b2_t24000/wav_in"]).cuda()
    f, c = m.encode_infer(wav, bandwidth_id=BW)
    assert m.decode(f, bandwidth_id=BW).shape == (2, (f.shape[2] - 1) * arch.hop_length)
    with pytest.raises(Exception):
        m.decode(f[:, :, :1].contiguous(), bandwidth_id=BW)


def test_errors_like_the_reference(gpu_model):
    name, m, _sd = gpu_model
    feats = torch.zeros(1, 512, 4, device="cuda")
    with pytest.raises(AssertionError):
        m.decode(feats)                     # missing bandwidth_id: decoder/models.py:227
    from wavtokenizer_amd._capi import WavTokError
    with pytest.raises(WavTokError):
        m.decode(feats, bandwidth_id=torch.tensor([7]))


# ------------------------------------------------------------------- single-stage entry points
def _ptr(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


@pytest.mark.parametrize("B,T,Cin,Cout,k,stride,dil,elu", [
    (2, 1000, 32, 16, 3, 1, 1, 1), (2, 1000, 16, 32, 1, 1, 1, 1), (3, 777, 32, 64, 8, 4, 1, 1),
    (2, 333, 64, 128, 10, 5, 1, 1), (1, 50, 256, 512, 12, 6, 1, 1), (2, 120, 512, 512, 7, 1, 1, 0),
    (2, 200, 32, 32, 3, 1, 2, 1), (2, 200, 32, 32, 3, 1, 4, 0), (1, 2, 32, 32, 7, 1, 1, 1), (1, 1, 64, 64, 8, 4, 1, 0),
    (1, 3, 32, 64, 16, 8, 1, 1),
])
def test_sconv1d_kernel(B, T, Cin, Cout, k, stride, dil, elu):
    """wt_sconv1d (the implicit-GEMM conv) against SConv1d semantics (conv.py:195-211) on CPU,
    including dilation and inputs shorter than the reflect pad."""
    import torch.nn.functional as F
    from oracle.cpu_ref import get_extra_padding_for_conv1d, pad1d_reflect
    from wavtokenizer_amd._capi import lib, check
    gen = torch.Generator().manual_seed(B * 1000 + T + k)
    x = torch.randn(B, Cin, T, generator=gen)
    w = torch.randn(Cout, Cin, k, generator=gen) / (Cin * k) ** 0.5
    b = torch.randn(Cout, generator=gen)
    keff = (k - 1) * dil + 1
    pt = keff - stride
    extra = get_extra_padding_for_conv1d(T, keff, stride, pt)
    pr = pt // 2
    xin = F.elu(x) if elu else x
    want = F.conv1d(pad1d_reflect(xin, (pt - pr, pr + extra)), w, b, stride=stride, dilation=dil)
    Tout = want.shape[-1]
    xg = x.permute(0, 2, 1).contiguous().cuda()
    wg = w.permute(0, 2, 1).contiguous().cuda()      # [Cout][k][Cin]
    bg = b.cuda()
    y = torch.empty(B, Tout, Cout, device="cuda")
    check(lib.wt_sconv1d(_ptr(xg), _ptr(wg), _ptr(bg), _ptr(y), B, T, Cin, Cout, k, stride, dil, elu, None), "wt_sconv1d")
    torch.cuda.synchronize()
    assert rel_l2(y.permute(0, 2, 1).cpu().numpy(), want.numpy()) < 1e-5


def _decode_s32(y, M, N):
    """S32 (gemm16s.hip): every 32 values of a row are 128 bytes [32 x f16 hi | 32 x f16 lo], value = hi + lo / 2048."""
    h = y.view(torch.float16).view(M, N // 32, 2, 32).double()
    return (h[:, :, 0, :] + h[:, :, 1, :] / 2048.0).reshape(M, N)


@pytest.mark.parametrize("M,N,K", [(7680, 2304, 768), (1000, 800, 96), (1, 32, 32), (129, 196, 64), (300, 96, 2304)])
def test_linear_s32_kernel(M, N, K):
    """wt_linear on the S32 split-f16 / LDS-DMA kernel (modes 2, 3) against float64: fp32-equivalent products,
    ragged row and column tiles, S32-encoded output."""
    from wavtokenizer_amd._capi import lib, check
    gen = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=gen) * torch.exp(0.5 * torch.randn(M, K, generator=gen))).cuda()
    w = (torch.randn(N, K, generator=gen) / K ** 0.5).cuda()
    b = torch.randn(N, generator=gen).cuda()
    ws = torch.empty(4 * (M + N) * K + 1024, dtype=torch.uint8, device="cuda")
    ref = x.double() @ w.double().t() + b.double()
    for mode in (2, 3):
        if mode == 3 and N % 32:
            continue
        y = torch.full((M, N), float("nan"), device="cuda")
        check(lib.wt_linear(_ptr(x), _ptr(w), _ptr(b), _ptr(y), M, N, K, mode, _ptr(ws), None), "wt_linear")
        torch.cuda.synchronize()
        got = _decode_s32(y, M, N) if mode == 3 else y.double()
        assert torch.isfinite(got).all()
        assert ((got - ref).norm() / ref.norm()).item() < 1e-6
        assert (got - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("B,T,Cin,Cout,k,stride,zero_same", [
    (4, 120, 768, 768, 3, 1, 1), (3, 225, 512, 768, 7, 1, 1), (2, 1, 768, 768, 3, 1, 1), (2, 2, 64, 96, 7, 1, 1),
    (3, 777, 32, 64, 8, 4, 0), (2, 333, 64, 128, 10, 5, 0), (1, 50, 256, 512, 12, 6, 0), (2, 5, 512, 512, 7, 1, 0),
    (1, 3, 32, 64, 16, 8, 0),
])
def test_conv1d_s32_kernel(B, T, Cin, Cout, k, stride, zero_same):
    """wt_conv1d_s32 (S32 operands, LDS-DMA gather: an out-of-range row offset must read as zeros) against
    nn.Conv1d 'same' zero padding (decoder/models.py:29-43) and SConv1d reflect padding (conv.py:195-211)."""
    import torch.nn.functional as F
    from oracle.cpu_ref import get_extra_padding_for_conv1d, pad1d_reflect
    from wavtokenizer_amd._capi import lib, check
    gen = torch.Generator().manual_seed(B * 1000 + T + k + Cin)
    x = torch.randn(B, Cin, T, generator=gen)
    w = torch.randn(Cout, Cin, k, generator=gen) / (Cin * k) ** 0.5
    b = torch.randn(Cout, generator=gen)
    if zero_same:
        want = F.conv1d(x.double(), w.double(), b.double(), padding=(k - 1) // 2)
    else:
        pt = k - stride
        extra = get_extra_padding_for_conv1d(T, k, stride, pt)
        pr = pt // 2
        want = F.conv1d(pad1d_reflect(x, (pt - pr, pr + extra)).double(), w.double(), b.double(), stride=stride)
    Tout = want.shape[-1]
    xg = x.permute(0, 2, 1).contiguous().cuda()
    wg = w.permute(0, 2, 1).contiguous().cuda()      # [Cout][k][Cin]
    bg = b.cuda()
    y = torch.full((B, Tout, Cout), float("nan"), device="cuda")
    ws = torch.empty(4 * (B * T * Cin + Cout * k * Cin) + 1024, dtype=torch.uint8, device="cuda")
    check(lib.wt_conv1d_s32(_ptr(xg), _ptr(wg), _ptr(bg), _ptr(y), B, T, Cin, Cout, k, stride, zero_same, _ptr(ws), None),
          "wt_conv1d_s32")
    torch.cuda.synchronize()
    assert rel_l2(y.permute(0, 2, 1).cpu().numpy(), want.numpy()) < 1e-6


def test_s32_decode_path_matches_the_fp32_chain(gpu_model):
    """decode() on the default plan (S32 operands end to end) against the same model forced onto the fp32 MFMA chain."""
    name, m, sd = gpu_model
    from wavtokenizer_amd import synth
    wav = torch.from_numpy(synth.make_clips(3, 24000, seed=77)).cuda()
    feats, codes = m.encode_infer(wav, bandwidth_id=BW)
    out = m.decode(feats, bandwidth_id=BW)
    m.set_gemm_precision("f32")
    try:
        feats32, codes32 = m.encode_infer(wav, bandwidth_id=BW)
        out32 = m.decode(feats32, bandwidth_id=BW)
    finally:
        m.set_gemm_precision("f16x3")
    assert torch.equal(codes, codes32)
    assert rel_l2(out.cpu().numpy(), out32.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("kernel", ["shipped", "fp32"])
def test_vq_nearest_kernel_and_ties(kernel):
    """EuclideanCodebook.quantize (core_vq.py:175-183) on the encoder plan's own kernels (wt_vq_nearest: distances on
    gemm16s.hip with the per-slab argmax epilogue EPI_ARGMAX, then vq_finalize) and on the fp32 twin.  Exactly
    duplicated codebook rows tie: torch.max returns the lowest index, and so must every merge level of the kernel:
    inside a lane's 4-column run, across the lane pair (columns n and n + 4), across 8-column groups and 32-column
    MFMA tiles of a wave, across the 96-column wave slabs, across 192-column block tiles, and across the eight lanes
    (parts q, q + 8, ...) and part order of vq_finalize."""
    from wavtokenizer_amd._capi import lib, check
    gen = torch.Generator().manual_seed(11)
    bins, D, N = 4096, 512, 1000
    embed = torch.randn(bins, D, generator=gen)
    pairs = [(1, 2), (8, 12), (3, 11), (5, 37), (40, 70), (95, 96), (10, 100), (191, 192), (50, 250), (200, 968),
             (300, 1068), (17, 3000), (2048, 4095), (385, 1153), (4000, 4090), (2100, 2101)]
    used = set()
    for a, b in pairs:
        assert a < b and not ({a, b} & used)
        used |= {a, b}
        embed[b] = embed[a]                                # exact duplicate: the tie must resolve to a
    x = torch.randn(N, D, generator=gen) * 0.7
    for r, (a, b) in enumerate(pairs):
        x[r] = embed[a]                                    # distance 0 to both copies, far from everything else
        x[len(pairs) + r] = embed[b] * 1.0                 # the same through the other row
    e = embed.t()
    dist = -(x.pow(2).sum(1, keepdim=True) - 2 * x @ e + e.pow(2).sum(0, keepdim=True))
    want = dist.max(dim=-1).indices
    top2 = dist.topk(2, dim=-1).values
    margin = (top2[:, 0] - top2[:, 1]).numpy()
    ws = torch.empty(lib.wt_vq_workspace_bytes(N, D, bins), dtype=torch.uint8, device="cuda")
    codes = torch.full((N,), -7, dtype=torch.int64, device="cuda")
    fn = lib.wt_vq_nearest if kernel == "shipped" else lib.wt_vq_nearest_f32
    check(fn(_ptr(x.cuda()), _ptr(embed.cuda()), N, D, bins, _ptr(codes), _ptr(ws), None), "wt_vq_nearest")
    torch.cuda.synchronize()
    got = codes.cpu()
    for r, (a, b) in enumerate(pairs):
        assert int(got[r]) == a and int(got[len(pairs) + r]) == a, (kernel, a, b, int(got[r]), int(got[len(pairs) + r]))
    # every other row has a healthy margin and must match exactly
    bad = (got != want).nonzero().flatten().tolist()
    assert all(margin[i] < 1e-3 for i in bad), bad
    assert len(bad) <= 2 * len(pairs)


def test_seanet_decoder():
    """Secondary path: feature_extractor.encodec.decoder(features) (seanet.py:147-238) against the
    reference's captured output, and against the oracle on a second input."""
    m, sd = _model("hop600", with_seanet_decoder=True)
    g = load_case("hop600", "seanet_decoder")
    got = m.feature_extractor.encodec.decoder(torch.from_numpy(g["z"]).cuda())
    assert tuple(got.shape) == g["wav_out"].shape == (2, 1, 20 * 600)
    assert rel_l2(got.cpu().numpy(), g["wav_out"]) < WAV_REL_TOL
    orc = _oracle("hop600", sd)
    gen = torch.Generator().manual_seed(5)
    z = torch.randn(3, 512, 33, generator=gen) * 0.6
    with torch.inference_mode():
        want = orc.seanet_decoder(z)
    got = m.feature_extractor.encodec.decoder(z.cuda())
    assert rel_l2(got.cpu().numpy(), want.numpy()) < WAV_REL_TOL


@pytest.mark.parametrize("arch", ["hop600", "hop320"])
def test_seanet_decoder_small_and_odd_shapes(arch):
    """SEANetDecoder against the oracle on one- and two-frame inputs (every transposed conv is all edge), odd batch
    sizes, both architectures (stride sets 6,5,5,4 and 8,5,4,2)."""
    m, sd = _model(arch, with_seanet_decoder=True)
    orc = _oracle(arch, sd)
    for i, (B, L) in enumerate(((1, 1), (2, 2), (5, 7), (3, 41))):
        z = torch.randn(B, 512, L, generator=torch.Generator().manual_seed(70 + i)) * 0.6
        with torch.inference_mode():
            want = orc.seanet_decoder(z)
        got = m.feature_extractor.encodec.decoder(z.cuda())
        assert tuple(got.shape) == tuple(want.shape), (B, L)
        assert rel_l2(got.cpu().numpy(), want.numpy()) < WAV_REL_TOL, (B, L)


def test_persistent_lstm_matches_step_lstm(gpu_model):
    """The one-launch LSTM (per-XCD clip groups, XCD-local step barrier) against the launch-per-step kernel: same
    codes, features equal, waveform to rounding; batch sizes that leave XCDs empty, partly filled and full (100: the
    last XCD holds 9 of 13 clip rows; 128: 16 per XCD, the state is polled in two rounds)."""
    name, m, sd = gpu_model
    from wavtokenizer_amd import synth
    for B in (1, 3, 8, 20, 64, 100, 128):
        wav = torch.from_numpy(synth.make_clips(B, 7200 if B > 8 else 24000, seed=500 + B)).cuda()
        f1, c1 = m.encode_infer(wav, bandwidth_id=BW)
        m.set_lstm_mode("step")
        try:
            f2, c2 = m.encode_infer(wav, bandwidth_id=BW)
        finally:
            m.set_lstm_mode("persistent")
        assert torch.equal(c1, c2), B
        assert torch.equal(f1, f2), B


def test_nan_audio_does_not_fault(gpu_model):
    """NaN samples poison every VQ distance of their frames: the argmax then selects nothing, and the code must still
    be a valid index (0, like torch.max on an all-NaN row) instead of an out-of-range codebook gather."""
    name, m, sd = gpu_model
    from wavtokenizer_amd import synth
    wav = torch.from_numpy(synth.make_clips(2, 6000, seed=3)).cuda()
    wav[1, 100:200] = float("nan")
    feats, codes = m.encode_infer(wav, bandwidth_id=BW)
    torch.cuda.synchronize()
    assert int(codes.min()) >= 0 and int(codes.max()) < 4096
    assert torch.isfinite(feats).all()                       # features are codebook rows
    clean_f, clean_c = m.encode_infer(wav[:1], bandwidth_id=BW)
    assert torch.equal(codes[:, :1], clean_c)                # the clean clip is untouched by its neighbour


def test_head_alone_matches_decode(gpu_model):
    """model.head(model.backbone(f)) == model.decode(f) (decoder/pretrained.py:203-206 composes exactly these)."""
    name, m, sd = gpu_model
    from wavtokenizer_amd import synth
    wav = torch.from_numpy(synth.make_clips(2, 12000, seed=31)).cuda()
    feats, _ = m.encode_infer(wav, bandwidth_id=BW)
    full = m.decode(feats, bandwidth_id=BW)
    x = m.backbone(feats, bandwidth_id=BW)
    assert x.shape == (2, feats.shape[-1], m.arch.dim)
    alone = m.head(x)
    assert alone.shape == full.shape
    assert rel_l2(alone.cpu().numpy(), full.cpu().numpy()) < 2e-6      # the fp32 backbone output re-split vs split in place


def test_plan_introspection_and_timing_hook(gpu_model):
    """wt_plan_* accessors the bench and the debug taps rely on."""
    import ctypes
    from wavtokenizer_amd import _capi
    name, m, _sd = gpu_model
    wav = torch.from_numpy(load_case(name, "b2_t72000")["wav_in"]).cuda()
    feats, _ = m.encode_infer(wav, bandwidth_id=BW)
    m.decode(feats, bandwidth_id=BW)
    L = feats.shape[-1]
    plan = m._engine.plans[(_capi.WT_PLAN_DECODE, 2, L, m._graph_flags(2))][0]
    n = _capi.lib.wt_plan_num_steps(plan)
    names = []
    for i in range(n):
        p = ctypes.c_char_p()
        assert _capi.lib.wt_plan_step_name(plan, i, ctypes.byref(p)) == 0
        names.append(p.value.decode())
    assert names.count("cnx.pwconv1") == m.arch.num_layers and "head.istft" in names and "head.ola" in names
    assert _capi.lib.wt_plan_frames(plan) == L and _capi.lib.wt_plan_num_launches(plan) >= n
    assert _capi.lib.wt_plan_workspace_bytes(plan) > 0
    _capi.check(_capi.lib.wt_plan_set_timing(plan, b"cnx.pwconv1"), "set_timing")
    out1 = m.decode(feats, bandwidth_id=BW)
    ms, cnt = ctypes.c_double(), ctypes.c_int64()
    _capi.check(_capi.lib.wt_plan_read_timing(plan, ctypes.byref(ms), ctypes.byref(cnt), 1), "read_timing")
    _capi.lib.wt_plan_set_timing(plan, b"")
    assert cnt.value == m.arch.num_layers and 0.0 < ms.value < 1000.0
    assert torch.equal(out1, m.decode(feats, bandwidth_id=BW))      # timing does not change results
    off, numel = ctypes.c_size_t(), ctypes.c_size_t()
    assert _capi.lib.wt_plan_find_buffer(plan, b"bb.out", ctypes.byref(off), ctypes.byref(numel)) == 0
    assert numel.value == 2 * L * m.arch.dim
    assert _capi.lib.wt_plan_find_buffer(plan, b"no.such.buffer", ctypes.byref(off), ctypes.byref(numel)) != 0


def test_graph_replay_matches_direct_launches(gpu_model):
    """Small batches replay a recorded hipGraph (WT_PLAN_FLAG_GRAPH): first call direct, second call records, later calls
    replay.  Results must be bit-identical to direct launches, for fresh inputs too (the graph reads fixed staging
    buffers), and the replay counter must show that the graph really served the calls."""
    from wavtokenizer_amd import _capi, synth
    name, m, _sd = gpu_model
    B, T = 3, 9000
    wavs = [torch.from_numpy(synth.make_clips(B, T, seed=900 + i)).cuda() for i in range(4)]
    m.set_graph_max_clips(0)
    try:
        want = []
        for w in wavs:
            f, c = m.encode_infer(w, bandwidth_id=BW)
            want.append((f, c, m.decode(f, bandwidth_id=BW)))
    finally:
        m.set_graph_max_clips(16)
    for w, (f0, c0, o0) in zip(wavs, want):
        f, c = m.encode_infer(w, bandwidth_id=BW)
        o = m.decode(f, bandwidth_id=BW)
        assert torch.equal(c, c0) and torch.equal(f, f0) and torch.equal(o, o0)
    L = want[0][0].shape[-1]
    flags = m._graph_flags(B)
    assert flags & _capi.WT_PLAN_FLAG_GRAPH
    eplan = m._engine.plans[(_capi.WT_PLAN_ENCODE, B, T, flags)][0]
    dplan = m._engine.plans[(_capi.WT_PLAN_DECODE, B, L, flags)][0]
    assert _capi.lib.wt_plan_graph_replays(eplan) >= 3 and _capi.lib.wt_plan_graph_replays(dplan) >= 3
    # results handed out earlier are copies: a later call must not change them
    f1, c1 = m.encode_infer(wavs[0], bandwidth_id=BW)
    keep = f1.clone()
    m.encode_infer(wavs[1], bandwidth_id=BW)
    assert torch.equal(f1, keep)


def test_persistent_lstm_long_sequence(gpu_model):
    """400 and 1200 recurrent steps (10 s / 30 s clips): the three exchange buffers of the persistent LSTM rotate hundreds
    of times; codes must still equal the launch-per-step kernel's."""
    name, m, sd = gpu_model
    from wavtokenizer_amd import synth
    hop = m.arch.hop
    for B, L in ((5, 400), (2, 1200)):
        wav = torch.from_numpy(synth.make_clips(B, L * hop, seed=640 + B)).cuda()
        _f1, c1 = m.encode_infer(wav, bandwidth_id=BW)
        m.set_lstm_mode("step")
        try:
            _f2, c2 = m.encode_infer(wav, bandwidth_id=BW)
        finally:
            m.set_lstm_mode("persistent")
        assert c1.shape == (1, B, L) and torch.equal(c1, c2), (B, L)


def test_decode_long_sequences_against_oracle(gpu_model):
    """Sequences beyond the one-slab GroupNorm kernel (L > 256: chunk statistics merged in chunk order, attention over a
    long window) against the oracle; 777 is not a multiple of the 128-row chunk."""
    name, m, sd = gpu_model
    orc = _oracle(name, sd)
    for i, (B, L) in enumerate(((2, 300), (1, 777))):
        feats = torch.randn(B, 512, L, generator=torch.Generator().manual_seed(4200 + i)) * 0.5
        with torch.inference_mode():
            want = orc.decode(feats, BW)
        got = m.decode(feats.cuda(), bandwidth_id=BW)
        assert tuple(got.shape) == tuple(want.shape)
        assert rel_l2(got.cpu().numpy(), want.numpy()) < WAV_REL_TOL, (B, L)


# ------------------------------------------------------ shipped kernels, one at a time, against the oracle
def _fold(sd, prefix):
    """weight_norm fold as wt_model_create does it (fp64): w = g * v / ||v|| (conv.py:25-34); returns [Cout][k][Cin]."""
    g = torch.from_numpy(sd[prefix + ".weight_g"]).double()
    v = torch.from_numpy(sd[prefix + ".weight_v"]).double()
    w = g * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)
    return w.float().permute(0, 2, 1).contiguous(), torch.from_numpy(sd[prefix + ".bias"])


ENC = "feature_extractor.encodec.encoder.model."


@pytest.mark.parametrize("T", [1, 2, 126, 127, 128, 129, 18000])
@pytest.mark.parametrize("stage,fold", [(1, True), (1, False), (4, False)])
def test_resblock16_kernel_against_oracle(stage, fold, T):
    """resblock16_kernel (the fused SEANetResnetBlock the encoder plan launches: C = 32 with the first conv folded into
    the tile fill, C = 32 plain, C = 64) against oracle.resblock (seanet.py:21-63) on tile-edge lengths: one frame, one
    tile +- 1, many tiles; fp32, S32 and elu'd outputs."""
    import torch.nn.functional as F
    from wavtokenizer_amd._capi import lib, check
    from tests import parity_log
    sd = synth_state_dict("hop600")
    orc = _oracle("hop600", sd)
    C = 32 if stage == 1 else 64
    B = 3 if T < 1000 else 2
    gen = torch.Generator().manual_seed(100 * stage + T)
    w3, b3 = _fold(sd, ENC + f"{stage}.block.1.conv.conv")
    w1, b1 = _fold(sd, ENC + f"{stage}.block.3.conv.conv")
    ws, bs = _fold(sd, ENC + f"{stage}.shortcut.conv.conv")
    if fold:
        wav = torch.randn(B, T, generator=gen) * 0.3
        e0w, e0b = _fold(sd, ENC + "0.conv.conv")                    # [32][7][1]
        e0w = e0w[:, :, 0].t().contiguous()                         # [7][32]
        with torch.inference_mode():
            x_ref = orc.sconv1d(wav.unsqueeze(1), ENC + "0.conv.conv")
    else:
        x_ref = torch.randn(B, C, T, generator=gen) * 0.8
    with torch.inference_mode():
        want = orc.resblock(x_ref, ENC + f"{stage}")
    dev = lambda t: t.cuda().contiguous()
    w3d, b3d, w1d, b1d, wsd, bsd = map(dev, (w3, b3, w1.reshape(C, C // 2), b1, ws.reshape(C, C), bs))
    worst = 0.0
    for elu_out, out_s32 in ((0, 0), (1, 1), (0, 1)):
        y = torch.full((B, T, C), float("nan"), device="cuda")
        if fold:
            wd, e0wd, e0bd = dev(wav), dev(e0w), dev(e0b)
            check(lib.wt_resblock(None, _ptr(wd), _ptr(e0wd), _ptr(e0bd), _ptr(w3d), _ptr(b3d), _ptr(w1d), _ptr(b1d), _ptr(wsd),
                                  _ptr(bsd), _ptr(y), B, T, C, elu_out, out_s32, 0, None), "wt_resblock")
        else:
            xd = dev(x_ref.permute(0, 2, 1))
            check(lib.wt_resblock(_ptr(xd), None, None, None, _ptr(w3d), _ptr(b3d), _ptr(w1d), _ptr(b1d), _ptr(wsd), _ptr(bsd),
                                  _ptr(y), B, T, C, elu_out, out_s32, 0, None), "wt_resblock")
        torch.cuda.synchronize()
        got = _decode_s32(y.view(B * T, C), B * T, C).float().view(B, T, C) if out_s32 else y
        ref = F.elu(want) if elu_out else want
        err = rel_l2(got.permute(0, 2, 1).cpu().numpy(), ref.numpy())
        worst = max(worst, err)
        assert err < 1e-5, (stage, fold, T, elu_out, out_s32, err)
    parity_log.record(f"resblock16[stage{stage},fold={int(fold)},T={T}]", rel_l2=worst)


@pytest.mark.parametrize("arch,T", [("hop600", 1024), ("hop600", 1025), ("hop600", 1027), ("hop600", 1080), ("hop600", 1081),
                                    ("hop600", 1200), ("hop600", 1203), ("hop600", 61920), ("hop600", 72000),
                                    ("hop320", 1024), ("hop320", 1116), ("hop320", 1117), ("hop320", 1241), ("hop320", 72000)])
def test_resblock16_down_kernel_against_oracle(arch, T):
    """The shipped stage-1 kernel (first conv + SEANetResnetBlock + ELU + the stage's strided conv in one launch,
    resblock16_kernel<DOWN = r>) against the oracle's three modules (seanet.py:117-127) on lengths around its 31- / 63-frame
    output tiles: an exact number of tiles, one frame more, a ragged tail, lengths that are no multiple of the stride (the
    last window is completed by extra reflected padding, conv.py:54-61), the benchmark length; r = 4 (hop600) and r = 2
    (hop320).  The first and last output frames take their reflected taps from inside the tile."""
    import torch.nn.functional as F
    from wavtokenizer_amd._capi import lib, check
    from tests import parity_log
    sd = synth_state_dict(arch)
    orc = _oracle(arch, sd)
    r = 4 if arch == "hop600" else 2
    B = 3 if T < 10000 else 2
    gen = torch.Generator().manual_seed(7 * T + r)
    wav = torch.randn(B, T, generator=gen) * 0.3
    w3, b3 = _fold(sd, ENC + "1.block.1.conv.conv")
    w1, b1 = _fold(sd, ENC + "1.block.3.conv.conv")
    ws, bs = _fold(sd, ENC + "1.shortcut.conv.conv")
    wdn, bdn = _fold(sd, ENC + "3.conv.conv")                       # [64][2r][32]
    e0w, e0b = _fold(sd, ENC + "0.conv.conv")
    e0w = e0w[:, :, 0].t().contiguous()
    assert tuple(wdn.shape) == (64, 2 * r, 32), wdn.shape
    with torch.inference_mode():
        x = orc.sconv1d(wav.unsqueeze(1), ENC + "0.conv.conv")
        want = orc.sconv1d(F.elu(orc.resblock(x, ENC + "1")), ENC + "3.conv.conv", stride=r)
    dev = lambda t: t.cuda().contiguous()
    args = list(map(dev, (wav, e0w, e0b, w3, b3, w1.reshape(32, 16), b1, ws.reshape(32, 32), bs, wdn, bdn)))
    Td = -(-T // r)
    y = torch.full((B, Td, 64), float("nan"), device="cuda")
    check(lib.wt_resblock_down(*[_ptr(t) for t in args], _ptr(y), B, T, r, None), "wt_resblock_down")
    torch.cuda.synchronize()
    assert tuple(want.shape) == (B, 64, Td), want.shape
    got = y.permute(0, 2, 1).cpu().numpy()
    err = rel_l2(got, want.numpy())
    edge = max(rel_l2(got[:, :, :2], want.numpy()[:, :, :2]), rel_l2(got[:, :, -2:], want.numpy()[:, :, -2:]))
    parity_log.record(f"resblock16_down[{arch},T={T}]", rel_l2=err, edge_rel_l2=edge)
    assert err < 1e-5 and edge < 1e-5, (arch, T, err, edge)


@pytest.mark.parametrize("mode", ["persistent", "step"])
def test_lstm_kernels_against_oracle(mode):
    """SLSTM (lstm.py:31-39) alone, on the kernels the encoder plan launches (the S32 input-projection GEMM, then
    lstm_persist_kernel or the launch-per-step kernel), against oracle.slstm: batch sizes that leave XCDs empty (1), ragged
    (9, 100), full (64) and at the 16-clips-per-XCD limit (128); sequences of one, two and 120 steps."""
    from tests import parity_log
    m, sd = _model("hop600")
    orc = _oracle("hop600", sd)
    m.set_lstm_mode(mode)
    worst = 0.0
    for B in (1, 9, 64, 100, 128):
        for L in (1, 2, 120):
            x = torch.randn(B, 512, L, generator=torch.Generator().manual_seed(B * 1000 + L)) * 0.7
            with torch.inference_mode():
                want = orc.slstm(x, ENC + "13")                            # (B, 512, L)
            got = m._run_unit_lstm(x.permute(0, 2, 1).contiguous().cuda())
            torch.cuda.synchronize()
            err = rel_l2(got.permute(0, 2, 1).cpu().numpy(), want.numpy())
            worst = max(worst, err)
            assert err < 1e-5, (mode, B, L, err)
    parity_log.record(f"lstm[{mode}]", rel_l2=worst)


# --------------------------------------------------------------- range of the split-f16 representation
@pytest.mark.parametrize("scale_x,scale_w", [(1e5, 1.0), (1e6, 1e3), (1e-7, 1.0), (1.0, 1e-7), (3e-5, 2e4)])
def test_linear_s32_operand_range(scale_x, scale_w):
    """wt_linear modes 2 / 3 with operands far from 1: the split form's f16 hi half would overflow at 65504 and loses
    relative precision below 6e-5, so the entry point scales each operand by a per-tensor power of two (exact) and
    brings the accumulators back; results must stay fp32-equivalent against float64 (decoder/modules.py:52-54 is a
    plain fp32 addmm, which has no such range limit)."""
    from wavtokenizer_amd._capi import lib, check
    M, N, K = 300, 96, 768
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(M, K, generator=gen) * scale_x).cuda()
    w = (torch.randn(N, K, generator=gen) / K ** 0.5 * scale_w).cuda()
    b = (torch.randn(N, generator=gen) * scale_x * scale_w).cuda()
    ws = torch.empty(4 * (M + N) * K + 1024, dtype=torch.uint8, device="cuda")
    ref = x.double() @ w.double().t() + b.double()
    for mode in (2, 3):
        y = torch.full((M, N), float("nan"), device="cuda")
        check(lib.wt_linear(_ptr(x), _ptr(w), _ptr(b), _ptr(y), M, N, K, mode, _ptr(ws), None), "wt_linear")
        torch.cuda.synchronize()
        if mode == 3 and not (6e-5 < scale_x * scale_w < 6e4):
            continue            # an S32-encoded OUTPUT is itself limited to the f16 range: only the fp32 output is checked here
        got = _decode_s32(y, M, N) if mode == 3 else y.double()
        assert torch.isfinite(got).all(), (scale_x, scale_w, mode)
        assert ((got - ref).norm() / ref.norm()).item() < 1e-6, (scale_x, scale_w, mode)


def test_activation_beyond_f16_range_is_loud_then_falls_back(gpu_model):
    """Features 1e5 times the usual scale overflow the f16 hi half of the first S32 producer.  The call must not hand
    out silent garbage: its output is overwritten with NaN and the status word says WT_STATUS_BIT_RANGE; the next call
    re-plans on fp32 GEMMs by itself and is correct; in strict mode the failing call itself is repeated."""
    import ctypes
    from wavtokenizer_amd import _capi, WavTokenizer, NAMED_ARCHS
    from tests import parity_log
    name, _m, sd = gpu_model
    m = WavTokenizer.from_arch(NAMED_ARCHS[name])           # a fresh model: the fallback is sticky
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m = m.eval().to("cuda")
    orc = _oracle(name, sd)
    feats = torch.randn(2, 512, 50, generator=torch.Generator().manual_seed(9)) * 0.5
    big = feats * 1e5
    with torch.inference_mode():
        want_big = orc.decode(big, BW)
        want = orc.decode(feats, BW)
    assert torch.isfinite(want_big).all()                  # the reference has no such limit
    m.set_graph_max_clips(0)
    out = m.decode(big.cuda(), bandwidth_id=BW)
    torch.cuda.synchronize()
    assert torch.isnan(out).all()                          # poisoned, not plausible
    plan = next(p for k, (p, _w) in m._engine.plans.items() if k[0] == _capi.WT_PLAN_DECODE)
    bits = ctypes.c_int32()
    _capi.check(_capi.lib.wt_plan_status(plan, ctypes.byref(bits), 0), "wt_plan_status")
    assert bits.value & _capi.WT_STATUS_BIT_RANGE
    mbits = ctypes.c_int32()
    _capi.check(_capi.lib.wt_model_status(m._engine.model, ctypes.byref(mbits), 0), "wt_model_status")
    assert mbits.value & _capi.WT_STATUS_BIT_RANGE         # the model's word carries it too (any plan's next call sees it)
    sites = ctypes.c_uint64()
    _capi.check(_capi.lib.wt_plan_range_sites(plan, ctypes.byref(sites), 0), "wt_plan_range_sites")
    assert sites.value & (1 << _capi.WT_SITE_BB_EMBED), hex(sites.value)     # the first S32 producer: the input transpose
    # WT_ERR_RANGE inside -> the reporting site goes to fp32 operands -> the call runs; each call answers the lowest site that
    # reported, so an input that overflows several sites converges over as many calls (asynchronous mode)
    for _ in range(6):
        out2 = m.decode(big.cuda(), bandwidth_id=BW)
        torch.cuda.synchronize()
        if torch.isfinite(out2).all():
            break
    assert m._fp32_sites & (1 << _capi.WT_SITE_BB_EMBED) and not (m._plan_flags & _capi.WT_PLAN_FLAG_FP32_GEMM)
    n_sites = bin(m._fp32_sites).count("1")
    assert n_sites <= 4, bin(m._fp32_sites)                  # the norms bring everything behind the embedding back into range
    e_big = rel_l2(out2.cpu().numpy(), want_big.numpy())
    assert e_big < WAV_REL_TOL, e_big
    # strict mode on a second fresh model: the failing call itself comes back correct
    m2 = WavTokenizer.from_arch(NAMED_ARCHS[name])
    m2.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m2 = m2.eval().to("cuda")
    m2.set_strict_status(True)
    e_strict = rel_l2(m2.decode(big.cuda(), bandwidth_id=BW).cpu().numpy(), want_big.numpy())
    assert e_strict < WAV_REL_TOL, e_strict
    # small magnitudes need no fallback: absolute floor 2^-36 of the split form, far below fp32 rounding of the sums
    m3, _ = _model(name)
    tiny = feats * 1e-7
    with torch.inference_mode():
        want_tiny = orc.decode(tiny, BW)
    out3 = m3.decode(tiny.cuda(), bandwidth_id=BW)
    e_tiny = rel_l2(out3.cpu().numpy(), want_tiny.numpy())
    m3.check_status()
    assert e_tiny < WAV_REL_TOL, e_tiny
    e_norm = rel_l2(m3.decode(feats.cuda(), bandwidth_id=BW).cpu().numpy(), want.numpy())
    parity_log.record(f"range[{name}]", wav_rel_l2_x1e5_fp32_fallback=e_big, wav_rel_l2_x1e5_strict=e_strict,
                      wav_rel_l2_x1e_7=e_tiny, wav_rel_l2=e_norm, fp32_sites_after_x1e5=n_sites)


def test_range_overflow_in_one_convnext_block_costs_one_block():
    """an activation beyond the f16 range inside ONE ConvNeXt block (forced: +1e5 on that block's pwconv1
    bias, so gelu(.) ~ 1e5 > 65504 in the S32 tensor between pwconv1 and pwconv2) must put that block on fp32 operands,
    not the model: the waveform still matches the oracle (same weights), no other site falls back, and the round trip
    at 64 x 3 s gets slower by less than 10 % (r03: 2.8x, the whole model on the fp32 chain)."""
    import time
    from wavtokenizer_amd import _capi, WavTokenizer, NAMED_ARCHS, synth
    from tests import parity_log
    name = "hop600"
    sd = dict(synth_state_dict(name))
    key = "backbone.convnext.5.pwconv1.bias"
    sd[key] = (sd[key] + 1e5).astype(np.float32)
    orc = _oracle(name, sd)
    m = _fresh_model(name, sd)
    m0 = _fresh_model(name, synth_state_dict(name))           # the unmodified model: the timing baseline
    wav = torch.from_numpy(synth.make_clips(64, 72000, seed=4242)).cuda()

    def step_ms(model, n=10):
        for _ in range(3):
            f, _c = model.encode_infer(wav, bandwidth_id=BW)
            model.decode(f, bandwidth_id=BW)
        blocks = []
        for _ in range(3):                    # median of three blocks: a clock ramp or a neighbour's burst must not decide the test
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                f, _c = model.encode_infer(wav, bandwidth_id=BW)
                model.decode(f, bandwidth_id=BW)
            torch.cuda.synchronize()
            blocks.append(1e3 * (time.perf_counter() - t0) / n)
        return sorted(blocks)[1]

    m.set_strict_status(True)                 # the failing call itself is repeated on the fallback path
    f, c = m.encode_infer(wav[:2], bandwidth_id=BW)
    out = m.decode(f, bandwidth_id=BW)
    assert torch.isfinite(out).all()
    assert m._fp32_sites == 1 << (_capi.WT_SITE_CNX0 + 5), bin(m._fp32_sites)
    assert not (m._plan_flags & _capi.WT_PLAN_FLAG_FP32_GEMM)
    with torch.inference_mode():
        fo, _co = orc.encode_infer(wav[:2].cpu(), BW)
        want = orc.decode(fo, BW)
    err = rel_l2(m.decode(fo.cuda(), bandwidth_id=BW).cpu().numpy(), want.numpy())
    assert err < WAV_REL_TOL, err
    with pytest.raises(_capi.WavTokError, match="fallback"):
        m.check_status()
    m.set_strict_status(None)
    t_base, t_fb = step_ms(m0), step_ms(m)
    if t_fb >= 1.10 * t_base:                 # once more, the other way round, before failing on a timing
        t_fb, t_base = step_ms(m), step_ms(m0)
    m.check_status()
    parity_log.record("range_one_block", wav_rel_l2=err, ms_step_all_s32=t_base, ms_step_block5_fp32=t_fb, slowdown=t_fb / t_base)
    assert t_fb < 1.10 * t_base, (t_base, t_fb)


def test_range_report_lists_every_s32_operand(gpu_model):
    """model.range_report: one entry per (step, S32 buffer) of the encode and the decode plan, finite head-room on the
    fixture input, and the encoder / every ConvNeXt block present."""
    from wavtokenizer_amd import synth
    from tests import parity_log
    name, m, _sd = gpu_model
    wav = torch.from_numpy(synth.make_clips(2, 24000, seed=88)).cuda()
    rep = m.range_report(wav, bandwidth_id=BW)
    assert len(rep) > 60
    steps = {r["step"] for r in rep}
    assert "cnx.pwconv1" in steps and "vq.argmin" in steps and "head.out" in steps
    worst = min(rep, key=lambda r: r["headroom_bits"])
    assert np.isfinite(worst["headroom_bits"]) and worst["headroom_bits"] > 2.0, worst
    m.check_status()
    parity_log.record(f"range_report[{name}]", entries=len(rep), least_headroom_bits=worst["headroom_bits"],
                      least_headroom_at=worst["step"] + ":" + worst["buffer"])


def test_codes_out_of_range_raise_like_embedding(gpu_model):
    """F.embedding raises IndexError on a code outside the codebook (decoder/pretrained.py:236); so does the drop-in."""
    name, m, _sd = gpu_model
    codes = torch.zeros(1, 2, 7, dtype=torch.int64, device="cuda")
    m.set_check_codes("sync")
    try:
        m.codes_to_features(codes)
        for bad in (4096, -1, 1 << 40):
            c = codes.clone()
            c[0, 1, 3] = bad
            with pytest.raises(IndexError):
                m.codes_to_features(c)
        assert torch.isfinite(m.codes_to_features(codes)).all()      # the flag does not stick
    finally:
        m.set_check_codes("deferred")
    # default mode: no stream synchronisation inside codes_to_features (the reference's CUDA F.embedding does not
    # synchronise either); the bad index gives NaN features at once and IndexError on the next call or in check_status()
    c = codes.clone()
    c[0, 0, 2] = 5000
    f = m.codes_to_features(c)
    torch.cuda.synchronize()
    assert torch.isnan(f[0, :, 2]).all() and torch.isfinite(f[1]).all()
    with pytest.raises(IndexError):
        m.codes_to_features(codes)
    assert torch.isfinite(m.codes_to_features(codes)).all()
    m.check_status()


def test_graph_replay_single_clip_alternating_plans(gpu_model):
    """The reference's own usage (infer.py:44-70): one clip per call.  Encode and decode graphs replay alternately; every
    replay must equal the direct launches bit for bit.  (Round 2 regression: hipMemsetAsync nodes replayed from a hipGraph
    left garbage at the start of their destination — the V^T pad fill and the LSTM exchange buffers; fills are kernels now.)"""
    from wavtokenizer_amd import synth
    name, m, _sd = gpu_model
    for T in (24000, 36963):
        x = torch.from_numpy(synth.make_clips(1, T, seed=77)).cuda()
        m.set_graph_max_clips(0)
        try:
            f0, c0 = m.encode_infer(x, bandwidth_id=BW)
            y0 = m.decode(f0, bandwidth_id=BW)
        finally:
            m.set_graph_max_clips(16)
        for rep in range(5):
            f, c = m.encode_infer(x, bandwidth_id=BW)
            y = m.decode(f, bandwidth_id=BW)
            assert torch.equal(c, c0), (T, rep)
            assert torch.equal(f, f0), (T, rep)
            assert torch.equal(y, y0), (T, rep)
    m.check_status()


def test_packed_image_round_trip(gpu_model, tmp_path):
    """save_packed / from_packed: the image of the model as it sits in HBM loads without recomputation into a
    model whose own image is byte-identical (every folded / packed / split array and the pointer table) and whose outputs
    equal the original's bit for bit; the load times of both routes are recorded."""
    import hashlib
    import time
    import yaml
    from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS, synth
    from tests import parity_log
    name, m, sd = gpu_model
    arch = NAMED_ARCHS[name]
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text(yaml.safe_dump({"model": {"init_args": arch.to_yaml_node()}}))
    path = str(tmp_path / "model.wtpk")
    m.save_packed(path)
    t0 = time.perf_counter()
    m2 = WavTokenizer.from_packed(str(cfg), path)
    torch.cuda.synchronize()
    t_packed = time.perf_counter() - t0
    t0 = time.perf_counter()
    m3 = WavTokenizer.from_arch(arch)
    m3.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m3 = m3.eval().to("cuda")
    m3._ensure_engine()
    torch.cuda.synchronize()
    t_state = time.perf_counter() - t0
    img1, img2 = m._engine.export(), m2._engine.export()
    assert hashlib.sha256(img1.tobytes()).hexdigest() == hashlib.sha256(img2.tobytes()).hexdigest()
    wav = torch.from_numpy(synth.make_clips(3, 12000, seed=41)).cuda()
    f1, c1 = m.encode_infer(wav, bandwidth_id=BW)
    f2, c2 = m2.encode_infer(wav, bandwidth_id=BW)
    assert torch.equal(c1, c2) and torch.equal(f1, f2)
    assert torch.equal(m.decode(f1, bandwidth_id=BW), m2.decode(f2, bandwidth_id=BW))
    with pytest.raises(RuntimeError):
        m2.state_dict()
    # the image holds no fp32 copies of the GEMM weights (they are rebuilt from the S32 copies when a plan on the fp32
    # chain is first created: 22 of 24 significant bits): the fp32 chain of the packed model still meets the usual bar
    m2.set_gemm_precision("f32")
    m.set_gemm_precision("f32")
    try:
        fa, ca = m.encode_infer(wav, bandwidth_id=BW)
        fb, cb = m2.encode_infer(wav, bandwidth_id=BW)
        wa, wb = m.decode(fa, bandwidth_id=BW), m2.decode(fa, bandwidth_id=BW)
    finally:
        m.set_gemm_precision("f16x3")
        m2.set_gemm_precision("f16x3")
    assert torch.equal(ca, cb)
    e32 = rel_l2(wb.cpu().numpy(), wa.cpu().numpy())
    assert e32 < 2e-5, e32
    m.check_status()
    m2.check_status()
    assert img1.nbytes < 650e6, img1.nbytes
    parity_log.record(f"packed_load[{name}]", image_mb=img1.nbytes / 1e6, from_packed_s=t_packed, from_state_dict_s=t_state,
                      fp32_chain_packed_vs_exact_rel_l2=e32)


# ------------------------------------------------------------------------------------------ round 3
def _fresh_model(name, sd):
    from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS
    m = WavTokenizer.from_arch(NAMED_ARCHS[name])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return m.eval().to("cuda")


def test_precision_against_float64(gpu_model):
    """What "fp32-equivalent" means end to end, measured: the oracle runs in FLOAT64 (same op sequence, weights and inputs
    cast up) and three fp32-class implementations are compared with it on the b2 fixture inputs: the CPU fp32 oracle
    (= the reference), the shipped GPU path (split-f16 products, S32 operands between dense layers) and the GPU's plain
    fp32 MFMA chain (WT_PLAN_FLAG_FP32_GEMM).  Recorded per arch: rel-L2 of the encoder output (pre-VQ), of the decoded
    waveform (same features on every side), and code agreement.  Bar: the shipped path's error stays within 4x the CPU
    fp32 oracle's own error (both are rounding noise around the same real-arithmetic result), and far inside 1e-4."""
    from tests import parity_log
    name, m, sd = gpu_model
    g = load_case(name, "b2_t72000")
    wav_np = g["wav_in"]
    o32 = _oracle(name, sd)
    from oracle.cpu_ref import OracleWavTokenizer
    from wavtokenizer_amd import NAMED_ARCHS
    o64 = OracleWavTokenizer(NAMED_ARCHS[name], {k: torch.from_numpy(v).double() for k, v in sd.items()})
    t32, t64 = {}, {}
    with torch.inference_mode():
        f32, c32 = o32.encode_infer(torch.from_numpy(wav_np), BW, t32)
        f64, c64 = o64.encode_infer(torch.from_numpy(wav_np).double(), BW, t64)
        w64 = o64.decode(f32.double(), BW)              # the same (fp32 oracle's) features on every side
        w32 = o32.decode(f32, BW)
    emb_key = max((k for k in t64 if k.startswith("enc.")), key=lambda k: int(k.split(".")[1]))     # last encoder tap = pre-VQ embedding
    e64 = t64[emb_key].numpy()
    res = {"cpu_fp32_oracle": {"emb": rel_l2(t32[emb_key].numpy(), e64), "wav": rel_l2(w32.numpy(), w64.numpy()),
                               "codes_equal_fp64": bool(torch.equal(c32, c64))}}
    wav = torch.from_numpy(wav_np).cuda()
    for label, mode in (("gpu_f16x3_shipped", "f16x3"), ("gpu_fp32_chain", "f32")):
        m.set_gemm_precision(mode)
        try:
            emb = m.feature_extractor.encodec.encoder(wav.unsqueeze(1)).cpu().numpy()
            _f, codes = m.encode_infer(wav, bandwidth_id=BW)
            wg = m.decode(f32.cuda(), bandwidth_id=BW).cpu().numpy()
        finally:
            m.set_gemm_precision("f16x3")
        res[label] = {"emb": rel_l2(emb, e64), "wav": rel_l2(wg, w64.numpy()), "codes_equal_fp64": bool(torch.equal(codes.cpu(), c64))}
    m.check_status()
    ratio_wav = res["gpu_f16x3_shipped"]["wav"] / res["cpu_fp32_oracle"]["wav"]
    ratio_emb = res["gpu_f16x3_shipped"]["emb"] / res["cpu_fp32_oracle"]["emb"]
    parity_log.record(f"vs_float64[{name}]", **{f"{k}.{kk}": vv for k, v in res.items() for kk, vv in v.items()},
                      shipped_over_cpu_fp32_wav=ratio_wav, shipped_over_cpu_fp32_emb=ratio_emb)
    print(f"[{name}] vs float64: {res}")
    assert all(v["codes_equal_fp64"] for v in res.values()), res
    assert res["gpu_f16x3_shipped"]["wav"] < 2e-5 and res["gpu_f16x3_shipped"]["emb"] < 2e-5, res
    assert ratio_wav < 4.0 and ratio_emb < 4.0, res


def test_30s_batch32():
    """BASELINE configs[4] at its per-GPU batch: 32 clips x 30 s (L = 1200).  The fixture clip (hop600_b1_t720000, captured
    from the reference) sits in slot k of a batch of other clips: its codes must equal the fixture's (so the result of a
    clip does not depend on its slot or its neighbours: 4 clips per XCD in lstm_persist, the chunked GroupNorm, the long
    attention), its waveform head / tail / norm must match, the status word stays clean; p50 encode latency is recorded."""
    import time
    from wavtokenizer_amd import synth
    from tests import parity_log
    m, _sd = _model("hop600")
    g = load_case("hop600", "b1_t720000")
    fill = synth.make_clips(32, 720000, seed=3100)
    ref_codes = None
    lat = []
    for k in (0, 17, 31):
        batch = fill.copy()
        batch[k] = g["wav_in"][0]
        wav = torch.from_numpy(batch).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        feats, codes = m.encode_infer(wav, bandwidth_id=BW)
        torch.cuda.synchronize()
        lat.append(1e3 * (time.perf_counter() - t0))
        out = m.decode(feats, bandwidth_id=BW)
        ck = codes[:, k:k + 1].cpu().numpy()
        assert check_codes(ck, g["codes"], g["margin"], f"30s slot {k}") == 0
        o = out[k:k + 1].cpu().numpy()
        assert rel_l2(o[:, :4096], g["wav_out_head"]) < WAV_REL_TOL, k
        assert rel_l2(o[:, -4096:], g["wav_out_tail"]) < WAV_REL_TOL, k
        assert abs(np.sqrt((o.astype(np.float64) ** 2).sum()) - float(g["wav_out_l2"])) < WAV_REL_TOL * float(g["wav_out_l2"])
        # the other 31 clips do not depend on slot k either: clips outside {0, 17, 31} are identical in all three batches
        others = codes[:, 1:17].cpu()
        if ref_codes is None:
            ref_codes = others
        assert torch.equal(others, ref_codes), k
        assert torch.isfinite(out).all()
    m.check_status()
    parity_log.record("batch32x30s[hop600]", slots=3, encode_ms_first=lat[0], encode_ms_p50=sorted(lat)[1])


@pytest.mark.parametrize("name", ["hop600", "hop320"])
def test_trained_like_weights(name):
    """Weights with the statistics of a trained checkpoint instead of an init (synth.make_trained_like_state_dict: log-normal
    weight_g, heavy-tailed matrices with outliers, layer scale up to 10, spread norm scales; no trained checkpoint exists
    offline).  Fixture = the reference's outputs on them (tests/golden/make_golden_trained_like.py).  The shipped path and
    the fp32 chain are both held to the usual bars, and it is recorded whether the f16-range guard fired."""
    import os
    from wavtokenizer_amd import NAMED_ARCHS, synth
    from tests.util import GOLDEN
    from tests import parity_log
    g = np.load(os.path.join(GOLDEN, f"{name}_trained_like.npz"))
    sd = synth.make_trained_like_state_dict(NAMED_ARCHS[name], seed=int(g["weight_seed"]))
    bw = torch.tensor([int(g["bandwidth_id"])])
    wav = torch.from_numpy(g["wav_in"]).cuda()
    rec = {}
    for label, mode in (("f16x3", "f16x3"), ("fp32_chain", "f32")):
        m = _fresh_model(name, sd)
        m.set_gemm_precision(mode)
        feats, codes = m.encode_infer(wav, bandwidth_id=bw)
        out = m.decode(feats, bandwidth_id=bw)
        torch.cuda.synchronize()
        fired = False
        try:
            m.check_status()
        except Exception:                      # the range guard fired: the class has switched to fp32 GEMMs, repeat
            fired = True
            feats, codes = m.encode_infer(wav, bandwidth_id=bw)
            out = m.decode(feats, bandwidth_id=bw)
            m.check_status()
        flips = check_codes(codes.cpu().numpy(), g["codes"], g["margin"], f"trained-like {name} {label}")
        err = rel_l2(out.cpu().numpy(), g["wav_out"]) if flips == 0 else float("nan")
        rec[label] = (flips, err, fired)
        assert flips == 0 and err < WAV_REL_TOL, (label, flips, err)
    parity_log.record(f"trained_like[{name}]", code_flips=rec["f16x3"][0], frames=int(g["codes"].size), wav_rel_l2=rec["f16x3"][1],
                      range_guard_fired=rec["f16x3"][2], wav_rel_l2_fp32_chain=rec["fp32_chain"][1], min_margin=float(g["margin"].min()))


def test_bandwidth_id_tensor_created_under_inference_mode(gpu_model):
    """infer.py builds bandwidth_id on the GPU; a tensor created under torch.inference_mode() has no version counter."""
    name, m, _sd = gpu_model
    feats = torch.randn(1, 512, 20, generator=torch.Generator().manual_seed(3)).cuda()
    want = m.decode(feats, bandwidth_id=torch.tensor([1]))
    with torch.inference_mode():
        bw = torch.tensor([1]).cuda()
        got = m.decode(feats, bandwidth_id=bw)
        got2 = m.decode(feats, bandwidth_id=bw)
    assert torch.equal(got, want) and torch.equal(got2, want)
    bw_plain = torch.tensor([1]).cuda()                 # the cached path (a normal tensor, same object passed twice)
    assert torch.equal(m.decode(feats, bandwidth_id=bw_plain), want) and torch.equal(m.decode(feats, bandwidth_id=bw_plain), want)


def test_corrupt_packed_images_are_refused(gpu_model):
    """The packed image is not trusted: truncation, a flipped payload bit, a flipped bit in the model section and a pointer
    offset outside its allocation all end in an error return (never a crash, never a model that faults later)."""
    import ctypes
    from wavtokenizer_amd import _capi
    name, m, _sd = gpu_model
    img = m._engine.export()
    lib = _capi.lib

    def load(buf):
        h = ctypes.c_void_p()
        rc = lib.wt_model_create_packed(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes, torch.cuda.current_device(), ctypes.byref(h))
        if rc == 0:
            lib.wt_model_destroy(h)
        return rc, lib.wt_last_error().decode()

    assert load(img)[0] == 0
    assert lib.wt_packed_verify(img.ctypes.data_as(ctypes.c_void_p), img.nbytes) == 0
    for cut in (img.nbytes - 1, img.nbytes // 2, 4096, 100):
        rc, msg = load(img[:cut].copy())
        assert rc != 0, cut
    flip = img.copy()
    flip[img.nbytes - 12345] ^= 0x10                     # payload
    rc, msg = load(flip)
    assert rc != 0 and "hash" in msg, msg
    flip = img.copy()
    flip[128 + 8 * 4 + 40] ^= 0x01                      # allocation table / model section
    assert load(flip)[0] != 0
    # a consistent-looking file (hash recomputed is not possible from here): at least the verify entry point agrees
    assert lib.wt_packed_verify(flip.ctypes.data_as(ctypes.c_void_p), flip.nbytes) != 0


def test_batch_beyond_the_persistent_lstm_limit(gpu_model):
    """More than 128 clips per call do not fit the persistent LSTM's per-XCD clip groups (16 per XCD): the plan then runs the
    launch-per-step kernel (documented in DESIGN section 7).  Results must not depend on that: 130 clips in one call equal
    the same clips in two calls of 65."""
    from wavtokenizer_amd import synth
    name, m, _sd = gpu_model
    wav = torch.from_numpy(synth.make_clips(130, 4800, seed=1300)).cuda()
    f, c = m.encode_infer(wav, bandwidth_id=BW)
    fa, ca = m.encode_infer(wav[:65].contiguous(), bandwidth_id=BW)
    fb, cb = m.encode_infer(wav[65:].contiguous(), bandwidth_id=BW)
    assert torch.equal(c, torch.cat([ca, cb], dim=1)) and torch.equal(f, torch.cat([fa, fb], dim=0))
    m.check_status()


def test_two_lanes_match_one_stream(gpu_model):
    """One model called from two HIP streams at once (sharding.StepRunner(lanes=2): step i+1's encode_infer beside step i's
    decode).  Every stream gets its own plan and workspace (pretrained._Engine._key) and the library chains the persistent
    LSTM launches across streams (capi.cpp LstmChain), so the results are those of the single-stream calls, bit for bit, for
    the whole-batch form (B = 20, persistent LSTM) and for the graph-replay form (B = 2)."""
    from wavtokenizer_amd import synth
    from wavtokenizer_amd.sharding import StepRunner
    name, m, _sd = gpu_model
    for B, T in ((20, 9600), (2, 7200)):
        wav = torch.from_numpy(synth.make_clips(B, T, seed=4200 + B)).cuda()
        f0, c0 = m.encode_infer(wav, bandwidth_id=BW)
        w0 = m.decode(f0, bandwidth_id=BW)
        torch.cuda.synchronize()
        runner = StepRunner(m, wav, BW, None, 1, 0, gather=False, lanes=2)
        outs = [runner.step() for _ in range(6)]
        runner.drain()
        torch.cuda.synchronize()
        for codes, out, _ in outs:
            assert torch.equal(codes, c0) and torch.equal(out, w0)
        keys = [k for k in m._engine.plans if len(k) == 5 and k[1] == B]
        assert len({k[4] for k in keys}) == 2, keys                  # one plan set per lane stream
        m.check_status()


def test_device_clock_timing_of_a_gemm_step(gpu_model):
    """bench.py's roofline leg: wt_plan_set_timing("@name") makes the step's gemm16s launch stamp its own entry / exit on the
    device clock (no events in the stream).  Every timed launch is counted once, the durations are sane and not larger than
    what bracketing HIP events report for the same launches (those include the event packets), a step that is no gemm16s
    launch is refused, and the results of a timed call are those of an untimed one."""
    import ctypes
    from wavtokenizer_amd import _capi, synth
    lib = _capi.lib
    name, m, _sd = gpu_model
    B, T = 20, 24000
    wav = torch.from_numpy(synth.make_clips(B, T, seed=5100)).cuda()
    f, _c = m.encode_infer(wav, bandwidth_id=BW)
    ref = m.decode(f, bandwidth_id=BW)
    plan = m._engine.plans[(_capi.WT_PLAN_DECODE, B, f.shape[2], m._graph_flags(B))][0]

    def timed(flt, calls=3):
        _capi.check(lib.wt_plan_set_timing(plan, flt), "set_timing")
        outs = [m.decode(f, bandwidth_id=BW) for _ in range(calls)]
        tot, n = ctypes.c_double(), ctypes.c_int64()
        _capi.check(lib.wt_plan_read_timing(plan, ctypes.byref(tot), ctypes.byref(n), 1), "read_timing")
        lib.wt_plan_set_timing(plan, b"")
        return outs, tot.value, n.value

    outs, ms_dev, n_dev = timed(b"@cnx.pwconv1")
    assert n_dev == 3 * m._arch.num_layers
    assert all(torch.equal(o, ref) for o in outs)
    _outs, ms_ev, n_ev = timed(b"cnx.pwconv1")
    assert n_ev == n_dev
    per_dev, per_ev = ms_dev / n_dev, ms_ev / n_ev
    assert 1e-3 < per_dev < 5.0 and per_dev < per_ev * 1.05, (per_dev, per_ev)
    lib.wt_plan_set_timing(plan, b"@bb.cnx.norm")              # a row kernel, not a gemm16s launch
    with pytest.raises(Exception, match="gemm16s"):
        m.decode(f, bandwidth_id=BW)
    lib.wt_plan_set_timing(plan, b"")
    assert torch.equal(m.decode(f, bandwidth_id=BW), ref)
    m.check_status()


def test_fault_injection_on_the_lab_library():
    """The product library carries no fault hook, so the tests that force the persistent LSTM to lose
    co-residency (tests/lab/test_lab_faults.py: poisoned outputs, loud error, step-kernel retry, the error reaching the
    next call on another plan, graph replays resuming after the fallback) run ONCE in a child process on the LAB build
    (tools/lib/libwavtok_hip_lab.so through WAVTOK_HIP_LIB).  One child, not one per case."""
    import os
    import subprocess
    import sys
    from tests.util import ROOT
    lab = os.path.join(ROOT, "tools", "lib", "libwavtok_hip_lab.so")
    assert os.path.exists(lab), "build it: make -C wavtokenizer_amd/csrc lab (or __graft_entry__.build())"
    torch.cuda.synchronize()
    env = dict(os.environ, WAVTOK_HIP_LIB=lab, WAVTOK_PARITY_SUMMARY="gpu_parity_summary_lab.json")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "lab"), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout or "")[-3000:] + (r.stderr or "")[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], tail


def test_rccl_world1_exchange_beside_the_persistent_lstm():
    """the `nccl` branch had never executed anywhere.  On the one GPU of this box: a world-1 RCCL process group
    (device_id set), StepRunner(gather=True, force_collectives=True) for 20 steps of 8 clips beside the persistent LSTM;
    asserts the exchange order on the real codec's log, the gathered tensors against the steps' own, no
    status bit (an RCCL kernel in flight must not cost the persistent LSTM its co-residency), persistent LSTM still on."""
    import os
    import torch.distributed as dist
    from wavtokenizer_amd import synth
    from wavtokenizer_amd.sharding import StepRunner, check_exchange_order
    from tests import parity_log
    name = "hop600"
    m, _sd = _model(name)
    m.set_graph_max_clips(0)                   # direct launches: the persistent kernel itself, not a replayed graph
    dev = torch.device("cuda", torch.cuda.current_device())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 2000))
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        wav = torch.from_numpy(synth.make_clips(8, 24000, seed=911)).cuda()
        f0, c0 = m.encode_infer(wav, bandwidth_id=BW)
        w0 = m.decode(f0, bandwidth_id=BW)
        log = []
        r = StepRunner(m, wav, BW, dist, 1, 0, gather=True, backend="nccl", log=log, force_collectives=True)
        assert r.gather
        prev = None
        for i in range(20):
            codes, out, res = r.step()
            assert torch.equal(codes, c0) and torch.equal(out, w0)
            if prev is not None:
                assert res is not None and torch.equal(res[0], prev[0]) and torch.equal(res[1], prev[1])
            prev = (codes, out)
        res = r.drain()
        assert torch.equal(res[0], prev[0]) and torch.equal(res[1], prev[1])
        torch.cuda.synchronize()
        check_exchange_order(log, 20)
        m.check_status()
        assert m.persistent_lstm and r.exchanges == 20
        parity_log.record("rccl_world1", exchanges=r.exchanges, backend=dist.get_backend(), exchange_wait_ms=1e3 * r.wait_s / r.exchanges)
    finally:
        if own:
            dist.destroy_process_group()


def test_codes_to_features_sums_several_codebooks():
    """decoder/pretrained.py:230-237 sums the rows of K concatenated codebooks.  Every YAML of the reference has
    num_quantizers = 1; a checkpoint with more (K <= num_quantizers codes per frame) goes through the same gather: bit-equal
    to the oracle's F.embedding(...).sum(0) for K = 1, 2 and 3, (K, L) and (K, B, L) layouts, and encode_infer still uses the
    first codebook alone (vq.py:137)."""
    import dataclasses
    from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS, synth
    from oracle.cpu_ref import OracleWavTokenizer
    arch = dataclasses.replace(NAMED_ARCHS["hop600"], num_quantizers=3)
    sd = synth.make_state_dict(arch, seed=manifest()["weight_seed"])
    m = WavTokenizer.from_arch(arch)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m = m.eval().to("cuda")
    orc = OracleWavTokenizer(arch, sd)
    gen = torch.Generator().manual_seed(3)
    for K in (1, 2, 3):
        codes = torch.randint(0, arch.vq_bins, (K, 5, 37), generator=gen)
        assert torch.equal(m.codes_to_features(codes.cuda()).cpu(), orc.codes_to_features(codes))
    c2 = torch.randint(0, arch.vq_bins, (2, 19), generator=gen)
    assert torch.equal(m.codes_to_features(c2.cuda()).cpu(), orc.codes_to_features(c2))
    import os
    from tests.util import GOLDEN
    g = np.load(os.path.join(GOLDEN, "hop600_codebooks3.npz"))          # captured from the reference class itself
    for tag in ("k1", "k2", "k3", "k2_2d"):
        assert np.array_equal(m.codes_to_features(torch.from_numpy(g[f"{tag}/codes"]).cuda()).cpu().numpy(), g[f"{tag}/features"]), tag
    with pytest.raises(Exception):
        m.codes_to_features(torch.zeros(4, 1, 7, dtype=torch.int64, device="cuda"))      # more code rows than codebooks
    # the encode path of this checkpoint equals the single-codebook model's (same layer-0 weights)
    m1, _sd1 = _model("hop600")
    wav = torch.from_numpy(synth.make_clips(2, 9000, seed=12)).cuda()
    f3, c3 = m.encode_infer(wav, bandwidth_id=BW)
    f1, c1 = m1.encode_infer(wav, bandwidth_id=BW)
    assert c3.shape == (1, 2, 15) and torch.equal(c3, c1) and torch.equal(f3, f1)
    assert torch.equal(m.codes_to_features(c3), f3)
    m.check_status()


def test_range_overflow_in_the_encoder_puts_only_the_encoder_on_fp32(gpu_model):
    """The encoder is one range site (WT_SITE_ENCODER).  A waveform 10^6 times full scale overflows the f16 range of its
    first S32 tensor; the reference has no such limit.  With the default (strict for small batches) status the failing call
    itself comes back correct from the fp32 chain - codes equal the oracle's by the margin rule - only bit 0 of the site mask
    is set, and decode keeps running on the split-f16 kernels (its plans carry no fp32 site)."""
    from wavtokenizer_amd import _capi, synth
    from tests import parity_log
    name, _m, sd = gpu_model
    m = _fresh_model(name, sd)
    orc = _oracle(name, sd)
    wav = synth.make_clips(2, 12000, seed=77) * np.float32(1e6)
    taps = {}
    with torch.inference_mode():
        fo, co = orc.encode_infer(torch.from_numpy(wav), BW, taps)
        wo = orc.decode(fo, BW)
    assert torch.isfinite(fo).all()
    feats, codes = m.encode_infer(torch.from_numpy(wav).cuda(), bandwidth_id=BW)
    assert m._fp32_sites == 1 << _capi.WT_SITE_ENCODER, bin(m._fp32_sites)
    flips = check_codes(codes.cpu().numpy(), co.numpy(), taps["vq.margin"].numpy(), f"{name} x1e6")
    out = m.decode(fo.cuda(), bandwidth_id=BW)
    err = rel_l2(out.cpu().numpy(), wo.numpy())
    assert err < WAV_REL_TOL, err
    assert m._fp32_sites == 1 << _capi.WT_SITE_ENCODER                      # the decoder did not fall back
    dkeys = [k for k in m._engine.plans if k[0] == _capi.WT_PLAN_DECODE]
    assert dkeys and all(len(k) < 6 for k in dkeys), dkeys                   # decode plans without an fp32 site mask
    with pytest.raises(_capi.WavTokError, match="fallback"):
        m.check_status()
    parity_log.record(f"range_encoder_site[{name}]", code_flips=flips, frames=int(co.numel()), wav_rel_l2=err)
