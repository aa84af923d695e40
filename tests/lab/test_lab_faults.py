"""Fault-injection tests.  They need the LAB build of the library (make -C wavtokenizer_amd/csrc lab ->
tools/lib/libwavtok_hip_lab.so): the product library carries no fault hook (WT_LSTM_PERSIST_FAULT is read by lab_env(), a
constant in product builds).  tests/test_gpu_parity.py::test_fault_injection_on_the_lab_library runs this module ONCE in a
child process with WAVTOK_HIP_LIB pointing at the LAB build; collected in any other process the tests skip themselves."""
import numpy as np
import pytest
import torch

from tests.util import synth_state_dict

pytestmark = pytest.mark.gpu
BW = torch.tensor([0])


@pytest.fixture(autouse=True)
def _needs_lab_library():
    from wavtokenizer_amd import _capi
    if b"LAB build" not in _capi.lib.wt_version():
        pytest.skip("needs WAVTOK_HIP_LIB=tools/lib/libwavtok_hip_lab.so (run through test_fault_injection_on_the_lab_library)")


@pytest.fixture(scope="module", params=["hop600", "hop320"])
def gpu_model(request):
    from wavtokenizer_amd import _capi
    if b"LAB build" not in _capi.lib.wt_version():
        return request.param, None, None
    sd = synth_state_dict(request.param)
    return request.param, _fresh_model(request.param, sd), sd


def _fresh_model(name, sd):
    from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS
    m = WavTokenizer.from_arch(NAMED_ARCHS[name])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return m.eval().to("cuda")


def test_persistent_lstm_lost_coresidency(gpu_model):
    """A persistent LSTM launch that cannot get its 32 workgroups per XCD resident (forced here: WT_LSTM_PERSIST_FAULT=1
    launches eight too few and shortens the spin bound) must fail loudly on the call that failed: codes = -1, features
    NaN, status bit set; the retry runs the launch-per-step kernel and gives the step kernel's codes."""
    import ctypes
    import os
    from wavtokenizer_amd import _capi, synth
    name, _shared, sd = gpu_model
    m = _fresh_model(name, sd)              # the fallback is sticky for a model: not on the module's shared one
    wav = torch.from_numpy(synth.make_clips(20, 7200, seed=520)).cuda()
    m.set_lstm_mode("step")
    try:
        f_ref, c_ref = m.encode_infer(wav, bandwidth_id=BW)
    finally:
        m.set_lstm_mode("persistent")
    # drop cached plans so that the encode plan below is fresh (persistent)
    m._engine.drop(lambda k: k[0] == _capi.WT_PLAN_ENCODE and k[1] == 20 and not (k[3] & _capi.WT_PLAN_FLAG_STEP_LSTM))
    os.environ["WT_LSTM_PERSIST_FAULT"] = "1"
    try:
        f1, c1 = m.encode_infer(wav, bandwidth_id=BW)
        torch.cuda.synchronize()
    finally:
        del os.environ["WT_LSTM_PERSIST_FAULT"]
    assert int(c1.max()) == -1 and int(c1.min()) == -1, "a failed call must not hand out plausible codes"
    assert torch.isnan(f1).all()
    plan = m._engine.plans[(_capi.WT_PLAN_ENCODE, 20, 7200, m._graph_flags(20))][0]
    bits = ctypes.c_int32()
    _capi.check(_capi.lib.wt_plan_status(plan, ctypes.byref(bits), 0), "wt_plan_status")
    assert bits.value & _capi.WT_STATUS_BIT_LSTM
    f2, c2 = m.encode_infer(wav, bandwidth_id=BW)          # WT_ERR_LSTM_SYNC inside -> step kernel -> runs
    assert torch.equal(c2, c_ref) and torch.equal(f2, f_ref)
    with pytest.raises(_capi.WavTokError, match="fallback"):
        m.check_status()                                   # the answered failure is still reported once
    m.check_status()
    # strict mode repeats the failing call itself (a fresh model: the first one runs the step LSTM for good now)
    m = _fresh_model(name, sd)
    m.set_strict_status(True)
    os.environ["WT_LSTM_PERSIST_FAULT"] = "1"
    try:
        f3, c3 = m.encode_infer(wav, bandwidth_id=BW)
    finally:
        del os.environ["WT_LSTM_PERSIST_FAULT"]
        m.set_strict_status(False)
        m._engine.drop(lambda k: k[0] == _capi.WT_PLAN_ENCODE)
    assert torch.equal(c3, c_ref) and torch.equal(f3, f_ref)


def test_device_failure_reaches_the_next_call_on_another_plan(gpu_model):
    """A file-by-file caller (infer.py: one clip per call, a new length and so a new plan per file) never uses a plan twice:
    the failure of one call must surface on the NEXT call on the model whatever its shape, and the fallback must stick
    for the whole model.  First call (length A): persistent LSTM forced to fail -> poisoned outputs.  Second call
    (length B, a different plan): returns correct results from the step kernel without another timeout, and a third
    plan (length C) does not launch the persistent kernel again."""
    import os
    import time
    from wavtokenizer_amd import synth, _capi
    name, _m, sd = gpu_model
    m = _fresh_model(name, sd)
    m.set_graph_max_clips(0)
    A, Bl, C = 7200, 9000, 10100
    wa, wb, wc = (torch.from_numpy(synth.make_clips(3, T, seed=620 + T)).cuda() for T in (A, Bl, C))
    m.set_lstm_mode("step")
    try:
        refs = [m.encode_infer(w, bandwidth_id=BW) for w in (wb, wc)]
    finally:
        m.set_lstm_mode("persistent")
    m._engine.drop(lambda k: True)
    os.environ["WT_LSTM_PERSIST_FAULT"] = "1"
    try:
        f1, c1 = m.encode_infer(wa, bandwidth_id=BW)
        torch.cuda.synchronize()
        assert int(c1.max()) == -1 and torch.isnan(f1).all()
        # the fault hook is still armed: a second persistent launch would time out again and poison this call too
        t0 = time.perf_counter()
        f2, c2 = m.encode_infer(wb, bandwidth_id=BW)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        f3, c3 = m.encode_infer(wc, bandwidth_id=BW)
        torch.cuda.synchronize()
    finally:
        del os.environ["WT_LSTM_PERSIST_FAULT"]
    assert torch.equal(c2, refs[0][1]) and torch.equal(f2, refs[0][0]), "the next call on another plan must fall back and be correct"
    assert torch.equal(c3, refs[1][1]) and torch.equal(f3, refs[1][0])
    assert dt2 < 5.0
    assert len(m.fallback_events) == 1                  # one failure, answered once, for the whole model
    with pytest.raises(_capi.WavTokError, match="fallback"):
        m.check_status()                                # reported once (the first call's poisoned outputs were handed out) ...
    m.check_status()                                    # ... and nothing is left pending: plan A's stale word does not come back




def test_graph_replays_resume_after_the_lstm_fallback(gpu_model):
    """after a lost-co-residency fallback a graph plan (B <= 16) records the launch-per-step LSTM; that recording
    must be kept and replayed, not destroyed and re-captured on every other call (graph_persist used to be set from
    uses_persist alone)."""
    import ctypes
    import os
    from wavtokenizer_amd import _capi, synth
    name, _m, sd = gpu_model
    m = _fresh_model(name, sd)
    wav = torch.from_numpy(synth.make_clips(2, 7200, seed=733)).cuda()
    m.set_lstm_mode("step")
    try:
        f_ref, c_ref = m.encode_infer(wav, bandwidth_id=BW)
    finally:
        m.set_lstm_mode("persistent")
    m._engine.drop(lambda k: True)
    m.set_strict_status(False)          # (the default for two clips is strict: the failed call would be repeated at once)
    os.environ["WT_LSTM_PERSIST_FAULT"] = "1"
    try:
        f1, c1 = m.encode_infer(wav, bandwidth_id=BW)          # eager first call of a new plan: the forced fault
        torch.cuda.synchronize()
    finally:
        del os.environ["WT_LSTM_PERSIST_FAULT"]
    assert int(c1.max()) == -1
    key = (_capi.WT_PLAN_ENCODE, 2, 7200, m._graph_flags(2))
    replays = []
    for _ in range(8):
        f, c = m.encode_infer(wav, bandwidth_id=BW)
        torch.cuda.synchronize()
        assert torch.equal(c, c_ref) and torch.equal(f, f_ref)
        replays.append(int(_capi.lib.wt_plan_graph_replays(m._engine.plans[key][0])))
    assert replays[-1] >= 5 and replays[-1] > replays[2], f"the recorded graph is not being replayed after the fallback: {replays}"
    assert not _capi.lib.wt_model_persistent_lstm(m._engine.model)
    with pytest.raises(_capi.WavTokError, match="fallback"):
        m.check_status()


def test_small_batches_never_hand_out_poisoned_tensors_by_default(gpu_model):
    """VERDICT r03 #3c: with the default (automatic) strict status an infer.py-style caller (one clip per call) gets the
    CORRECT result from the very call whose persistent LSTM failed: the class synchronises, sees the status bit, falls back
    and repeats the call before returning."""
    import os
    from wavtokenizer_amd import synth
    name, _m, sd = gpu_model
    m = _fresh_model(name, sd)
    wav = torch.from_numpy(synth.make_clips(1, 9100, seed=734)).cuda()
    m.set_lstm_mode("step")
    try:
        f_ref, c_ref = m.encode_infer(wav, bandwidth_id=BW)
    finally:
        m.set_lstm_mode("persistent")
    m._engine.drop(lambda k: True)
    os.environ["WT_LSTM_PERSIST_FAULT"] = "1"
    try:
        f1, c1 = m.encode_infer(wav, bandwidth_id=BW)
    finally:
        del os.environ["WT_LSTM_PERSIST_FAULT"]
    assert torch.equal(c1, c_ref) and torch.equal(f1, f_ref)
    assert not m.persistent_lstm
    with pytest.raises(Exception, match="fallback"):
        m.check_status()
