"""Import the real reference (read-only at /root/reference) in the build container.

Used ONLY by make_golden.py to validate oracle/cpu_ref.py and emit fixtures; never
at test/bench time (the reference does not exist on the GPU box).

torchaudio is absent from the image; the reference imports it at module top
(decoder/feature_extractors.py:4, decoder/heads.py:3) but never calls it on the
encode/decode path, so an empty placeholder package is registered first.
"""
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def _stub_torchaudio():
    if "torchaudio" in sys.modules:
        return
    ta = types.ModuleType("torchaudio")
    tr = types.ModuleType("torchaudio.transforms")

    class _Placeholder(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    tr.MelSpectrogram = _Placeholder
    tr.Resample = _Placeholder
    fn = types.ModuleType("torchaudio.functional")
    ff = types.ModuleType("torchaudio.functional.functional")
    ff._hz_to_mel = lambda *a, **k: None
    ff._mel_to_hz = lambda *a, **k: None
    fn.functional = ff
    ta.transforms = tr
    ta.functional = fn
    for name, mod in (("torchaudio", ta), ("torchaudio.transforms", tr),
                      ("torchaudio.functional", fn), ("torchaudio.functional.functional", ff)):
        sys.modules[name] = mod


def load_reference_class():
    _stub_torchaudio()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from decoder.pretrained import WavTokenizer  # noqa: E402
    return WavTokenizer


def build_reference(yaml_path: str, state_dict_np):
    """Reference model in eval mode with our synthetic hot-path weights loaded through
    load_state_dict (keys we do not generate keep the reference's own init)."""
    cls = load_reference_class()
    m = cls.from_hparams0802(yaml_path)
    sd = m.state_dict()
    for k, v in state_dict_np.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
        sd[k] = torch.from_numpy(v.copy())
    m.load_state_dict(sd)
    m.eval()
    return m
