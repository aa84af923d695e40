"""Audio helpers on either side of the hot path, on the GPU (mirrors of the reference's encoder/utils.py).

    convert_audio(wav, sr, target_sr, target_channels)   encoder/utils.py:79-92   (mono target; resample on the GPU)
    to_pcm16(wav, rescale=False)                         encoder/utils.py:95-103 + the PCM_S 16 conversion of torchaudio.save
    linear_overlap_add(frames, stride)                   encoder/utils.py:17-56   (bit-identical)
    segment_offsets(length, segment_length, stride)      encoder/model.py:133-145 (the frame offsets of EncodecModel.encode)

All take and return torch tensors on the GPU; there is no CPU fallback."""
import ctypes
from typing import Dict, List, Sequence, Tuple

import torch

from ._capi import check, lib

_resamplers: Dict[Tuple[int, int, int], ctypes.c_void_p] = {}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _need_gpu(t: torch.Tensor):
    if t.device.type != "cuda":
        raise RuntimeError("wavtokenizer_amd.audio runs on the GPU only (move the tensor with .to('cuda'))")


def convert_audio(wav: torch.Tensor, sr: int, target_sr: int, target_channels: int = 1) -> torch.Tensor:
    assert wav.dim() >= 2, "Audio tensor must have at least 2 dimensions"
    assert wav.shape[-2] in [1, 2], "Audio must be mono or stereo."
    if target_channels != 1:
        raise RuntimeError("only target_channels = 1 is implemented (what every caller of the codec path asks for)")
    _need_gpu(wav)
    *shape, channels, length = wav.shape
    dev = wav.device
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (int(sr), int(target_sr), idx)
    if key not in _resamplers:
        r = ctypes.c_void_p()
        check(lib.wt_resampler_create(int(sr), int(target_sr), idx, ctypes.byref(r)), "wt_resampler_create")
        _resamplers[key] = r
    r = _resamplers[key]
    B = 1
    for s in shape:
        B *= int(s)
    x = wav.to(torch.float32).contiguous().view(max(B, 1), channels, length)
    Tout = int(lib.wt_resampler_out_length(r, length))
    out = torch.empty((max(B, 1), Tout), dtype=torch.float32, device=dev)
    check(lib.wt_convert_audio(r, _ptr(x), max(B, 1), channels, length, _ptr(out), _stream(dev)), "wt_convert_audio")
    return out.view(*shape, 1, Tout)


def to_pcm16(wav: torch.Tensor, rescale: bool = False, limit: float = 0.99) -> torch.Tensor:
    _need_gpu(wav)
    x = wav.to(torch.float32).contiguous()
    out = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    ws = torch.zeros(1, dtype=torch.int32, device=x.device)
    check(lib.wt_pcm16(_ptr(x), x.numel(), float(limit), 1 if rescale else 0, _ptr(out), _ptr(ws), _stream(x.device)), "wt_pcm16")
    return out


def linear_overlap_add(frames: Sequence[torch.Tensor], stride: int) -> torch.Tensor:
    assert len(frames)
    _need_gpu(frames[0])
    dev = frames[0].device
    shape = frames[0].shape[:-1]
    flen, last = int(frames[0].shape[-1]), int(frames[-1].shape[-1])
    rows = 1
    for s in shape:
        rows *= int(s)
    buf = torch.zeros((len(frames), rows, flen), dtype=torch.float32, device=dev)
    for i, f in enumerate(frames):
        buf[i, :, : f.shape[-1]] = f.reshape(rows, -1)
    # the reference's triangle, from the same torch.linspace
    t = torch.linspace(0, 1, flen + 2, dtype=torch.float32)[1:-1]
    weight = (0.5 - (t - 0.5).abs()).to(dev)
    total = stride * (len(frames) - 1) + last
    out = torch.empty((rows, total), dtype=torch.float32, device=dev)
    check(lib.wt_linear_overlap_add(_ptr(buf), _ptr(weight), len(frames), rows, flen, last, int(stride), _ptr(out), _stream(dev)),
          "wt_linear_overlap_add")
    return out.view(*shape, total)


def segment_offsets(length: int, segment_length: int, stride: int) -> List[int]:
    """Offsets of the frames EncodecModel.encode cuts (encoder/model.py:139-145)."""
    return list(range(0, length, stride))


def segmented_round_trip(model, wav: torch.Tensor, segment_length: int, stride: int, bandwidth_id=None) -> torch.Tensor:
    """encode_infer + decode of a long clip in overlapping segments, cross-faded with the reference's linear
    overlap-add (EncodecModel.encode / .decode, encoder/model.py:122-190): wav [B, T] -> [B, T]."""
    _need_gpu(wav)
    bw = bandwidth_id if bandwidth_id is not None else torch.tensor([0])
    T = wav.shape[-1]
    outs = []
    for off in segment_offsets(T, segment_length, stride):
        frame = wav[..., off: off + segment_length]
        feats, _codes = model.encode_infer(frame, bandwidth_id=bw)
        outs.append(model.decode(feats, bandwidth_id=bw)[..., : frame.shape[-1]])
    return linear_overlap_add(outs, stride)[..., :T]
