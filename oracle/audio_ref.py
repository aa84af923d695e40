"""TEST ORACLE (not shipped on the product path) for the helpers around the hot path.

  linear_overlap_add  — restates encoder/utils.py:17-56 (`_linear_overlap_add`).  PINNED: tests/golden/overlap_add.npz
                        holds outputs of the reference function itself (tests/golden/make_golden_audio.py imports it).
  segmented_round_trip — restates EncodecModel.forward's segment loop (encoder/model.py:122-145 `encode`: frames at offsets
                        range(0, length, stride) of segment_length samples; :167-178 `decode`: every frame decoded, then
                        `_linear_overlap_add(frames, stride)`; :189-191 `forward`: trimmed to the input length) around the
                        WavTokenizer codec path (oracle/cpu_ref.py `encode_infer` / `decode` in place of EncodecModel's
                        `_encode_frame` / `_decode_frame`, which WavTokenizer never calls).  Pinned through its parts: the
                        codec oracle by the reference fixtures, the overlap-add by overlap_add.npz.
  convert_audio       — restates encoder/utils.py:79-92: channel mix, then `torchaudio.transforms.Resample(sr, target_sr)`.
                        PARITY UNPINNED for the resampler: torchaudio (the reference pins no version; requirements.txt
                        lists `torchaudio`) is absent from this image and the reference holds no fixture for it.  The
                        restatement follows torchaudio's published algorithm (functional.resample: sinc_interp_hann,
                        lowpass_filter_width = 6, rolloff = 0.99, gcd-reduced polyphase kernel, conv1d with stride
                        orig_freq, output length ceil(new * length / orig)); tests check it through properties
                        (identity at equal rates, exactness on band-limited tones, linearity, length rule).
  to_pcm16            — restates save_audio's clamp / rescale (encoder/utils.py:95-103) followed by the 16-bit
                        conversion torchaudio.save(..., encoding='PCM_S', bits_per_sample=16) performs in its backend.
                        PARITY UNPINNED for the rounding rule (backend dependent); this oracle and the kernel use
                        round-half-to-even of x * 32768, clipped to [-32768, 32767].
"""
import math

import numpy as np


def linear_overlap_add(frames, stride):
    """frames: list of arrays [..., len_i] (all but the last of equal length) -> [..., total]."""
    assert len(frames)
    dtype = frames[0].dtype
    shape = frames[0].shape[:-1]
    total_size = stride * (len(frames) - 1) + frames[-1].shape[-1]
    frame_length = frames[0].shape[-1]
    # the reference builds the triangle with torch.linspace (whose vectorised CPU kernel is not reproduced bit for
    # bit by a scalar start + i * step), so the oracle calls the same function
    import torch
    t = torch.linspace(0, 1, frame_length + 2, dtype=torch.from_numpy(np.zeros(1, dtype)).dtype)[1:-1].numpy()
    weight = (np.asarray(0.5, dtype) - np.abs(t - np.asarray(0.5, dtype))).astype(dtype)
    sum_weight = np.zeros(total_size, dtype)
    out = np.zeros(shape + (total_size,), dtype)
    offset = 0
    for frame in frames:
        fl = frame.shape[-1]
        out[..., offset:offset + fl] += weight[:fl] * frame
        sum_weight[offset:offset + fl] += weight[:fl]
        offset += stride
    assert sum_weight.min() > 0
    return out / sum_weight


def resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """Polyphase windowed-sinc table [new][2*width + orig] (float64) and width, for gcd-reduced rates."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = t * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base / orig
    with np.errstate(divide="ignore", invalid="ignore"):
        k = np.where(t == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t))
    return k * window * scale, width, orig, new


def convert_audio(wav, sr, target_sr, target_channels=1):
    """wav [..., channels, length] float32 -> [..., target_channels, length'] (encoder/utils.py:79-92)."""
    wav = np.asarray(wav, np.float32)
    assert wav.ndim >= 2 and wav.shape[-2] in (1, 2)
    channels = wav.shape[-2]
    if target_channels == 1:
        wav = wav.mean(-2, keepdims=True, dtype=np.float32)
    elif target_channels == 2:
        wav = np.broadcast_to(wav, wav.shape[:-2] + (2, wav.shape[-1])) if channels == 1 else wav
    else:
        raise RuntimeError(f"Impossible to convert from {channels} to {target_channels}")
    if int(sr) == int(target_sr):
        return wav.copy()
    kern, width, orig, new = resample_kernel(sr, target_sr)
    kern = kern.astype(np.float32)
    length = wav.shape[-1]
    target_length = int(math.ceil(new * length / orig))
    padded = np.pad(wav, [(0, 0)] * (wav.ndim - 1) + [(width, width + orig)])
    K = kern.shape[1]
    nfr = (padded.shape[-1] - K) // orig + 1
    frames = np.lib.stride_tricks.sliding_window_view(padded, K, axis=-1)[..., ::orig, :][..., :nfr, :]
    out = np.einsum("...fk,pk->...fp", frames.astype(np.float32), kern, dtype=np.float32)     # [..., frames, new]
    out = out.reshape(out.shape[:-2] + (nfr * new,))
    return out[..., :target_length]


def to_pcm16(wav, rescale=False, limit=0.99):
    wav = np.asarray(wav, np.float32)
    if rescale:
        mx = np.abs(wav).max()
        wav = wav * np.float32(min(limit / mx, 1.0)) if mx > 0 else wav
    else:
        wav = np.clip(wav, -limit, limit)
    return np.clip(np.rint(wav.astype(np.float32) * np.float32(32768.0)), -32768, 32767).astype(np.int16)


def segmented_round_trip(oracle_model, wav, segment_length, stride, bandwidth_id=None):
    """wav: torch tensor [B, T] -> numpy [B, T]: encoder/model.py:139-145 (frame offsets), :174-178 (decode every frame,
    linear overlap-add) and :191 (trim to the input length), with the WavTokenizer oracle as the codec."""
    import torch
    bw = bandwidth_id if bandwidth_id is not None else torch.tensor([0])
    length = wav.shape[-1]
    frames = []
    with torch.inference_mode():
        for offset in range(0, length, stride):
            frame = wav[..., offset: offset + segment_length]
            feats, _codes = oracle_model.encode_infer(frame, bw)
            frames.append(oracle_model.decode(feats, bw)[..., : frame.shape[-1]].numpy())
    return linear_overlap_add(frames, stride)[..., :length]
