"""Deterministic synthetic weights and clips (numpy only, no torch RNG).

There is no trained checkpoint offline, so parity fixtures, tests and
``bench.py`` run on random-init weights of the real architecture.  Every tensor
is generated from a Philox counter stream keyed by a hash of its state-dict key,
using integer arithmetic only (an Irwin-Hall sum of four 16-bit uniforms), so
the container that wrote the golden fixtures and the GPU box regenerate
bit-identical bytes whatever their libm / SIMD level.  ``weights_manifest``
gives a SHA-256 per tensor; the fixtures pin it.

State-dict keys and shapes follow the reference checkpoint layout
(decoder/pretrained.py:103-105 keeps the
``feature_extractor.`` / ``backbone.`` / ``head.`` prefixes).
"""
from __future__ import annotations

import hashlib
from typing import Dict, List, Tuple

import numpy as np

from .config import ArchConfig, SEANET_DIMENSION, SEANET_N_FILTERS

ENC = "feature_extractor.encodec.encoder.model."
DEC = "feature_extractor.encodec.decoder.model."
VQ = "feature_extractor.encodec.quantizer.vq.layers.0._codebook."


# --------------------------------------------------------------------------- RNG
def _stream(key: str, n: int, seed: int) -> np.ndarray:
    """n raw uint64 from a Philox stream keyed by (key, seed)."""
    h = hashlib.sha256(f"{seed}:{key}".encode()).digest()
    k = np.frombuffer(h[:16], dtype=np.uint64).copy()
    return np.random.Philox(key=k).random_raw(n)


def _approx_normal(key: str, n: int, seed: int) -> np.ndarray:
    """Unit-variance, zero-mean, bell-shaped float64; exact integer arithmetic."""
    raw = _stream(key, n, seed)
    s = ((raw & 0xFFFF) + ((raw >> 16) & 0xFFFF) + ((raw >> 32) & 0xFFFF) + ((raw >> 48) & 0xFFFF))
    # sum of 4 U{0..65535}: mean 2*65535, var 4*(65536^2-1)/12
    return (s.astype(np.int64) - 2 * 65535).astype(np.float64) * (1.0 / 37837.22)


def _uniform(key: str, n: int, seed: int) -> np.ndarray:
    """U[0,1) float64 with 32 bits."""
    raw = _stream(key, n, seed)
    return (raw >> 32).astype(np.float64) * (1.0 / 4294967296.0)


def normal(key: str, shape, std: float, seed: int, mean: float = 0.0) -> np.ndarray:
    n = int(np.prod(shape))
    return (mean + std * _approx_normal(key, n, seed)).astype(np.float32).reshape(shape)


# ------------------------------------------------------------------ key/shape spec
def encoder_conv_specs(arch: ArchConfig) -> List[Tuple[str, int, int, int]]:
    """(key prefix, Cout, Cin, k) for every weight-normed encoder conv."""
    specs = []
    nf = SEANET_N_FILTERS
    specs.append((ENC + "0.conv.conv", nf, 1, 7))
    idx = 1
    mult = 1
    for r in arch.enc_ratios:
        c = mult * nf
        specs.append((ENC + f"{idx}.block.1.conv.conv", c // 2, c, 3))
        specs.append((ENC + f"{idx}.block.3.conv.conv", c, c // 2, 1))
        specs.append((ENC + f"{idx}.shortcut.conv.conv", c, c, 1))
        specs.append((ENC + f"{idx + 2}.conv.conv", 2 * c, c, 2 * r))
        idx += 3
        mult *= 2
    # idx -> SLSTM, idx+1 -> ELU, idx+2 -> final conv
    specs.append((ENC + f"{idx + 2}.conv.conv", SEANET_DIMENSION, mult * nf, 7))
    return specs


def lstm_index(arch: ArchConfig) -> int:
    return 1 + 3 * len(arch.ratios)


def seanet_decoder_specs(arch: ArchConfig):
    """SEANetDecoder layout (encoder/modules/seanet.py:147-238): list of
    (kind, key prefix, a, b, k) with kind in {'conv','convtr'}; for convtr the
    weight tensor is (Cin, Cout, k) and weight_g is per input channel."""
    nf = SEANET_N_FILTERS
    n = len(arch.ratios)
    mult = 2 ** n
    out = [("conv", DEC + "0.conv.conv", mult * nf, SEANET_DIMENSION, 7)]
    # index 1 = SLSTM
    idx = 2
    for r in arch.ratios:
        c = mult * nf
        out.append(("convtr", DEC + f"{idx + 1}.convtr.convtr", c, c // 2, 2 * r))
        h = c // 2
        out.append(("conv", DEC + f"{idx + 2}.block.1.conv.conv", h // 2, h, 3))
        out.append(("conv", DEC + f"{idx + 2}.block.3.conv.conv", h, h // 2, 1))
        out.append(("conv", DEC + f"{idx + 2}.shortcut.conv.conv", h, h, 1))
        idx += 3
        mult //= 2
    out.append(("conv", DEC + f"{idx + 1}.conv.conv", 1, nf, 7))
    return out


def make_state_dict(arch: ArchConfig, seed: int = 0, with_seanet_decoder: bool = False) -> Dict[str, np.ndarray]:
    """Synthetic hot-path weights as float32 numpy arrays keyed like a checkpoint."""
    sd: Dict[str, np.ndarray] = {}

    def wn_conv(prefix, cout, cin, k, gain):
        v = normal(prefix + ".weight_v", (cout, cin, k), 1.0 / np.sqrt(cin * k), seed)
        vn = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(1, 2)))
        jitter = 0.8 + 0.4 * _uniform(prefix + ".weight_g", cout, seed)
        sd[prefix + ".weight_g"] = (gain * vn * jitter).astype(np.float32).reshape(cout, 1, 1)
        sd[prefix + ".weight_v"] = v
        sd[prefix + ".bias"] = normal(prefix + ".bias", (cout,), 0.05, seed)

    for prefix, cout, cin, k in encoder_conv_specs(arch):
        # gains picked so activations stay O(1) through the stack (first conv lifts the
        # ~0.15-rms clip; residual branches add, so they get < 1)
        if cin == 1:
            gain = 3.0
        elif ".block." in prefix or ".shortcut." in prefix:
            gain = 0.85
        else:
            gain = 1.25
        wn_conv(prefix, cout, cin, k, gain=gain)

    li = lstm_index(arch)
    H = SEANET_DIMENSION
    for layer in range(2):
        for nm in ("weight_ih", "weight_hh"):
            key = ENC + f"{li}.lstm.{nm}_l{layer}"
            sd[key] = normal(key, (4 * H, H), 2.0 / np.sqrt(H) / np.sqrt(3.0), seed)
        for nm in ("bias_ih", "bias_hh"):
            key = ENC + f"{li}.lstm.{nm}_l{layer}"
            sd[key] = normal(key, (4 * H,), 0.1, seed)

    # codebook: N(0, sigma_e^2); sigma_e chosen near the encoder-output std for this recipe
    sd[VQ + "embed"] = normal(VQ + "embed", (arch.vq_bins, H), 0.6, seed)
    sd[VQ + "inited"] = np.ones((1,), np.float32)
    sd[VQ + "cluster_size"] = np.ones((arch.vq_bins,), np.float32)
    sd[VQ + "embed_avg"] = sd[VQ + "embed"].copy()
    for q in range(1, arch.num_quantizers):        # further codebooks of a checkpoint with num_quantizers > 1 (codes_to_features sums them)
        vq = VQ.replace("layers.0.", f"layers.{q}.")
        sd[vq + "embed"] = normal(vq + "embed", (arch.vq_bins, H), 0.6 / (q + 1), seed)
        sd[vq + "inited"] = np.ones((1,), np.float32)
        sd[vq + "cluster_size"] = np.ones((arch.vq_bins,), np.float32)
        sd[vq + "embed_avg"] = sd[vq + "embed"].copy()

    if with_seanet_decoder:
        for kind, prefix, a, b, k in seanet_decoder_specs(arch):
            if kind == "conv":
                wn_conv(prefix, a, b, k, gain=1.5)
            else:  # convtr: weight (Cin=a, Cout=b, k); weight_norm dim=0 -> g per input channel
                v = normal(prefix + ".weight_v", (a, b, k), 1.0 / np.sqrt(a * k), seed)
                vn = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(1, 2)))
                jitter = 0.8 + 0.4 * _uniform(prefix + ".weight_g", a, seed)
                sd[prefix + ".weight_g"] = (1.5 * vn * jitter).astype(np.float32).reshape(a, 1, 1)
                sd[prefix + ".weight_v"] = v
                sd[prefix + ".bias"] = normal(prefix + ".bias", (b,), 0.05, seed)
        for layer in range(2):
            for nm in ("weight_ih", "weight_hh"):
                key = DEC + f"1.lstm.{nm}_l{layer}"
                sd[key] = normal(key, (4 * H, H), 2.0 / np.sqrt(H) / np.sqrt(3.0), seed)
            for nm in ("bias_ih", "bias_hh"):
                key = DEC + f"1.lstm.{nm}_l{layer}"
                sd[key] = normal(key, (4 * H,), 0.1, seed)

    # ---- backbone (decoder/models.py:166-216)
    D, I, C = arch.dim, arch.intermediate_dim, arch.input_channels
    A = arch.adanorm_num_embeddings

    def dense(key, shape, fan_in, gain=1.0, bias_std=0.05):
        sd[key + ".weight"] = normal(key + ".weight", shape, gain / np.sqrt(fan_in), seed)
        sd[key + ".bias"] = normal(key + ".bias", (shape[0],), bias_std, seed)

    def affine(key, n):
        sd[key + ".weight"] = normal(key + ".weight", (n,), 0.1, seed, mean=1.0)
        sd[key + ".bias"] = normal(key + ".bias", (n,), 0.1, seed)

    def adanorm(key):
        sd[key + ".scale.weight"] = normal(key + ".scale.weight", (A, D), 0.1, seed, mean=1.0)
        sd[key + ".shift.weight"] = normal(key + ".shift.weight", (A, D), 0.1, seed)

    dense("backbone.embed", (D, C, 7), C * 7)
    adanorm("backbone.norm")
    for i in range(arch.num_layers):
        p = f"backbone.convnext.{i}"
        dense(p + ".dwconv", (D, 1, 7), 7)
        adanorm(p + ".norm")
        dense(p + ".pwconv1", (I, D), D, gain=1.4)
        dense(p + ".pwconv2", (D, I), I, gain=1.4)
        sd[p + ".gamma"] = (0.1 + 0.3 * _uniform(p + ".gamma", D, seed)).astype(np.float32)
    affine("backbone.final_layer_norm", D)
    for i in (0, 1, 3, 4):
        p = f"backbone.pos_net.{i}"
        affine(p + ".norm1", D)
        dense(p + ".conv1", (D, D, 3), D * 3, gain=1.4)
        affine(p + ".norm2", D)
        dense(p + ".conv2", (D, D, 3), D * 3, gain=1.0)
    p = "backbone.pos_net.2"
    affine(p + ".norm", D)
    for nm in ("q", "k", "v", "proj_out"):
        # q/k gain > 1 so softmax rows are peaked, not uniform
        dense(p + "." + nm, (D, D, 1), D, gain=2.0 if nm in ("q", "k") else 1.0)
    affine("backbone.pos_net.5", D)

    # ---- head (decoder/heads.py:36-40): rows [0, n_fft/2+1) = log-magnitude, rest = phase
    nb = arch.n_fft // 2 + 1
    w = normal("head.out.weight", (arch.n_fft + 2, D), 1.0 / np.sqrt(D), seed)
    w[:nb] *= 2.0   # log-mag std ~2: ~0.5 % of bins pass the clip(max=1e2) at exp(4.6)
    w[nb:] *= 2.0   # phases spread over several periods
    sd["head.out.weight"] = w
    b = normal("head.out.bias", (arch.n_fft + 2,), 0.1, seed)
    b[:nb] -= 0.5
    sd["head.out.bias"] = b
    n = np.arange(arch.n_fft, dtype=np.float64)
    # periodic Hann, as torch.hann_window(win_length) (decoder/spectral_ops.py:30)
    sd["head.istft.window"] = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / arch.n_fft)).astype(np.float32)
    return sd


def make_trained_like_state_dict(arch: ArchConfig, seed: int = 7, with_seanet_decoder: bool = False) -> Dict[str, np.ndarray]:
    """A second synthetic recipe whose STATISTICS resemble a trained checkpoint rather than an init (no trained weights
    exist offline): starting from make_state_dict, every tensor class gets the spread that training produces and that the
    split-f16 (S32) arithmetic has to survive:
      * weight_g of the weight-normed convs: log-normal per channel (sigma 0.7: some channels 4x up, some 4x down)
      * conv / linear weight matrices: heavy-tailed (element-wise log-normal factor, sigma 0.5, plus 0.1 % outliers x8),
        each output row renormalised to its former L2 norm so the activations stay O(1)
      * ConvNeXt layer scale gamma: log-normal around 0.3 with a tail up to 10
      * LayerNorm / AdaLayerNorm / GroupNorm scales: log-normal, sigma 0.5 (0.25 .. 4); shifts and biases 3x larger
      * codebook rows: per-row log-normal radius (sigma 0.3)
    tests/golden/make_golden_trained_like.py pins the reference's outputs on it; tests/test_gpu_parity.py compares."""
    sd = make_state_dict(arch, seed=seed, with_seanet_decoder=with_seanet_decoder)

    def lognorm(key, n, sigma):
        return np.exp(sigma * _approx_normal(key + "#ln", n, seed))

    for k in list(sd.keys()):
        v = sd[k]
        if k.endswith(".weight_g"):
            sd[k] = (v.astype(np.float64) * lognorm(k, v.size, 0.7).reshape(v.shape)).astype(np.float32)
        elif k.endswith(".gamma"):
            g = 0.3 * lognorm(k, v.size, 1.2)
            sd[k] = np.minimum(g, 10.0).astype(np.float32)
        elif (k.endswith(".weight_v") or k.endswith(".weight")) and v.ndim >= 2 and v.shape[0] > 4 and "norm" not in k and "lstm" not in k:
            w = v.astype(np.float64)
            rows = w.reshape(w.shape[0], -1)
            n0 = np.sqrt((rows ** 2).sum(axis=1, keepdims=True))
            f = lognorm(k, rows.size, 0.5).reshape(rows.shape)
            out_mask = _uniform(k + "#out", rows.size, seed).reshape(rows.shape) < 1e-3
            rows2 = rows * f * np.where(out_mask, 8.0, 1.0)
            n1 = np.sqrt((rows2 ** 2).sum(axis=1, keepdims=True))
            sd[k] = (rows2 * (n0 / np.maximum(n1, 1e-30))).reshape(w.shape).astype(np.float32)
        elif ("norm" in k or k.startswith("backbone.pos_net.5")) and k.endswith(".weight"):
            # GroupNorm / LayerNorm weights (n,) and AdaLayerNorm scale/shift tables (A, D)
            if ".shift." in k:
                sd[k] = (3.0 * v).astype(np.float32)
            else:
                sd[k] = (np.sign(v) * np.abs(v.astype(np.float64)) * lognorm(k, v.size, 0.5).reshape(v.shape)).astype(np.float32)
        elif k.endswith(".bias") and "lstm" not in k and "head.out" not in k:
            sd[k] = (3.0 * v).astype(np.float32)
        elif k.endswith("_codebook.embed"):
            r = lognorm(k, v.shape[0], 0.3).reshape(-1, 1)
            sd[k] = (v.astype(np.float64) * r).astype(np.float32)
    sd[VQ + "embed_avg"] = sd[VQ + "embed"].copy()
    return sd


def weights_manifest(sd: Dict[str, np.ndarray]) -> Dict[str, Dict]:
    return {k: {"shape": list(v.shape), "sha256": hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()}
            for k, v in sorted(sd.items())}


# ---------------------------------------------------------------------- input clips
def make_clips(n_clips: int, T: int, seed: int, sample_rate: int = 24000) -> np.ndarray:
    """AM/FM tone + noise clips, float32 in about [-0.5, 0.5].
    Uses float64 sin; fixtures commit the arrays they pin, so last-ulp libm
    differences between machines cannot matter."""
    out = np.empty((n_clips, T), np.float32)
    t = np.arange(T, dtype=np.float64) / sample_rate
    for i in range(n_clips):
        u = _uniform(f"clip{i}", 4, seed)
        f0 = 100.0 + 300.0 * u[0]
        am_f = 1.0 + 3.0 * u[1]
        ph = 2 * np.pi * u[2]
        am = 0.5 * (1.0 + np.sin(2 * np.pi * am_f * t + ph))
        tone = np.sin(2 * np.pi * f0 * t * (1.0 + 0.2 * np.sin(2 * np.pi * 0.7 * t)) + 2 * np.pi * u[3])
        noise = _approx_normal(f"clipnoise{i}", T, seed)
        out[i] = (0.3 * am * tone + 0.05 * noise).astype(np.float32)
    return out
