"""The reference model's full ``state_dict`` layout (key -> shape), derived from the
architecture alone.  ``tests/golden/manifest.json`` holds the key/shape
list dumped from the real reference; ``tests/test_host_logic.py`` checks this module against it,
so ``load_state_dict(strict=True)`` accepts exactly what the reference accepts.
"""
from __future__ import annotations

from typing import Dict, Tuple

from .config import ArchConfig, SEANET_DIMENSION
from .synth import DEC, ENC, VQ, encoder_conv_specs, lstm_index, seanet_decoder_specs

Shape = Tuple[int, ...]


def _lstm(spec: Dict[str, Shape], prefix: str, H: int):
    # nn.LSTM parameter order: weight_ih, weight_hh, bias_ih, bias_hh per layer
    for layer in range(2):
        spec[f"{prefix}.lstm.weight_ih_l{layer}"] = (4 * H, H)
        spec[f"{prefix}.lstm.weight_hh_l{layer}"] = (4 * H, H)
        spec[f"{prefix}.lstm.bias_ih_l{layer}"] = (4 * H,)
        spec[f"{prefix}.lstm.bias_hh_l{layer}"] = (4 * H,)


def full_state_spec(arch: ArchConfig) -> Dict[str, Shape]:
    H = SEANET_DIMENSION
    spec: Dict[str, Shape] = {}
    # ---- SEANetEncoder; weight_norm (old API) registers weight_g, weight_v after bias
    enc = encoder_conv_specs(arch)
    li = lstm_index(arch)
    for prefix, cout, cin, k in enc[:-1]:
        spec[prefix + ".bias"] = (cout,)
        spec[prefix + ".weight_g"] = (cout, 1, 1)
        spec[prefix + ".weight_v"] = (cout, cin, k)
    _lstm(spec, ENC + str(li), H)
    prefix, cout, cin, k = enc[-1]
    spec[prefix + ".bias"] = (cout,)
    spec[prefix + ".weight_g"] = (cout, 1, 1)
    spec[prefix + ".weight_v"] = (cout, cin, k)
    # ---- SEANetDecoder
    sdec = seanet_decoder_specs(arch)
    first = True
    for kind, prefix, a, b, k in sdec:
        if kind == "conv":
            spec[prefix + ".bias"] = (a,)
            spec[prefix + ".weight_g"] = (a, 1, 1)
            spec[prefix + ".weight_v"] = (a, b, k)
        else:
            spec[prefix + ".bias"] = (b,)
            spec[prefix + ".weight_g"] = (a, 1, 1)
            spec[prefix + ".weight_v"] = (a, b, k)
        if first:
            _lstm(spec, DEC + "1", H)
            first = False
    # ---- quantizer buffers (core_vq.py:135-138), one layer per quantizer
    for q in range(arch.num_quantizers):
        p = VQ.replace("layers.0.", f"layers.{q}.")
        spec[p + "inited"] = (1,)
        spec[p + "cluster_size"] = (arch.vq_bins,)
        spec[p + "embed"] = (arch.vq_bins, H)
        spec[p + "embed_avg"] = (arch.vq_bins, H)
    # ---- backbone
    D, I, C, A = arch.dim, arch.intermediate_dim, arch.input_channels, arch.adanorm_num_embeddings
    spec["backbone.embed.weight"] = (D, C, 7)
    spec["backbone.embed.bias"] = (D,)
    spec["backbone.norm.scale.weight"] = (A, D)
    spec["backbone.norm.shift.weight"] = (A, D)
    for i in range(arch.num_layers):
        p = f"backbone.convnext.{i}"
        spec[p + ".gamma"] = (D,)
        spec[p + ".dwconv.weight"] = (D, 1, 7)
        spec[p + ".dwconv.bias"] = (D,)
        spec[p + ".norm.scale.weight"] = (A, D)
        spec[p + ".norm.shift.weight"] = (A, D)
        spec[p + ".pwconv1.weight"] = (I, D)
        spec[p + ".pwconv1.bias"] = (I,)
        spec[p + ".pwconv2.weight"] = (D, I)
        spec[p + ".pwconv2.bias"] = (D,)
    spec["backbone.final_layer_norm.weight"] = (D,)
    spec["backbone.final_layer_norm.bias"] = (D,)
    for i in (0, 1, 3, 4):
        p = f"backbone.pos_net.{i}"
        for nm in ("norm1", "conv1", "norm2", "conv2"):
            if nm.startswith("norm"):
                spec[f"{p}.{nm}.weight"] = (D,)
            else:
                spec[f"{p}.{nm}.weight"] = (D, D, 3)
            spec[f"{p}.{nm}.bias"] = (D,)
        if i == 1:
            q = "backbone.pos_net.2"
            spec[q + ".norm.weight"] = (D,)
            spec[q + ".norm.bias"] = (D,)
            for nm in ("q", "k", "v", "proj_out"):
                spec[f"{q}.{nm}.weight"] = (D, D, 1)
                spec[f"{q}.{nm}.bias"] = (D,)
    spec["backbone.pos_net.5.weight"] = (D,)
    spec["backbone.pos_net.5.bias"] = (D,)
    # ---- head
    spec["head.out.weight"] = (arch.n_fft + 2, D)
    spec["head.out.bias"] = (arch.n_fft + 2,)
    spec["head.istft.window"] = (arch.n_fft,)
    return spec


# buffers (not parameters) in the reference module tree
BUFFER_SUFFIXES = ("_codebook.inited", "_codebook.cluster_size", "_codebook.embed", "_codebook.embed_avg",
                   "istft.window")


def is_buffer(key: str) -> bool:
    return key.endswith(BUFFER_SUFFIXES)
