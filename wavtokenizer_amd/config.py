"""Architecture description for the WavTokenizer encode/decode path.

The reference builds its three sub-modules from a Lightning-CLI YAML
(``decoder/pretrained.py:81-92`` ``from_hparams0802`` reads
``config['model']['init_args'][{feature_extractor,backbone,head}]``).  This
module parses the same keys into a flat :class:`ArchConfig` that sizes the HIP
kernels; nothing is instantiated by class path.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Any, Dict, List, Sequence

import yaml

# Fixed SEANet hyper-parameters (decoder/feature_extractors.py:71-75).
SEANET_DIMENSION = 512
SEANET_N_FILTERS = 32
SEANET_KERNEL = 7
SEANET_RES_KERNEL = 3
SEANET_LAST_KERNEL = 7
SEANET_LSTM_LAYERS = 2
SAMPLE_RATE = 24000


@dataclasses.dataclass(frozen=True)
class ArchConfig:
    """Everything the kernels need to know about one WavTokenizer variant."""

    ratios: Sequence[int]              # `dowmsamples` as written in the YAML (decoder order)
    vq_bins: int = 4096
    num_quantizers: int = 1
    bandwidths: Sequence[float] = (6.6, 6.6, 6.6, 6.6)
    input_channels: int = 512
    dim: int = 768
    intermediate_dim: int = 2304
    num_layers: int = 12
    adanorm_num_embeddings: int = 4
    n_fft: int = 2400
    hop_length: int = 600
    padding: str = "same"

    @property
    def enc_ratios(self) -> List[int]:
        """Encoder uses the ratios reversed (encoder/modules/seanet.py:100)."""
        return list(reversed(list(self.ratios)))

    @property
    def hop(self) -> int:
        return int(math.prod(self.ratios))

    def frames(self, T: int) -> int:
        """L = ceil(T / hop): every SConv1d pads so the last window is full
        (encoder/modules/conv.py:54-61)."""
        return -(-T // self.hop)

    def to_yaml_node(self) -> Dict[str, Any]:
        """The `model.init_args` sub-tree of a Lightning-CLI YAML that arch_from_yaml_dict parses back into this
        architecture (the keys decoder/pretrained.py:81-92 reads)."""
        return {
            "feature_extractor": {"class_path": "decoder.feature_extractors.EncodecFeatures",
                                  "init_args": {"encodec_model": "encodec_24khz", "bandwidths": list(self.bandwidths),
                                                "dowmsamples": list(self.ratios), "vq_bins": self.vq_bins,
                                                "num_quantizers": self.num_quantizers}},
            "backbone": {"class_path": "decoder.models.VocosBackbone",
                         "init_args": {"input_channels": self.input_channels, "dim": self.dim,
                                       "intermediate_dim": self.intermediate_dim, "num_layers": self.num_layers,
                                       "adanorm_num_embeddings": self.adanorm_num_embeddings}},
            "head": {"class_path": "decoder.heads.ISTFTHead",
                     "init_args": {"dim": self.dim, "n_fft": self.n_fft, "hop_length": self.hop_length, "padding": self.padding}},
        }

    def to_dict(self) -> Dict[str, Any]:
        d = dataclasses.asdict(self)
        d["ratios"] = list(self.ratios)
        d["bandwidths"] = list(self.bandwidths)
        return d


def _init_args(node: Dict[str, Any], expect_suffix: str) -> Dict[str, Any]:
    cp = node.get("class_path", "")
    if not cp.endswith(expect_suffix):
        raise ValueError(
            f"unsupported class_path {cp!r}: this build implements only {expect_suffix} "
            "(the classes every YAML under the reference's configs/ selects)")
    return node.get("init_args", {}) or {}


def arch_from_yaml_dict(config: Dict[str, Any]) -> ArchConfig:
    ia = config["model"]["init_args"]
    fe = _init_args(ia["feature_extractor"], "EncodecFeatures")
    bb = _init_args(ia["backbone"], "VocosBackbone")
    hd = _init_args(ia["head"], "ISTFTHead")
    if fe.get("encodec_model", "encodec_24khz") != "encodec_24khz":
        # decoder/feature_extractors.py:87-90
        raise ValueError(f"Unsupported encodec_model: {fe.get('encodec_model')}. "
                         "Supported options are 'encodec_24khz'.")
    padding = hd.get("padding", "same")
    if padding not in ("center", "same"):
        raise ValueError("Padding must be 'center' or 'same'.")  # decoder/spectral_ops.py:24-25
    return ArchConfig(
        ratios=tuple(fe.get("dowmsamples", [6, 5, 5, 4])),
        vq_bins=int(fe.get("vq_bins", 16384)),
        num_quantizers=int(fe.get("num_quantizers", 1)),
        bandwidths=tuple(fe.get("bandwidths", [1.5, 3.0, 6.0, 12.0])),
        input_channels=int(bb["input_channels"]),
        dim=int(bb["dim"]),
        intermediate_dim=int(bb["intermediate_dim"]),
        num_layers=int(bb["num_layers"]),
        adanorm_num_embeddings=int(bb.get("adanorm_num_embeddings") or 0),
        n_fft=int(hd["n_fft"]),
        hop_length=int(hd["hop_length"]),
        padding=padding,
    )


def arch_from_yaml(path: str) -> ArchConfig:
    with open(path, "r") as f:
        return arch_from_yaml_dict(yaml.safe_load(f))


# The two architectures the reference ships YAMLs for.
ARCH_HOP600 = ArchConfig(ratios=(6, 5, 5, 4), n_fft=2400, hop_length=600)   # small/large-600, 40 tok/s
ARCH_HOP320 = ArchConfig(ratios=(8, 5, 4, 2), n_fft=1280, hop_length=320)   # small/medium/large-320, 75 tok/s

NAMED_ARCHS = {"hop600": ARCH_HOP600, "hop320": ARCH_HOP320}
