"""wavtokenizer_amd — MI355X-native WavTokenizer encode/decode path.

``from wavtokenizer_amd import WavTokenizer`` is a drop-in for the reference's
``from decoder.pretrained import WavTokenizer`` (decoder/pretrained.py:32).
Importing the class loads the HIP C-ABI library; it raises if the library is missing.
"""
from .config import ArchConfig, ARCH_HOP600, ARCH_HOP320, NAMED_ARCHS, arch_from_yaml  # noqa: F401

__all__ = ["WavTokenizer", "ArchConfig", "ARCH_HOP600", "ARCH_HOP320", "NAMED_ARCHS", "arch_from_yaml"]


def __getattr__(name):
    if name == "WavTokenizer":
        from .pretrained import WavTokenizer
        return WavTokenizer
    raise AttributeError(name)
