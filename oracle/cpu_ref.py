"""ORACLE — test infrastructure only, never the product path.

A flat CPU restatement of the reference's encode/decode path
(``WavTokenizer.encode_infer`` -> ``codes_to_features`` -> ``decode``,
/root/reference/decoder/pretrained.py:186-239) written as straight-line calls
of the same ATen ops the reference's ``nn.Module`` tree issues (the reference is
100 % PyTorch, so "the reference's CPU algorithm" *is* this op sequence; a C or
numpy port would only add a second, differently-rounded floating-point
implementation between us and the reference).  It takes a checkpoint-layout
state dict and an ``ArchConfig``; it never imports the
reference, so it travels to the GPU box.

Pinning: ``tests/golden/make_golden.py`` imports the real reference in the build
container, loads the same synthetic weights through ``load_state_dict``, and
asserts this file's outputs are BIT-IDENTICAL (codes, features, backbone output,
waveform, and every stage checkpoint) before writing the fixtures under
``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks the oracle against
those fixtures on every run.  Parity on *trained* weights is unpinned (no
checkpoint exists offline).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

ENC = "feature_extractor.encodec.encoder.model."
DEC = "feature_extractor.encodec.decoder.model."
VQ = "feature_extractor.encodec.quantizer.vq.layers.0._codebook."


# ----------------------------------------------------------------- padding helpers
def get_extra_padding_for_conv1d(length: int, kernel_size: int, stride: int, padding_total: int = 0) -> int:
    """encoder/modules/conv.py:54-61."""
    n_frames = (length - kernel_size + padding_total) / stride + 1
    ideal_length = (math.ceil(n_frames) - 1) * stride + (kernel_size - padding_total)
    return ideal_length - length


def pad1d_reflect(x: torch.Tensor, paddings: Tuple[int, int]) -> torch.Tensor:
    """encoder/modules/conv.py:79-96 (mode='reflect' branch, incl. the short-input zero extension)."""
    length = x.shape[-1]
    padding_left, padding_right = paddings
    assert padding_left >= 0 and padding_right >= 0, (padding_left, padding_right)
    max_pad = max(padding_left, padding_right)
    extra_pad = 0
    if length <= max_pad:
        extra_pad = max_pad - length + 1
        x = F.pad(x, (0, extra_pad))
    padded = F.pad(x, paddings, "reflect")
    end = padded.shape[-1] - extra_pad
    return padded[..., :end]


def reflect_index_map(T: int, pad_left: int, pad_right: int) -> List[int]:
    """Source index (or -1 for an inserted zero) of every element of pad1d_reflect's output.
    Pure-Python known-answer helper for the index arithmetic the HIP loader does."""
    max_pad = max(pad_left, pad_right)
    Tp = T
    if T <= max_pad:
        Tp = max_pad + 1
    out = []
    for p in range(pad_left + T + pad_right):
        j = p - pad_left
        if j < 0:
            j = -j
        if j >= Tp:
            j = 2 * (Tp - 1) - j
        out.append(j if j < T else -1)
    return out


# ------------------------------------------------------------------------- oracle
class OracleWavTokenizer:
    """Functional restatement; methods mirror decoder/pretrained.py's surface."""

    def __init__(self, arch, state_dict: Dict[str, torch.Tensor]):
        self.arch = arch
        self.sd = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(v)) for k, v in state_dict.items()}
        self.n_lstm = 1 + 3 * len(arch.ratios)

    # -- building blocks ------------------------------------------------------
    def _wn_weight(self, prefix: str) -> torch.Tensor:
        # torch.nn.utils.weight_norm pre-forward hook (conv.py:28): w = g * v / ||v||, dim=0
        return torch._weight_norm(self.sd[prefix + ".weight_v"], self.sd[prefix + ".weight_g"], 0)

    def sconv1d(self, x: torch.Tensor, prefix: str, stride: int = 1, dilation: int = 1) -> torch.Tensor:
        """SConv1d.forward, non-causal branch (conv.py:195-211)."""
        w = self._wn_weight(prefix)
        k = (w.shape[-1] - 1) * dilation + 1
        padding_total = k - stride
        extra = get_extra_padding_for_conv1d(x.shape[-1], k, stride, padding_total)
        pr = padding_total // 2
        pl = padding_total - pr
        x = pad1d_reflect(x, (pl, pr + extra))
        return F.conv1d(x, w, self.sd[prefix + ".bias"], stride=stride, dilation=dilation)

    def resblock(self, x: torch.Tensor, prefix: str) -> torch.Tensor:
        """SEANetResnetBlock (seanet.py:21-63), true_skip=False, kernel sizes [3,1]."""
        h = F.elu(x, alpha=1.0)
        h = self.sconv1d(h, prefix + ".block.1.conv.conv")
        h = F.elu(h, alpha=1.0)
        h = self.sconv1d(h, prefix + ".block.3.conv.conv")
        return self.sconv1d(x, prefix + ".shortcut.conv.conv") + h

    def slstm(self, x: torch.Tensor, prefix: str) -> torch.Tensor:
        """SLSTM.forward (lstm.py:31-39): nn.LSTM(512,512,2) with zero state + skip."""
        x1 = x.permute(2, 0, 1)
        H = x.shape[1]
        flat = []
        for layer in range(2):
            for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                flat.append(self.sd[f"{prefix}.lstm.{nm}_l{layer}"])
        h0 = torch.zeros(2, x.shape[0], H, dtype=x.dtype)
        c0 = torch.zeros(2, x.shape[0], H, dtype=x.dtype)
        y, _, _ = torch._VF.lstm(x1, (h0, c0), flat, True, 2, 0.0, False, False, False)
        y = y.permute(1, 2, 0)
        return y + x

    # -- encoder ------------------------------------------------------------------
    def encoder(self, x: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        """SEANetEncoder.forward (seanet.py:143); x (B,1,T) -> (B,512,L)."""
        def tap(name, v):
            if taps is not None:
                taps[name] = v
        x = self.sconv1d(x, ENC + "0.conv.conv")
        tap("enc.0", x)
        idx = 1
        for r in self.arch.enc_ratios:
            x = self.resblock(x, ENC + f"{idx}")
            tap(f"enc.{idx}", x)
            x = F.elu(x, alpha=1.0)
            x = self.sconv1d(x, ENC + f"{idx + 2}.conv.conv", stride=r)
            tap(f"enc.{idx + 2}", x)
            idx += 3
        x = self.slstm(x, ENC + f"{idx}")
        tap(f"enc.{idx}", x)
        x = F.elu(x, alpha=1.0)
        x = self.sconv1d(x, ENC + f"{idx + 2}.conv.conv")
        tap(f"enc.{idx + 2}", x)
        return x

    # -- quantiser ----------------------------------------------------------------
    def quantize(self, emb: torch.Tensor, taps: Optional[dict] = None):
        """ResidualVectorQuantizer.infer (vq.py:115-140, n_q forced to 1) ->
        LanguageVectorQuantization.forward (core_vq.py:378-401) ->
        VectorQuantization.forward (:294-315) -> EuclideanCodebook.forward (:206-231)."""
        embed_w = self.sd[VQ + "embed"]
        assert float(self.sd[VQ + "inited"].reshape(-1)[0]) == 1.0, "codebook not initialised (core_vq.py:140-151 would run k-means)"
        x = emb.permute(0, 2, 1)                       # rearrange b d n -> b n d
        shape = x.shape
        flat = x.reshape(-1, shape[-1])
        embed = embed_w.t()
        dist = -(flat.pow(2).sum(1, keepdim=True) - 2 * flat @ embed + embed.pow(2).sum(0, keepdim=True))
        ind = dist.max(dim=-1).indices
        if taps is not None:
            top2 = dist.topk(2, dim=-1).values
            taps["vq.margin"] = (top2[:, 0] - top2[:, 1]).reshape(shape[:-1])
        ind = ind.view(*shape[:-1])
        q = F.embedding(ind, embed_w)
        q = q.permute(0, 2, 1)                          # b n d -> b d n
        codes = torch.stack([ind])                      # (n_q=1, B, L)
        return q, codes

    # -- API: encode ------------------------------------------------------------
    def encode_infer(self, audio: torch.Tensor, bandwidth_id=None, taps: Optional[dict] = None):
        """WavTokenizer.encode_infer (pretrained.py:186-189) -> EncodecFeatures.infer
        (feature_extractors.py:131-142)."""
        emb = self.encoder(audio.unsqueeze(1), taps)
        q, codes = self.quantize(emb, taps)
        return q, codes

    def codes_to_features(self, codes: torch.Tensor) -> torch.Tensor:
        """pretrained.py:209-239."""
        if codes.dim() == 2:
            codes = codes.unsqueeze(1)
        n_bins = self.arch.vq_bins
        offsets = torch.arange(0, n_bins * len(codes), n_bins)
        idx = codes + offsets.view(-1, 1, 1)
        tmp = torch.cat([self.sd[VQ.replace("layers.0.", f"layers.{q}.") + "embed"] for q in range(self.arch.num_quantizers)], dim=0)
        feats = F.embedding(idx, tmp).sum(dim=0)
        return feats.transpose(1, 2)

    # -- decoder backbone -----------------------------------------------------------
    def _gn(self, x, prefix):
        return F.group_norm(x, 32, self.sd[prefix + ".weight"], self.sd[prefix + ".bias"], eps=1e-6)

    def pos_resnet(self, x, prefix):
        """ResnetBlock.forward (models.py:58-78), temb=None, dropout=identity in eval."""
        h = self._gn(x, prefix + ".norm1")
        h = h * torch.sigmoid(h)
        h = F.conv1d(h, self.sd[prefix + ".conv1.weight"], self.sd[prefix + ".conv1.bias"], padding=1)
        h = self._gn(h, prefix + ".norm2")
        h = h * torch.sigmoid(h)
        h = F.conv1d(h, self.sd[prefix + ".conv2.weight"], self.sd[prefix + ".conv2.bias"], padding=1)
        return x + h

    def pos_attn(self, x, prefix):
        """AttnBlock.forward (models.py:107-127)."""
        h_ = self._gn(x, prefix + ".norm")
        q = F.conv1d(h_, self.sd[prefix + ".q.weight"], self.sd[prefix + ".q.bias"])
        k = F.conv1d(h_, self.sd[prefix + ".k.weight"], self.sd[prefix + ".k.bias"])
        v = F.conv1d(h_, self.sd[prefix + ".v.weight"], self.sd[prefix + ".v.bias"])
        b, c, h = q.shape
        q = q.permute(0, 2, 1)
        w_ = torch.bmm(q, k)
        w_ = w_ * (int(c) ** (-0.5))
        w_ = F.softmax(w_, dim=2)
        w_ = w_.permute(0, 2, 1)
        h_ = torch.bmm(v, w_)
        h_ = F.conv1d(h_, self.sd[prefix + ".proj_out.weight"], self.sd[prefix + ".proj_out.bias"])
        return x + h_

    def adanorm(self, x, prefix, cond_id):
        """AdaLayerNorm.forward (modules.py:81-86)."""
        scale = F.embedding(cond_id, self.sd[prefix + ".scale.weight"])
        shift = F.embedding(cond_id, self.sd[prefix + ".shift.weight"])
        x = F.layer_norm(x, (self.arch.dim,), eps=1e-6)
        return x * scale + shift

    def convnext(self, x, prefix, cond_id):
        """ConvNeXtBlock.forward (modules.py:43-60)."""
        residual = x
        x = F.conv1d(x, self.sd[prefix + ".dwconv.weight"], self.sd[prefix + ".dwconv.bias"], padding=3,
                     groups=self.arch.dim)
        x = x.transpose(1, 2)
        x = self.adanorm(x, prefix + ".norm", cond_id)
        x = F.linear(x, self.sd[prefix + ".pwconv1.weight"], self.sd[prefix + ".pwconv1.bias"])
        x = F.gelu(x)
        x = F.linear(x, self.sd[prefix + ".pwconv2.weight"], self.sd[prefix + ".pwconv2.bias"])
        x = self.sd[prefix + ".gamma"] * x
        x = x.transpose(1, 2)
        return residual + x

    def backbone(self, x: torch.Tensor, bandwidth_id: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        """VocosBackbone.forward (models.py:223-235): (B,512,L) -> (B,L,768)."""
        def tap(name, v):
            if taps is not None:
                taps[name] = v
        assert bandwidth_id is not None
        x = F.conv1d(x, self.sd["backbone.embed.weight"], self.sd["backbone.embed.bias"], padding=3)
        tap("bb.embed", x)
        for i in (0, 1):
            x = self.pos_resnet(x, f"backbone.pos_net.{i}")
            tap(f"bb.pos_net.{i}", x)
        x = self.pos_attn(x, "backbone.pos_net.2")
        tap("bb.pos_net.2", x)
        for i in (3, 4):
            x = self.pos_resnet(x, f"backbone.pos_net.{i}")
            tap(f"bb.pos_net.{i}", x)
        x = self._gn(x, "backbone.pos_net.5")
        tap("bb.pos_net.5", x)
        x = self.adanorm(x.transpose(1, 2), "backbone.norm", bandwidth_id)
        x = x.transpose(1, 2)
        tap("bb.norm", x)
        for i in range(self.arch.num_layers):
            x = self.convnext(x, f"backbone.convnext.{i}", bandwidth_id)
            if i in (0, self.arch.num_layers // 2 - 1, self.arch.num_layers - 1):
                tap(f"bb.convnext.{i}", x)
        x = F.layer_norm(x.transpose(1, 2), (self.arch.dim,), self.sd["backbone.final_layer_norm.weight"],
                         self.sd["backbone.final_layer_norm.bias"], eps=1e-6)
        tap("bb.out", x)
        return x

    # -- head -----------------------------------------------------------------------
    def head(self, x: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        """ISTFTHead.forward (heads.py:42-67) + ISTFT.forward 'same' (spectral_ops.py:33-75)."""
        n_fft, hop = self.arch.n_fft, self.arch.hop_length
        win = self.sd["head.istft.window"]
        x = F.linear(x, self.sd["head.out.weight"], self.sd["head.out.bias"]).transpose(1, 2)
        if taps is not None:
            taps["head.out"] = x
        mag, p = x.chunk(2, dim=1)
        mag = torch.exp(mag)
        mag = torch.clip(mag, max=1e2)
        xr = torch.cos(p)
        yi = torch.sin(p)
        S = mag * (xr + 1j * yi)
        if self.arch.padding == "center":
            return torch.istft(S, n_fft, hop, n_fft, win, center=True)
        pad = (n_fft - hop) // 2
        assert S.dim() == 3
        B, N, T = S.shape
        ifft = torch.fft.irfft(S, n_fft, dim=1, norm="backward")
        ifft = ifft * win[None, :, None]
        output_size = (T - 1) * hop + n_fft
        y = F.fold(ifft, output_size=(1, output_size), kernel_size=(1, n_fft), stride=(1, hop))[:, 0, 0, pad:-pad]
        window_sq = win.square().expand(1, T, -1).transpose(1, 2)
        env = F.fold(window_sq, output_size=(1, output_size), kernel_size=(1, n_fft), stride=(1, hop)).squeeze()[pad:-pad]
        assert (env > 1e-11).all()
        return y / env

    # -- API: decode ------------------------------------------------------------------
    def decode(self, features: torch.Tensor, bandwidth_id: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        """WavTokenizer.decode (pretrained.py:192-207)."""
        x = self.backbone(features, bandwidth_id, taps)
        return self.head(x, taps)

    def forward(self, audio: torch.Tensor, bandwidth_id: torch.Tensor) -> torch.Tensor:
        """WavTokenizer.forward (pretrained.py:159-175); in eval the training-mode quantiser
        forward picks n_q=1 too (vq.py:98-111), so features equal encode_infer's."""
        feats, _ = self.encode_infer(audio, bandwidth_id)
        return self.decode(feats, bandwidth_id)

    # -- secondary: SEANetDecoder (seanet.py:147-238) --------------------------------------
    def sconvtr1d(self, x: torch.Tensor, prefix: str, stride: int) -> torch.Tensor:
        """SConvTranspose1d.forward, non-causal (conv.py:232-253)."""
        w = self._wn_weight(prefix)       # (Cin, Cout, k); weight_norm dim=0 -> per input channel
        k = w.shape[-1]
        padding_total = k - stride
        y = F.conv_transpose1d(x, w, self.sd[prefix + ".bias"], stride=stride)
        pr = padding_total // 2
        pl = padding_total - pr
        return y[..., pl: y.shape[-1] - pr]

    def seanet_decoder(self, z: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        def tap(name, v):
            if taps is not None:
                taps[name] = v
        x = self.sconv1d(z, DEC + "0.conv.conv")
        x = self.slstm(x, DEC + "1")
        tap("sdec.1", x)
        idx = 2
        for r in self.arch.ratios:
            x = F.elu(x, alpha=1.0)
            x = self.sconvtr1d(x, DEC + f"{idx + 1}.convtr.convtr", r)
            tap(f"sdec.{idx + 1}", x)
            x = self.resblock(x, DEC + f"{idx + 2}")
            idx += 3
        x = F.elu(x, alpha=1.0)
        x = self.sconv1d(x, DEC + f"{idx + 1}.conv.conv")
        return x
