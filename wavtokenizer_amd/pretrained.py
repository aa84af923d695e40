"""Drop-in ``WavTokenizer`` (mirror of the reference's decoder/pretrained.py:32-239).

Same constructor classmethods, method names, argument meaning, tensor layouts and error
behaviour as the reference class; underneath, every method enqueues hand-written HIP kernels
through the C-ABI library (``include/wavtokenizer_amd.h``).  PyTorch supplies device memory,
the current HIP stream and the ``nn.Module`` parameter container (so ``state_dict()`` /
``load_state_dict()`` / ``.to(device)`` keep working with reference checkpoints) — no torch op
runs on the compute path, and there is no CPU fallback: calling a method while the module sits
on the CPU raises.
"""
from __future__ import annotations

import ctypes
import os
import weakref
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch
import yaml
from torch import nn

from . import _capi
from ._capi import WavTokError, WtArch, WtTensor, check, lib
from .config import ArchConfig, arch_from_yaml_dict
from .state_spec import full_state_spec, is_buffer


# ------------------------------------------------------------------ parameter containers
class _Holder(nn.Module):
    """Parameter/buffer container addressed by the reference's dotted state-dict keys."""

    def _put(self, dotted: str, tensor: torch.Tensor, buffer: bool):
        head, _, rest = dotted.partition(".")
        if rest:
            if head not in self._modules:
                self.add_module(head, _Holder())
            self._modules[head]._put(rest, tensor, buffer)
        elif buffer:
            self.register_buffer(head, tensor)
        else:
            self.register_parameter(head, nn.Parameter(tensor, requires_grad=False))

    def __getitem__(self, i):
        return self._modules[str(i)]

    def __len__(self):
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules.values())

    def _bind(self, root):
        object.__setattr__(self, "_root", weakref.ref(root))
        for m in self._modules.values():
            if isinstance(m, _Holder):
                m._bind(root)


class SEANetEncoder(_Holder):
    """encodec.encoder: callable (B,1,T) -> (B,512,L) (encoder/modules/seanet.py:143)."""

    @torch.inference_mode()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        assert x.dim() == 3 and x.shape[1] == 1, "expected (B, 1, T)"
        return self._root()._run_encode(x[:, 0, :])[2]


class SEANetDecoder(_Holder):
    """encodec.decoder: callable (B,512,L) -> (B,1,L*hop) (encoder/modules/seanet.py:236-238)."""

    @torch.inference_mode()
    def forward(self, z: torch.Tensor) -> torch.Tensor:
        return self._root()._run_seanet_decoder(z)


class _CodebookLayer(_Holder):
    @property
    def codebook(self):                 # core_vq.py:274-276
        return self._modules["_codebook"].embed


class ResidualVectorQuantizer(_Holder):
    pass


class EncodecModel(_Holder):
    sample_rate = 24000
    channels = 1


class EncodecFeatures(_Holder):
    """decoder/feature_extractors.py:55-142 (container + infer)."""

    def forward(self, audio: torch.Tensor, bandwidth_id: torch.Tensor):
        # eval-mode quantiser forward picks n_q = 1 as well (vq.py:98-111), so it equals infer()
        return self.infer(audio, bandwidth_id)

    @torch.inference_mode()
    def infer(self, audio: torch.Tensor, bandwidth_id: torch.Tensor):
        _ = self.bandwidths[self._root()._bandwidth_index(bandwidth_id)] if bandwidth_id is not None else None
        feats, codes, _emb = self._root()._run_encode(audio, want_emb=False)
        # the third element of the reference's tuple (feature_extractors.py:141) is a zero scalar in eval mode; one cached
        # tensor per device instead of a fill kernel per call
        z = getattr(self, "_zero_loss", None)
        if z is None or z.device != audio.device:
            z = torch.zeros((), device=audio.device)
            object.__setattr__(self, "_zero_loss", z)
        return feats, codes, z


class VocosBackbone(_Holder):
    """decoder/models.py:223-235: callable (B,512,L) -> (B,L,dim)."""

    @torch.inference_mode()
    def forward(self, x: torch.Tensor, bandwidth_id: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert bandwidth_id is not None      # models.py:227
        return self._root()._run_decode(x, bandwidth_id, want_backbone=True)[1]


class ISTFTHead(_Holder):
    """decoder/heads.py:42-67: callable (B, L, dim) -> (B, L*hop)."""

    @torch.inference_mode()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._root()._run_head(x)


def _build_tree(arch: ArchConfig) -> Tuple[EncodecFeatures, VocosBackbone, ISTFTHead]:
    fe, bb, hd = EncodecFeatures(), VocosBackbone(), ISTFTHead()
    enc_model = EncodecModel()
    enc_model.add_module("encoder", SEANetEncoder())
    enc_model.add_module("quantizer", ResidualVectorQuantizer())
    enc_model.add_module("decoder", SEANetDecoder())
    fe.add_module("encodec", enc_model)
    q = enc_model.quantizer
    q.add_module("vq", _Holder())
    q.vq.add_module("layers", _Holder())
    for i in range(arch.num_quantizers):
        q.vq.layers.add_module(str(i), _CodebookLayer())
    q.bins = arch.vq_bins
    q.n_q = arch.num_quantizers
    q.dimension = 512
    fe.bandwidths = list(arch.bandwidths)
    fe.frame_rate = 25
    roots = {"feature_extractor": fe, "backbone": bb, "head": hd}
    for key, shape in full_state_spec(arch).items():
        top, _, rest = key.partition(".")
        t = torch.zeros(shape, dtype=torch.float32)
        roots[top]._put(rest, t, is_buffer(key))
    return fe, bb, hd


# --------------------------------------------------------------------------------- engine
class _Engine:
    """Owns the wt_model handle, the (kind, B, len) plans and their workspaces."""

    def __init__(self):
        self.model = ctypes.c_void_p()
        # cached plans (+ workspaces): least recently used ones go first once there are more than max_plans of them or
        # their workspaces exceed max_ws_bytes (a file-by-file caller meets a new length, hence a new plan, per file)
        self.max_plans = int(os.environ.get("WAVTOK_MAX_PLANS", "64"))
        self.max_ws_bytes = int(float(os.environ.get("WAVTOK_MAX_WORKSPACE_GB", "16")) * (1 << 30))
        # plans are per HIP stream (a plan owns a workspace): at most this many non-default streams keep plans at a time; a
        # caller that makes a new stream (or takes one of torch's 32 pooled side streams) per request would otherwise fill the
        # LRU with plans + workspaces + recorded graphs of streams it never uses again (the least recently used stream goes first)
        self.max_streams = int(os.environ.get("WAVTOK_MAX_STREAMS", "4"))
        self.stream_lru: List[int] = []
        self.plans: Dict[tuple, Tuple[ctypes.c_void_p, torch.Tensor]] = {}
        self.io: Dict[Tuple[int, int, int, int], Dict[str, torch.Tensor]] = {}     # fixed I/O buffers of graph plans
        self.device_index = -1
        self._keepalive: List[Any] = []

    def close(self):
        for plan, _ws in self.plans.values():
            lib.wt_plan_destroy(plan)
        self.plans.clear()
        self.io.clear()
        if self.model:
            lib.wt_model_destroy(self.model)
            self.model = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load(self, arch: ArchConfig, state: Dict[str, torch.Tensor], device_index: int):
        self.close()
        wa = WtArch()
        wa.n_ratios = len(arch.ratios)
        for i, r in enumerate(arch.ratios):
            wa.ratios[i] = int(r)
        wa.vq_bins, wa.num_quantizers, wa.input_channels = arch.vq_bins, arch.num_quantizers, arch.input_channels
        wa.dim, wa.intermediate_dim, wa.num_layers = arch.dim, arch.intermediate_dim, arch.num_layers
        wa.adanorm_num_embeddings = arch.adanorm_num_embeddings
        wa.n_fft, wa.hop_length = arch.n_fft, arch.hop_length
        wa.padding_same = 1 if arch.padding == "same" else 0
        arrs = []
        tens = (WtTensor * len(state))()
        for i, (k, v) in enumerate(state.items()):
            a = np.ascontiguousarray(v.detach().to("cpu", torch.float32).numpy())
            arrs.append(a)
            tens[i].name = k.encode()
            tens[i].data = a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
            tens[i].numel = a.size
        check(lib.wt_model_create(ctypes.byref(wa), tens, len(state), device_index, ctypes.byref(self.model)),
              "wt_model_create")
        self.device_index = device_index

    def load_packed(self, image: np.ndarray, device_index: int):
        """Upload a packed image (WavTokenizer.save_packed): nothing is folded, packed or split again."""
        self.close()
        check(lib.wt_model_create_packed(image.ctypes.data_as(ctypes.c_void_p), image.nbytes, device_index,
                                         ctypes.byref(self.model)), "wt_model_create_packed")
        self.device_index = device_index

    def export(self) -> np.ndarray:
        n = lib.wt_model_export_bytes(self.model)
        buf = np.empty(n, dtype=np.uint8)
        check(lib.wt_model_export(self.model, buf.ctypes.data_as(ctypes.c_void_p), n), "wt_model_export")
        used = lib.wt_packed_bytes(buf.ctypes.data_as(ctypes.c_void_p), n)       # wt_model_export_bytes is an upper bound
        return buf[:used] if 0 < used <= n else buf

    @staticmethod
    def _key(kind: int, B: int, length: int, flags: int, device, sites: int = 0) -> tuple:
        """A plan owns a workspace, so it serves ONE stream: calls made under another current stream than the default one
        get plans (and staging buffers) of their own, keyed (..., stream handle).  Two streams can then run the same model
        side by side (sharding.StepRunner(lanes=2)); the library orders their persistent LSTM launches itself (capi.cpp).
        Key: (kind, B, length, flags) on the default stream, + (stream,) on another one, + (stream, fp32 site mask) for a
        plan with range sites on fp32 operands (stream 0 = the default stream)."""
        sp = torch.cuda.current_stream(device).cuda_stream if device is not None else 0
        if sites:
            return (kind, B, length, flags, sp, sites)
        return (kind, B, length, flags) if not sp else (kind, B, length, flags, sp)

    def _touch_stream(self, sp: int):
        if not sp:
            return
        if sp in self.stream_lru:
            self.stream_lru.remove(sp)
        self.stream_lru.append(sp)
        while len(self.stream_lru) > self.max_streams:
            old = self.stream_lru.pop(0)
            self.drop(lambda k: len(k) >= 5 and k[4] == old)

    def plan(self, kind: int, B: int, length: int, flags: int, device: torch.device, sites: int = 0):
        key = self._key(kind, B, length, flags, device, sites)
        hit = self.plans.pop(key, None)
        if hit is not None:
            self.plans[key] = hit                     # most recently used last (dicts keep insertion order)
            return hit
        if len(key) >= 5:
            self._touch_stream(key[4])
        p = ctypes.c_void_p()
        check(lib.wt_plan_create_ex(self.model, kind, B, length, flags, sites, ctypes.byref(p)), "wt_plan_create")
        need = lib.wt_plan_workspace_bytes(p)
        while self.plans and (len(self.plans) >= self.max_plans or
                              sum(w.numel() for _p, w in self.plans.values()) + need > self.max_ws_bytes):
            old = next(iter(self.plans))              # LRU: the least recently used plan + workspace
            lib.wt_plan_destroy(self.plans.pop(old)[0])
            self.io.pop(old, None)
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        self.plans[key] = (p, ws)
        return p, ws

    def drop(self, pred):
        """Destroys the cached plans (and their workspaces) whose key (kind, B, length, flags) satisfies pred."""
        for k in [k for k in self.plans if pred(k)]:
            lib.wt_plan_destroy(self.plans.pop(k)[0])
            self.io.pop(k, None)

    def staging(self, kind: int, B: int, length: int, flags: int, make, device=None, sites: int = 0) -> Dict[str, torch.Tensor]:
        """Fixed input/output tensors of a graph plan: a recorded hipGraph replays fixed addresses, so calls copy their
        input in and hand out copies of the results (a few hundred KB at the batch sizes graphs are used for)."""
        key = self._key(kind, B, length, flags, device, sites)
        io = self.io.get(key)
        if io is None:
            io = self.io[key] = make()
        return io


def site_name(site: int) -> str:
    """Range site (include/wavtokenizer_amd.h wt_range_site) -> the reference module it covers."""
    names = {_capi.WT_SITE_ENCODER: "feature_extractor.encodec.encoder + quantizer", _capi.WT_SITE_BB_EMBED: "backbone.embed",
             _capi.WT_SITE_RES0: "backbone.pos_net.0", _capi.WT_SITE_RES1: "backbone.pos_net.1", _capi.WT_SITE_ATTN: "backbone.pos_net.2",
             _capi.WT_SITE_RES2: "backbone.pos_net.3", _capi.WT_SITE_RES3: "backbone.pos_net.4", _capi.WT_SITE_HEAD: "backbone.final_layer_norm + head",
             _capi.WT_SITE_SEANET_DECODER: "feature_extractor.encodec.decoder"}
    if _capi.WT_SITE_CNX0 <= site < _capi.WT_SITE_HEAD:
        return "backbone.convnext.%d" % (site - _capi.WT_SITE_CNX0)
    return names.get(site, "site %d" % site)


def _stream_ptr(device: torch.device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


# ------------------------------------------------------------------------------ public class
class WavTokenizer(nn.Module):
    """Same surface as the reference class (decoder/pretrained.py:32)."""

    def __init__(self, feature_extractor: EncodecFeatures, backbone: VocosBackbone, head: ISTFTHead,
                 arch: Optional[ArchConfig] = None):
        super().__init__()
        self.feature_extractor = feature_extractor
        self.backbone = backbone
        self.head = head
        self._arch = arch
        self._engine = _Engine()
        self._dirty = True
        self._plan_flags = 0
        self._fp32_sites = 0                 # range sites (_capi.WT_SITE_*) that have left the split-f16 form after an overflow
        # strict status: None (default) = automatic: calls of up to _graph_max_clips clips (graph-replayed, bound by the host
        # anyway: the reference's own file-by-file usage) synchronise, check and REPEAT a failed call on the fallback path, so
        # an infer.py-style caller is never handed poisoned tensors; larger batches stay asynchronous and report on the next
        # call.  WAVTOK_STRICT_STATUS=1 / 0 or set_strict_status force it on / off for every size
        env_strict = os.environ.get("WAVTOK_STRICT_STATUS")
        self._strict = None if env_strict is None else env_strict == "1"
        # codes_to_features and indices outside the codebook (F.embedding raises IndexError, pretrained.py:236): "sync"
        # (default) synchronises after the gather (a few microseconds of work) and raises for the offending call, like the
        # reference on the CPU; "deferred" (opt-in, for pipelines that must not synchronise) raises on the NEXT call on this
        # model or in check_status(); "off" never looks.  The gathered features of a bad index are NaN in every mode.
        env_cc = os.environ.get("WAVTOK_CHECK_CODES", "sync")
        env_cc = {"1": "sync", "0": "off"}.get(env_cc, env_cc)
        if env_cc not in ("sync", "deferred", "off"):
            raise ValueError(f"WAVTOK_CHECK_CODES={env_cc!r}: use sync (or 1), deferred or off (or 0)")
        self._check_codes = env_cc
        self._bw_cache = None                # (tensor ref, version, index): bandwidth_id tensors living on the GPU
        self.fallback_events: List[str] = []   # device-side failures this model has answered by falling back (check_status reports them)
        # batches up to this many clips are replayed as one hipGraph per (shape) plan: they are bound by the host's
        # launch rate (about 100 launches per call), not by the GPU; 0 turns graphs off
        self._graph_max_clips = int(os.environ.get("WAVTOK_GRAPH_MAX_CLIPS", "16"))
        for m in (feature_extractor, backbone, head):
            m._bind(self)

    # -- construction (pretrained.py:46-156) ---------------------------------------------------
    @classmethod
    def _from_config_node(cls, node: Dict[str, Any]) -> "WavTokenizer":
        arch = arch_from_yaml_dict({"model": {"init_args": node}})
        fe, bb, hd = _build_tree(arch)
        return cls(feature_extractor=fe, backbone=bb, head=hd, arch=arch)

    @classmethod
    def from_arch(cls, arch: ArchConfig) -> "WavTokenizer":
        fe, bb, hd = _build_tree(arch)
        return cls(feature_extractor=fe, backbone=bb, head=hd, arch=arch)

    @classmethod
    def from_hparams(cls, config_path: str) -> "WavTokenizer":
        with open(config_path, "r") as f:
            config = yaml.safe_load(f)
        return cls._from_config_node(config)

    @classmethod
    def from_hparams0802(cls, config_path: str) -> "WavTokenizer":
        with open(config_path, "r") as f:
            config = yaml.safe_load(f)
        return cls._from_config_node(config["model"]["init_args"])

    @staticmethod
    def _filter_state(state_dict_raw: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        # pretrained.py:103-105: drops the discriminators of the Lightning checkpoint
        return {k: v for k, v in state_dict_raw.items()
                if k.startswith("backbone.") or k.startswith("head.") or k.startswith("feature_extractor.")}

    @staticmethod
    def _read_ckpt(model_path: str) -> Dict[str, torch.Tensor]:
        # Lightning .ckpt: a pickle; only tensors are read (weights_only), nothing else is executed
        return torch.load(model_path, map_location="cpu", weights_only=True)["state_dict"]

    @classmethod
    def from_pretrained0802(cls, config_path: str, model_path: str) -> "WavTokenizer":
        model = cls.from_hparams0802(config_path)
        model.load_state_dict(cls._filter_state(cls._read_ckpt(model_path)))
        model.eval()
        return model

    @staticmethod
    def _best_checkpoints(folder: str, keep: int = 3) -> List[str]:
        """File names of the `vocos_*` checkpoints to average: the reference ranks them by the six characters that
        precede the extension (the validation loss printed into the name) as STRINGS and keeps every file whose tag is
        among the `keep` smallest, in directory order (pretrained.py:122-138)."""
        names = [n for n in os.listdir(folder) if n.startswith("vocos_")]
        loss_tag = lambda n: n[-11:-5]
        chosen = set(sorted(loss_tag(n) for n in names)[:keep])
        return [n for n in names if loss_tag(n) in chosen]

    @classmethod
    def from_pretrained0911(cls, config_path: str, model_folder_path: str) -> "WavTokenizer":
        """Mean of the best `vocos_*` checkpoints of a folder (pretrained.py:117-156): tensors are summed in directory
        order and divided by the count, in their own dtype, exactly as the reference does."""
        model = cls.from_hparams0802(config_path)
        parts = [cls._filter_state(cls._read_ckpt(os.path.join(model_folder_path, n)))
                 for n in cls._best_checkpoints(model_folder_path)]
        if not parts:
            raise FileNotFoundError(f"no vocos_* checkpoint in {model_folder_path}")
        mean_state = {}
        for key, first in parts[0].items():
            total = first.clone()
            for other in parts[1:]:
                total += other[key]
            mean_state[key] = total / len(parts)
        model.load_state_dict(mean_state)
        model.eval()
        return model

    # -- hot-path state: one pickle-free file holding exactly the hot-path tensors ------------------------------------
    def save_hot_state(self, path: str) -> None:
        """Write the hot-path state (the 289 reference keys this class keeps; discriminators, optimizer state and
        the rest of a Lightning checkpoint are gone) as one safetensors file: nothing is executed when it is read
        back, and it loads without unpickling a multi-GB training checkpoint."""
        from safetensors.torch import save_file
        sd = {k: v.detach().to("cpu", copy=True).contiguous() for k, v in self.state_dict().items()}
        save_file(sd, path, metadata={"format": "wavtokenizer_amd.hot_state.v1", "hop": str(self._arch.hop)})

    @classmethod
    def from_hot_state(cls, config_path: str, path: str) -> "WavTokenizer":
        """Counterpart of from_pretrained0802 (pretrained.py:95-114) for a file written by save_hot_state."""
        from safetensors.torch import load_file
        model = cls.from_hparams0802(config_path)
        model.load_state_dict(load_file(path, device="cpu"))
        model.eval()
        return model

    # -- packed image: what sits in HBM after loading, ready to upload again ------------------------------
    def save_packed(self, path: str) -> None:
        """Write the model as it sits in HBM (folded conv weights in [Cout][tap][Cin], LSTM lane packings, the packed
        ISTFT head and inverse-DFT basis, the S32 split copies and their scales) behind a header with a layout
        version and an architecture hash.  from_packed uploads it without folding, packing or splitting anything again.
        The model must be on the GPU (the image is read back from there)."""
        self._ensure_engine()
        self._engine.export().tofile(path)

    @classmethod
    def from_packed(cls, config_path: str, packed_path: str, device="cuda") -> "WavTokenizer":
        """Counterpart of from_pretrained0802 (pretrained.py:95-114) for a file written by save_packed: the file is
        memory-mapped and uploaded as it is.  The returned model keeps no copy of the reference's raw tensors (its
        nn.Module parameters are placeholders): state_dict() raises, load_state_dict() turns it into a normal model."""
        model = cls.from_hparams0802(config_path)
        image = np.memmap(packed_path, dtype=np.uint8, mode="r")
        wa, ver, ah = WtArch(), ctypes.c_int32(), ctypes.c_uint64()
        check(lib.wt_packed_info(image.ctypes.data_as(ctypes.c_void_p), image.nbytes, ctypes.byref(wa), ctypes.byref(ver),
                                 ctypes.byref(ah)), "wt_packed_info")
        a = model._arch
        # every field of wt_arch: the library sizes its launches from the image's architecture while this class sizes the
        # tensors it hands over from the config's (a 'same' image under a 'center' config would be written B*hop floats
        # past the end of the waveform tensor), so any difference is an error
        mine = {"ratios": tuple(a.ratios), "vq_bins": a.vq_bins, "num_quantizers": a.num_quantizers,
                "input_channels": a.input_channels, "dim": a.dim, "intermediate_dim": a.intermediate_dim,
                "num_layers": a.num_layers, "adanorm_num_embeddings": a.adanorm_num_embeddings, "n_fft": a.n_fft,
                "hop_length": a.hop_length, "padding": a.padding}
        theirs = {"ratios": tuple(wa.ratios[i] for i in range(wa.n_ratios)), "vq_bins": wa.vq_bins,
                  "num_quantizers": wa.num_quantizers, "input_channels": wa.input_channels, "dim": wa.dim,
                  "intermediate_dim": wa.intermediate_dim, "num_layers": wa.num_layers,
                  "adanorm_num_embeddings": wa.adanorm_num_embeddings, "n_fft": wa.n_fft, "hop_length": wa.hop_length,
                  "padding": "same" if wa.padding_same else "center"}
        diff = {k: (theirs[k], mine[k]) for k in mine if mine[k] != theirs[k]}
        if diff:
            raise ValueError(f"packed file was written for another architecture (field: (file, config)): {diff}")
        dev = torch.device(device)
        model.eval()
        model = model.to(dev)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        model._engine.load_packed(np.asarray(image), idx)
        model._dirty = False
        model._packed_only = True
        return model

    def state_dict(self, *a, **kw):
        if getattr(self, "_packed_only", False):
            raise RuntimeError("this model was loaded from a packed image (from_packed): it holds the folded and packed "
                               "weights in HBM, not the reference's raw tensors; load a checkpoint to get a state_dict")
        return super().state_dict(*a, **kw)

    @classmethod
    def from_pretrained(cls, repo_id: str) -> "WavTokenizer":
        from huggingface_hub import hf_hub_download
        config_path = hf_hub_download(repo_id=repo_id, filename="config.yaml")
        model_path = hf_hub_download(repo_id=repo_id, filename="pytorch_model.bin")
        model = cls.from_hparams(config_path)
        state_dict = torch.load(model_path, map_location="cpu", weights_only=True)
        model.load_state_dict(state_dict, strict=False)
        model.eval()
        return model

    # -- nn.Module protocol: any weight or device change invalidates the packed HBM copy -------
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._dirty = True
        self._packed_only = False
        return out

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        if not getattr(self, "_packed_only", False):      # a packed model has nothing to re-pack from
            self._dirty = True
        return out

    def refresh_weights(self):
        """Call after mutating parameters in place."""
        self._dirty = True

    def set_debug_keep_stages(self, on: bool, unfused: bool = False):
        """Parity tests: keep every stage buffer of the next plans distinct in the workspace (and snapshot the in-place
        residual stream).  The kernels stay the shipped ones; `unfused=True` selects the unfused debug twin instead
        (raw fp32 tensors between stages)."""
        f = self._plan_flags & ~(_capi.WT_PLAN_FLAG_KEEP_STAGES | _capi.WT_PLAN_FLAG_UNFUSED)
        if on:
            f |= _capi.WT_PLAN_FLAG_KEEP_STAGES | (_capi.WT_PLAN_FLAG_UNFUSED if unfused else 0)
        self._plan_flags = f

    def set_strict_status(self, on: bool):
        """Device-side failures of a call (an activation beyond the f16 range of the split-f16 form; a persistent-LSTM step
        barrier that timed out) always overwrite that call's outputs (codes = -1, NaN) and surface as an error on the
        NEXT call on this model, whatever its shape (the library keeps one host-mapped status word per model beside the
        per-plan ones), which this class answers by falling back for good (fp32 GEMMs / launch-per-step LSTM for every
        plan of the model) and running that next call.  strict=True additionally synchronises after every call, checks, falls back and REPEATS
        the failed call itself, so no poisoned result is ever handed out (costs the host/GPU overlap between calls).
        Default (None): strict for calls of up to the graph batch limit (16 clips), asynchronous above."""
        self._strict = None if on is None else bool(on)

    def _is_strict(self, B: int) -> bool:
        if self._strict is None:
            return 0 < B <= self._graph_max_clips
        return self._strict

    def check_status(self):
        """Synchronise and raise WavTokError if any call since the last check failed on the device: failures still pending
        in the status words, and failures that a later call has already consumed and answered by falling back
        (self.fallback_events; in non-strict mode the failed call's poisoned outputs were handed out)."""
        dev = self._device()
        torch.cuda.current_stream(dev).synchronize()
        # the model's word first: it is what decides (every plan's guard step reports into it, and it outlives plans the
        # LRU has destroyed); the plans' own words only say which plans reported since the last check and may hold
        # failures that a later call has already consumed and answered by falling back
        mbits = ctypes.c_int32()
        if self._engine.model:
            check(lib.wt_model_status(self._engine.model, ctypes.byref(mbits), 1), "wt_model_status")
        seen = []
        for key, (plan, _ws) in self._engine.plans.items():
            bits = ctypes.c_int32()
            check(lib.wt_plan_status(plan, ctypes.byref(bits), 1), "wt_plan_status")
            if bits.value:
                seen.append((key, bits.value))
        if mbits.value & _capi.WT_STATUS_BIT_RANGE:
            self._answer_range()
        bad = (seen or [("model", mbits.value)]) if mbits.value else []
        events, self.fallback_events = self.fallback_events, []
        if events and not bad:
            raise WavTokError("device-side failures since the last check, already answered by a fallback: %s" % events)
        self._poll_bad_codes()
        if bad:
            raise WavTokError("device-side failure in earlier calls (plan key, status bits): %s; their outputs were "
                              "overwritten with -1 / NaN; later calls fall back (fp32 operands at the reporting site / step LSTM)" % bad)

    def _answer_range(self) -> str:
        """An S32 producer met |v| >= 65504.  The plans know which range sites reported (wt_plan_range_sites): the LOWEST
        one that is still on split-f16 operands goes to fp32 operands (its GEMMs then run on the fp32 MFMA chain; sites
        behind it usually report as well, because infinities propagate, and are left alone until they report on their
        own).  Without attribution (the plan is gone) the whole model falls back, as in round 3."""
        mask = 0
        for plan, _ws in self._engine.plans.values():
            m = ctypes.c_uint64()
            check(lib.wt_plan_range_sites(plan, ctypes.byref(m), 1), "wt_plan_range_sites")
            mask |= m.value
        new = mask & ~self._fp32_sites
        if new:
            site = (new & -new).bit_length() - 1
            self._fp32_sites |= 1 << site
            return "range site %d (%s) now keeps fp32 operands" % (site, site_name(site))
        self._plan_flags |= _capi.WT_PLAN_FLAG_FP32_GEMM
        return "no site attribution left: the whole model now runs fp32 GEMMs"

    def set_gemm_precision(self, mode: str):
        """"f16x3" (default): dense layers on the fp32-equivalent split-f16 MFMA kernel; "f32": the plain
        fp32 MFMA chain everywhere.  Both accumulate in fp32; measured error vs float64 is lower for f16x3."""
        if mode not in ("f16x3", "f32"):
            raise ValueError("mode must be 'f16x3' or 'f32'")
        self._plan_flags = (self._plan_flags | _capi.WT_PLAN_FLAG_FP32_GEMM) if mode == "f32" else \
            (self._plan_flags & ~_capi.WT_PLAN_FLAG_FP32_GEMM)

    def set_check_codes(self, mode: str):
        """How codes_to_features reports an index outside the codebook: "sync" (default: synchronise and raise for the
        offending call, like the reference), "deferred" (IndexError on the next call on this model or in check_status(), no
        stream synchronisation) or "off"."""
        if mode not in ("deferred", "sync", "off"):
            raise ValueError("mode must be 'deferred', 'sync' or 'off'")
        self._check_codes = mode

    @property
    def persistent_lstm(self) -> bool:
        """True while this model's plans may launch the persistent LSTM kernel (the library's own word: a 256-CU device and
        no lost-co-residency report so far)."""
        return bool(self._engine.model) and bool(lib.wt_model_persistent_lstm(self._engine.model))

    def _poll_bad_codes(self):
        if self._check_codes != "off" and self._engine.model and lib.wt_model_take_bad_codes(self._engine.model):
            raise IndexError("index out of range in self")

    def set_graph_max_clips(self, n: int):
        """Largest batch whose encode / decode plans are recorded and replayed as a hipGraph (default 16; 0 = never)."""
        self._graph_max_clips = int(n)

    def _graph_flags(self, B: int) -> int:
        if 0 < B <= self._graph_max_clips and not (self._plan_flags & _capi.WT_PLAN_FLAG_KEEP_STAGES):
            return self._plan_flags | _capi.WT_PLAN_FLAG_GRAPH
        return self._plan_flags

    def _guarded(self, dev: torch.device, get_plan, launch, strict: bool = False):
        """Runs launch(plan, ws) with the fallbacks for device-side failures (set_strict_status).  get_plan() builds the
        plan from the CURRENT flags and fp32 sites, so a fallback that changes them re-plans."""
        for attempt in range(8):
            plan, ws = get_plan()
            try:
                out = launch(plan, ws)
            except WavTokError as e:
                if e.status == _capi.WT_ERR_LSTM_SYNC and attempt < 7:
                    self.fallback_events.append("persistent LSTM lost co-residency in an earlier call (its outputs were poisoned): "
                                                "the model now runs the launch-per-step LSTM")
                    continue                                  # the model's plans now run the step LSTM
                if e.status == _capi.WT_ERR_RANGE and attempt < 7:
                    self.fallback_events.append("an earlier call left the f16 range of the split-f16 form (its outputs were "
                                                "poisoned): " + self._answer_range())
                    continue
                raise
            if not strict:
                return out
            torch.cuda.current_stream(dev).synchronize()
            bits = ctypes.c_int32()
            check(lib.wt_plan_status(plan, ctypes.byref(bits), 1), "wt_plan_status")
            if not bits.value:
                return out
            what = self._answer_range() if bits.value & _capi.WT_STATUS_BIT_RANGE else "launch-per-step LSTM"
            self.fallback_events.append("strict mode: the call failed on the device (status bits %d) and was repeated on the fallback path: %s" % (bits.value, what))
        raise WavTokError("the call kept failing on the device after the fp32 / step-LSTM fallbacks")

    def _sites(self, kind: int) -> int:
        """The fp32 range sites that matter to a plan kind (so that an overflow in the decoder does not re-plan the encoder)."""
        m = self._fp32_sites
        if kind in (_capi.WT_PLAN_ENCODE, _capi.WT_PLAN_UNIT_LSTM):
            return m & (1 << _capi.WT_SITE_ENCODER)
        if kind == _capi.WT_PLAN_HEAD:
            return m & (1 << _capi.WT_SITE_HEAD)
        if kind == _capi.WT_PLAN_SEANET_DECODER:
            return m & (1 << _capi.WT_SITE_SEANET_DECODER)
        return m & ~((1 << _capi.WT_SITE_ENCODER) | (1 << _capi.WT_SITE_SEANET_DECODER))

    def range_report(self, audio_input: torch.Tensor, bandwidth_id=None) -> List[Dict[str, Any]]:
        """How far every split-f16 (S32) operand of the dense layers sits below the f16 limit on THIS input with THESE
        weights: one encode_infer + decode pass on plans created with WT_PLAN_FLAG_RANGE_REPORT (same kernels; behind every
        step the largest magnitude of each S32 buffer the step touches is measured).  Returns a list of
        {"plan", "step", "buffer", "amax", "headroom_bits" = log2(65504 / amax)} in execution order; an entry with
        amax = inf marks a tensor that left the range (the call's outputs are then poisoned, as always)."""
        import math
        dev = self._ensure_engine()
        bw = self._bandwidth_index(bandwidth_id if bandwidth_id is not None else torch.tensor([0]))
        audio = self._as_input(audio_input, dev)
        B, T = audio.shape
        L = self._arch.frames(T)
        flags = (self._plan_flags | _capi.WT_PLAN_FLAG_RANGE_REPORT) & ~_capi.WT_PLAN_FLAG_GRAPH
        feats = torch.empty((B, 512, L), dtype=torch.float32, device=dev)
        codes = torch.empty((1, B, L), dtype=torch.int64, device=dev)
        wav = torch.empty((B, self._wave_len(L)), dtype=torch.float32, device=dev)
        out: List[Dict[str, Any]] = []

        def collect(plan, what):
            i = 0
            step, buf, amax = ctypes.c_char_p(), ctypes.c_char_p(), ctypes.c_float()
            while lib.wt_plan_range_report(plan, i, ctypes.byref(step), ctypes.byref(buf), ctypes.byref(amax)) == 0:
                a = float(amax.value)
                out.append({"plan": what, "step": step.value.decode(), "buffer": buf.value.decode(), "amax": a,
                            "headroom_bits": (math.log2(65504.0 / a) if 0.0 < a < float("inf") else (float("inf") if a == 0.0 else float("-inf")))})
                i += 1

        pe, wse = self._engine.plan(_capi.WT_PLAN_ENCODE, B, T, flags, dev, self._sites(_capi.WT_PLAN_ENCODE))
        check(lib.wt_encode(pe, _ptr(audio), _ptr(feats), _ptr(codes), _ptr(None), _ptr(wse), _stream_ptr(dev)), "wt_encode")
        collect(pe, "encode")
        # the decoder is measured on the features the codes select (finite even if the encoder's own report shows an overflow)
        torch.cuda.current_stream(dev).synchronize()
        bits = ctypes.c_int32()
        check(lib.wt_plan_status(pe, ctypes.byref(bits), 1), "wt_plan_status")
        check(lib.wt_model_status(self._engine.model, ctypes.byref(bits), 1), "wt_model_status")
        pd, wsd = self._engine.plan(_capi.WT_PLAN_DECODE, B, L, flags, dev, self._sites(_capi.WT_PLAN_DECODE))
        fin = feats if torch.isfinite(feats).all() else torch.zeros_like(feats)
        check(lib.wt_decode(pd, _ptr(fin), bw, _ptr(wav), _ptr(None), _ptr(wsd), _stream_ptr(dev)), "wt_decode")
        collect(pd, "decode")
        torch.cuda.current_stream(dev).synchronize()
        check(lib.wt_plan_status(pd, ctypes.byref(bits), 1), "wt_plan_status")
        check(lib.wt_model_status(self._engine.model, ctypes.byref(bits), 1), "wt_model_status")
        m = ctypes.c_uint64()
        for p in (pe, pd):
            check(lib.wt_plan_range_sites(p, ctypes.byref(m), 1), "wt_plan_range_sites")
        self._engine.drop(lambda k: k[3] & _capi.WT_PLAN_FLAG_RANGE_REPORT)
        return out

    def set_lstm_mode(self, mode: str):
        """"persistent" (default): the whole LSTM recurrence in one launch (per-XCD clip groups, weights resident);
        "step": one launch per time step."""
        if mode not in ("persistent", "step"):
            raise ValueError("mode must be 'persistent' or 'step'")
        self._plan_flags = (self._plan_flags | _capi.WT_PLAN_FLAG_STEP_LSTM) if mode == "step" else \
            (self._plan_flags & ~_capi.WT_PLAN_FLAG_STEP_LSTM)

    @property
    def arch(self) -> ArchConfig:
        return self._arch

    def _wave_len(self, L: int) -> int:
        """Samples per clip the ISTFT head returns for L frames (spectral_ops.py:43-47): 'same' L * hop, 'center' (L - 1) * hop."""
        return L * self._arch.hop_length if self._arch.padding == "same" else (L - 1) * self._arch.hop_length

    @property
    def hop_length(self) -> int:
        return self._arch.hop

    def _device(self) -> torch.device:
        return self.backbone.embed.weight.device

    def _ensure_engine(self) -> torch.device:
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("wavtokenizer_amd runs only on an AMD GPU: move the model with .to('cuda'). "
                               "There is no CPU path in this package.")
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if self._dirty or self._engine.device_index != idx:
            self._engine.load(self._arch, self.state_dict(), idx)
            self._dirty = False
        self._poll_bad_codes()          # an earlier codes_to_features met a bad index (deferred mode: found without a sync)
        return torch.device("cuda", idx)

    @staticmethod
    def _as_input(x: torch.Tensor, dev: torch.device, dtype=torch.float32) -> torch.Tensor:
        if x.device != dev:
            raise RuntimeError(f"input is on {x.device} but the model is on {dev}")
        return x.to(dtype).contiguous()

    # -- kernels ------------------------------------------------------------------------------------
    def _run_encode(self, audio: torch.Tensor, want_emb: bool = True):
        dev = self._ensure_engine()
        assert audio.dim() == 2, "expected audio of shape (B, T)"
        audio = self._as_input(audio, dev)
        B, T = audio.shape
        L = self._arch.frames(T)

        def launch(plan, ws):
            flags = self._graph_flags(B)
            if flags & _capi.WT_PLAN_FLAG_GRAPH:
                io = self._engine.staging(_capi.WT_PLAN_ENCODE, B, T, flags, lambda: {
                    "in": torch.empty((B, T), dtype=torch.float32, device=dev),
                    "feats": torch.empty((B, 512, L), dtype=torch.float32, device=dev),
                    "codes": torch.empty((1, B, L), dtype=torch.int64, device=dev),
                    "emb": torch.empty((B, 512, L), dtype=torch.float32, device=dev)}, dev, self._sites(_capi.WT_PLAN_ENCODE))
                io["in"].copy_(audio)
                check(lib.wt_encode(plan, _ptr(io["in"]), _ptr(io["feats"]), _ptr(io["codes"]), _ptr(io["emb"]), _ptr(ws),
                                    _stream_ptr(dev)), "wt_encode")
                return io["feats"].clone(), io["codes"].clone(), (io["emb"].clone() if want_emb else None)
            feats = torch.empty((B, 512, L), dtype=torch.float32, device=dev)
            codes = torch.empty((1, B, L), dtype=torch.int64, device=dev)
            emb = torch.empty((B, 512, L), dtype=torch.float32, device=dev) if want_emb else None
            check(lib.wt_encode(plan, _ptr(audio), _ptr(feats), _ptr(codes), _ptr(emb), _ptr(ws), _stream_ptr(dev)),
                  "wt_encode")
            return feats, codes, emb

        return self._guarded(dev, lambda: self._engine.plan(_capi.WT_PLAN_ENCODE, B, T, self._graph_flags(B), dev,
                                                            self._sites(_capi.WT_PLAN_ENCODE)), launch, self._is_strict(B))

    def _bandwidth_index(self, bandwidth_id) -> int:
        if bandwidth_id is None:
            raise AssertionError("bandwidth_id is required (decoder/models.py:227)")
        if isinstance(bandwidth_id, torch.Tensor):
            if bandwidth_id.numel() != 1:
                raise ValueError("bandwidth_id must hold one index (the reference broadcasts a (1, dim) embedding row)")
            if bandwidth_id.device.type == "cpu":
                return int(bandwidth_id.reshape(-1)[0])
            # a tensor on the GPU: reading it is a device synchronisation, so the value is remembered per tensor object and
            # version for callers that build it once and pass it to every call.  Tensors made under torch.inference_mode()
            # carry no version counter (reading _version raises): those are read every time, like the reference's
            # infer.py:60-62 does with the new tensor it builds per file
            if bandwidth_id.is_inference():
                return int(bandwidth_id.reshape(-1)[0])
            c = self._bw_cache
            if c is not None and c[0]() is bandwidth_id and c[1] == bandwidth_id._version:
                return c[2]
            v = int(bandwidth_id.reshape(-1)[0])
            self._bw_cache = (weakref.ref(bandwidth_id), bandwidth_id._version, v)
            return v
        return int(bandwidth_id)

    def _run_decode(self, features: torch.Tensor, bandwidth_id, want_backbone: bool = False):
        dev = self._ensure_engine()
        assert features.dim() == 3 and features.shape[1] == self._arch.input_channels, "expected features (B, 512, L)"
        bw = self._bandwidth_index(bandwidth_id)
        features = self._as_input(features, dev)
        B, _, L = features.shape
        cur_flags = lambda: self._graph_flags(B) if not want_backbone else self._plan_flags

        def launch(plan, ws):
            flags = cur_flags()
            if flags & _capi.WT_PLAN_FLAG_GRAPH:
                io = self._engine.staging(_capi.WT_PLAN_DECODE, B, L, flags, lambda: {
                    "in": torch.empty((B, self._arch.input_channels, L), dtype=torch.float32, device=dev),
                    "wav": torch.empty((B, self._wave_len(L)), dtype=torch.float32, device=dev)}, dev, self._sites(_capi.WT_PLAN_DECODE))
                io["in"].copy_(features)
                check(lib.wt_decode(plan, _ptr(io["in"]), bw, _ptr(io["wav"]), _ptr(None), _ptr(ws), _stream_ptr(dev)), "wt_decode")
                return io["wav"].clone(), None
            wav = torch.empty((B, self._wave_len(L)), dtype=torch.float32, device=dev)
            bb = torch.empty((B, L, self._arch.dim), dtype=torch.float32, device=dev) if want_backbone else None
            check(lib.wt_decode(plan, _ptr(features), bw, _ptr(wav), _ptr(bb), _ptr(ws), _stream_ptr(dev)), "wt_decode")
            return wav, bb

        return self._guarded(dev, lambda: self._engine.plan(_capi.WT_PLAN_DECODE, B, L, cur_flags(), dev, self._sites(_capi.WT_PLAN_DECODE)),
                             launch, self._is_strict(B))

    def _run_head(self, x: torch.Tensor) -> torch.Tensor:
        dev = self._ensure_engine()
        assert x.dim() == 3 and x.shape[2] == self._arch.dim, "expected the backbone output (B, L, dim)"
        x = self._as_input(x, dev)
        B, L, _ = x.shape

        def launch(plan, ws):
            wav = torch.empty((B, self._wave_len(L)), dtype=torch.float32, device=dev)
            check(lib.wt_head(plan, _ptr(x), _ptr(wav), _ptr(ws), _stream_ptr(dev)), "wt_head")
            return wav

        return self._guarded(dev, lambda: self._engine.plan(_capi.WT_PLAN_HEAD, B, L, self._plan_flags, dev, self._sites(_capi.WT_PLAN_HEAD)),
                             launch, self._is_strict(B))

    def _run_seanet_decoder(self, z: torch.Tensor) -> torch.Tensor:
        dev = self._ensure_engine()
        z = self._as_input(z, dev)
        B, _, L = z.shape

        def launch(plan, ws):
            out = torch.empty((B, 1, L * self._arch.hop), dtype=torch.float32, device=dev)
            check(lib.wt_seanet_decode(plan, _ptr(z), _ptr(out), _ptr(ws), _stream_ptr(dev)), "wt_seanet_decode")
            return out

        return self._guarded(dev, lambda: self._engine.plan(_capi.WT_PLAN_SEANET_DECODER, B, L, self._plan_flags, dev,
                                                            self._sites(_capi.WT_PLAN_SEANET_DECODER)), launch, self._is_strict(B))

    def _run_unit_lstm(self, x: torch.Tensor) -> torch.Tensor:
        """Unit tests: the encoder's SLSTM alone, x (B, L, 512) time-major -> lstm(x) + x, on the plan's kernels."""
        dev = self._ensure_engine()
        x = self._as_input(x, dev)
        B, L, _ = x.shape
        plan, ws = self._engine.plan(_capi.WT_PLAN_UNIT_LSTM, B, L, self._plan_flags, dev, self._sites(_capi.WT_PLAN_UNIT_LSTM))
        y = torch.empty_like(x)
        check(lib.wt_unit_run(plan, _ptr(x), _ptr(y), _ptr(ws), _stream_ptr(dev)), "wt_unit_run")
        return y

    def debug_stage(self, kind: int, B: int, length: int, name: str, rows: Optional[int] = None):
        """A named stage buffer of the cached plan (after a KEEP_STAGES run), decoded to fp32: returns (flat tensor,
        format bits); format & BUF_ELU says the buffer holds elu() of the reference's tensor.  S32 buffers are decoded
        (rows x cols with cols = numel / rows: pass `rows` for them)."""
        plan, ws = self._engine.plans[(kind, B, length, self._plan_flags)]
        off, n, fmt = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_int32()
        check(lib.wt_plan_buffer_info(plan, name.encode(), ctypes.byref(off), ctypes.byref(n), ctypes.byref(fmt)),
              "wt_plan_buffer_info")
        raw = ws[off.value: off.value + 4 * n.value]
        if fmt.value & _capi.BUF_S32:
            h = raw.view(torch.float16).view(-1, 2, 32).float()        # groups of [32 hi | 32 lo]
            return (h[:, 0, :] + h[:, 1, :] / 2048.0).reshape(-1), fmt.value
        return raw.view(torch.float32), fmt.value

    # -- reference API (pretrained.py:159-239) --------------------------------------------------------
    @torch.inference_mode()
    def forward(self, audio_input: torch.Tensor, **kwargs: Any) -> torch.Tensor:
        features, _, _ = self.feature_extractor(audio_input, **kwargs)
        return self.decode(features, **kwargs)

    @torch.inference_mode()
    def encode(self, audio_input: torch.Tensor, **kwargs: Any):
        features, discrete_codes, _ = self.feature_extractor(audio_input, **kwargs)
        return features, discrete_codes

    @torch.inference_mode()
    def encode_infer(self, audio_input: torch.Tensor, **kwargs: Any):
        features, discrete_codes, _ = self.feature_extractor.infer(audio_input, **kwargs)
        return features, discrete_codes

    @torch.inference_mode()
    def decode(self, features_input: torch.Tensor, **kwargs: Any) -> torch.Tensor:
        return self._run_decode(features_input, kwargs.get("bandwidth_id"))[0]

    @torch.inference_mode()
    def codes_to_features(self, codes: torch.Tensor) -> torch.Tensor:
        assert isinstance(self.feature_extractor, EncodecFeatures), \
            "Feature extractor should be an instance of EncodecFeatures"
        dev = self._ensure_engine()
        if codes.dim() == 2:
            codes = codes.unsqueeze(1)
        codes = self._as_input(codes, dev, torch.int64)
        K, B, L = codes.shape
        feats = torch.empty((B, 512, L), dtype=torch.float32, device=dev)
        check(lib.wt_codes_to_features(self._engine.model, _ptr(codes), K, B, L, _ptr(feats), _stream_ptr(dev)),
              "wt_codes_to_features")
        if self._check_codes == "sync":
            # F.embedding raises on an index outside the codebook (pretrained.py:236); the kernel flags it instead (and
            # writes NaN), which is read here after the (few microseconds of) work has completed
            torch.cuda.current_stream(dev).synchronize()
            self._poll_bad_codes()
        return feats
