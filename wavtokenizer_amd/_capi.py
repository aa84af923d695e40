"""ctypes binding of libwavtok_hip.so (see include/wavtokenizer_amd.h).

There is no CPU fallback: if the library is missing or fails to load, importing this
module raises, and every caller of the product path fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# WAVTOK_HIP_LIB names another build of the same library (A/B timing of two kernel versions); never a fallback
LIB_PATH = os.environ.get("WAVTOK_HIP_LIB") or os.path.join(_HERE, "libwavtok_hip.so")

# every symbol include/wavtokenizer_amd.h declares
EXPORTS = [
    "wt_last_error", "wt_version", "wt_model_create", "wt_model_destroy", "wt_model_export_bytes", "wt_model_export", "wt_packed_info", "wt_packed_verify", "wt_packed_bytes",
    "wt_model_create_packed", "wt_model_hop", "wt_model_weight_bytes",
    "wt_plan_create", "wt_plan_create_ex", "wt_plan_range_sites", "wt_plan_range_report", "wt_model_persistent_lstm", "wt_device_info", "wt_plan_destroy", "wt_plan_workspace_bytes", "wt_plan_frames", "wt_plan_num_launches", "wt_plan_graph_replays",
    "wt_plan_find_buffer", "wt_plan_buffer_info", "wt_plan_buffer_name", "wt_plan_status", "wt_plan_num_steps", "wt_plan_step_name",
    "wt_plan_set_timing", "wt_plan_read_timing", "wt_model_split_ok", "wt_model_status", "wt_model_take_bad_codes", "wt_encode", "wt_codes_to_features",
    "wt_decode", "wt_seanet_decode", "wt_head", "wt_unit_run", "wt_sconv1d", "wt_linear", "wt_conv1d_s32", "wt_vq_workspace_bytes",
    "wt_vq_nearest", "wt_vq_nearest_f32", "wt_resblock", "wt_resblock_down",
    "wt_resampler_create", "wt_resampler_destroy", "wt_resampler_out_length", "wt_convert_audio", "wt_pcm16",
    "wt_linear_overlap_add",
]

WT_PLAN_ENCODE, WT_PLAN_DECODE, WT_PLAN_SEANET_DECODER, WT_PLAN_HEAD, WT_PLAN_UNIT_LSTM = 0, 1, 2, 3, 4
WT_PLAN_FLAG_KEEP_STAGES = 1
WT_PLAN_FLAG_FP32_GEMM = 2
WT_PLAN_FLAG_STEP_LSTM = 4
WT_PLAN_FLAG_GRAPH = 8
WT_PLAN_FLAG_UNFUSED = 16
WT_PLAN_FLAG_RANGE_REPORT = 32
WT_SITE_ENCODER, WT_SITE_BB_EMBED, WT_SITE_RES0, WT_SITE_RES1, WT_SITE_ATTN, WT_SITE_RES2, WT_SITE_RES3 = 0, 1, 2, 3, 4, 5, 6
WT_SITE_CNX0, WT_SITE_HEAD, WT_SITE_SEANET_DECODER = 7, 40, 41
WT_ERR_RANGE, WT_ERR_LSTM_SYNC, WT_ERR_INDEX = -6, -7, -8
WT_STATUS_BIT_LSTM, WT_STATUS_BIT_RANGE = 1, 2
BUF_S32, BUF_ELU = 1, 2


class WtArch(ctypes.Structure):
    _fields_ = [("n_ratios", c_int32), ("ratios", c_int32 * 8), ("vq_bins", c_int32), ("num_quantizers", c_int32),
                ("input_channels", c_int32), ("dim", c_int32), ("intermediate_dim", c_int32),
                ("num_layers", c_int32), ("adanorm_num_embeddings", c_int32), ("n_fft", c_int32),
                ("hop_length", c_int32), ("padding_same", c_int32)]


class WtTensor(ctypes.Structure):
    _fields_ = [("name", c_char_p), ("data", POINTER(c_float)), ("numel", c_int64)]


class WavTokError(RuntimeError):
    def __init__(self, msg, status=0):
        super().__init__(msg)
        self.status = status


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C wavtokenizer_amd/csrc). There is no CPU fallback for the product path.")
    lib = ctypes.CDLL(LIB_PATH)
    if os.environ.get("WAVTOK_HIP_LIB"):
        # A/B timing against an older build of the library (tools/build_prev.sh): entry points it does not have yet are
        # replaced by a stub that fails when called, so that the timing tools can still bind the ones they use
        class _Missing:
            def __init__(self, name):
                self.name, self.argtypes, self.restype = name, None, None

            def __call__(self, *a):
                raise WavTokError(f"{LIB_PATH} has no {self.name}")
        for sym in EXPORTS:
            try:
                getattr(lib, sym)
            except AttributeError:
                setattr(lib, sym, _Missing(sym))
    lib.wt_last_error.restype = c_char_p
    lib.wt_version.restype = c_char_p
    lib.wt_model_create.argtypes = [POINTER(WtArch), POINTER(WtTensor), c_int32, c_int32, POINTER(c_void_p)]
    lib.wt_model_destroy.argtypes = [c_void_p]
    lib.wt_model_destroy.restype = None
    lib.wt_model_export_bytes.argtypes = [c_void_p]
    lib.wt_model_export_bytes.restype = c_size_t
    lib.wt_model_export.argtypes = [c_void_p, c_void_p, c_size_t]
    lib.wt_packed_info.argtypes = [c_void_p, c_size_t, POINTER(WtArch), POINTER(c_int32), POINTER(ctypes.c_uint64)]
    lib.wt_packed_verify.argtypes = [c_void_p, c_size_t]
    lib.wt_packed_bytes.argtypes = [c_void_p, c_size_t]
    lib.wt_packed_bytes.restype = c_size_t
    lib.wt_model_create_packed.argtypes = [c_void_p, c_size_t, c_int32, POINTER(c_void_p)]
    lib.wt_model_hop.argtypes = [c_void_p]
    lib.wt_model_weight_bytes.argtypes = [c_void_p]
    lib.wt_model_weight_bytes.restype = c_int64
    lib.wt_plan_create.argtypes = [c_void_p, c_int32, c_int32, c_int64, c_int32, POINTER(c_void_p)]
    lib.wt_plan_create_ex.argtypes = [c_void_p, c_int32, c_int32, c_int64, c_int32, ctypes.c_uint64, POINTER(c_void_p)]
    lib.wt_plan_range_sites.argtypes = [c_void_p, POINTER(ctypes.c_uint64), c_int32]
    lib.wt_plan_range_report.argtypes = [c_void_p, c_int32, POINTER(c_char_p), POINTER(c_char_p), POINTER(c_float)]
    lib.wt_model_persistent_lstm.argtypes = [c_void_p]
    lib.wt_device_info.argtypes = [c_int32, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]
    lib.wt_plan_destroy.argtypes = [c_void_p]
    lib.wt_plan_destroy.restype = None
    lib.wt_plan_workspace_bytes.argtypes = [c_void_p]
    lib.wt_plan_workspace_bytes.restype = c_size_t
    lib.wt_plan_frames.argtypes = [c_void_p]
    lib.wt_plan_frames.restype = c_int64
    lib.wt_plan_num_launches.argtypes = [c_void_p]
    lib.wt_plan_graph_replays.argtypes = [c_void_p]
    lib.wt_plan_graph_replays.restype = c_int64
    lib.wt_plan_find_buffer.argtypes = [c_void_p, c_char_p, POINTER(c_size_t), POINTER(c_size_t)]
    lib.wt_plan_buffer_info.argtypes = [c_void_p, c_char_p, POINTER(c_size_t), POINTER(c_size_t), POINTER(c_int32)]
    lib.wt_plan_status.argtypes = [c_void_p, POINTER(c_int32), c_int32]
    lib.wt_model_split_ok.argtypes = [c_void_p]
    lib.wt_model_status.argtypes = [c_void_p, POINTER(c_int32), c_int32]
    lib.wt_model_take_bad_codes.argtypes = [c_void_p]
    lib.wt_unit_run.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.wt_plan_buffer_name.argtypes = [c_void_p, c_int32, POINTER(c_char_p)]
    lib.wt_plan_num_steps.argtypes = [c_void_p]
    lib.wt_plan_step_name.argtypes = [c_void_p, c_int32, POINTER(c_char_p)]
    lib.wt_plan_set_timing.argtypes = [c_void_p, c_char_p]
    lib.wt_plan_read_timing.argtypes = [c_void_p, POINTER(ctypes.c_double), POINTER(c_int64), c_int32]
    lib.wt_encode.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.wt_codes_to_features.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int64, c_void_p, c_void_p]
    lib.wt_decode.argtypes = [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.wt_seanet_decode.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.wt_head.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.wt_sconv1d.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_int32,
                               c_int32, c_int32, c_int32, c_void_p]
    lib.wt_linear.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p]
    lib.wt_conv1d_s32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_int32,
                                  c_int32, c_int32, c_void_p, c_void_p]
    lib.wt_vq_workspace_bytes.argtypes = [c_int64, c_int32, c_int32]
    lib.wt_vq_workspace_bytes.restype = c_size_t
    lib.wt_vq_nearest.argtypes = [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]
    lib.wt_vq_nearest_f32.argtypes = [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]
    lib.wt_resblock.argtypes = [c_void_p] * 11 + [c_int32, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p]
    lib.wt_resblock_down.argtypes = [c_void_p] * 12 + [c_int32, c_int64, c_int32, c_void_p]
    lib.wt_resampler_create.argtypes = [c_int32, c_int32, c_int32, POINTER(c_void_p)]
    lib.wt_resampler_destroy.argtypes = [c_void_p]
    lib.wt_resampler_destroy.restype = None
    lib.wt_resampler_out_length.argtypes = [c_void_p, c_int64]
    lib.wt_resampler_out_length.restype = c_int64
    lib.wt_convert_audio.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int64, c_void_p, c_void_p]
    lib.wt_pcm16.argtypes = [c_void_p, c_int64, ctypes.c_float, c_int32, c_void_p, c_void_p, c_void_p]
    lib.wt_linear_overlap_add.argtypes = [c_void_p, c_void_p, c_int32, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p]
    return lib


lib = _load()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.wt_last_error()
        raise WavTokError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}", rc)
