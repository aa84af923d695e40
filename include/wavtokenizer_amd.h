/*
 * wavtokenizer_amd.h — C-ABI of libwavtok_hip.so (MI355X / gfx950).
 *
 * The reference (Rita-zi/WavTokenizer) is 100 % Python; it has no FFI boundary of its own.
 * The drop-in boundary is the Python class decoder/pretrained.py:32 `WavTokenizer`; this
 * library is what our same-named class binds (ctypes) underneath.  Each entry point cites the
 * reference function whose work it performs.  Conventions:
 *
 *   - plain C: raw device pointers, explicit sizes, `void* stream` = hipStream_t
 *     (NULL = the null stream).  No torch types.
 *   - every function returns 0 on success, a negative wt_status otherwise; nothing throws
 *     across the ABI; wt_last_error() gives a thread-local message.
 *   - the caller owns every buffer, including the workspace (size from
 *     wt_plan_workspace_bytes).  Kernels are enqueued on `stream`; nothing synchronises,
 *     allocates or frees inside wt_encode / wt_decode / wt_codes_to_features.
 *   - a wt_model is immutable after creation (folded + packed weights in HBM) and may be shared
 *     by host threads.  A wt_plan may be shared too: host calls on one plan are serialised by a
 *     per-plan lock (they only enqueue), and calls that may be in flight on the GPU at the same
 *     time (two streams) need a plan and a workspace each (stage buffers and the call's status
 *     word live in the workspace).  Calls that launch the persistent LSTM kernel are chained per
 *     device by the library: one made on another stream than the previous one first waits on the
 *     GPU for that previous call (two such launches must never share the CUs).
 *   - a call that fails ON THE DEVICE (wt_status_bits) never hands out plausible data: the guard
 *     step that ends every plan overwrites its outputs (codes = -1, floats = NaN), and the next
 *     host call on ANY plan of the same model returns the matching error once, without running
 *     (wt_plan_status, wt_model_status): a caller that makes a new plan per input length still meets it.
 *   - activations inside the library are time-major [clip][frame][channel] fp32; the API
 *     tensors keep the reference's layouts (wav (B,T); features (B,512,L); codes (K,B,L) int64).
 */
#ifndef WAVTOKENIZER_AMD_H
#define WAVTOKENIZER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    WT_OK = 0,
    WT_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
    WT_ERR_MISSING_TENSOR = -2,
    WT_ERR_SHAPE = -3,
    WT_ERR_HIP = -4,          /* a HIP runtime call failed */
    WT_ERR_NOT_INITED = -5,   /* codebook buffer `inited` != 1 (core_vq.py:140-151 would run k-means) */
    WT_ERR_RANGE = -6,        /* an EARLIER call on this model met a value outside the f16 range of the split-f16 form; its
                                 outputs were poisoned; re-plan with WT_PLAN_FLAG_FP32_GEMM and repeat both calls */
    WT_ERR_LSTM_SYNC = -7,    /* an EARLIER call's persistent LSTM lost co-residency (a step barrier timed out); its outputs
                                 were poisoned; every plan of the model now runs the LSTM one launch per step: repeat both calls */
    WT_ERR_INDEX = -8         /* a code outside [0, bins) (F.embedding raises IndexError: decoder/pretrained.py:236) */
} wt_status;

/* Device-side failure bits of a call (wt_plan_status). */
enum wt_status_bits {
    WT_STATUS_BIT_LSTM = 1,   /* persistent LSTM: a step barrier timed out */
    WT_STATUS_BIT_RANGE = 2   /* an S32 (split-f16) producer met |v| >= 65504 */
};

/* Architecture: the YAML keys decoder/pretrained.py:81-92 (from_hparams0802) reads. */
typedef struct {
    int32_t n_ratios;
    int32_t ratios[8];               /* `dowmsamples`, decoder order as written in the YAML */
    int32_t vq_bins;                 /* 4096 */
    int32_t num_quantizers;          /* 1 */
    int32_t input_channels;          /* 512 */
    int32_t dim;                     /* 768 */
    int32_t intermediate_dim;        /* 2304 */
    int32_t num_layers;              /* 12 */
    int32_t adanorm_num_embeddings;  /* 4 */
    int32_t n_fft;                   /* 2400 / 1280 */
    int32_t hop_length;              /* 600 / 320 */
    int32_t padding_same;            /* ISTFT padding (spectral_ops.py:33-47): 1 = "same" (every YAML; audio = L * hop samples), 0 = "center"
                                        (torch.istft(center=True): (L - 1) * hop samples per clip, L >= 2) */
} wt_arch;

/* One named fp32 host array of a checkpoint `state_dict` (keys: SURVEY.md Appendix A). */
typedef struct {
    const char*  name;
    const float* data;    /* host pointer, fp32, contiguous */
    int64_t      numel;
} wt_tensor;

typedef struct wt_model wt_model;
typedef struct wt_plan  wt_plan;

typedef enum {
    WT_PLAN_ENCODE = 0,          /* audio (B,T)       -> features (B,512,L) + codes (1,B,L)  */
    WT_PLAN_DECODE = 1,          /* features (B,512,L) -> audio (B, L*hop)                     */
    WT_PLAN_SEANET_DECODER = 2,  /* features (B,512,L) -> audio (B,1,L*hop): encodec.decoder   */
    WT_PLAN_HEAD = 3,            /* backbone output (B,L,dim) -> audio (B, L*hop): model.head   */
    WT_PLAN_UNIT_LSTM = 4        /* unit tests: x [B][L][512] -> SLSTM(x) [B][L][512], time-major (wt_unit_run) */
} wt_plan_kind;

enum {
    WT_PLAN_FLAG_KEEP_STAGES = 1,  /* never alias stage buffers and snapshot the in-place residual stream (debug taps; bigger
                                      workspace).  The kernels are the ones the default plan launches: taps of the shipped path;
                                      a tap may therefore hold S32-encoded and / or ELU-applied data (wt_plan_buffer_info) */
    WT_PLAN_FLAG_UNFUSED = 16,     /* debug twin: unfused stages, raw fp32 tensors between them (first conv, three-GEMM
                                      resblocks, consumer-side ELU); with FP32_GEMM it is the plain fp32 restatement */
    WT_PLAN_FLAG_STEP_LSTM = 4,    /* LSTM as one launch per time step instead of the persistent per-XCD kernel */
    WT_PLAN_FLAG_GRAPH = 8,        /* small batches: the second call in a row with the same buffers records the plan's
                                      launches as a hipGraph, later calls with those buffers replay it (one
                                      hipGraphLaunch on the caller's stream); any other call launches directly.  Such a
                                      plan must not be run from two host threads at once */
    WT_PLAN_FLAG_FP32_GEMM = 2,    /* every dense layer on the fp32 MFMA chain; default: the fp32-equivalent
                                      split-f16 kernel (3 f16 MFMAs per product, fp32 accumulate) where covered */
    WT_PLAN_FLAG_RANGE_REPORT = 32 /* diagnostic: behind every step, the largest magnitude held by each S32 (split-f16) buffer the
                                      step touches is measured (wt_plan_range_report): the head-room of every dense layer's
                                      operands below the f16 limit 65504.  Costs a pass per buffer; never graph-replayed */
};

/* Range sites: the units in which a plan can leave the split-f16 form on its own (wt_plan_create_ex, wt_plan_range_sites).
 * Every S32 tensor is produced and consumed inside one site.  Bit i of a site mask = site i. */
enum wt_range_site {
    WT_SITE_ENCODER = 0,       /* the whole encode plan: SEANetEncoder + VQ (encoder/modules/seanet.py:66-144) */
    WT_SITE_BB_EMBED = 1,      /* backbone.embed (decoder/models.py:177) */
    WT_SITE_RES0 = 2, WT_SITE_RES1 = 3,   /* pos_net ResnetBlocks (decoder/models.py:19-78) */
    WT_SITE_ATTN = 4,          /* AttnBlock (decoder/models.py:80-127) */
    WT_SITE_RES2 = 5, WT_SITE_RES3 = 6,
    WT_SITE_CNX0 = 7,          /* ConvNeXtBlock i = WT_SITE_CNX0 + i (decoder/modules.py:8-60), i < 32 */
    WT_SITE_HEAD = 40,         /* final_layer_norm output + ISTFTHead (decoder/heads.py:24-67) */
    WT_SITE_SEANET_DECODER = 41
};

const char* wt_last_error(void);
const char* wt_version(void);

/* Replaces: WavTokenizer.from_pretrained0802's load_state_dict (decoder/pretrained.py:95-114)
 * plus the weight_norm pre-forward hook of every SConv1d (encoder/modules/conv.py:25-34), done
 * once here: w = g * v / ||v||.  Copies, folds and packs the weights into HBM on `device`. */
int  wt_model_create(const wt_arch* arch, const wt_tensor* tensors, int32_t n_tensors, int32_t device,
                     wt_model** out);
void wt_model_destroy(wt_model* m);

/* Packed weight image ("a packed on-disk weight format, folded, pre-tiled, for fast start"): everything
 * wt_model_create computes and leaves in HBM — weight-norm folded conv weights in [Cout][tap][Cin], LSTM gate-row
 * packings, the packed ISTFT head and inverse-DFT basis, the S32 split copies with their per-tensor scales —
 * behind a header (magic, layout version, the wt_arch, a hash of both, a hash of everything behind the header).
 * wt_model_create_packed allocates, uploads and fixes up pointers: nothing is folded, packed or split again (replaces
 * the load_state_dict + per-forward weight_norm of decoder/pretrained.py:95-114 a second time over).  The file is not
 * trusted: sizes are checked without wrap-around, element counts are bounded, every pointer offset must lie inside
 * its allocation, and the content hash must match before anything is uploaded.  wt_packed_info validates a header
 * and wt_packed_verify the whole image (bounds + content hash) without a GPU. */
size_t wt_model_export_bytes(const wt_model* m);
int  wt_model_export(const wt_model* m, void* buf, size_t n);
int  wt_packed_info(const void* buf, size_t n, wt_arch* arch, int32_t* version, uint64_t* arch_hash);
int  wt_packed_verify(const void* buf, size_t n);
/* exact length of the image that starts at buf (wt_model_export_bytes is an upper bound); 0 when the header is not valid */
size_t wt_packed_bytes(const void* buf, size_t n);
int  wt_model_create_packed(const void* buf, size_t n, int32_t device, wt_model** out);
int  wt_model_hop(const wt_model* m);                 /* prod(ratios) */
/* 1 while the model's plans may launch the persistent LSTM kernel (a 256-CU device and no lost-co-residency report so far) */
int  wt_model_persistent_lstm(const wt_model* m);
/* What the library sees of a device: compute units, whether it is gfx950 (wt_model_create refuses anything else), whether
 * the persistent LSTM can run there (256 CUs = 8 XCDs x 32; a compute partition runs the launch-per-step kernel and sizes
 * every persistent launch by its own CU count). */
int  wt_device_info(int32_t device, int32_t* compute_units, int32_t* is_gfx950, int32_t* persistent_lstm);
int64_t wt_model_weight_bytes(const wt_model* m);     /* packed fp32 bytes resident in HBM */

/* A plan fixes (kind, B, T or L): kernel launch list + workspace layout.
 * `len` = T (samples) for WT_PLAN_ENCODE, L (frames) for the two decode kinds. */
int    wt_plan_create(const wt_model* m, int32_t kind, int32_t B, int64_t len, int32_t flags, wt_plan** out);
/* ... with a mask of range sites (wt_range_site) that keep fp32 operands and run their GEMMs on the fp32 MFMA chain while
 * every other site stays on the split-f16 kernel: the answer to WT_ERR_RANGE that costs one block, not the model. */
int    wt_plan_create_ex(const wt_model* m, int32_t kind, int32_t B, int64_t len, int32_t flags, uint64_t fp32_sites,
                         wt_plan** out);
void   wt_plan_destroy(wt_plan* p);
size_t wt_plan_workspace_bytes(const wt_plan* p);
int64_t wt_plan_frames(const wt_plan* p);             /* L = ceil(T / hop) (conv.py:54-61) */
int    wt_plan_num_launches(const wt_plan* p);
/* calls of this plan that were served by a graph replay (WT_PLAN_FLAG_GRAPH); 0 for a plan without the flag */
int64_t wt_plan_graph_replays(const wt_plan* p);
/* Debug taps: byte offset / element count of a named stage buffer inside the workspace. */
int    wt_plan_find_buffer(const wt_plan* p, const char* name, size_t* offset, size_t* numel);
/* ... and what it holds: format bit 0: the S32 split-f16 encoding (every 32 values of a row = 128 bytes
 * [32 x f16 hi | 32 x f16 lo], value = hi + lo * 2^-11) instead of fp32; bit 1: elu() of the reference's tensor. */
int    wt_plan_buffer_info(const wt_plan* p, const char* name, size_t* offset, size_t* numel, int32_t* format);
/* Device-side failure bits (wt_status_bits) that calls on this plan have reported since the last clear.  The bits of
 * a call are visible once its stream work has completed: synchronise first to learn about the call just made.
 * clear != 0 consumes them together with the model's word (and switches the model to the launch-per-step LSTM after
 * WT_STATUS_BIT_LSTM); bits left unconsumed make the next wt_encode / wt_decode / ... return WT_ERR_LSTM_SYNC /
 * WT_ERR_RANGE once. */
int    wt_plan_status(const wt_plan* p, int32_t* bits, int32_t clear);
/* Which range sites of this plan reported WT_STATUS_BIT_RANGE since the last clear (a site downstream of the first one
 * that overflowed usually reports too: infinities propagate; put the LOWEST set site on fp32 and run again). */
int    wt_plan_range_sites(const wt_plan* p, uint64_t* sites, int32_t clear);
/* WT_PLAN_FLAG_RANGE_REPORT plans, after a call: entry `index` of the report = (step name, S32 buffer name, largest
 * magnitude found in that buffer right behind that step; +inf if a value left the f16 range).  Synchronises the device on the
 * first query after a call.  Returns WT_ERR_INVALID past the last entry. */
int    wt_plan_range_report(const wt_plan* p, int32_t index, const char** step, const char** buffer, float* amax);
/* The same bits collected over ALL plans of the model (each plan's guard step reports into this word too, and it
 * outlives plans that were destroyed): what the next call on any plan of the model will consume. */
int    wt_model_status(const wt_model* m, int32_t* bits, int32_t clear);
/* 1 when every GEMM weight fits the split-f16 form; 0: all plans of this model run the fp32 MFMA chain. */
int    wt_model_split_ok(const wt_model* m);
int    wt_plan_buffer_name(const wt_plan* p, int32_t index, const char** name);

/* Measurement hook (bench.py's roofline leg; no counterpart in the reference): HIP events are
 * recorded on the call's stream around every step whose name contains `name_substr` ("" / NULL
 * turns it off).  A leading '@' ("@cnx.pwconv1") uses no events: the step's gemm16s launch
 * records its own duration on the device (every workgroup takes the constant 100 MHz clock on
 * entry and exit; earliest entry to latest exit), so no packet is added between the launches (an
 * event record costs 4-7 us there and is counted into the bracket).  wt_plan_read_timing waits
 * for the work and returns the summed milliseconds and the number of timed launches since the
 * last reset.  Not thread-safe. */
int wt_plan_num_steps(const wt_plan* p);
int wt_plan_step_name(const wt_plan* p, int32_t index, const char** name);
int wt_plan_set_timing(const wt_plan* p, const char* name_substr);
int wt_plan_read_timing(const wt_plan* p, double* total_ms, int64_t* launches, int32_t reset);

/* Replaces: WavTokenizer.encode_infer (decoder/pretrained.py:186-189) ->
 * EncodecFeatures.infer (decoder/feature_extractors.py:131-142): SEANetEncoder
 * (encoder/modules/seanet.py:143) + ResidualVectorQuantizer.infer (encoder/quantization/vq.py:115-140).
 *   wav      [B][T] fp32 (device)
 *   features [B][512][L] fp32 (device, out) — quantized embedding
 *   codes    [1][B][L] int64 (device, out)
 *   emb_out  optional [B][512][L] fp32: encoder output before quantisation (may be NULL) */
int wt_encode(const wt_plan* p, const float* wav, float* features, int64_t* codes, float* emb_out,
              void* workspace, void* stream);

/* Replaces: WavTokenizer.codes_to_features (decoder/pretrained.py:209-239).
 *   codes [K][B][L] int64, features [B][512][L] fp32; K <= num_quantizers. */
int wt_codes_to_features(const wt_model* m, const int64_t* codes, int32_t K, int32_t B, int64_t L,
                         float* features, void* stream);

/* Replaces: WavTokenizer.decode (decoder/pretrained.py:192-207): VocosBackbone.forward
 * (decoder/models.py:223-235) + ISTFTHead.forward (decoder/heads.py:42-67) + ISTFT.forward
 * (decoder/spectral_ops.py:33-75).
 *   features [B][512][L] fp32, bandwidth_id in [0, adanorm_num_embeddings), wav_out [B][L*hop] (padding "same";
 *   "center": [B][(L-1)*hop], L >= 2).
 *   backbone_out optional [B][L][dim] fp32 (may be NULL). */
int wt_decode(const wt_plan* p, const float* features, int32_t bandwidth_id, float* wav_out,
              float* backbone_out, void* workspace, void* stream);

/* Replaces: ISTFTHead.forward (decoder/heads.py:42-67) + ISTFT.forward (decoder/spectral_ops.py:33-75) on its own,
 * reached by callers as model.head(x).  x [B][L][dim] fp32 (the backbone output), wav_out [B][L*hop] ("center":
 * [B][(L-1)*hop]). */
int wt_head(const wt_plan* p, const float* x, float* wav_out, void* workspace, void* stream);

/* Replaces: SEANetDecoder.forward (encoder/modules/seanet.py:236-238), reached by callers as
 * model.feature_extractor.encodec.decoder(features).  wav_out [B][1][L*hop]. */
int wt_seanet_decode(const wt_plan* p, const float* features, float* wav_out, void* workspace, void* stream);

/* ---- single-stage entry points (unit parity tests) -------------------------------------------
 * Shipped kernels (what the default plans launch): wt_linear modes 2/3, wt_conv1d_s32, wt_vq_nearest (gemm16s.hip +
 * vq_finalize), wt_resblock (resblock16.hip), WT_PLAN_UNIT_LSTM (lstm_persist.hip / the step kernel).  fp32 twins
 * (gemm.hip, resblock.hip: the WT_PLAN_FLAG_FP32_GEMM path): wt_sconv1d, wt_linear mode 0, wt_vq_nearest_f32,
 * wt_resblock with fp32_chain = 1.  The S32 entry points scale each operand by a per-tensor power of two chosen on
 * the device (the plans' producers write S32 unscaled and report |v| >= 65504 through the status word instead). */

/* Replaces: SConv1d.forward with weight-normed Conv1d (encoder/modules/conv.py:195-211), time-major
 * tensors: x [B][T][Cin] -> y [B][Tout][Cout], w [Cout][k][Cin] (already folded), reflect padding,
 * optional ELU on the input (seanet.py:49,124,136).  Tout = ceil(T/stride).  fp32 MFMA chain (gemm.hip). */
int wt_sconv1d(const float* x, const float* w, const float* bias, float* y, int32_t B, int64_t T,
               int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t dilation, int32_t elu_input,
               void* stream);

/* Replaces: nn.Linear.forward as used by ConvNeXtBlock.pwconv1/2 (decoder/modules.py:52,54): y [M][N] =
 * x [M][K] . w[N][K]^T + bias.  f16x3 = 0: fp32 MFMA chain; 1: removed (round 1's in-loop split kernel); 2: the
 * fp32-equivalent split-f16 arithmetic (x = hi + lo * 2^-11, three f16 MFMAs per product, fp32 accumulation) on
 * pre-split "S32" operands staged by LDS-DMA (gemm16s.hip: the plans' producers write S32 directly; here x and w
 * are split into the workspace first, 4*(M+N)*K bytes, K % 32 == 0); 3: as 2 and y is written in S32 too
 * (N % 32 == 0; every 32 outputs of a row = 128 bytes [32 x f16 hi | 32 x f16 lo], value = hi + lo * 2^-11);
 * 4: as 3 with the exact-erf GELU of decoder/modules.py:53 applied first (pwconv1's epilogue as the decode plan runs it).
 * Modes 2 / 3 need 8 KB more workspace (per-tensor scales; clock stamps of the timing-experiment builds). */
int wt_linear(const float* x, const float* w, const float* bias, float* y, int64_t M, int32_t N, int32_t K,
              int32_t f16x3, void* workspace, void* stream);

/* Conv1d on the S32 split-f16 kernel (gemm16s.hip), time-major x [B][T][Cin] -> y [B][Tout][Cout] fp32,
 * w [Cout][k][Cin]: zero_same = 1: nn.Conv1d(k, padding=(k-1)/2) as in decoder/models.py:29-43,177 (stride 1);
 * zero_same = 0: SConv1d reflect padding (conv.py:195-211).  Cin % 32 == 0; workspace 4*(B*T*Cin + Cout*k*Cin) + 256 bytes. */
int wt_conv1d_s32(const float* x, const float* w, const float* bias, float* y, int32_t B, int64_t T, int32_t Cin,
                  int32_t Cout, int32_t k, int32_t stride, int32_t zero_same, void* workspace, void* stream);

/* Replaces: EuclideanCodebook.quantize (encoder/quantization/core_vq.py:175-183): x [N][D] rows,
 * embed [bins][D]; codes_out [N] int64 = argmax_j -(|x|^2 - 2 x.e_j + |e_j|^2), ties -> lowest j.
 * wt_vq_nearest: the encoder plan's kernels (distances on gemm16s.hip with the per-slab argmax epilogue, then
 * vq_finalize; D % 32 == 0); wt_vq_nearest_f32: the fp32 MFMA chain.  workspace: wt_vq_workspace_bytes(N, D, bins). */
size_t wt_vq_workspace_bytes(int64_t N, int32_t D, int32_t bins);
int wt_vq_nearest(const float* x, const float* embed, int64_t N, int32_t D, int32_t bins, int64_t* codes_out,
                  void* workspace, void* stream);
int wt_vq_nearest_f32(const float* x, const float* embed, int64_t N, int32_t D, int32_t bins, int64_t* codes_out,
                      void* workspace, void* stream);

/* Replaces: SEANetResnetBlock.forward (encoder/modules/seanet.py:62-63): y = shortcut(x) + conv1(elu(conv3(elu(x)))),
 * one fused launch, time-major x [B][T][C] -> y [B][T][C], C = 32 or 64; folded weights w3 [C/2][3][C], w1 [C][C/2],
 * ws [C][C].  wav != NULL (C = 32): x is not read; the tile is built from the waveform wav [B][T] through
 * SEANetEncoder.model[0] (seanet.py:107-110; e0_w [7][32], e0_b [32]).  elu_out: y = elu(.);  out_s32: y in the S32
 * encoding;  fp32_chain = 0: resblock16.hip (split-f16 MFMAs, the shipped kernel), 1: resblock.hip. */
int wt_resblock(const float* x, const float* wav, const float* e0_w, const float* e0_b, const float* w3, const float* b3,
                const float* w1, const float* b1, const float* ws, const float* bs, float* y, int32_t B, int64_t T,
                int32_t C, int32_t elu_out, int32_t out_s32, int32_t fp32_chain, void* stream);
/* wt_resblock's shipped stage-1 form: first conv + resblock + ELU + the stage's strided conv (seanet.py:123-127) in one
 * launch.  wd [64][2r][32] (out, tap, in), y_down [B][ceil(T / r)][64] fp32; r in {2, 4}, T >= 1024. */
int wt_resblock_down(const float* wav, const float* e0_w, const float* e0_b, const float* w3, const float* b3, const float* w1,
                     const float* b1, const float* ws, const float* bs, const float* wd, const float* bd, float* y_down,
                     int32_t B, int64_t T, int32_t r, void* stream);

/* Runs a WT_PLAN_UNIT_LSTM plan: x, y [B][L][512] fp32 time-major; y = SLSTM(x) (encoder/modules/lstm.py:31-39) with
 * the encoder's LSTM weights of the plan's model. */
int wt_unit_run(const wt_plan* p, const float* x, float* y, void* workspace, void* stream);

/* After wt_codes_to_features has completed on its stream: 1 if a call since the last query met a code outside
 * [0, bins) (the frames it touched were written as NaN), else 0; the flag is cleared. */
int wt_model_take_bad_codes(const wt_model* m);

/* ---- helpers on either side of the hot path ------------------------------------- */

/* Replaces: convert_audio (encoder/utils.py:79-92) with target_channels = 1: mean over the C (1 or 2) channels, then
 * torchaudio.transforms.Resample(orig_sr, new_sr) (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99; torchaudio
 * is not in this image: restated from its published algorithm, parity unpinned).  wav [B][C][T] -> out [B][T'],
 * T' = wt_resampler_out_length(T) = ceil(new * T / orig).  Equal rates give the plain channel mean. */
typedef struct wt_resampler wt_resampler;
int     wt_resampler_create(int32_t orig_sr, int32_t new_sr, int32_t device, wt_resampler** out);
void    wt_resampler_destroy(wt_resampler* r);
int64_t wt_resampler_out_length(const wt_resampler* r, int64_t T);
int     wt_convert_audio(const wt_resampler* r, const float* wav, int32_t B, int32_t C, int64_t T, float* out, void* stream);

/* Replaces: save_audio's clamp / rescale (encoder/utils.py:95-103) + the PCM_S 16 conversion of torchaudio.save
 * (infer.py:70): rescale = 0: clamp to [-limit, limit]; 1: scale by min(limit / max|x|, 1) (workspace: 4 bytes);
 * then round-half-even(x * 32768) clipped to int16 (the backend's rounding rule is not pinned by the reference). */
int wt_pcm16(const float* x, int64_t n, float limit, int32_t rescale, int16_t* out, void* workspace, void* stream);

/* Replaces: _linear_overlap_add (encoder/utils.py:17-56), bit for bit: frames [n_frames][rows][frame_len] (the last
 * one holds last_len valid samples), weight [frame_len] = the reference's triangle 0.5 - |linspace(0,1,len+2)[1:-1] - 0.5|,
 * out [rows][stride * (n_frames - 1) + last_len]. */
int wt_linear_overlap_add(const float* frames, const float* weight, int32_t n_frames, int64_t rows, int64_t frame_len,
                          int64_t last_len, int64_t stride, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WAVTOKENIZER_AMD_H */
