// Internal declarations shared by the HIP translation units of libwavtok_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <string>

namespace wt {

void set_error(const std::string& msg);

#define WT_HIP_CHECK(expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            ::wt::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
            return -4;                                                                          \
        }                                                                                       \
    } while (0)

// hipFuncSetAttribute (dynamic LDS size) is per device: a process that drives several GPUs must set it on each one.
// run(f) calls f once per device (under a lock: plans may be run from several host threads) and again later if it failed
struct PerDeviceOnce {
    std::mutex mu;
    bool seen[64] = {};
    template <class F> int run(F&& f) {
        int d = 0;
        (void)hipGetDevice(&d);
        d &= 63;
        std::lock_guard<std::mutex> g(mu);
        if (seen[d]) return 0;
        if (int rc = f()) return rc;
        seen[d] = true;
        return 0;
    }
};

// ---------------------------------------------------------------------------------------------
// Call status.  A plan run owns one 32-bit status word in its workspace (zeroed at the start of the call).  Kernels
// OR bits into it; the guard step that ends every plan poisons the call's outputs (codes = -1, floats = NaN) when
// it is non-zero and copies it to a host-visible word, so a failed call can never hand out plausible-looking data.
//   WT_STATUS_LSTM  : a step barrier of the persistent LSTM timed out (lost co-residency)
//   WT_STATUS_RANGE : an S32 producer met |v| >= 65504: the f16 hi half of the split representation would be inf
enum : unsigned { WT_STATUS_LSTM = 1u, WT_STATUS_RANGE = 2u };
// The launch functions below take the status pointer from this per-thread context, which run_plan sets for the
// duration of a call (nullptr outside a plan: the single-stage entry points have no status word)
// stamp_start / stamp_end: when set (wt_plan_set_timing("@name")), the next gemm16s launch records its own duration on the
// device: every workgroup takes the constant 100 MHz clock (s_memrealtime) on entry and on exit, atomic min / max into the two
// words; stamp_used reports that a launch took them.  No packet is added to the stream, so the neighbours do not move
struct LaunchCtx {
    unsigned* status = nullptr;
    unsigned long long* stamp_start = nullptr;
    unsigned long long* stamp_end = nullptr;
    bool stamp_used = false;
};
extern thread_local LaunchCtx g_launch;

// Compute units of the current device (cached per device).  Launch geometry is derived from it: the persistent GEMM and
// resblock launches run one (or two) workgroups per CU of THIS device; only the persistent LSTM is tied to the full
// 256-CU / 8-XCD MI355X (plan.cpp) and falls back to the launch-per-step kernel anywhere else (a CPX / NPS partition).
int device_cus();

// Environment switches.  The PRODUCT library reads none on any launch path: lab_env() is a constant there, so the sweeps,
// A/B switches and fault hooks below cost nothing and cannot be reached (tests/test_host_logic.py checks that no object of
// the product build references getenv).  A LAB build (make LAB=1: -DWT_LAB, tools/lib/libwavtok_hip_lab.so, what tools/*.py,
// tools/micro/gemm_lab.hip and the fault-injection tests load through WAVTOK_HIP_LIB) reads them per call.
#ifdef WT_LAB
inline const char* lab_env(const char* name) { return getenv(name); }
#else
inline const char* lab_env(const char*) { return nullptr; }
#endif

#if defined(__HIPCC__)
// largest |v| of a value that is being converted to the split-f16 form; NaNs are ignored (they propagate by themselves)
#ifdef WT_NO_RANGE_TRACK        // A/B timing builds only (tools/ab_lib.sh): what the tracking costs
__device__ __forceinline__ float amax1(float m, float) { return m; }
__device__ __forceinline__ float amax4(float m, float, float, float, float) { return m; }
#else
__device__ __forceinline__ float amax1(float m, float v) { return fmaxf(m, fabsf(v)); }
__device__ __forceinline__ float amax4(float m, float a, float b, float c, float d) {      // two v_max3_f32 with |.| source modifiers
    m = fmaxf(fmaxf(m, fabsf(a)), fabsf(b));
    return fmaxf(fmaxf(m, fabsf(c)), fabsf(d));
}
#endif
// The split-f16 form of two values, packed: hi = f16(v) (round to nearest even), lo = f16((v - hi) * 2048).  v_cvt_pk_f16_f32
// for the hi pair, then lo = fma(hi, -2048, v * 2048) on v_fma_mixlo/mixhi_f16, which read hi as f16 straight from the packed
// register and round the fp32 result to f16 themselves: 2 instructions per element instead of 3 (no v_cvt_f32_f16 back,
// no separate convert).  The fma is exact in fp32 (v - hi is representable, the scale a power of two), so the bits equal
// those of (_Float16)((v - (float)hi) * 2048.f).
__device__ __forceinline__ void split2_f16(float a, float b, unsigned& hi_pk, unsigned& lo_pk) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t ab = {a, b};
    hi_pk = __builtin_bit_cast(unsigned, __builtin_convertvector(ab, f16x2_t));
    const f32x2_t s = ab * 2048.f;
    const float m = -2048.f;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lo_pk) : "v"(hi_pk), "s"(m), "v"(s[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo_pk) : "v"(hi_pk), "s"(m), "v"(s[1]));
}
typedef _Float16 wt_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split4_f16(float a, float b, float c, float d, wt_f16x4& hi, wt_f16x4& lo) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    unsigned h0, l0, h1, l1;
    split2_f16(a, b, h0, l0);
    split2_f16(c, d, h1, l1);
    hi = __builtin_bit_cast(wt_f16x4, (u32x2_t){h0, h1});
    lo = __builtin_bit_cast(wt_f16x4, (u32x2_t){l0, l1});
}
// ELU(alpha = 1): e = exp(x) - 1 >= x everywhere, so the median of (x, e, 0) is x for x > 0 and e otherwise: one v_med3_f32
// instead of a compare and a select
__device__ __forceinline__ float elu_med3(float x) { return __builtin_amdgcn_fmed3f(x, __expf(x) - 1.f, 0.f); }
// four at once, written on vectors so that the scale by log2(e) and the -1 become v_pk_mul_f32 / v_pk_add_f32
typedef float wt_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ wt_f32x4 elu_med3_x4(wt_f32x4 x) {
    const wt_f32x4 t = x * 1.44269504088896340736f;
    wt_f32x4 e;
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(t[i]);
    e = e - 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_fmed3f(x[i], e[i], 0.f);
    return e;
}
__device__ __forceinline__ void range_report(unsigned* status, float amax) {
    if (status && amax >= 65504.f) __hip_atomic_fetch_or(status, (unsigned)WT_STATUS_RANGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

// ---------------------------------------------------------------------------------------------
// The one dense-contraction kernel of the path: C = epilogue(prologue(gather(A)) . W^T)
//   A  : activations, time-major rows [clip][time][channel]; a row of the im2col matrix for
//        output frame t is the `taps` input rows (t*stride + tap*dil - pad_left), each `Cin` long,
//        so K = taps*Cin and no im2col buffer exists.  Plain matrices are taps=1, stride=1.
//   W  : [N][K] fp32, K contiguous (nn.Linear layout; conv weights repacked to [Cout][tap][Cin]).
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 multiply-add chain).
// ---------------------------------------------------------------------------------------------
enum Pro : int { PRO_NONE = 0, PRO_ELU = 1 };
enum Epi : int {
    EPI_BIAS = 0,            // C = acc + bias[n]                        (bias may be null)
    EPI_BIAS_RES = 1,        // C = (acc + bias[n]) + R[m][n]
    EPI_BIAS_GELU = 2,       // C = gelu_erf(acc + bias[n])
    EPI_BIAS_GAMMA_RES = 3,  // C = R[m][n] + gamma[n] * (acc + bias[n])
    EPI_HEAD = 4,            // ISTFTHead: paired (log-mag, phase) column tiles -> re/im spectrum
    EPI_ARGMAX = 6,          // VQ: per-row argmax of -(xx - 2 acc + ee[n]) over this wave's columns
    EPI_SCALE = 7,           // C = alpha * acc
    EPI_BIAS_ROW = 8,        // C = acc + bias[m]
    EPI_BIAS_RES_ELU = 9,    // C = elu((acc + bias[n]) + R[m][n])   (the only consumer applies ELU)
    EPI_BIAS_ELU = 10        // C = elu(acc + bias[n])               (gemm16s only)
};
// gemm16s output formats: what is written to C (and C2)
enum Out16s : int {
    OUT_F32 = 0,             // C fp32
    OUT_S32 = 1,             // C in the S32 split-f16 layout
    OUT_S32_DUAL_ELU = 2,    // C = S32(v) and C2 = S32(elu(v)): a tensor read both raw (shortcut) and through ELU (conv)
    OUT_F32_AND_S32 = 3      // C = fp32 v and C2 = S32(v): a tensor read by fp32 kernels and by a split-f16 GEMM
};
enum PadMode : int { PAD_ZERO = 0, PAD_REFLECT = 1 };

struct GemmArgs {
    // A gather
    const float* A = nullptr;
    long a_bstride = 0;   // elements between clips
    long a_rstride = 0;   // elements between consecutive time rows of a clip
    int T_in = 0, T_out = 0;     // rows per clip in / out;  M = nclips * T_out
    int Cin = 0, taps = 1, stride = 1, dil = 1, pad_left = 0, pad_mode = PAD_ZERO;
    int Tp = 0;           // reflect: effective length max(T_in, max_pad + 1) (conv.py:86-91)
    // gemm16s, plain row-major operands only (taps = 1, stride 1, no padding): K columns [K1, K) come from a second S32
    // tensor A2 with its own strides ([x | elu(h)] of a resblock's shortcut + conv1 in one contraction)
    const float* A2 = nullptr;
    long a2_bstride = 0, a2_rstride = 0;
    int K1 = 0;
    // W
    const float* W = nullptr;
    long w_rstride = 0;
    const void* W_hi = nullptr;    // gemm16s: the S32 copy of W (same strides as the fp32 array)
    const float* bias = nullptr;
    int M = 0, N = 0, K = 0;
    // C
    float* C = nullptr;
    long c_rstride = 0;
    float* C2 = nullptr;  // gemm16s second output (Out16s), row stride c_rstride
    const float* R = nullptr;
    long r_rstride = 0;
    const float* gamma = nullptr;
    float alpha = 1.f;
    // batched over blockIdx.z
    int nz = 1;
    long zA = 0, zW = 0, zC = 0;
    // EPI_HEAD
    int head_kb = 0;      // padded bins per half; C row = [re (kb) | im (kb)]
    // EPI_ARGMAX
    const float* vq_xx = nullptr;     // [M] row |x|^2
    const float* vq_ee = nullptr;     // [N] |e|^2
    float* vq_pval = nullptr;         // [M][vq_nparts]
    int*   vq_pidx = nullptr;
    int vq_nparts = 0;
    int group_m = 8;      // tile scheduling group (set by launch_gemm)
    int group_n = 0;      // gemm16s: column tiles per scheduling block (0 = all): an XCD's share of tiles then spans fewer weight panels
    int tap_pair = 0;     // gemm16s: K slot q holds tap (q >> 1) + (q & 1) * stride (k = 2 * stride convs; weights packed alike):
                          // the two output frames that share an input frame read it in adjacent K steps (L2 hit, not a re-fetch)
    unsigned long long* stamp_start = nullptr;   // gemm16s timing hook (LaunchCtx): device clock of the first entry / last exit
    unsigned long long* stamp_end = nullptr;
    int stage_epi = 0;    // gemm16s: epilogue staged through per-wave LDS scratch at byte offset stage_off (set by the launcher)
    int stage_off = 0;
    int pc_off = 0;       // gemm16s: byte offset of the per-wave bias (and gamma) cache in LDS, 0 = vectors read from global memory
    float acc_scale = 1.f;                 // gemm16s: the accumulators are multiplied by this (a power of two: operands were
    const float* acc_scale_dev = nullptr;  // stored scaled) before bias / activation; the device copy, when set, wins
    unsigned* status = nullptr;            // gemm16s: call status word (range report of the S32 epilogues); launcher default: g_launch
    unsigned long long* dbg_stamps = nullptr;   // gemm16s timing-experiment builds: per workgroup {s_memtime, s_memrealtime} spans
};

int gemm_vq_parts(int N);   // partial (val, idx) slots per row written by EPI_ARGMAX
int launch_gemm(const GemmArgs& a, int pro, int epi, hipStream_t s);
// gemm16s.hip: both operands pre-split in the S32 layout (128-byte groups [32 x f16 hi | 32 x f16 lo], same
// footprint and strides as the fp32 array); a.A / a.W_hi point at S32 data, out_s32 selects an S32 C
int launch_gemm16s(const GemmArgs& a, int epi, int out, hipStream_t s);     // out: Out16s
int gemm16s_vq_parts(int N);
int launch_split_s32(const float* x, void* out, long n, hipStream_t s, const float* scale_dev = nullptr);
int launch_unsplit_s32(const void* s32, float* out, long n, float inv_scale, hipStream_t s);     // fp32 = (hi + lo * 2^-11) * inv_scale
// per-tensor power-of-two scales of two tensors on the device: out3 = {scale_a, scale_b, 1 / (scale_a * scale_b)}
int launch_pow2_scales(const float* a, long na, const float* b, long nb, unsigned* bits2, float* out3, hipStream_t s);

// ------------------------------------------------------------------------ non-GEMM kernels
int launch_conv_first(const float* wav, const float* w /*[7][Cout]*/, const float* bias, float* y, int B, long T,
                      int k, int Cout, hipStream_t s);
int launch_conv_last(const float* x /*[B][T][Cin]*/, const float* w /*[k][Cin]*/, const float* bias, float* y /*[B][T]*/,
                     int B, long T, int Cin, int k, int elu_in, hipStream_t s);
int launch_transpose(const float* in, float* out, int B, int R, int C, hipStream_t s, int out_s32 = 0);  // [B][R][C] -> [B][C][R]
// `part`: gn_part_floats() floats of scratch for sequences too long for the one-slab kernel (chunk statistics)
size_t gn_part_floats(int B, int L, int groups);
int launch_gn_stats(const float* x, const float* gamma, const float* beta, float* scale, float* shift, int B, int L,
                    int C, int groups, float eps, hipStream_t s, float* part = nullptr);
int launch_gn_apply(const float* x, const float* gamma, const float* beta, float* scale, float* shift, float* y,
                    int swish, int B, int L, int C, int groups, float eps, hipStream_t s, int out_s32 = 0, float* part = nullptr);
enum RowNormMode : int { RN_DWCONV = 0, RN_PLAIN = 1, RN_AFFINE_IN = 2 };
int launch_rownorm(int mode, const float* x, float* y, int B, int L, int C, const float* dw_w /*[7][C]*/,
                   const float* dw_b, const float* in_scale, const float* in_shift, const float* out_scale,
                   const float* out_shift, float eps, hipStream_t s, int out_s32 = 0);
int launch_istft_ola(const float* parts, const float* win, const float* wsq, float* out, int B, int L, int n_fft, int hop,
                     int Kq, int center, hipStream_t s);
int launch_softmax(float* S, int rows, int L, int ld, hipStream_t s, float* P_s32 = nullptr);
int launch_row_sumsq(const float* x, float* out, long rows, int D, hipStream_t s);
int launch_vq_finalize(const float* pval, const int* pidx, int nparts, const float* embed, int64_t* codes,
                       float* feat_ncl, int B, int L, int D, int bins, hipStream_t s);
int launch_codes_to_features(const int64_t* codes, const float* embed, int K, int bins, int B, long L, int D,
                             float* feat_ncl, hipStream_t s, unsigned* bad = nullptr);
// buffer fill as a kernel (hipMemsetAsync nodes misbehave under hipGraph replay: ops.hip); 16-byte aligned pointer and size
int launch_fill_u32(void* p, unsigned value, size_t n_bytes, hipStream_t s);
// last step of every plan: on a non-zero status word poison the outputs (codes = -1, floats = NaN) and publish the bits
int launch_plan_guard(const unsigned* status, int nwords, int site0, unsigned* host_status, unsigned* model_status, int64_t* codes,
                      long n_codes, float* f0, long n0, float* f1, long n1, float* f2, long n2, hipStream_t s);
// max |value| of an S32 array as the bit pattern of a float, atomicMax-ed into *out_bits (range report)
int launch_s32_amax(const void* s32, long numel, unsigned* out_bits, hipStream_t s);
struct LstmArgs {
    const float* xg0;     // [L][B][4H] (time-major) layer-0 input projection (+ both biases), packed gate order
    const float* W0;      // W_hh_l0, per 16 packed gate rows: [H/16][64 lanes][4] (ops.hip lstm_step_kernel)
    const float* W1;      // [W_ih_l1 | W_hh_l1], same packing over K = 2H
    const float* b1;      // [4H]      packed b_ih_l1 + b_hh_l1
    float* h0;            // [2][H][Bp]  K-major, clip pitch Bp = B rounded up to 64 (zero-filled before step 0)
    float* h1;            // [2][H][Bp]
    float* c0;            // [B][H]
    float* c1;            // [B][H]
    const float* x;       // [B][L][H] skip input
    float* y;             // [B][L][H] output = h1 + x
    int B, L, H;
    int elu_out;          // store elu(h1 + x): the only consumer is ELU -> conv (seanet.py:136-139)
    int out_s32;          // write y in the S32 split-f16 layout (its consumer is a gemm16s conv)
    int f16x3;            // W0 / W1 are the f16 (hi, lo) packings and the state is kept quad-split (ops.hip)
    unsigned* status = nullptr;   // call status word (range report of the S32 output); launcher default: g_launch
};
int launch_lstm_step(const LstmArgs& a, int s, hipStream_t stream);
// the whole recurrence in one persistent launch (lstm_persist.hip): per-XCD clip groups, weights resident in registers
struct LstmPersistArgs {
    const float* xg0;     // [L][B][4H] layer-0 input projection (+ both biases), packed gate order
    const void* Wp;       // [3 roles][32 wg][4 tiles][16 blk][hi, lo][64 lanes][8 halves]: W_hh_l0, W_ih_l1, W_hh_l1
    const float* b1;      // [4H] packed b_ih_l1 + b_hh_l1
    const float* x;       // [B][L][H] skip input
    float* y;             // [B][L][H] output
    void* hx;             // lstm_persist_hx_bytes(): per-XCD exchange buffers  } filled before the launch with 0xFF bytes
    unsigned* ctl;        // lstm_persist_ctl_bytes(): tickets / counters / error } (data_flag) or zeros (counter form)
    unsigned* status = nullptr;   // call status word: WT_STATUS_LSTM when a step barrier times out, WT_STATUS_RANGE; launcher default: g_launch
    int dbg_spin_shift = 0;       // test hook: the spin bounds are divided by 2^this
    int B, L, H, Bx;      // Bx = clips per XCD = ceil(B / 8) <= 16
    int elu_out, out_s32;
    int data_flag;        // bit 0: the exchanged state carries its own readiness marks (default), else arrival counter per step;
                          // bit 2: workgroup 0 of XCD 0 writes phase timestamps of steps 64..71 into ctl[520..] (tools/lstm_trace.py)
};
size_t lstm_persist_hx_bytes();
size_t lstm_persist_ctl_bytes();
int launch_lstm_persist(const LstmPersistArgs& a, hipStream_t stream);
struct ResblockArgs {
    const float* x;       // [B][T][C] raw block input (unused when wav is set)
    const float* wav;     // optional [B][T]: fold SEANetEncoder model[0] (k=7, 1 -> 32) into the tile load
    const float* e0_w;    // [7][32]
    const float* e0_b;    // [32]
    const float* W3;      // [C/2][3][C]
    const float* b3;
    const float* W1;      // [C][C/2]
    const float* b1;
    const float* Ws;      // [C][C]
    const float* bs;
    float* y;             // [B][T][C]
    int B, T, C;
    long x_bstride;       // elements between clips of x (0 = T*C); x may be a trimmed view of a longer buffer
    int elu_out;          // store elu(y) (the only consumer is ELU -> down conv)
    int out_s32;          // resblock16 only: write y in the S32 split-f16 layout (gemm16s.hip) instead of fp32
    int dbg;              // resblock16 timing experiments only (WT_RB16_DBG)
    unsigned* status;     // resblock16: call status word (range report of its split-f16 conversions); launcher default: g_launch
    // resblock16 with the stage's down conv fused in (launch_resblock16_down): Wd [64][2R][32] folded, bd [64], stride R,
    // y_down [B][T / R][64] fp32 is the only output
    const float* Wd;
    const float* bd;
    float* y_down;
    int R;
};
bool resblock_fusable(int C);
int launch_resblock(const ResblockArgs& a, hipStream_t s);      // fp32 MFMA chain (resblock.hip)
int launch_resblock16(const ResblockArgs& a, hipStream_t s);    // split-f16 MFMAs, fp32-equivalent (resblock16.hip)
bool resblock16_down_fusable(int C, long T, int r, int k);
int launch_resblock16_down(const ResblockArgs& a, hipStream_t s);   // stage 1 + ELU + down conv in one launch
int launch_convtr(const float* x, const float* w /*[k][Cin][Cout]*/, const float* bias, float* y, int B, int Tin,
                  int Cin, int Cout, int k, int stride, int elu_in, hipStream_t s);

}  // namespace wt
