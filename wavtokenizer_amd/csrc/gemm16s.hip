// Split-f16 GEMM on PRE-SPLIT operands, staged global -> LDS by LDS-DMA (buffer_load ... lds).
//
// Arithmetic: every fp32 operand x is carried as two f16 numbers, x = hi + lo * 2^-11, and a product is formed by three
// f16 MFMAs into a main and a correction fp32 accumulator (hi.hi, then hi.lo + lo.hi scaled by 2^-11): fp32-equivalent
// products.  Both operands arrive already split, in the "S32" layout: every run of 32 consecutive fp32 elements of a row becomes one 128-byte group
//      [32 x f16 hi | 32 x f16 lo]
// so an S32 array has exactly the footprint and the row strides of the fp32 array it stands for, element e of a
// row starts at byte (e & ~31) * 4 + (e & 31) * 2, and one K step (32 deep) of one row is one full 128-byte line.
// Producers write S32 directly (norm kernels, the GELU / head epilogues here, weights once at load), so the K loop
// has no conversion work at all: per step a wave issues a few LDS-DMA loads (8 rows x 128 B each, no VGPRs), 16
// ds_read_b128 and 18 MFMAs.  The conv gather (taps, stride, reflect / zero padding, ragged edges) is the same
// per-(tap, row) byte-offset table as gemm.hip; an out-of-range offset makes the DMA write zeros.
//
// LDS image of a stage: (BM + BN) rows x 128 B; the eight 16-byte chunks of row r are XOR-swizzled with
// (r >> 1) & 7, applied on the SOURCE address of the DMA (its LDS side is lane-linear) and on the ds_read address:
// every ds_read_b128 lane group then covers all 64 banks.
#include "common.h"

#include <cmath>
#include <stdlib.h>
#include <type_traits>

namespace wt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
static constexpr int SBK = 32;          // k elements per step = one 128-byte S32 group per row

__device__ __forceinline__ int xcd_remap_s(int orig, int nwg) {
    int q = nwg >> 3, r = nwg & 7;
    int xcd = orig & 7, idx = orig >> 3;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
__device__ __forceinline__ float elu_s(float x) { return elu_med3(x); }
// erf to < 1 ulp (max abs error 6.3e-8 against float64 erf over [-6, 6], checked on the host), branch-free: the
// two minimax pieces |a| <= 475/512 (odd polynomial) and beyond (1 - exp(polynomial)) are both evaluated and selected
__device__ __forceinline__ float erf_s(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - __expf(r), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    const float small = fmaf(q, a, a);
    return t > 0.927734375f ? big : small;
}
__device__ __forceinline__ float gelu_erf_s(float x) { return x * 0.5f * (1.f + erf_s(x * 0.70710678118654752440f)); }
// the same arithmetic on two elements at a time: every polynomial step is one v_pk_fma_f32 / v_pk_mul_f32
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 gelu_erf_s2(f32x2 x) {
    const f32x2 a = x * 0.70710678118654752440f;
    const f32x2 t = __builtin_elementwise_abs(a), s = a * a;
    f32x2 r = fma2((f32x2)(-1.72853470e-5f), t, (f32x2)(3.83197126e-4f));
    const f32x2 u = fma2((f32x2)(-3.88396438e-3f), t, (f32x2)(2.42546219e-2f));
    r = fma2(r, s, u);
    r = fma2(r, t, (f32x2)(-1.06777877e-1f));
    r = fma2(r, t, (f32x2)(-6.34846687e-1f));
    r = fma2(r, t, (f32x2)(-1.28717512e-1f));
    r = fma2(r, t, -t);
    f32x2 q = fma2((f32x2)(-5.96761703e-4f), s, (f32x2)(4.99119423e-3f));
    q = fma2(q, s, (f32x2)(-2.67681349e-2f));
    q = fma2(q, s, (f32x2)(1.12819925e-1f));
    q = fma2(q, s, (f32x2)(-3.76125336e-1f));
    q = fma2(q, s, (f32x2)(1.28379166e-1f));
    const f32x2 small = fma2(q, a, a);
    f32x2 e;
    e.x = t.x > 0.927734375f ? copysignf(1.0f - __expf(r.x), a.x) : small.x;
    e.y = t.y > 0.927734375f ? copysignf(1.0f - __expf(r.y), a.y) : small.y;
    const f32x2 hx = x * 0.5f;
    return fma2(hx, e, hx);
}
__device__ __forceinline__ f32x4 gelu_erf_s4(f32x4 v) {
    const f32x2 lo = gelu_erf_s2((f32x2){v.x, v.y}), hi = gelu_erf_s2((f32x2){v.z, v.w});
    return (f32x4){lo.x, lo.y, hi.x, hi.y};
}

// GELU through erfc on ONE branch: with a = x / sqrt(2), gelu(x) = x * (1 - h) for x >= 0 and x * h for x < 0, where
// h = erfc(|a|) / 2 = 2^p(|a|): p is a degree-9 minimax fit of log2(erfc(t)) - 1 on [0, 4.3] (weighted for the absolute
// error of erfc: 2e-9 in exact arithmetic; |a| is clamped to 4.3, beyond which h < 6e-10).  No second polynomial, no select
// between two formulas, and no cancellation on the negative side (h is formed with relative accuracy, where the
// reference's own fp32 1 + erf(a) carries the rounding of erf near -1): simulated in fp32 against float64 on 2.4 M
// inputs the maximum absolute error is 3.9e-7 (torch's fp32 gelu: 1.2e-6), rms relative 1.0e-7 (2.3e-6).
__device__ __forceinline__ f32x2 gelu_erfc_s2(f32x2 x) {
    const f32x2 a = x * 0.70710678118654752440f;
    f32x2 t = __builtin_elementwise_abs(a);
    t.x = fminf(t.x, 4.3f); t.y = fminf(t.y, 4.3f);
    f32x2 p = fma2((f32x2)(1.146792511e-05f), t, (f32x2)(-1.515573094e-04f));
    p = fma2(p, t, (f32x2)(8.423082181e-04f));
    p = fma2(p, t, (f32x2)(-2.261521295e-03f));
    p = fma2(p, t, (f32x2)(6.768874300e-05f));
    p = fma2(p, t, (f32x2)(2.773740143e-02f));
    p = fma2(p, t, (f32x2)(-1.483134478e-01f));
    p = fma2(p, t, (f32x2)(-9.184416533e-01f));
    p = fma2(p, t, (f32x2)(-1.627907395e+00f));
    p = fma2(p, t, (f32x2)(-1.0f));
    f32x2 h;
    h.x = __builtin_amdgcn_exp2f(p.x);
    h.y = __builtin_amdgcn_exp2f(p.y);
    f32x2 s;
    s.x = a.x >= 0.f ? 1.f - h.x : h.x;
    s.y = a.y >= 0.f ? 1.f - h.y : h.y;
    return x * s;
}
__device__ __forceinline__ f32x4 gelu_erfc_s4(f32x4 v) {
    const f32x2 lo = gelu_erfc_s2((f32x2){v.x, v.y}), hi = gelu_erfc_s2((f32x2){v.z, v.w});
    return (f32x4){lo.x, lo.y, hi.x, hi.y};
}
// The same arithmetic on the four sub-runs of a 32 x 32 block at once, written step by step ACROSS the eight packed pairs:
// a packed fp32 instruction that reads the result of the one in front of it costs a wait state (hipcc pads with s_nop), so
// one Horner chain at a time runs at half rate; eight independent chains need no padding (same results bit for bit)
__device__ __forceinline__ void gelu_erfc_x16(f32x4 (&v)[4]) {
    f32x2 x[8], a[8], t[8], p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        x[i] = (f32x2){v[i >> 1][2 * (i & 1)], v[i >> 1][2 * (i & 1) + 1]};
        a[i] = x[i] * 0.70710678118654752440f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        t[i] = __builtin_elementwise_abs(a[i]);
        t[i].x = fminf(t[i].x, 4.3f); t[i].y = fminf(t[i].y, 4.3f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = fma2((f32x2)(1.146792511e-05f), t[i], (f32x2)(-1.515573094e-04f));
    constexpr float c[8] = {8.423082181e-04f, -2.261521295e-03f, 6.768874300e-05f, 2.773740143e-02f, -1.483134478e-01f,
                            -9.184416533e-01f, -1.627907395e+00f, -1.0f};
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = fma2(p[i], t[i], (f32x2)(c[s]));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f32x2 h, sgn;
        h.x = __builtin_amdgcn_exp2f(p[i].x);
        h.y = __builtin_amdgcn_exp2f(p[i].y);
        sgn.x = a[i].x >= 0.f ? 1.f - h.x : h.x;
        sgn.y = a[i].y >= 0.f ? 1.f - h.y : h.y;
        const f32x2 r = x[i] * sgn;
        v[i >> 1][2 * (i & 1)] = r.x;
        v[i >> 1][2 * (i & 1) + 1] = r.y;
    }
}

// sin and cos of one argument for the ISTFT head (heads.py:58-59): three-constant Cody-Waite reduction by pi/2 (exact for the
// |x| < 2^15 it is used for: the first constant has 8 significant bits, every step is one fma) and the classic degree-7 / 8
// minimax polynomials on [-pi/4, pi/4] (about 1 ulp).  libm's sincosf carries a Payne-Hanek path for huge arguments that
// hipcc evaluates branch-free for every lane: 128 64-bit multiply-adds and 450 selects per 16 values in the r02 epilogue.
// Phases beyond 2^15 (never seen; a Linear output) take libm's path.
__device__ __forceinline__ void sincos_head(float x, float& s, float& c) {
    if (__builtin_expect(!(fabsf(x) < 32768.f), 0)) { sincosf(x, &s, &c); return; }
    const float k = rintf(x * 0.63661977236758134308f);
    float r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188216e-8f, r);
    const float z = r * r;
    float sp = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(sp, z, -1.6666654611e-1f);
    const float sr = fmaf(sp * z, r, r);
    float cp = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(cp, z, 4.166664568298827e-2f);
    const float cr = fmaf(cp * z, z, fmaf(z, -0.5f, 1.f));
    const int q = (int)k;
    const float s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}
// e^x to about 2 ulp: x = n ln2 + r with a two-constant ln2, 2^n by ldexp (heads.py:55)
__device__ __forceinline__ float exp_head(float x) {
    const float n = rintf(x * 1.44269504088896340736f);
    float r = fmaf(-n, 0.693145751953125f, x);
    r = fmaf(-n, 1.42860682030941723212e-6f, r);
    return ldexpf(__builtin_amdgcn_exp2f(r * 1.44269504088896340736f), (int)n);
}

// fp32 value -> S32 slots of element n (n & 31 = slot) in the group that starts at `grp` (a _Float16*)
__device__ __forceinline__ void store_s32(_Float16* grp, int slot, float v) {
    const _Float16 h = (_Float16)v;
    grp[slot] = h;
    grp[32 + slot] = (_Float16)((v - (float)h) * 2048.f);
}

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// four consecutive elements n .. n+3 (n % 4 == 0) of a row in S32: 8 bytes of hi halves, 8 bytes of lo halves
__device__ __forceinline__ void store_s32_x4(float* row, int n, const f32x4 v, float& amax) {
    amax = amax4(amax, v.x, v.y, v.z, v.w);
    f16x4 hi, lo;
    split4_f16(v.x, v.y, v.z, v.w, hi, lo);
    _Float16* g = reinterpret_cast<_Float16*>(row) + ((n >> 5) * 64 + (n & 31));
    *reinterpret_cast<f16x4*>(g) = hi;
    *reinterpret_cast<f16x4*>(g + 32) = lo;
}

// 16 bytes of an output tile.  SC1: write-through store that does not keep the line in this XCD's L2 (MI355X_MICROARCH.md,
// "stores of each flavour"): the C tile is never read again by this kernel, and 71 MB of it per launch otherwise push the
// A / W panels, which ARE re-read, out of the 4 MiB L2 (pwconv1: -1.4 us of 93, tools/micro/gemm_lab.hip)
template <bool SC1>
__device__ __forceinline__ void store_c16(float* dst, const f32x4 v) {
    if constexpr (SC1) {
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t qv = __builtin_bit_cast(u32x4_t, v);
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(qv) : "memory");
    } else {
        *reinterpret_cast<f32x4*>(dst) = v;
    }
}

template <int N>
__device__ __forceinline__ void wait_vm_lgkm() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

// DBG: timing-experiment builds (WT_GEMM16S_DBG, tools/gemm16s_bench.py): a compile-time mask of phases to leave out
// (1 DMA, 2 MFMA, 4 epilogue, 8 barrier, 16 LDS fragment reads, 32 waits, 64 static priority for the younger half, 128
// stores, 256 bias, 512 GELU; 1024: stamp s_memtime / s_memrealtime around the tile loop into GemmArgs::dbg_stamps: the
// clock the chip holds).  The shipped instantiations have DBG = 0.  The mask is a template argument because a run-time
// test in front of every phase (round 1) cuts the K loop and the epilogue into dozens of basic blocks, across which
// hipcc neither overlaps LDS reads with MFMAs nor one 4-column run's GELU with the next one's: 108.7 -> 102.7 us on pwconv1.
#ifndef WT_GEMM16S_MF
#define WT_GEMM16S_MF 1          // MFMA shape of every instantiation (0: 32x32x16, for A/B timing builds; 1: 16x16x32)
#endif
// KS: K tiles per barrier.  1 = the pipeline described at the loop (one barrier in the middle of every K step).  2 = the
// small-problem form (a handful of narrow tiles, one wave per SIMD): two K tiles are made visible by one barrier and their
// fragment reads and MFMAs run back to back - such a launch is a chain of DMA-wait, barrier, LDS-read and MFMA latencies per
// barrier, not of work (tools/step_times.py at B = 1), so half the barriers is most of the time.  Every accumulator still
// sees its K tiles in ascending order: results are bit-identical to KS = 1
// PROD = 1, 2 (with KS = 2): the workgroup carries PROD further sets of WAVES_M x WAVES_N waves that do nothing but the loader's
// work (DMA issue and the wait for it, the pieces shared out among them); the first set does nothing but fragment reads,
// MFMAs and the epilogue.  With one
// wave per SIMD a wave pays for its DMA issue (5 pieces per K tile), its LDS reads and its MFMAs one after the other - at
// B = 1 the sum IS the launch time (ladder in tools/micro/gemm_lab.hip: MFMA 7, LDS reads 4, barriers 3, DMA issue 5, data 8
// of 35 us) - while two waves per SIMD with the roles split run the DMA side under the MFMA side
template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, int EPI, int OUT, int DBG = 0, int MF = WT_GEMM16S_MF, int WPS = 2, int KS = 1,
          int PROD = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * (1 + PROD), WPS) void gemm16s_kernel(const GemmArgs p) {
    constexpr int dbg = DBG;
    static_assert(KS == 1 || (KS == 2 && NSTAGE == 6 && MF == 1 && DBG == 0), "two K tiles per barrier: 6 stages (three pairs), 16x16x32 MFMA, no experiment masks");
    static_assert(!PROD || KS == 2, "loader waves: the two-tiles-per-barrier form only");
    constexpr int NW = WAVES_M * WAVES_N, NT = 64 * NW * (1 + PROD);        // NW: MFMA waves
    constexpr int NL = PROD ? NW * PROD : NW;                               // waves that issue the DMA pieces
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int STG = (BM + BN) * 128;                   // bytes per stage
    constexpr int NPA = BM / 8 / NL, NPB = BN / 8 / NL;    // DMA pieces (8 rows x 128 B) per loading wave and K step
    constexpr int NPT = NPA + NPB;
    static_assert(BM % (8 * NL) == 0 && BN % (8 * NL) == 0, "pieces must divide evenly among the loading waves");
    static_assert(WM % 32 == 0 && WN % 32 == 0 && BM % 16 == 0, "32x32 MFMA tiles");
    static_assert((NSTAGE - 2) * NPT < 64, "vmcnt field");
    extern __shared__ __attribute__((aligned(1024))) char smem_s[];

    // the wave index as a SCALAR: with threadIdx.x >> 6 in a vector register every LDS-DMA destination (M0) went through
    // v_add + v_readfirstlane + s_mov per piece, and every per-wave offset cost vector instructions
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = (dbg & 65536) ? (tid >> 6) : __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader_wave = PROD && wave_all >= NW;          // (scalar)
    const bool mfma_wave = !loader_wave;
    const bool dma_wave = !PROD || loader_wave;
    const int wave = loader_wave ? wave_all - NW : wave_all;  // index within the role: a loader wave issues the pieces of its twin
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;                 // persistent: this workgroup owns tiles blockIdx.x, + G, + 2G, ...
    if (p.stamp_start && tid == 0)           // timing hook: earliest entry of any workgroup (constant 100 MHz clock)
        __hip_atomic_fetch_min(p.stamp_start, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int z = blockIdx.z;
    const int nclips = p.M / p.T_out;
    constexpr unsigned OOB = 0x80000000u;

    // virtual block -> (row tile, column tile): XCD-contiguous, then grouped GM row tiles at a time (gemm.hip)
    auto tile_coords = [&](int vb, int& bm, int& bn) {
        int tile = xcd_remap_s(vb, ntiles);
        // column blocks of group_n tiles (the last one may be narrower), inside a block groups of group_m row tiles
        const int GN = p.group_n > 0 && p.group_n < tiles_n ? p.group_n : tiles_n;
        const int blk_full = tiles_m * GN, nfull = tiles_n / GN;
        int nblk = tile / blk_full, gn = GN;
        if (nblk >= nfull) { nblk = nfull; gn = tiles_n - nfull * GN; }
        tile -= nblk * blk_full;
        const int per_group = p.group_m * gn;
        const int grp = tile / per_group;
        const int first_m = grp * p.group_m;
        const int gsz = tiles_m - first_m < p.group_m ? tiles_m - first_m : p.group_m;
        const int in_grp = tile - grp * per_group;
        bm = first_m + in_grp % gsz;
        bn = nblk * GN + in_grp / gsz;
    };
    auto first_clip = [&](int bm) { return (bm * BM < p.M ? bm * BM : p.M - 1) / p.T_out; };

    // S32 arrays are addressed in bytes = 4 x the fp32 element offset
    const char* Ag = reinterpret_cast<const char*>(p.A) + (long)z * p.zA * 4;
    const char* Ag2 = reinterpret_cast<const char*>(p.A2);          // second K source (nz = 1: host)
    const char* Wg = reinterpret_cast<const char*>(p.W_hi) + (long)z * p.zW * 4;
    const long w_span = (long)p.N * p.w_rstride * 4;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(Wg), 0, (int)(w_span < 0x7fffffffL ? w_span : 0x7fffffffL), 0x00020000);

    // bias / gamma cache (see the persistent loop): the vectors of the epilogues that store between their loads
    constexpr bool PCACHE = EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_ELU || EPI == EPI_BIAS_RES ||
                            EPI == EPI_BIAS_RES_ELU || EPI == EPI_BIAS_GAMMA_RES || EPI == EPI_HEAD;
    constexpr int PC_BYTES = WN * 4 * (EPI == EPI_BIAS_GAMMA_RES ? 2 : 1);
    const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias ? p.bias : p.C), 0, p.bias ? p.N * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsGamma = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.gamma ? p.gamma : p.C), 0, p.gamma ? p.N * 4 : 0, 0x00020000);
    const char* pcw = smem_s + p.pc_off + wave * PC_BYTES;

    // two [taps][BM] tables of per-(tap, row) byte offsets: the tile being loaded and the one after it
    unsigned* s_rowoff = reinterpret_cast<unsigned*>(smem_s + NSTAGE * STG);
    const int tab_sz = p.taps * BM;
    auto build_table = [&](int vb, int par) {
        if (vb >= ntiles) return;
        int bm, bn;
        tile_coords(vb, bm, bn);
        const int clip0 = first_clip(bm);
        unsigned* tab = s_rowoff + par * tab_sz;
        for (int e = tid; e < tab_sz; e += NT) {
            const int tp = e / BM, r = e - tp * BM;
            const int m = bm * BM + r;
            unsigned off = OOB;
            if (m < p.M) {
                const int b = m / p.T_out;
                const int t = m - b * p.T_out;
                const int tap = p.tap_pair ? (tp >> 1) + (tp & 1) * p.stride : tp;
                int pos = t * p.stride - p.pad_left + tap * p.dil;
                bool ok;
                if (p.pad_mode == PAD_REFLECT) {
                    pos = pos < 0 ? -pos : pos;
                    pos = pos >= p.Tp ? 2 * (p.Tp - 1) - pos : pos;
                    ok = pos < p.T_in;
                } else {
                    ok = (pos >= 0) && (pos < p.T_in);
                }
                if (ok) off = (unsigned)(((long)(b - clip0) * p.a_bstride + (long)pos * p.a_rstride) * 4);
                // second source (taps = 1, stride 1, no padding: pos = t)
                if (p.A2) s_rowoff[2 * tab_sz + par * BM + r] = (unsigned)(((long)(b - clip0) * p.a2_bstride + (long)t * p.a2_rstride) * 4);
            } else if (p.A2) {
                s_rowoff[2 * tab_sz + par * BM + r] = OOB;
            }
            tab[e] = off;
        }
    };

    // ---- loader: a continuous stream of K tiles that runs NSTAGE-1 steps ahead of the MFMAs and crosses from one
    // output tile into the next without a seam.  DMA piece q = 8 image rows; lane -> (row q*8 + lane/8, physical
    // chunk lane%8); the logical chunk it must fetch is physical ^ ((row >> 1) & 7)
    const int prow = lane >> 3;
    int a_row[NPA];
    unsigned a_chunk[NPA], w_chunk[NPB];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int q = wave + NL * i;
        a_row[i] = q * 8 + prow;
        a_chunk[i] = (unsigned)(((lane & 7) ^ (((q & 1) << 2) | (lane >> 4))) * 16);
    }
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
        const int q = BM / 8 + wave + NL * j;
        w_chunk[j] = (unsigned)(((lane & 7) ^ (((q & 1) << 2) | (lane >> 4))) * 16);
    }
    unsigned a_voff[NPA], a2_voff[NPA], w_voff[NPB];
    __amdgpu_buffer_rsrc_t rsA, rsA2;
    int l_vb = blockIdx.x, l_par = 0, tapL = 0, ciL = 0, kL = 0;
    unsigned l_mask = 0;
    auto set_tap = [&](int tap) {
        const unsigned* tab = s_rowoff + l_par * tab_sz;
#pragma unroll
        for (int i = 0; i < NPA; ++i) a_voff[i] = tab[tap * BM + a_row[i]] + a_chunk[i];
    };
    auto loader_set_tile = [&](int vb, int par) {       // the table of `vb` (parity par) must be visible
        l_vb = vb; l_par = par; tapL = 0; ciL = 0; kL = 0;
        if (dbg & 131072) l_mask = OOB;          // timing experiment: every DMA is issued but fetches nothing (zero fill)
        if (vb >= ntiles) { l_mask = OOB; return; }
        int bm, bn;
        tile_coords(vb, bm, bn);
        const int clip0 = first_clip(bm);
        const char* Ablk = Ag + (long)clip0 * p.a_bstride * 4;
        const long a_span = ((long)(nclips - clip0 - 1) * p.a_bstride + (long)p.T_in * p.a_rstride) * 4;
        rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Ablk), 0,
                                                (int)(a_span < 0x7fffffffL ? a_span : 0x7fffffffL), 0x00020000);
        if (p.A2) {
            const long a2_span = ((long)(nclips - clip0 - 1) * p.a2_bstride + (long)p.T_in * p.a2_rstride) * 4;
            rsA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Ag2 + (long)clip0 * p.a2_bstride * 4), 0,
                                                     (int)(a2_span < 0x7fffffffL ? a2_span : 0x7fffffffL), 0x00020000);
            const unsigned* tab2 = s_rowoff + 2 * tab_sz + par * BM;
#pragma unroll
            for (int i = 0; i < NPA; ++i) a2_voff[i] = tab2[a_row[i]] + a_chunk[i];
        }
#pragma unroll
        for (int j = 0; j < NPB; ++j) {
            const int n = bn * BN + (wave + NL * j) * 8 + prow;
            w_voff[j] = n < p.N ? (unsigned)((long)n * p.w_rstride * 4) + w_chunk[j] : OOB;
        }
        set_tap(0);
    };
    // one DMA piece of the K tile the loader stands at (idx < NPA: activation rows, else weight rows); load_advance() moves on
    auto load_piece = [&](int stage, auto idx_c) {
        constexpr int idx = decltype(idx_c)::value;
        char* sbase = smem_s + stage * STG + wave * 1024;
        if constexpr (idx < NPA) {
            // the K advance rides in the instruction's scalar offset (it is not part of the range check, which the
            // out-of-range marker in the vector offset still fails): no vector add per piece
            if (dbg & 1048576)      // timing experiment: real (non-zero) data, but always the same 8 KB: the cost of the traffic itself
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(sbase + NL * idx * 1024), 16, (int)(a_voff[idx] & 0x1fffu), 0, 0, 0);
            else if (p.A2 && kL >= p.K1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (lds_ptr_t)(sbase + NL * idx * 1024), 16,
                                                         (int)(a2_voff[idx] | l_mask), (kL - p.K1) * 4, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(sbase + NL * idx * 1024), 16,
                                                         (int)(a_voff[idx] | l_mask), ciL * 4, 0, 0);
        } else {
            constexpr int j = idx - NPA;
            if (dbg & 1048576)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(sbase + (BM / 8 + NL * j) * 1024), 16, (int)(w_voff[j] & 0x1fffu), 0, 0, 0);
            else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(sbase + (BM / 8 + NL * j) * 1024), 16,
                                                     (int)(w_voff[j] | l_mask), kL * 4, 0, 0);
        }
    };
    auto load_advance = [&]() {
        kL += SBK; ciL += SBK;
        if (kL >= p.K) {
            if (l_mask == 0) loader_set_tile(l_vb + G, l_par ^ 1);     // on into the next output tile
        } else if (p.taps > 1 && ciL >= p.Cin) {
            ciL = 0; ++tapL; set_tap(tapL);
        }
    };
    auto load_tile = [&](int stage) {
        char* sbase = smem_s + stage * STG + wave * 1024;
        if (p.A2 && kL >= p.K1) {                                    // wave-uniform: this K tile comes from the second source
#pragma unroll
            for (int i = 0; i < NPA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (lds_ptr_t)(sbase + NL * i * 1024), 16, (int)(a2_voff[i] | l_mask), (kL - p.K1) * 4, 0, 0);
        } else
#pragma unroll
        for (int i = 0; i < NPA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(sbase + NL * i * 1024), 16, (int)(a_voff[i] | l_mask), ciL * 4, 0, 0);
#pragma unroll
        for (int j = 0; j < NPB; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(sbase + (BM / 8 + NL * j) * 1024), 16, (int)(w_voff[j] | l_mask), kL * 4, 0, 0);
        kL += SBK; ciL += SBK;
        if (kL >= p.K) {
            if (l_mask == 0) loader_set_tile(l_vb + G, l_par ^ 1);     // on into the next output tile
        } else if (p.taps > 1 && ciL >= p.Cin) {
            ciL = 0; ++tapL; set_tap(tapL);
        }
    };

    // Two MFMA shapes (template argument MF):
    //   MF = 0  v_mfma_f32_32x32x16_f16: lane (r = lane & 31, h = lane >> 5) holds k = 16 s + 8 h .. + 7 of row r: logical
    //           chunk 2 s + h of the hi half, 4 + 2 s + h of the lo half; a K step is two 16-deep halves
    //   MF = 1  v_mfma_f32_16x16x32_f16 (the default): lane (r = lane & 15, q = lane >> 4) holds k = 8 q .. 8 q + 7 of row r:
    //           chunk q of the hi half, 4 + q of the lo half; a K step is ONE 32-deep MFMA per 16 x 16 tile.  Same LDS
    //           image, same reads per step (16 ds_read_b128 per wave), conflict-free under the same swizzle, same
    //           accumulator count.  The chip holds a higher clock on this shape for the same work: the bare MFMA + LDS
    //           loop of this kernel ran 215 vs 245 us per unit of work at 1.82 vs 1.56 GHz (tools/micro/mfma_shape.hip).
    const int frow = (lane & 31) * 128, fsw = ((lane & 31) >> 1) & 7, fh = lane >> 5;
    int fo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) fo[c] = frow + (((2 * c + fh) ^ fsw) * 16);
    const int r16 = lane & 15, q4 = lane >> 4;
    // 16-row tiles start at multiples of 16 rows, so the swizzle term (row >> 1) & 7 depends on r16 alone
    const int fo16h = r16 * 128 + ((q4 ^ ((r16 >> 1) & 7)) * 16), fo16l = r16 * 128 + (((4 + q4) ^ ((r16 >> 1) & 7)) * 16);
    const int offA = wm * WM * 128, offB = (BM + wn * WN) * 128;

    // Operand order: the weight fragment is the MFMA's A operand and the activation fragment its B operand, so the
    // accumulator comes out transposed: lane -> output ROW, registers -> 4-column runs.  A 32 x 32 block of the wave tile
    // is four "sub-runs" s = 0..3 of 4 columns per lane:
    //   MF = 0: row = lane & 31,               col = 8 s + 4 (lane >> 5)          (registers 4 s .. 4 s + 3 of the 32x32 tile)
    //   MF = 1: row = 16 (s >> 1) + (lane & 15), col = 16 (s & 1) + 4 (lane >> 4)   (the 16x16 tile (s >> 1, s & 1) of the block)
    // The epilogue then moves 16 bytes (fp32) or 8 + 8 bytes (S32) per lane and store, and bias / gamma are per-register vectors.
    constexpr int TM16 = WM / 16, TN16 = WN / 16, TNH = TN16 / 2;
    f32x16 accm[MF ? 1 : TM][MF ? 1 : TN], accc[MF ? 1 : TM][MF ? 1 : TN];
    f32x4 am[MF ? TM16 : 1][MF ? TN16 : 1], ac[MF ? TM16 : 1][MF ? TN16 : 1];
    struct Frags { f16x8 ah[TM], al[TM], bh[TN], bl[TN]; };
    auto read_frags = [&](int stage, int s, Frags& F) {
        const char* sA = smem_s + stage * STG + offA;
        const char* sB = smem_s + stage * STG + offB;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            F.ah[i] = *reinterpret_cast<const f16x8*>(sA + i * 32 * 128 + fo[s]);
            F.al[i] = *reinterpret_cast<const f16x8*>(sA + i * 32 * 128 + fo[2 + s]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            F.bh[j] = *reinterpret_cast<const f16x8*>(sB + j * 32 * 128 + fo[s]);
            F.bl[j] = *reinterpret_cast<const f16x8*>(sB + j * 32 * 128 + fo[2 + s]);
        }
    };
    auto mfma_block = [&](const Frags& F) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                accm[MF ? 0 : i][MF ? 0 : j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.bh[j], F.ah[i], accm[MF ? 0 : i][MF ? 0 : j], 0, 0, 0);
                accc[MF ? 0 : i][MF ? 0 : j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.bl[j], F.ah[i], accc[MF ? 0 : i][MF ? 0 : j], 0, 0, 0);
                accc[MF ? 0 : i][MF ? 0 : j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.bh[j], F.al[i], accc[MF ? 0 : i][MF ? 0 : j], 0, 0, 0);
            }
    };
    // MF = 1: the activation fragments of a step (all TM16 row tiles) and the weight fragments in two halves of TNH column tiles
    struct FragA { f16x8 h[TM16], l[TM16]; };
    struct FragB { f16x8 h[TNH], l[TNH]; };
    auto read_a16 = [&](int stage, FragA& F) {
        const char* sA = smem_s + stage * STG + offA;
#pragma unroll
        for (int i = 0; i < TM16; ++i) {
            F.h[i] = *reinterpret_cast<const f16x8*>(sA + i * 16 * 128 + fo16h);
            F.l[i] = *reinterpret_cast<const f16x8*>(sA + i * 16 * 128 + fo16l);
        }
    };
    auto read_b16 = [&](int stage, int half, FragB& F) {
        const char* sB = smem_s + stage * STG + offB + half * TNH * 16 * 128;
#pragma unroll
        for (int j = 0; j < TNH; ++j) {
            F.h[j] = *reinterpret_cast<const f16x8*>(sB + j * 16 * 128 + fo16h);
            F.l[j] = *reinterpret_cast<const f16x8*>(sB + j * 16 * 128 + fo16l);
        }
    };
    auto mfma16_block = [&](const FragA& A, const FragB& Bf, auto half_c, auto&& between) {
        constexpr int half = decltype(half_c)::value;          // a constant: the accumulators must stay in registers
#pragma unroll
        for (int jj = 0; jj < TNH; ++jj)
#pragma unroll
            for (int i = 0; i < TM16; ++i) {
                between(jj * TM16 + i);
                constexpr int jbase = half * TNH;
                const int j = jbase + jj;
                f32x4& m = am[MF ? i : 0][MF ? j : 0];
                f32x4& c = ac[MF ? i : 0][MF ? j : 0];
                m = __builtin_amdgcn_mfma_f32_16x16x32_f16(Bf.h[jj], A.h[i], m, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Bf.l[jj], A.h[i], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Bf.h[jj], A.l[i], c, 0, 0, 0);
            }
    };

    // NSTAGE-deep ring of LDS stages, ONE barrier per K step, placed in the MIDDLE of the step:
    //   top    : DMA of the K tile NSTAGE-1 steps ahead -> the stage the previous step's tile occupied (all its reads
    //            retired before the previous barrier); read the second-half fragments of this step's tile; MFMAs on
    //            its first half (already in registers)
    //   middle : each wave waits for ITS OWN pieces of the next K tile (counted vmcnt: younger tiles stay in flight)
    //            and for its LDS reads, then the barrier makes that tile visible to everyone
    //   bottom : read the first-half fragments of the next K tile, MFMAs on the second half of this one
    // so no wave sits behind a barrier with nothing to issue: fragments always arrive under the other half's MFMAs.
    // (MF = 0: the halves are the two 16-deep k halves of the step.  MF = 1: the halves are the first and the last TNH
    // weight column tiles; the activation fragments serve both halves, so the next step's are read last in the bottom
    // phase, into the registers the second half's MFMAs have just read.)
    // The stream does not stop at an output-tile boundary: the first K tiles of the workgroup's next output tile are
    // already landing while the last steps of this one run, its first fragments are read before the epilogue, and
    // the epilogue's own loads and stores simply queue behind them.  Past the last tile the DMAs are issued all the
    // same with out-of-range offsets (constant wait counts).
    const int nk = p.K / SBK;
    f32x4 fake = {0.1f * lane, 0.2f, -0.3f, 0.01f * lane};
    const __amdgpu_buffer_rsrc_t rsFake = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 1024, 0x00020000);
    float amax = 0.f;            // largest magnitude this wave converts to the split-f16 form (range_report at the end)
    // operands may carry a per-tensor power-of-two scale (weights at load, the single-stage entry points): the
    // accumulators are brought back by acc_s, exactly (a power of two), before bias and activation
    const float acc_s = p.acc_scale_dev ? *p.acc_scale_dev : p.acc_scale;
    const float lo_s = acc_s * (1.f / 2048.f);
    if ((dbg & 64) && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);     // experiment: static priority for the younger half
    unsigned long long st_c0 = 0, st_r0 = 0;
    if (dbg & 1024) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    build_table(blockIdx.x, 0);
    __syncthreads();
    loader_set_tile(blockIdx.x, 0);
    if (dma_wave) {
#pragma unroll
        for (int s = 0; s < (KS == 2 ? 4 : NSTAGE - 1); ++s) load_tile(s);   // needs nk >= NSTAGE - 1 when a next tile exists (host)
    }
    if (KS == 2) wait_vm_lgkm<2 * NPT>(); else wait_vm_lgkm<(NSTAGE - 2) * NPT>();
    __builtin_amdgcn_s_barrier();
    Frags F0, F1;
    FragA Fa;
    FragB F0b, F1b;
    if (MF) { if (mfma_wave) { read_a16(0, Fa); read_b16(0, 0, F0b); } }
    else read_frags(0, 0, F0);
    int rs = 0, ws = KS == 2 ? 4 : NSTAGE - 1;
    int c_par = 0;
    FragA Fa2;           // KS = 2: fragments of the second K tile of a pair
    FragB F0b2, F1b2;
    for (int vb = blockIdx.x; vb < ntiles; vb += G, c_par ^= 1) {
        // the table of this workgroup's next output tile: the loader turns to it NSTAGE-1 steps before this tile's
        // K loop ends, i.e. after at least one of the barriers below (host: nk >= NSTAGE + 1 in persistent launches)
        build_table(vb + G, c_par ^ 1);
        int bm, bn;
        tile_coords(vb, bm, bn);
        // This wave's WN bias (and gamma) values go into a private LDS cache by DMA now, a K loop ahead of their use: a
        // global load in the epilogue would have to wait for vmcnt(0), i.e. for every store issued before it, and the
        // epilogue would run one store round trip at a time (it did: 15 of pwconv1's 110 us).  The K loop's counted
        // waits only ever leave the youngest DMA pieces outstanding, so the cache is complete long before it is read.
        // (loader waves skip the epilogue and would be here while the MFMA waves still read the previous tile's cache: in a
        // persistent launch the two roles meet before it is overwritten)
        if constexpr (PROD != 0 && PCACHE) { if (vb != (int)blockIdx.x) __syncthreads(); }
        if (PCACHE && dma_wave && wave < NW && lane < WN / 4) {
            const int nb4 = (bn * BN + wn * WN + 4 * lane) * 4;
            char* pc = smem_s + p.pc_off + wave * PC_BYTES;
            if (p.bias) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsBias, (lds_ptr_t)pc, 16, nb4, 0, 0, 0);
            if (EPI == EPI_BIAS_GAMMA_RES) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsGamma, (lds_ptr_t)(pc + WN * 4), 16, nb4, 0, 0, 0);
        }
        if (MF) {
#pragma unroll
            for (int i = 0; i < TM16; ++i)
#pragma unroll
                for (int j = 0; j < TN16; ++j) { am[MF ? i : 0][MF ? j : 0] = (f32x4){0.f, 0.f, 0.f, 0.f}; ac[MF ? i : 0][MF ? j : 0] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { accm[MF ? 0 : i][MF ? 0 : j][r] = 0.f; accc[MF ? 0 : i][MF ? 0 : j][r] = 0.f; }
        }
        if constexpr (KS == 2) {
            // K tiles kt, kt + 1 (stages rs, rs + 1) are resident and visible; the next pair is in flight since the previous
            // iteration and is waited for at the bottom of this one; the pair after it is requested now.  nk is even (host)
            for (int kt = 0; kt < nk; kt += 2) {
                const int rs1 = rs + 1;                          // rs is even, < 6
                if (dma_wave) {
                    load_tile(ws);
                    load_tile(ws + 1);
                }
                if (mfma_wave) {
                    // every fragment of both tiles is requested before the first MFMA (both tiles have been visible since the
                    // last barrier): the LDS latency is paid once per pair, the MFMAs then run back to back
                    read_b16(rs, 1, F1b);
                    read_a16(rs1, Fa2);
                    read_b16(rs1, 0, F0b2);
                    read_b16(rs1, 1, F1b2);
                    mfma16_block(Fa, F0b, std::integral_constant<int, 0>{}, [](int) {});
                    mfma16_block(Fa, F1b, std::integral_constant<int, 1>{}, [](int) {});
                    mfma16_block(Fa2, F0b2, std::integral_constant<int, 0>{}, [](int) {});
                    mfma16_block(Fa2, F1b2, std::integral_constant<int, 1>{}, [](int) {});
                }
                wait_vm_lgkm<2 * NPT>();     // all but the pair just requested; and this wave's LDS reads (their stages are the next DMA target)
                __builtin_amdgcn_s_barrier();
                rs = rs + 2 == 6 ? 0 : rs + 2;
                ws = ws + 2 == 6 ? 0 : ws + 2;
                if (mfma_wave) {
                    read_a16(rs, Fa);        // first fragments of the next pair (after the last pair: of the next output tile);
                    read_b16(rs, 0, F0b);    // their latency passes under the DMA issue at the top of the loop
                }
            }
        } else
        for (int kt = 0; kt < nk; ++kt) {
            // (tried: the younger half of the waves issuing its DMA pieces after the first-half MFMAs instead of before them,
            // so that one wave's DMA issue runs beside its SIMD partner's MFMAs: 5.79 vs 5.80 ms per step A/B on one box: nothing)
            constexpr bool SPREAD = MF && !(dbg & 2048) && NPT <= TNH * TM16;     // one DMA piece in front of each MFMA group (below)
            if (!(dbg & 1) && !SPREAD) load_tile(ws);
            if (MF) {
                if (!(dbg & 16)) read_b16(rs, 1, F1b);
                if (dbg & 8192) __builtin_amdgcn_s_setprio(1);
                if (!(dbg & 2)) mfma16_block(Fa, F0b, std::integral_constant<int, 0>{}, [&](int grp) {
                    if constexpr (SPREAD) {
                        if (!(dbg & 1)) {
                            // grp is a compile-time constant after unrolling; the pieces go out in front of groups 0 .. NPT-1
#define WT_PIECE(I) if (NPT > I && grp == I) load_piece(ws, std::integral_constant<int, (NPT > I ? I : 0)>{});
                            WT_PIECE(0) WT_PIECE(1) WT_PIECE(2) WT_PIECE(3) WT_PIECE(4) WT_PIECE(5)
                            WT_PIECE(6) WT_PIECE(7) WT_PIECE(8) WT_PIECE(9) WT_PIECE(10) WT_PIECE(11)
#undef WT_PIECE
                        }
                    }
                });
                if (dbg & 8192) __builtin_amdgcn_s_setprio(0);
                if (SPREAD && !(dbg & 1)) load_advance();
            } else {
                if (!(dbg & 16)) read_frags(rs, 1, F1);
                if (!(dbg & 2)) mfma_block(F0);
            }
            // timing experiment: what an epilogue drained inside the K loop would cost - per K step the GELU + split of one
            // 4-column run and its two 8-byte stores (out of range: counted, dropped), scheduled among the MFMAs.  With bit
            // 16777216 the two waves of a SIMD drain in different halves of the step (waves 0-3 after the barrier, 4-7 in front
            // of it), so that one wave's vector work meets the other's MFMAs
            auto fake_drain = [&]() {
                fake = gelu_erfc_s4(fake + (f32x4){1e-3f, 2e-3f, 3e-3f, 4e-3f});
                f16x4 fh, fl;
                split4_f16(fake.x, fake.y, fake.z, fake.w, fh, fl);
                amax = amax4(amax, fake.x, fake.y, fake.z, fake.w);
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, fh), rsFake, (int)0x7ffffff0, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, fl), rsFake, (int)0x7ffffff0, 0, 0);
            };
            if constexpr ((dbg & 2097152) != 0) {
                if (!(dbg & 16777216) || wave >= NW / 2) fake_drain();
            }
            if (dbg & 32768) { if (!(dbg & 32)) wait_vm_lgkm<63>(); }                 // timing experiment: DMA issued, never waited for (races)
            else if (!(dbg & 32)) {
                // (the drain's two stores are younger than this step's DMA pieces only where they were issued behind them)
                if ((dbg & 2097152) && (!(dbg & 16777216) || wave >= NW / 2)) wait_vm_lgkm<(NSTAGE - 2) * NPT + 2>();
                else wait_vm_lgkm<(NSTAGE - 2) * NPT>();
            }
            if (!(dbg & 8)) __builtin_amdgcn_s_barrier();
            rs = rs + 1 == NSTAGE ? 0 : rs + 1;
            ws = ws + 1 == NSTAGE ? 0 : ws + 1;
            if constexpr ((dbg & 2097152) != 0 && (dbg & 16777216) != 0) {
                if (wave < NW / 2) fake_drain();
            }
            if (MF) {
                if (!(dbg & 16)) read_b16(rs, 0, F0b);       // after the very last step: a harmless read of a zero-filled stage
                if (dbg & 8192) __builtin_amdgcn_s_setprio(1);
                if (!(dbg & 2)) mfma16_block(Fa, F1b, std::integral_constant<int, 1>{}, [](int) {});
                if (dbg & 8192) __builtin_amdgcn_s_setprio(0);
                if (!(dbg & 16)) read_a16(rs, Fa);
            } else {
                if (!(dbg & 16)) read_frags(rs, 0, F0);
                if (!(dbg & 2)) mfma_block(F1);
            }
        }
        if (dbg & 4) {      // timing builds without an epilogue: keep the accumulators (and so the MFMAs) alive through
                            // a store the compiler cannot rule out (alpha is never this value)
            if (p.alpha == -12345.f) {
                if (MF) {
#pragma unroll
                    for (int i = 0; i < TM16; ++i)
#pragma unroll
                        for (int j = 0; j < TN16; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) p.C[(tid * 4 + r) * 2] = am[MF ? i : 0][MF ? j : 0][r] + ac[MF ? i : 0][MF ? j : 0][r];
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) p.C[(tid * 16 + r) * 2] = accm[MF ? 0 : i][MF ? 0 : j][r] + accc[MF ? 0 : i][MF ? 0 : j][r];
                }
            }
            continue;
        }

    // ------------------------------------------------------------------------- epilogue
    if (loader_wave) continue;               // (no barrier from here to the end of the tile loop)
    const int m_w = bm * BM + wm * WM, n_w = bn * BN + wn * WN;
    float* __restrict__ Cg = p.C + (long)z * p.zC;
    // sub-run s of a 32 x 32 block: its row and first column inside the block (see the accumulator layouts above)
    auto sub_row = [&](int s) { return MF ? 16 * (s >> 1) + r16 : (lane & 31); };
    auto sub_col = [&](int s) { return MF ? 16 * (s & 1) + 4 * q4 : 8 * s + 4 * (lane >> 5); };
    auto acc4 = [&](int i, int j, int s) {
        f32x4 v;
        if (MF) {
            const f32x4 m = am[MF ? 2 * i + (s >> 1) : 0][MF ? 2 * j + (s & 1) : 0], c = ac[MF ? 2 * i + (s >> 1) : 0][MF ? 2 * j + (s & 1) : 0];
            v.x = fmaf(c.x, lo_s, m.x * acc_s);
            v.y = fmaf(c.y, lo_s, m.y * acc_s);
            v.z = fmaf(c.z, lo_s, m.z * acc_s);
            v.w = fmaf(c.w, lo_s, m.w * acc_s);
        } else {
            v.x = fmaf(accc[MF ? 0 : i][MF ? 0 : j][4 * s + 0], lo_s, accm[MF ? 0 : i][MF ? 0 : j][4 * s + 0] * acc_s);
            v.y = fmaf(accc[MF ? 0 : i][MF ? 0 : j][4 * s + 1], lo_s, accm[MF ? 0 : i][MF ? 0 : j][4 * s + 1] * acc_s);
            v.z = fmaf(accc[MF ? 0 : i][MF ? 0 : j][4 * s + 2], lo_s, accm[MF ? 0 : i][MF ? 0 : j][4 * s + 2] * acc_s);
            v.w = fmaf(accc[MF ? 0 : i][MF ? 0 : j][4 * s + 3], lo_s, accm[MF ? 0 : i][MF ? 0 : j][4 * s + 3] * acc_s);
        }
        return v;
    };

    if constexpr (EPI == EPI_ARGMAX) {
        // VQ (core_vq.py:176-182): per row, the best of this wave's WN columns of -(|x|^2 - 2 x.e + |e|^2), lowest
        // index on ties.  MF = 0: a lane holds one row, its partner lane + 32 the other half of the columns.  MF = 1: a
        // lane holds two rows (16 apart), and the four lanes r16, r16 + 16, + 32, + 48 share a row's columns.
        const int part = bn * WAVES_N + wn;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int rsel = 0; rsel < (MF ? 2 : 1); ++rsel) {
                const int m = m_w + i * 32 + (MF ? 16 * rsel + r16 : (lane & 31));
                const float xx = (m < p.M) ? p.vq_xx[m] : 0.f;
                float best = -INFINITY;
                int bidx = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int ss = 0; ss < (MF ? 2 : 4); ++ss) {
                        const int s = MF ? 2 * rsel + ss : ss;          // ascending columns within the lane either way
                        const int n = n_w + j * 32 + sub_col(s);
                        if (n >= p.N) continue;
                        const f32x4 dot = acc4(i, j, s);
                        const f32x4 ee = *reinterpret_cast<const f32x4*>(p.vq_ee + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float d = -((xx - 2.f * dot[e]) + ee[e]);
                            if (d > best) { best = d; bidx = n + e; }        // ascending n within the lane: strict > keeps the lowest
                        }
                    }
#pragma unroll
                for (int off = (MF ? 16 : 32); off <= 32; off <<= 1) {
                    const float ov = __shfl_xor(best, off, 64);
                    const int oi = __shfl_xor(bidx, off, 64);
                    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
                }
                if ((MF ? q4 == 0 : lane < 32) && m < p.M) {
                    p.vq_pval[(long)m * p.vq_nparts + part] = best;
                    p.vq_pidx[(long)m * p.vq_nparts + part] = bidx;
                }
            }
    } else if constexpr (EPI == EPI_HEAD) {
        // packed rows come in 32-row groups: 16 log-magnitude rows, then the 16 phase rows of the same spectrum slots
        // (weights.cpp), so every 32 x 32 block holds both halves of 16 slots whatever the tile width: the sub-run s of
        // columns < 16 pairs with the sub-run of the same rows 16 columns on
        auto head_run = [&](int i, int j, int h, f32x4& re, f32x4& im) {
            const int s = MF ? 2 * h : h, sp = MF ? s + 1 : s + 2;
            const int pc = n_w + j * 32 + sub_col(s);         // packed row of the log-magnitude; phase 16 later
            const f32x4 bmag = *reinterpret_cast<const f32x4*>(pcw + (pc - n_w) * 4);
            const f32x4 bph = *reinterpret_cast<const f32x4*>(pcw + (pc + 16 - n_w) * 4);
            const f32x4 lm = acc4(i, j, s) + bmag, ph = acc4(i, j, sp) + bph;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float mag = fminf((dbg & 4194304) ? expf(lm[e]) : exp_head(fminf(lm[e], 88.f)), 100.f);      // heads.py:55-56
                float sn, cs;
                if (dbg & 4194304) sincosf(ph[e], &sn, &cs); else sincos_head(ph[e], sn, cs);     // one range reduction for both
                re[e] = mag * cs;
                im[e] = mag * sn;
            }
        };
        if (OUT == OUT_S32 && WN % 64 == 0 && p.stage_epi && !(dbg & 8388608)) {
            // Staged: two neighbouring blocks hold the 32 slots of one S32 group (128 bytes of a spectrum row, re and im each).
            // The 32 x 32-slot tile goes through the wave's LDS scratch in the S32 row image and leaves as full lines, like
            // the staged epilogue of the hot layers below (direct form: 8-byte pieces, 16 stores per block)
            char* sc = smem_s + p.stage_off + wave * 4096;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jp = 0; jp + 1 < TN; jp += 2) {
                    const int n0 = n_w + jp * 32;
                    if (n0 >= p.N) continue;                       // wave-uniform; N % 64 == 0 (host)
                    f32x4 re[4], im[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) head_run(i, jp + (k >> 1), k & 1, re[k], im[k]);
#pragma unroll
                    for (int pz = 0; pz < 2; ++pz) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int sk = MF ? 2 * (k & 1) : (k & 1);
                            const int rw = sub_row(sk), cw = 16 * (k >> 1) + (sub_col(sk) & 15), sww = (rw >> 1) & 7;
                            const f32x4 v = pz ? im[k] : re[k];
                            amax = amax4(amax, v.x, v.y, v.z, v.w);
                            f16x4 hi, lo;
                            split4_f16(v.x, v.y, v.z, v.w, hi, lo);
                            *reinterpret_cast<f16x4*>(sc + rw * 128 + (((cw >> 3) ^ sww) * 16) + 2 * (cw & 7)) = hi;
                            *reinterpret_cast<f16x4*>(sc + rw * 128 + (((4 + (cw >> 3)) ^ sww) * 16) + 2 * (cw & 7)) = lo;
                        }
                        f32x4 q[4];
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int r = 8 * it + (lane >> 3), ch = lane & 7;
                            q[it] = *reinterpret_cast<const f32x4*>(sc + r * 128 + ((ch ^ ((r >> 1) & 7)) * 16));
                        }
                        const int f0 = (n0 >> 6) * 32 + pz * p.head_kb;           // first slot of the group, as a float index of the row
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int r = 8 * it + (lane >> 3), ch = lane & 7;
                            const int m = m_w + i * 32 + r;
                            if (m < p.M && !(dbg & 128)) store_c16<!(dbg & 4096)>(Cg + (long)m * p.c_rstride + f0 + 4 * ch, q[it]);
                        }
                    }
                }
        } else
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int s = MF ? 2 * h : h;
                    const int m = m_w + i * 32 + sub_row(s);
                    const int pc = n_w + j * 32 + sub_col(s);
                    if (m >= p.M || pc >= p.N) continue;
                    float* crow = Cg + (long)m * p.c_rstride;
                    f32x4 re, im;
                    head_run(i, j, h, re, im);
                    const int f = (pc >> 5) * 16 + (pc & 15);          // spectrum slot
                    if (dbg & 128) {           // timing experiment: no stores
                        amax = amax4(amax4(amax, re.x, re.y, re.z, re.w), im.x, im.y, im.z, im.w);
                    } else if (OUT == OUT_S32) {
                        store_s32_x4(crow, f, re, amax);
                        store_s32_x4(crow, p.head_kb + f, im, amax);
                    } else {
                        *reinterpret_cast<f32x4*>(crow + f) = re;
                        *reinterpret_cast<f32x4*>(crow + p.head_kb + f) = im;
                    }
                }
        }
    } else {
        // Staged form (the hot layers): a lane holds 4-column runs of ONE row, so direct stores touch 32 rows x 32 B
        // per instruction.  Each 32x32 sub-tile goes through a 4 KB wave-private LDS scratch instead (chunk-swizzled,
        // conflict-free both ways) and is read back with 8 lanes per 128-byte row: every store is then a full line
        // (measured: pwconv1 + GELU 114 -> 109 us; the residual epilogues gained nothing and stay direct).
        constexpr bool DUAL = OUT == OUT_S32_DUAL_ELU || OUT == OUT_F32_AND_S32;
        constexpr bool CAN_STAGE = ((OUT == OUT_F32 || DUAL) && EPI == EPI_BIAS) ||
                                   (OUT == OUT_S32 && (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_ELU));
        if (CAN_STAGE && p.stage_epi) {
            char* sc = smem_s + p.stage_off + wave * 4096;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n0 = n_w + j * 32;
                    if (n0 >= p.N) continue;                       // wave-uniform
                    // a dual-output launch sends the block through the scratch twice, once per destination format
#pragma unroll
                    for (int pz = 0; pz < (DUAL ? 2 : 1); ++pz) {
                        const bool as_f32 = OUT == OUT_F32 || (OUT == OUT_F32_AND_S32 && pz == 0);
                        // the block's four bias vectors in one LDS round trip
                        f32x4 bq[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int n = n0 + sub_col(g);
                            bq[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                            if (p.bias && n < p.N && !(dbg & 256))
                                bq[g] = PCACHE ? *reinterpret_cast<const f32x4*>(pcw + (n - n_w) * 4) : *reinterpret_cast<const f32x4*>(p.bias + n);
                        }
                        f32x4 vb[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) vb[g] = acc4(i, j, g) + bq[g];
                        if (EPI == EPI_BIAS_GELU && !(dbg & 512)) {
                            if (dbg & 16384) {
#pragma unroll
                                for (int g = 0; g < 4; ++g) vb[g] = gelu_erf_s4(vb[g]);
                            } else if (dbg & 524288) {
#pragma unroll
                                for (int g = 0; g < 4; ++g) vb[g] = gelu_erfc_s4(vb[g]);
                            } else {
                                gelu_erfc_x16(vb);
                            }
                        }
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int rw = sub_row(g), cw = sub_col(g), sww = (rw >> 1) & 7;
                            f32x4 v = vb[g];
                            if (EPI == EPI_BIAS_ELU || (OUT == OUT_S32_DUAL_ELU && pz == 1)) {
                                v.x = elu_s(v.x); v.y = elu_s(v.y); v.z = elu_s(v.z); v.w = elu_s(v.w);
                            }
                            if (as_f32) {
                                *reinterpret_cast<f32x4*>(sc + rw * 128 + (((cw >> 2) ^ sww) * 16)) = v;
                            } else {
                                amax = amax4(amax, v.x, v.y, v.z, v.w);
                                f16x4 hi, lo;
                                split4_f16(v.x, v.y, v.z, v.w, hi, lo);
                                *reinterpret_cast<f16x4*>(sc + rw * 128 + (((cw >> 3) ^ sww) * 16) + 2 * (cw & 7)) = hi;
                                *reinterpret_cast<f16x4*>(sc + rw * 128 + (((4 + (cw >> 3)) ^ sww) * 16) + 2 * (cw & 7)) = lo;
                            }
                        }
                        // a wave's LDS operations execute in order and the scratch is private to the wave: no barrier
                        float* dbase = pz == 0 ? Cg : p.C2 + (long)z * p.zC;
                        // all four read-backs first (one LDS round trip per 32x32 block, not four), then the stores
                        f32x4 q[4];
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int r = 8 * it + (lane >> 3), ch = lane & 7;
                            q[it] = *reinterpret_cast<const f32x4*>(sc + r * 128 + ((ch ^ ((r >> 1) & 7)) * 16));
                        }
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int r = 8 * it + (lane >> 3), ch = lane & 7;
                            const int m = m_w + i * 32 + r;
                            const int n = n0 + 4 * ch;             // fp32 columns; S32: byte ch * 16 of the group at n0
                            if (m < p.M && (!as_f32 || n < p.N) && !(dbg & 128)) {
                                float* dst = dbase + (long)m * p.c_rstride + n;
                                store_c16<!(dbg & 4096)>(dst, q[it]);
                            }
                        }
                    }
                }
            }
        } else
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m = m_w + i * 32 + sub_row(g);
                    const int n = n_w + j * 32 + sub_col(g);        // columns n .. n+3 (N % 4 == 0)
                    if (m >= p.M || n >= p.N) continue;
                    float* crow = Cg + (long)m * p.c_rstride;
                    f32x4 v = acc4(i, j, g);
                    if (EPI == EPI_SCALE || EPI == EPI_BIAS_ROW) {
                        v = EPI == EPI_SCALE ? v * p.alpha : v + p.bias[m];
                        // these two take any N: the columns of a partial last run are written as zeros (pad columns
                        // of the row pitch, which the consumers rely on being zero)
                        if (n + 3 >= p.N) { if (n + 1 >= p.N) v.y = 0.f; if (n + 2 >= p.N) v.z = 0.f; v.w = 0.f; }
                    }
                    else if (p.bias) v += PCACHE ? *reinterpret_cast<const f32x4*>(pcw + (n - n_w) * 4) : *reinterpret_cast<const f32x4*>(p.bias + n);
                    if (EPI == EPI_BIAS_RES) {
                        v = v + *reinterpret_cast<const f32x4*>(p.R + (long)m * p.r_rstride + n);
                    } else if (EPI == EPI_BIAS_RES_ELU) {
                        v = v + *reinterpret_cast<const f32x4*>(p.R + (long)m * p.r_rstride + n);
                        v.x = elu_s(v.x); v.y = elu_s(v.y); v.z = elu_s(v.z); v.w = elu_s(v.w);
                    } else if (EPI == EPI_BIAS_ELU) {
                        v.x = elu_s(v.x); v.y = elu_s(v.y); v.z = elu_s(v.z); v.w = elu_s(v.w);
                    } else if (EPI == EPI_BIAS_GELU) {
                        v = (dbg & 16384) ? gelu_erf_s4(v) : gelu_erfc_s4(v);
                    } else if (EPI == EPI_BIAS_GAMMA_RES) {
                        const f32x4 gm = *reinterpret_cast<const f32x4*>(pcw + WN * 4 + (n - n_w) * 4);      // EPI_BIAS_GAMMA_RES: always cached
                        v = *reinterpret_cast<const f32x4*>(p.R + (long)m * p.r_rstride + n) + gm * v;
                    }
                    if (OUT == OUT_S32 || OUT == OUT_S32_DUAL_ELU) store_s32_x4(crow, n, v, amax);
                    else store_c16<(dbg & 262144) != 0>(crow + n, v);
                    if (OUT == OUT_S32_DUAL_ELU) {
                        f32x4 ev;
                        ev.x = elu_s(v.x); ev.y = elu_s(v.y); ev.z = elu_s(v.z); ev.w = elu_s(v.w);
                        store_s32_x4(p.C2 + (long)z * p.zC + (long)m * p.c_rstride, n, ev, amax);
                    } else if (OUT == OUT_F32_AND_S32) {
                        store_s32_x4(p.C2 + (long)z * p.zC + (long)m * p.c_rstride, n, v, amax);
                    }
                }
        }
    }
    }   // persistent tile loop
    if ((dbg & 1024) && p.dbg_stamps && tid == 0) {
        p.dbg_stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_c0;
        p.dbg_stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
    range_report(p.status, amax);
    if (p.stamp_end) {                       // ... and the latest exit, after every wave's stores have been acknowledged
        wait_vm_lgkm<0>();
        __syncthreads();
        if (tid == 0)
            __hip_atomic_fetch_max(p.stamp_end, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    wait_vm_lgkm<0>();
}

// fp32 -> S32 (flat: rows are multiples of 32 elements, so the layout is a function of the flat index alone)
// `scale` (optional, device): the values are multiplied by scale[0], a power of two chosen by pow2_scale_kernel, first
__global__ __launch_bounds__(256) void split_s32_kernel(const float* __restrict__ x, _Float16* __restrict__ out, long n,
                                                        const float* __restrict__ scale, unsigned* status) {
    const float sc = scale ? scale[0] : 1.f;
    float amax = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i] * sc;
        amax = amax1(amax, v);
        const _Float16 h = (_Float16)v;
        const long g = (i >> 5) * 64 + (i & 31);
        out[g] = h;
        out[g + 32] = (_Float16)((v - (float)h) * 2048.f);
    }
    range_report(status, amax);
}

int launch_split_s32(const float* x, void* out, long n, hipStream_t s, const float* scale_dev) {
    if (n % 32) { set_error("split_s32: element count must be a multiple of 32"); return -1; }
    int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(split_s32_kernel, dim3(blocks), dim3(256), 0, s, x, static_cast<_Float16*>(out), n, scale_dev, g_launch.status);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// S32 -> fp32 (the lazy fp32 GEMM weights of a model that came from a packed image: weights.cpp ensure_f32_weights)
__global__ __launch_bounds__(256) void unsplit_s32_kernel(const _Float16* __restrict__ in, float* __restrict__ out, long n, float inv_scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long g = (i >> 5) * 64 + (i & 31);
        out[i] = ((float)in[g] + (float)in[g + 32] * (1.f / 2048.f)) * inv_scale;
    }
}
int launch_unsplit_s32(const void* s32, float* out, long n, float inv_scale, hipStream_t s) {
    if (n % 32) { set_error("unsplit_s32: element count must be a multiple of 32"); return -1; }
    int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(unsplit_s32_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const _Float16*>(s32), out, n, inv_scale);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// Per-tensor power-of-two scale for the single-stage entry points (the plans' producers write S32 directly, with
// the overflow report instead): amax_bits_kernel leaves max |x| (as its bit pattern: positive floats order like
// unsigned integers) in bits[0]; pow2_scale_kernel turns the maxima of two tensors into scale_a, scale_b = powers of two
// that bring each maximum into [1, 2) (1 for an all-zero or non-finite tensor), and out[2] = 1 / (scale_a * scale_b),
// the factor that restores the accumulators.
__global__ __launch_bounds__(256) void amax_bits_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ bits) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(bits, __float_as_uint(m));
}
__global__ void pow2_scale_kernel(const unsigned* __restrict__ bits, float* __restrict__ out) {
    float sc[2];
    for (int i = 0; i < 2; ++i) {
        const unsigned e = (bits[i] >> 23) & 0xffu;             // biased exponent of the maximum
        // 2^(127 - e) brings the maximum into [1, 2); keep the scale itself a normal float, and leave zero / inf / NaN alone
        sc[i] = (e == 0u || e == 255u || e >= 253u) ? 1.f : __uint_as_float((254u - e) << 23);
    }
    out[0] = sc[0];
    out[1] = sc[1];
    out[2] = (1.f / sc[0]) * (1.f / sc[1]);
}
int launch_pow2_scales(const float* a, long na, const float* b, long nb, unsigned* bits2, float* out3, hipStream_t s) {
    WT_HIP_CHECK(hipMemsetAsync(bits2, 0, 2 * sizeof(unsigned), s));
    const int ba = (int)((na + 255) / 256 < 2048 ? (na + 255) / 256 : 2048), bb = (int)((nb + 255) / 256 < 2048 ? (nb + 255) / 256 : 2048);
    hipLaunchKernelGGL(amax_bits_kernel, dim3(ba), dim3(256), 0, s, a, na, bits2);
    hipLaunchKernelGGL(amax_bits_kernel, dim3(bb), dim3(256), 0, s, b, nb, bits2 + 1);
    hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(1), 0, s, bits2, out3);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------- host side
template <int BM, int BN, int WMs, int WNs, int NSTAGE, int EPI, int OUT, int WPS = 2, int LABDBG = 0, int KS = 1, int PROD = 0>
static int launch16s_one(const GemmArgs& a, hipStream_t s) {
    if (KS == 2 && ((a.K / SBK) % 2 || a.K / SBK < 6)) { set_error("gemm16s: two K tiles per barrier need an even number (>= 6) of K tiles"); return -1; }
    static PerDeviceOnce attr_once;
    constexpr size_t stage_bytes = (size_t)NSTAGE * (BM + BN) * 128;
    size_t smem = stage_bytes + 2ull * a.taps * BM * sizeof(unsigned) + (a.A2 ? 2ull * BM * sizeof(unsigned) : 0);
    constexpr size_t smem_cap = 160 * 1024;
    constexpr size_t smem_want = stage_bytes + 2ull * 32 * BM * sizeof(unsigned) + (size_t)WMs * WNs * 4096 + 8192 + 256;
    constexpr size_t smem_max = smem_want < smem_cap ? smem_want : smem_cap;
    static_assert(stage_bytes + 2 * BM * sizeof(unsigned) <= smem_cap, "LDS budget");
    if (smem > smem_cap) { set_error("gemm16s: too many taps for this tile's LDS budget"); return -1; }
    using kern_t = void (*)(const GemmArgs);
    kern_t kern = gemm16s_kernel<BM, BN, WMs, WNs, NSTAGE, EPI, OUT, LABDBG, WT_GEMM16S_MF, WPS, KS, PROD>;
#ifdef WT_LAB
    // LAB builds only: the timing-experiment builds of the tile the ConvNeXt GEMMs run on (tools/gemm16s_bench.py dbg): the
    // masks of the ablation ladder, each with the clock stamps (+1024), selected by WT_GEMM16S_DBG per launch
    constexpr bool has_dbg = LABDBG == 0 && KS == 1 && BM == 128 && BN == 192 && WMs == 4 && NSTAGE == 3 && ((EPI == EPI_BIAS && OUT == OUT_F32) || (EPI == EPI_BIAS_GELU && OUT == OUT_S32));
    int dbg_req = 0;
    if (const char* e = lab_env("WT_GEMM16S_DBG")) dbg_req = atoi(e);
    kern_t dbg_kerns[8] = {};
    static constexpr int dbg_masks[8] = {1024, 1024 + 4, 1024 + 5, 1024 + 13, 1024 + 45, 1024 + 61, 1024 + 64, 1024 + 21};
    if constexpr (has_dbg) {
        dbg_kerns[0] = gemm16s_kernel<BM, BN, WMs, WNs, NSTAGE, EPI, OUT, 1024, WT_GEMM16S_MF, WPS>;
        dbg_kerns[1] = gemm16s_kernel<BM, BN, WMs, WNs, NSTAGE, EPI, OUT, 1024 + 4, WT_GEMM16S_MF, WPS>;
        dbg_kerns[2] = gemm16s_kernel<BM, BN, WMs, WNs, NSTAGE, EPI, OUT, 1024 + 5, WT_GEMM16S_MF, WPS>;
        dbg_kerns[3] = gemm16s_kernel<BM, BN, WMs, WNs, NSTAGE, EPI, OUT, 1024 + 13, WT_GEMM16S_MF, WPS>;
        dbg_kerns[4] = gemm16s_kernel<BM, BN, WMs, WNs, NSTAGE, EPI, OUT, 1024 + 45, WT_GEMM16S_MF, WPS>;
        dbg_kerns[5] = gemm16s_kernel<BM, BN, WMs, WNs, NSTAGE, EPI, OUT, 1024 + 61, WT_GEMM16S_MF, WPS>;
        dbg_kerns[6] = gemm16s_kernel<BM, BN, WMs, WNs, NSTAGE, EPI, OUT, 1024 + 64, WT_GEMM16S_MF, WPS>;
        dbg_kerns[7] = gemm16s_kernel<BM, BN, WMs, WNs, NSTAGE, EPI, OUT, 1024 + 21, WT_GEMM16S_MF, WPS>;
        if (dbg_req) {
            kern = nullptr;
            for (int i = 0; i < 8; ++i) if (dbg_masks[i] == (dbg_req | 1024)) kern = dbg_kerns[i];
            if (!kern) { set_error("gemm16s: no timing-experiment build for this WT_GEMM16S_DBG mask"); return -1; }
        }
    }
#endif
    if (int rc = attr_once.run([&]() -> int {
        WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm16s_kernel<BM, BN, WMs, WNs, NSTAGE, EPI, OUT, LABDBG, WT_GEMM16S_MF, WPS, KS, PROD>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_max));
#ifdef WT_LAB
        for (int i = 0; i < 8; ++i)
            if (dbg_kerns[i]) WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dbg_kerns[i]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_max));
#endif
        return 0;
    })) return rc;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int ntiles = tiles_m * tiles_n;
    // Persistent launch: one workgroup per slot (256 CUs x resident workgroups per CU), each walking tiles
    // b, b + G, ... with its loader streaming across the seams.  The slot count must stay a multiple of 8 so that a
    // workgroup's tiles stay on its XCD, and K must be deep enough for the table hand-over (see the kernel).
    const int per_cu = (WPS >= 2 && WMs * WNs <= 4 && smem * 2 <= smem_cap) ? 2 : 1;
    const int ncu = device_cus();
    int G = (ncu * per_cu / a.nz) & ~7;
    const char* np = lab_env("WT_GEMM16S_NONPERSISTENT");
    if (G < 8 || ntiles <= G || a.K / SBK < NSTAGE + 1 || (np && np[0] == '1')) G = ntiles;
    else {
        // the same number of rounds with the fewest workgroups (720 tiles: 240 x 3 instead of 208 x 3 + 48 x 2): the idle
        // CUs' power goes to the clock of the busy ones
        const bool bal = [] { const char* e = lab_env("WT_GEMM16S_BALANCE"); return !e || e[0] != '0'; }();
        const int rounds = (ntiles + G - 1) / G;
        const int g2 = (((ntiles + rounds - 1) / rounds) + 7) & ~7;
        if (bal && g2 < G) G = g2;
    }
    GemmArgs b = a;
    {   // per-wave bias (+ gamma) cache: WN floats each, filled by DMA at the top of every output tile
        constexpr bool pcache = EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_ELU || EPI == EPI_BIAS_RES ||
                                EPI == EPI_BIAS_RES_ELU || EPI == EPI_BIAS_GAMMA_RES || EPI == EPI_HEAD;
        constexpr size_t pc_bytes = (size_t)WMs * WNs * (BN / WNs) * 4 * (EPI == EPI_BIAS_GAMMA_RES ? 2 : 1);
        const size_t off = (smem + 127) / 128 * 128;
        if (pcache && (a.bias || EPI == EPI_BIAS_GAMMA_RES)) {
            if (off + pc_bytes > smem_cap) { set_error("gemm16s: no LDS left for the bias cache"); return -1; }
            if (EPI == EPI_BIAS_GAMMA_RES && !a.gamma) { set_error("gemm16s: this epilogue needs gamma"); return -1; }
            b.pc_off = (int)off;
            smem = off + pc_bytes;
        }
    }
    {   // staged epilogue: 4 KB of scratch per wave after the stages and tables, when it fits and the layout allows
        // (the head pairs two 32-column blocks into one 32-slot S32 group: wave tiles of whole 64-column pairs, N % 64 == 0)
        constexpr bool can_stage = ((OUT == OUT_F32 || OUT == OUT_S32_DUAL_ELU || OUT == OUT_F32_AND_S32) && EPI == EPI_BIAS) ||
                                   (OUT == OUT_S32 && (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_ELU)) ||
                                   (OUT == OUT_S32 && EPI == EPI_HEAD && (BN / WNs) % 64 == 0);
        const size_t off = (smem + 127) / 128 * 128;
        const char* ns = lab_env("WT_GEMM16S_NOSTAGE");
        constexpr bool dual = OUT == OUT_S32_DUAL_ELU || OUT == OUT_F32_AND_S32;
        if (can_stage && off + (size_t)WMs * WNs * 4096 <= smem_cap && a.N % (EPI == EPI_HEAD ? 64 : 32) == 0 &&
            !(EPI == EPI_HEAD && (a.head_kb % 32 || a.c_rstride % 32)) && !(ns && ns[0] == '1') &&
            !(dual && ns && ns[0] == '2')) {
            b.stage_epi = 1; b.stage_off = (int)off;
            smem = off + (size_t)WMs * WNs * 4096;
        }
    }
    if (!b.status) b.status = g_launch.status;
    if (g_launch.stamp_start && g_launch.stamp_end) {
        b.stamp_start = g_launch.stamp_start; b.stamp_end = g_launch.stamp_end;
        g_launch.stamp_used = true;
    }
    hipLaunchKernelGGL(kern, dim3(G, 1, a.nz), dim3(64 * WMs * WNs * (1 + PROD)), smem, s, b);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

#ifndef WT_GEMM16S_LAB      // tools/micro/gemm_lab.hip includes this file and instantiates the variants it compares itself
static int tile16s_override() {
    const char* e = lab_env("WT_GEMM16S_TILE");      // LAB builds: read per launch, tools/gemm16s_bench.py switches it in-process
    return e ? atoi(e) : -1;
}

template <int EPI, int OUT>
static int launch16s_tiled(const GemmArgs& a, hipStream_t s) {
    if constexpr (EPI == EPI_HEAD) {
        // wave tile 32 x 64: one whole 32-slot group per row block, staged full-line stores.  (7680 x 2432: 1140 tiles in 5
        // rounds of 232; 128 x 192 tiles need 4 rounds of 1.5 x the work each: 134 vs 127 us, tools/micro/gemm_lab.hip)
        {   // a few clips: the small-problem form (below); its 32-column wave tile holds whole (log-mag, phase) pairs of 16 slots
            const bool ks2_env = [] { const char* e = lab_env("WT_GEMM16S_KS"); return !e || e[0] != '1'; }();
            const int nkt = a.K / SBK;
            if (ks2_env && nkt % 2 == 0 && nkt >= 6 && ((a.M + 63) / 64) * ((a.N + 31) / 32) * a.nz <= device_cus())
                return launch16s_one<64, 32, 2, 1, 6, EPI, OUT, 2, 0, 2, 2>(a, s);
        }
        return launch16s_one<128, 128, 4, 2, 3, EPI, OUT>(a, s);
    } else if constexpr (EPI == EPI_ARGMAX) {
        return launch16s_one<128, 192, 4, 2, 3, EPI, OUT>(a, s);      // gemm16s_vq_parts() assumes this tile
    } else {
        switch (tile16s_override()) {       // experiment hook (tools/linear_bench.py)
            case 2: return launch16s_one<128, 192, 4, 2, 3, EPI, OUT>(a, s);
            case 3: return launch16s_one<128, 128, 4, 2, 3, EPI, OUT>(a, s);
            case 8: return launch16s_one<128, 32, 4, 1, 3, EPI, OUT>(a, s);
            case 9: return launch16s_one<128, 64, 4, 1, 3, EPI, OUT>(a, s);
            default: break;
        }
        if (a.N <= 64) return launch16s_one<256, 64, 8, 1, 3, EPI, OUT>(a, s);      // narrow outputs (down conv 1)
        // a handful of clips (M of a few hundred rows): with 128-column tiles fewer than 32 CUs would each walk the whole
        // K loop alone, so the problem is cut into 32-column tiles instead (4x the workgroups, a third of the work per
        // K step).  Every tile shape accumulates K in the same order: results do not depend on the choice
        const long t128 = ((a.M + 127) / 128) * ((a.N + 127) / 128) * a.nz;
        // ... and two K tiles per barrier where K allows (KS = 2: these launches are latency chains, see the kernel)
        const bool ks2_env = [] { const char* e = lab_env("WT_GEMM16S_KS"); return !e || e[0] != '1'; }();
        const int nkt = a.K / SBK;
        const bool ks2 = ks2_env && nkt % 2 == 0 && nkt >= 6;
        // (with it a 128 x 32 tile walks K faster than a 128 x 64 tile does: it is used as long as all of its tiles run at once)
        // ... and loader waves beside the MFMA waves (PROD).  Same-process ladder on pwconv2 at B = 1 (tools/micro/gemm_lab.hip):
        // 128x32 one tile per barrier 35.6 us, two per barrier 32.2, + 4 loader waves 28-30; 64x32 + 2 loader waves 25, + 4: 23.7
        const long cols32 = ((a.N + 31) / 32) * a.nz;
        const long t32 = ((a.M + 127) / 128) * cols32, t64 = ((a.M + 63) / 64) * cols32;
        const int ncu = device_cus();
        if (ks2 && t64 <= ncu) return launch16s_one<64, 32, 2, 1, 6, EPI, OUT, 2, 0, 2, 2>(a, s);
        if (ks2 && t32 <= ncu && a.nz == 1) return launch16s_one<128, 32, 4, 1, 6, EPI, OUT, 2, 0, 2, 1>(a, s);     // (batched per-clip
                                                    // problems - the attention scores at B = 64 - are better off on 128x64: 21 vs 25 us)
        if (t128 <= 32) return launch16s_one<128, 32, 4, 1, 3, EPI, OUT>(a, s);
        if (t128 <= 100) {     // up to ~16 clips: 2x the workgroups; with loader waves where six stages fit beside the offset tables
            if (ks2 && a.nz == 1 && a.taps <= 7 && !a.A2) return launch16s_one<128, 64, 4, 1, 6, EPI, OUT, 2, 0, 2, 1>(a, s);
            return launch16s_one<128, 64, 4, 1, 3, EPI, OUT>(a, s);
        }
        // one 8-wave workgroup per CU (256 slots): 128x192 unless its last round would be mostly idle
        const long tm = (a.M + 127) / 128;
        auto cost = [&](int bn, double eff) {
            const long t = tm * ((a.N + bn - 1) / bn) * a.nz;
            return std::ceil((double)t / (double)ncu) * bn / eff;
        };
        if (cost(128, 0.82) < cost(192, 1.0)) return launch16s_one<128, 128, 4, 2, 3, EPI, OUT>(a, s);
        return launch16s_one<128, 192, 4, 2, 3, EPI, OUT>(a, s);
    }
}

int gemm16s_vq_parts(int N) { return ((N + 191) / 192) * 2; }      // (column tiles of 192) x (2 wave columns)

// Contract: a.A = S32 activations (same strides as the fp32 array),
// a.W_hi = S32 weights [N][K]; out_s32 selects an S32 C (c_rstride in fp32 elements either way).
int launch_gemm16s(const GemmArgs& a_in, int epi, int out, hipStream_t s) {
    const bool out_s32 = out != OUT_F32;        // some S32 array is written: whole 32-column groups
    const GemmArgs& c = a_in;
    if (c.M <= 0 || c.N <= 0 || c.K <= 0 || c.K % SBK || c.Cin % SBK || c.K != c.taps * c.Cin || c.T_out <= 0 ||
        c.M % c.T_out || c.taps > 32 || !c.W_hi || !c.A || (c.w_rstride % 32) || (c.zW % 32) || (c.a_rstride % 32) ||
        (c.a_bstride % 32) || (c.zA % 32) || ((c.N % 4) && !((epi == EPI_SCALE || epi == EPI_BIAS_ROW) && c.c_rstride >= ((c.N + 3) & ~3))) || (c.c_rstride % 4) || (c.zC % 4) || (c.r_rstride % 4) || (out_s32 && ((c.c_rstride % 32) || (c.zC % 32)))) {
        set_error("gemm16s: unsupported problem (S32 operands need every extent and stride in multiples of 32)");
        return -1;
    }
    if ((reinterpret_cast<uintptr_t>(c.A) & 127) || (reinterpret_cast<uintptr_t>(c.W_hi) & 127)) {
        set_error("gemm16s: S32 operands must be 128-byte aligned"); return -1;
    }
    {
        const long clips_per_tile = 256 / c.T_out + 2;      // BM <= 256
        if ((clips_per_tile * c.a_bstride + (long)c.T_in * c.a_rstride) * 4 >= 0x40000000L ||
            (long)c.N * c.w_rstride * 4 >= 0x40000000L) {
            set_error("gemm16s: operand window exceeds the 1 GiB buffer-offset range"); return -1;
        }
    }
    if (c.pad_mode == PAD_REFLECT && c.Tp < c.T_in) { set_error("gemm16s: reflect Tp < T_in"); return -1; }
    if (c.A2) {
        const long clips_per_tile = 256 / c.T_out + 2;
        if (c.taps != 1 || c.stride != 1 || c.pad_left != 0 || c.nz != 1 || c.T_in != c.T_out || c.K1 <= 0 || c.K1 >= c.K || (c.K1 % SBK) ||
            (c.a2_rstride % 32) || (c.a2_bstride % 32) || (reinterpret_cast<uintptr_t>(c.A2) & 127) ||
            (clips_per_tile * c.a2_bstride + (long)c.T_in * c.a2_rstride) * 4 >= 0x40000000L) {
            set_error("gemm16s: a second K source needs a plain row-major problem (taps 1, stride 1, no padding) and K1 in multiples of 32");
            return -1;
        }
    }
    GemmArgs a = a_in;
    // Tile order.  An XCD runs 30 tiles at a time (240 persistent workgroups / 8); a round pulls the A panels of its row tiles
    // and the W panels of its column tiles through that XCD's L2 once.  Wide outputs (ConvNeXt pwconv1: 60 x 12 tiles) walk
    // blocks of 5 row x 6 column tiles = one XCD round each: 132 MB read per launch instead of 156 with groups of 8 row
    // tiles over all 12 columns, at the same launch time (profiles/r03_gmn_traffic.txt; DESIGN section 3 has the 127 MB floor)
    const int tiles_n192 = (a.N + 191) / 192;
    a.group_m = tiles_n192 > 8 ? 8 : 1;
    if (tiles_n192 >= 12 && tiles_n192 % 6 == 0 && a.nz == 1) { a.group_m = 5; a.group_n = 6; }
    if (const char* e = lab_env("WT_GEMM16S_GM")) { a.group_m = atoi(e) > 0 ? atoi(e) : a.group_m; a.group_n = 0; }      // sweeps (tools/gemm16s_bench.py)
    if (const char* e = lab_env("WT_GEMM16S_GN")) a.group_n = atoi(e);
    if ((out == OUT_S32_DUAL_ELU || out == OUT_F32_AND_S32) && !c.C2) { set_error("gemm16s: this output format needs C2"); return -1; }
    if (epi == EPI_HEAD && (!c.bias || c.N % 32 || c.head_kb <= 0)) { set_error("gemm16s: head epilogue needs a bias, N % 32 == 0 and head_kb"); return -1; }
    if (epi == EPI_ARGMAX && (!c.vq_xx || !c.vq_ee || !c.vq_pval || !c.vq_pidx || c.vq_nparts != gemm16s_vq_parts(c.N))) {
        set_error("gemm16s: argmax epilogue needs xx, ee and (value, index) slots for gemm16s_vq_parts(N) parts"); return -1;
    }
#define WT_CASE16S(E, O) if (epi == E && out == O) return launch16s_tiled<E, O>(a, s);
    WT_CASE16S(EPI_BIAS, OUT_F32)
    WT_CASE16S(EPI_BIAS, OUT_S32)
    WT_CASE16S(EPI_BIAS, OUT_S32_DUAL_ELU)
    WT_CASE16S(EPI_BIAS, OUT_F32_AND_S32)
    WT_CASE16S(EPI_BIAS_RES, OUT_F32)
    WT_CASE16S(EPI_BIAS_ELU, OUT_S32)
    WT_CASE16S(EPI_BIAS_RES_ELU, OUT_S32)
    WT_CASE16S(EPI_BIAS_GELU, OUT_S32)
    WT_CASE16S(EPI_BIAS_GAMMA_RES, OUT_F32)
    WT_CASE16S(EPI_HEAD, OUT_S32)
    WT_CASE16S(EPI_ARGMAX, OUT_F32)
    WT_CASE16S(EPI_SCALE, OUT_F32)
    WT_CASE16S(EPI_BIAS_ROW, OUT_S32)
#undef WT_CASE16S
    set_error("gemm16s: unsupported epilogue / output-format pair");
    return -1;
}
#endif      // WT_GEMM16S_LAB

}  // namespace wt
