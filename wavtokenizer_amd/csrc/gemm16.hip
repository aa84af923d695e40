// fp32-equivalent GEMM on the f16 matrix pipe ("f16x3 split"), opt-in for the decoder's dense layers.
//
// Every fp32 operand x is carried as two f16 numbers: hi = f16(x), lo = f16((x - hi) * 2^11), so
// x = hi + lo * 2^-11 to 2^-22 relative.  A product a*b is evaluated as
//      a_hi*b_hi  +  2^-11 * (a_hi*b_lo + a_lo*b_hi)            (the lo*lo term is below 2^-22)
// with three v_mfma_f32_32x32x16_f16 per 16-deep k step into two fp32 accumulators (main, correction).
// The result differs from an fp32 FMA chain by less than the chain's own rounding (measured on
// K = 2304 heavy-tailed data: relative error 3.64e-7 vs 3.61e-7 for fp32), while the f16 pipe runs 16x
// the fp32 MFMA rate, i.e. 5.3x per fp32-equivalent product.  Scaling lo by 2^11 keeps it a normal f16
// number whenever hi is (no flush/underflow of the correction).
//
// Weights are split once at load time ([N][K] f16 hi and lo arrays); activations stay fp32 in HBM and
// are split while they are staged into LDS.  Loader, tile map and 3-stage pipeline are those of gemm.hip.
#include "common.h"

#include <cmath>
#include <stdlib.h>

namespace wt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

static constexpr int BK16 = 32;
static constexpr int PITCH16 = 40;       // halves per LDS row: 80-byte rows, conflict-free ds_read_b128

__device__ __forceinline__ int xcd_remap16(int orig, int nwg) {
    int q = nwg >> 3, r = nwg & 7;
    int xcd = orig & 7, idx = orig >> 3;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
__device__ __forceinline__ float elu16(float x) { return x > 0.f ? x : __expf(x) - 1.f; }
__device__ __forceinline__ float gelu_erf16(float x) { return x * 0.5f * (1.f + erff(x * 0.70710678118654752440f)); }

template <int BM, int BN, int WAVES_M, int WAVES_N, int PRO, int EPI>
__global__ __launch_bounds__(256) void gemm16_kernel(const GemmArgs p) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NA = BM / 32;                   // fp32 float4 staging loads per thread (A)
    constexpr int NBc = BN * 4 * 2 / 256;         // 16-byte chunks per thread (W hi + lo): BN rows x 4 chunks x 2 arrays
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) _Float16 smem16[];
    // per buffer: Ahi[BM][P], Alo[BM][P], Bhi[BN][P], Blo[BN][P]
    constexpr int BUF = (2 * BM + 2 * BN) * PITCH16;
    _Float16* lds = smem16;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int tile = xcd_remap16(blockIdx.x, tiles_m * tiles_n);
    const int GM = p.group_m;
    const int per_group = GM * tiles_n;
    const int grp = tile / per_group;
    const int first_m = grp * GM;
    const int gsz = tiles_m - first_m < GM ? tiles_m - first_m : GM;
    const int in_grp = tile - grp * per_group;
    const int bm = first_m + in_grp % gsz, bn = in_grp / gsz;
    const int z = blockIdx.z;

    const float* __restrict__ Ag = p.A + (long)z * p.zA;
    // one allocation holds the hi array(s) followed by the lo array(s): one buffer descriptor serves both
    const _Float16* __restrict__ Whi = reinterpret_cast<const _Float16*>(p.W_hi) + (long)z * p.zW;

    const int nclips = p.M / p.T_out;
    const int m_first = bm * BM < p.M ? bm * BM : p.M - 1;
    const int clip0 = m_first / p.T_out;
    const float* Ablk = Ag + (long)clip0 * p.a_bstride;
    const long a_span = ((long)(nclips - clip0 - 1) * p.a_bstride + (long)p.T_in * p.a_rstride) * 4;
    const long w_span = (p.w_lo_off + (long)p.N * p.w_rstride) * 2;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Ablk), 0, (int)(a_span < 0x7fffffffL ? a_span : 0x7fffffffL), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(Whi), 0, (int)(w_span < 0x7fffffffL ? w_span : 0x7fffffffL), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    unsigned* s_rowoff = reinterpret_cast<unsigned*>(smem16 + 2 * BUF);   // [taps][BM]
    for (int e = tid; e < p.taps * BM; e += 256) {
        const int tp = e / BM, r = e - tp * BM;
        const int m = bm * BM + r;
        unsigned off = OOB;
        if (m < p.M) {
            const int b = m / p.T_out;
            const int t = m - b * p.T_out;
            int pos = t * p.stride - p.pad_left + tp * p.dil;
            bool ok;
            if (p.pad_mode == PAD_REFLECT) {
                pos = pos < 0 ? -pos : pos;
                pos = pos >= p.Tp ? 2 * (p.Tp - 1) - pos : pos;
                ok = pos < p.T_in;
            } else {
                ok = (pos >= 0) && (pos < p.T_in);
            }
            if (ok) off = (unsigned)(((long)(b - clip0) * p.a_bstride + (long)pos * p.a_rstride) * 4);
        }
        s_rowoff[e] = off;
    }
    // A staging: thread -> (row srow + 32 i, 4 fp32 at k = kq4); W staging: chunk c = tid + 256 j ->
    // (array = c / (BN*4), row = (c / 4) % BN, 8 halves at k = (c % 4) * 8)
    const int srow = tid >> 3, kq4 = (tid & 7) * 4;
    unsigned w_off[NBc];
    int w_lds[NBc];
#pragma unroll
    for (int j = 0; j < NBc; ++j) {
        const int c = tid + 256 * j;
        const int arr = c / (BN * 4), row = (c >> 2) % BN, kc = (c & 3) * 8;
        const int n = bn * BN + row;
        w_off[j] = n < p.N ? (unsigned)(((long)arr * p.w_lo_off + (long)n * p.w_rstride + kc) * 2) : OOB;
        w_lds[j] = (2 * BM + arr * BN + row) * PITCH16 + kc;      // Bhi region follows Ahi, Alo
    }
    __syncthreads();
    const int nk_ = p.K / BK16;

    struct Stage { f32x4 a[NA]; i32x4 b[NBc]; };
    Stage st0, st1;
    unsigned a_off[NA];
    auto set_tap = [&](int tap) {
#pragma unroll
        for (int i = 0; i < NA; ++i) a_off[i] = s_rowoff[tap * BM + srow + 32 * i];
    };
    int tapL = 0, ciL = 0, kL = 0;
    auto load_tile = [&](Stage& st) {
        const unsigned kmask = (kL < p.K) ? 0u : OOB;
        const unsigned kadv = (unsigned)(ciL + kq4) * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            st.a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)((a_off[i] | kmask) + kadv), 0, 0));
#pragma unroll
        for (int j = 0; j < NBc; ++j) {
            st.b[j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)((w_off[j] | kmask) + (unsigned)kL * 2u), 0, 0);
        }
        kL += BK16; ciL += BK16;
        if (p.taps > 1 && ciL >= p.Cin && kL < p.K) { ciL = 0; ++tapL; set_tap(tapL); }
    };
    auto store_tile = [&](const Stage& st, int buf) {
        _Float16* base = lds + buf * BUF;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            f32x4 v = st.a[i];
            if (PRO == PRO_ELU) { v.x = elu16(v.x); v.y = elu16(v.y); v.z = elu16(v.z); v.w = elu16(v.w); }   // elu(0) = 0
            f16x4 hi, lo;
            hi.x = (_Float16)v.x; hi.y = (_Float16)v.y; hi.z = (_Float16)v.z; hi.w = (_Float16)v.w;
            lo.x = (_Float16)((v.x - (float)hi.x) * 2048.f);
            lo.y = (_Float16)((v.y - (float)hi.y) * 2048.f);
            lo.z = (_Float16)((v.z - (float)hi.z) * 2048.f);
            lo.w = (_Float16)((v.w - (float)hi.w) * 2048.f);
            const int r = srow + 32 * i;
            *reinterpret_cast<f16x4*>(base + r * PITCH16 + kq4) = hi;
            *reinterpret_cast<f16x4*>(base + (BM + r) * PITCH16 + kq4) = lo;
        }
#pragma unroll
        for (int j = 0; j < NBc; ++j) *reinterpret_cast<i32x4*>(base + w_lds[j]) = st.b[j];
    };

    f32x16 accm[TM][TN], accc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.f; accc[i][j][r] = 0.f; }

    // fragment: lane (row = lane & 31, half h = lane >> 5) holds k = 16*step + 8h .. +7 of its row
    const int frag = (lane & 31) * PITCH16 + 8 * (lane >> 5);
    auto mfma_step = [&](int buf, int step) {
        const _Float16* base = lds + buf * BUF + frag + step * 16;
        f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            ah[i] = *reinterpret_cast<const f16x8*>(base + (wm * WM + i * 32) * PITCH16);
            al[i] = *reinterpret_cast<const f16x8*>(base + (BM + wm * WM + i * 32) * PITCH16);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bh[j] = *reinterpret_cast<const f16x8*>(base + (2 * BM + wn * WN + j * 32) * PITCH16);
            bl[j] = *reinterpret_cast<const f16x8*>(base + (2 * BM + BN + wn * WN + j * 32) * PITCH16);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], accm[i][j], 0, 0, 0);
                accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accc[i][j], 0, 0, 0);
                accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accc[i][j], 0, 0, 0);
            }
    };
    auto k_step = [&](int buf, Stage& nxt, Stage& far) {
        load_tile(far);
        mfma_step(buf, 0);
        store_tile(nxt, buf ^ 1);
        mfma_step(buf, 1);
        __syncthreads();
    };

    set_tap(0);
    load_tile(st0);
    store_tile(st0, 0);
    load_tile(st1);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk_; kt += 2) {
        k_step(0, st1, st0);
        k_step(1, st0, st1);
    }
    if (kt < nk_) k_step(0, st1, st0);

    // ------------------------------------------------------------------------- epilogue
    const int col_l = lane & 31, row_h = 4 * (lane >> 5);
    const int m_w = bm * BM + wm * WM, n_w = bn * BN + wn * WN;
    float* __restrict__ Cg = p.C + (long)z * p.zC;
    constexpr float LO_SCALE = 1.f / 2048.f;

    if constexpr (EPI == EPI_ARGMAX) {
        // VQ (core_vq.py:176-182): best of this wave's WN columns for each of its WM rows
        const int part = bn * WAVES_N + wn;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_w + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
                const float xx = (m < p.M) ? p.vq_xx[m] : 0.f;
                float best = -INFINITY;
                int bidx = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n_w + j * 32 + col_l;
                    if (n < p.N) {
                        const float dot = accm[i][j][r] + accc[i][j][r] * LO_SCALE;
                        const float d = -((xx - 2.f * dot) + p.vq_ee[n]);
                        if (d > best || (d == best && n < bidx)) { best = d; bidx = n; }
                    }
                }
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) {
                    const float ov = __shfl_xor(best, off, 64);
                    const int oi = __shfl_xor(bidx, off, 64);
                    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
                }
                if (col_l == 0 && m < p.M) {
                    p.vq_pval[(long)m * p.vq_nparts + part] = best;
                    p.vq_pidx[(long)m * p.vq_nparts + part] = bidx;
                }
            }
        }
        return;
    }
    if constexpr (EPI == EPI_HEAD) {
        static_assert(TN % 2 == 0, "head epilogue pairs column tiles");
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j + 1 < TN; j += 2) {
                const int pc = n_w + j * 32 + col_l;
                if (pc < p.N) {
                    const float bmag = p.bias[pc], bph = p.bias[pc + 32];
                    const int f = (pc >> 6) * 32 + col_l;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m_w + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
                        if (m < p.M) {
                            float mag = expf(accm[i][j][r] + accc[i][j][r] * LO_SCALE + bmag);      // heads.py:55
                            mag = fminf(mag, 100.f);                                             // heads.py:56
                            const float ph = accm[i][j + 1][r] + accc[i][j + 1][r] * LO_SCALE + bph;
                            Cg[(long)m * p.c_rstride + f] = mag * cosf(ph);
                            Cg[(long)m * p.c_rstride + p.head_kb + f] = mag * sinf(ph);
                        }
                    }
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n_w + j * 32 + col_l;
            if (n >= p.N) continue;
            const float bn_ = p.bias ? p.bias[n] : 0.f;
            float gm = 1.f;
            if (EPI == EPI_BIAS_GAMMA_RES) gm = p.gamma[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_w + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
                if (m >= p.M) continue;
                const float v = accm[i][j][r] + accc[i][j][r] * LO_SCALE;
                if (EPI == EPI_BIAS) {
                    Cg[(long)m * p.c_rstride + n] = v + bn_;
                } else if (EPI == EPI_BIAS_RES) {
                    Cg[(long)m * p.c_rstride + n] = (v + bn_) + p.R[(long)m * p.r_rstride + n];
                } else if (EPI == EPI_BIAS_RES_ELU) {
                    Cg[(long)m * p.c_rstride + n] = elu16((v + bn_) + p.R[(long)m * p.r_rstride + n]);
                } else if (EPI == EPI_BIAS_GELU) {
                    Cg[(long)m * p.c_rstride + n] = gelu_erf16(v + bn_);
                } else if (EPI == EPI_BIAS_GAMMA_RES) {
                    Cg[(long)m * p.c_rstride + n] = p.R[(long)m * p.r_rstride + n] + gm * (v + bn_);
                }
            }
        }
}

// fp32 -> (hi, lo) f16 pair arrays; used once per weight matrix at model load
__global__ __launch_bounds__(256) void split_f16x2_kernel(const float* __restrict__ w, _Float16* __restrict__ hi,
                                                          _Float16* __restrict__ lo, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = w[i];
        const _Float16 h = (_Float16)v;
        hi[i] = h;
        lo[i] = (_Float16)((v - (float)h) * 2048.f);
    }
}

int launch_split_f16x2(const float* w, void* hi, void* lo, long n, hipStream_t s) {
    int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(split_f16x2_kernel, dim3(blocks), dim3(256), 0, s, w, static_cast<_Float16*>(hi),
                       static_cast<_Float16*>(lo), n);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------- host side
template <int BM, int BN, int WMs, int WNs, int PRO, int EPI>
static int launch16_one(const GemmArgs& a, hipStream_t s) {
    static PerDeviceOnce attr_once;
    const size_t smem = 2ull * (2 * BM + 2 * BN) * PITCH16 * sizeof(_Float16) + (size_t)a.taps * BM * sizeof(unsigned);
    constexpr size_t smem_max = 2ull * (2 * BM + 2 * BN) * PITCH16 * sizeof(_Float16) + 32ull * BM * sizeof(unsigned);
    auto kern = gemm16_kernel<BM, BN, WMs, WNs, PRO, EPI>;
    if (int rc = attr_once.run([&]() -> int {
        WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)smem_max));
        return 0;
    })) return rc;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, 1, a.nz), dim3(256), smem, s, a);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int PRO, int EPI>
static int launch16_tiled(const GemmArgs& a, hipStream_t s) {
    if constexpr (EPI == EPI_HEAD || EPI == EPI_ARGMAX) {
        // paired column tiles / per-wave column slabs: 128x64 with the waves stacked along M (wave tile 32x64)
        return launch16_one<128, 64, 4, 1, PRO, EPI>(a, s);
    } else {
        // 128x96 (wave tile 32x96) or 128x64 by the slot-rounding cost of gemm.hip; the two accumulator sets
        // rule out 128x128 (register budget)
        const long tm = (a.M + 127) / 128;
        auto cost = [&](int bn, double eff) {
            const long t = tm * ((a.N + bn - 1) / bn) * a.nz;
            return std::ceil((double)t / 512.0) * bn / eff;
        };
        if constexpr (PRO == PRO_NONE && EPI == EPI_BIAS) {       // experiment hook (tools/linear_bench.py)
            static int ov = -2;
            if (ov == -2) { const char* e = getenv("WT_GEMM16_TILE"); ov = e ? atoi(e) : -1; }
            if (ov == 128) return launch16_one<128, 128, 2, 2, PRO, EPI>(a, s);
            if (ov == 64) return launch16_one<128, 64, 2, 2, PRO, EPI>(a, s);
            if (ov == 96) return launch16_one<128, 96, 4, 1, PRO, EPI>(a, s);
        }
        if (cost(64, 0.85) < cost(96, 1.0)) return launch16_one<128, 64, 2, 2, PRO, EPI>(a, s);
        return launch16_one<128, 96, 4, 1, PRO, EPI>(a, s);
    }
}

int gemm16_vq_parts(int N) { return (N + 63) / 64; }

// same contract as launch_gemm; needs a.W_hi (f16 [N][K] hi array, lo array w_lo_off halves later) instead of a.W
int launch_gemm16(const GemmArgs& a_in, int pro, int epi, hipStream_t s) {
    const GemmArgs& c = a_in;
    if (c.M <= 0 || c.N <= 0 || c.K <= 0 || c.K % BK16 || c.Cin % 8 || (c.taps > 1 && c.Cin % BK16) || c.K != c.taps * c.Cin ||
        c.T_out <= 0 || c.M % c.T_out || c.taps > 32 || !c.W_hi || c.w_lo_off <= 0 || (c.w_lo_off % 8) || (c.w_rstride % 8) || (c.zW % 8) ||
        (c.a_rstride % 4) || (c.a_bstride % 4) || (c.zA % 4)) {
        set_error("gemm16: unsupported problem (K % 32, strides, split weights)");
        return -1;
    }
    {
        const long clips_per_tile = 128 / c.T_out + 2;
        if ((clips_per_tile * c.a_bstride + (long)c.T_in * c.a_rstride) * 4 >= 0x40000000L ||
            (c.w_lo_off + (long)c.N * c.w_rstride) * 2 >= 0x40000000L) {
            set_error("gemm16: operand window exceeds the 1 GiB buffer-offset range"); return -1;
        }
    }
    GemmArgs a = a_in;
    const int bn = 96;
    a.group_m = (a.N + bn - 1) / bn > 8 ? 8 : 1;
#define WT_CASE16(P, E) if (pro == P && epi == E) return launch16_tiled<P, E>(a, s);
    WT_CASE16(PRO_NONE, EPI_BIAS)
    WT_CASE16(PRO_NONE, EPI_BIAS_RES)
    WT_CASE16(PRO_NONE, EPI_BIAS_GELU)
    WT_CASE16(PRO_NONE, EPI_BIAS_GAMMA_RES)
    WT_CASE16(PRO_NONE, EPI_HEAD)
    WT_CASE16(PRO_NONE, EPI_ARGMAX)
    WT_CASE16(PRO_ELU, EPI_BIAS)
    WT_CASE16(PRO_ELU, EPI_BIAS_RES)
    WT_CASE16(PRO_ELU, EPI_BIAS_RES_ELU)
#undef WT_CASE16
    set_error("gemm16: unsupported prologue/epilogue pair");
    return -1;
}

}  // namespace wt
