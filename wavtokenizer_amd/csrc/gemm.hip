// fp32 MFMA implicit-GEMM for every dense contraction of the WavTokenizer path (gfx950).
//
//   C[m][n] = epi( sum_k pro(Agather[m][k]) * W[n][k] )
//
// * time-major activations make an im2col row a run of `taps` contiguous Cin-vectors, so a
//   strided / dilated Conv1d (encoder/modules/conv.py:195-211, reflect padded) and a zero
//   padded Conv1d (decoder/models.py:29-43,177) are this kernel with no im2col buffer;
// * 128xBN block tile, 4 waves (64 lanes each), v_mfma_f32_32x32x2_f32: exact fp32 FMA chains
//   at the fp32 matrix rate (157 TFLOP/s peak);
// * both operands staged global -> registers -> LDS as [row][k] with a 36-float pitch:
//   each lane fetches its four k-steps with one conflict-free ds_read_b128 (k order inside a
//   step group is permuted identically for A and B, which a sum over k does not see);
// * register-prefetched double buffering, one barrier per 32-deep K step;
// * workgroup id -> tile remap keeps tiles that share A rows on one XCD's L2.
#include "common.h"
#include <stdlib.h>
#include <cmath>

namespace wt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int BK = 32;
static constexpr int LDS_PITCH = 36;   // floats; 144 B rows keep 16-B alignment, spread banks

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    // bijective for any nwg (cdna guide T1): blocks orig, orig+8, ... share an XCD
    int q = nwg >> 3, r = nwg & 7;
    int xcd = orig & 7, idx = orig >> 3;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// elu: exp(x) - 1 as ATen's CPU kernel evaluates it; __expf keeps the absolute error ~1e-7
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : __expf(x) - 1.f; }
__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.f + erff(x * 0.70710678118654752440f)); }

template <int BM, int BN, int WAVES_M, int WAVES_N, int PRO, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NA = BM / 32, NB = BN / 32;     // float4 staging loads per thread
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(TM >= 1 && TN >= 1, "wave tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                               // [2][BM][PITCH]
    float* Bs = smem + 2 * BM * LDS_PITCH;          // [2][BN][PITCH]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    // grouped order: 8 row-tiles x all column-tiles at a time, column-major inside the group, so the
    // ~64 tiles an XCD runs together share 8 A panels and 8 W panels (its 4 MiB L2 holds them)
    const int GM = p.group_m;
    const int per_group = GM * tiles_n;
    const int grp = tile / per_group;
    const int first_m = grp * GM;
    const int gsz = tiles_m - first_m < GM ? tiles_m - first_m : GM;
    const int in_grp = tile - grp * per_group;
    const int bm = first_m + in_grp % gsz, bn = in_grp / gsz;
    const int z = blockIdx.z;

    const float* __restrict__ Ag = p.A + (long)z * p.zA;
    const float* __restrict__ Wg = p.W + (long)z * p.zW;

    // ---- operands are read through buffer resources: an out-of-range offset returns 0, so padding,
    //      ragged tile edges and K tails need no branches (cdna guide T8)
    const int nclips = p.M / p.T_out;
    const int m_first = bm * BM < p.M ? bm * BM : p.M - 1;
    const int clip0 = m_first / p.T_out;
    const float* Ablk = Ag + (long)clip0 * p.a_bstride;
    const long a_span = ((long)(nclips - clip0 - 1) * p.a_bstride + (long)p.T_in * p.a_rstride) * 4;
    const long w_span = (long)p.N * p.w_rstride * 4;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Ablk), 0, (int)(a_span < 0x7fffffffL ? a_span : 0x7fffffffL), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Wg), 0, (int)(w_span < 0x7fffffffL ? w_span : 0x7fffffffL), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;     // stays out of range after the (< 2^30) K advance

    // ---- per-(tap, tile row) byte offsets of the gathered A rows, resolved once into LDS:
    //      reflect / zero padding and ragged tile edges become the OOB marker, the K loop is branch-free
    unsigned* s_rowoff = reinterpret_cast<unsigned*>(smem + 2 * (BM + BN) * LDS_PITCH);   // [taps][BM]
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

    // ---- per-thread staging rows (fixed across the K loop)
    const int srow = tid >> 3;        // 0..31
    const int kq4 = (tid & 7) * 4;    // k offset of this thread's float4 inside a K step
    unsigned w_off[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = bn * BN + srow + 32 * i;
        w_off[i] = n < p.N ? (unsigned)((long)n * p.w_rstride * 4) + (unsigned)kq4 * 4u : OOB;
    }
    __syncthreads();
    const int nk_ = (p.K + BK - 1) / BK;

    // ---- 3-stage software pipeline over K: while the MFMAs of tile kt run from one LDS buffer, tile
    //      kt+1 (already in registers, loaded one iteration earlier) is written to the other LDS buffer
    //      in the MIDDLE of the MFMA block, and the global loads of tile kt+2 are in flight.  Two named
    //      register sets alternate (the loop is unrolled by two so every index is static).
    struct Stage { f32x4 a[NA]; f32x4 b[NB]; };
    Stage st0, st1;
    unsigned a_off[NA];               // row offsets of the tap being loaded (registers; refreshed on a tap change)
    auto set_tap = [&](int tap) {
#pragma unroll
        for (int i = 0; i < NA; ++i) a_off[i] = s_rowoff[tap * BM + srow + 32 * i];
    };
    int tapL = 0, ciL = 0, kL = 0;    // coordinates of the next tile to load; taps == 1 or Cin % 32 == 0 (host)
    auto load_tile = [&](Stage& st) {
        // K tail (only K = 16 has one): the OR keeps the offset >= 2^31 whatever is added
        const unsigned kmask = (kL + kq4 < p.K) ? 0u : OOB;
        const unsigned kadv = (unsigned)(ciL + kq4) * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const unsigned ro = a_off[i];
            const unsigned off = (ro | kmask) + kadv;
            st.a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)off, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const unsigned off = (w_off[i] | kmask) + (unsigned)kL * 4u;
            st.b[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)off, 0, 0));
        }
        kL += BK; ciL += BK;
        if (p.taps > 1 && ciL >= p.Cin && kL < p.K) { ciL = 0; ++tapL; set_tap(tapL); }
    };
    auto store_tile = [&](const Stage& st, int buf) {
        float* as = As + buf * BM * LDS_PITCH;
        float* bs = Bs + buf * BN * LDS_PITCH;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            f32x4 v = st.a[i];
            if (PRO == PRO_ELU) {                                  // elu(0) = 0: padding stays 0
                v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w);
            }
            *reinterpret_cast<f32x4*>(as + (srow + 32 * i) * LDS_PITCH + kq4) = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            *reinterpret_cast<f32x4*>(bs + (srow + 32 * i) * LDS_PITCH + kq4) = st.b[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_off = (lane & 31) * LDS_PITCH + 4 * (lane >> 5);
    auto mfma_q = [&](int buf, int q) {
        const float* as = As + buf * BM * LDS_PITCH + (wm * WM) * LDS_PITCH + frag_off;
        const float* bs = Bs + buf * BN * LDS_PITCH + (wn * WN) * LDS_PITCH + frag_off;
        f32x4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDS_PITCH + q * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LDS_PITCH + q * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    };
    // one K step: tile kt is in LDS buffer `buf`, tile kt+1 in `nxt`, tile kt+2 is fetched into `far`
    // Loads and LDS writes are unconditional: past the end of K the offsets are out of range (zeros come
    // back, nothing is fetched) and the written buffer is never read.  A conditional load would make hipcc
    // wait for vmcnt(0) at the LDS write (it merges the two paths' counters), i.e. for the loads just issued.
    auto k_step = [&](int buf, Stage& nxt, Stage& far) {
        load_tile(far);
        mfma_q(buf, 0);
        mfma_q(buf, 1);
        store_tile(nxt, buf ^ 1);
        mfma_q(buf, 2);
        mfma_q(buf, 3);
        __syncthreads();
    };

    set_tap(0);
    load_tile(st0);                       // tile 0
    store_tile(st0, 0);
    load_tile(st1);                       // tile 1
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk_; kt += 2) {
        k_step(0, st1, st0);              // even tile: next is in st1, tile kt+2 goes to st0
        k_step(1, st0, st1);
    }
    if (kt < nk_) k_step(0, st1, st0);    // odd tail

    // ------------------------------------------------------------------------- epilogue
    const int col_l = lane & 31;
    const int row_h = 4 * (lane >> 5);
    const int m_w = bm * BM + wm * WM;
    const int n_w = bn * BN + wn * WN;

    if constexpr (EPI == EPI_ARGMAX) {
        // best over this wave's WN columns for each of its WM rows
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
                        // core_vq.py:177-181: -(|x|^2 - 2 x.e + |e|^2), same association
                        const float d = -((xx - 2.f * acc[i][j][r]) + p.vq_ee[n]);
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

    float* __restrict__ Cg = p.C + (long)z * p.zC;
    if constexpr (EPI == EPI_HEAD) {
        // packed rows: 32-row groups = 16 log-magnitude rows, then the 16 phase rows of the same slots (weights.cpp).  A lane
        // holds one column of the 32-column tile, so the phase of lane l < 16 sits in lane l + 16 (same register)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int pc = n_w + j * 32 + col_l;        // packed column
                const bool is_mag = (col_l & 16) == 0;
                const bool in = pc < p.N;                   // N % 32 == 0: the partner column is in range with it
                const float bme = in ? p.bias[pc] : 0.f;
                const int f = (pc >> 5) * 16 + (col_l & 15);   // spectrum slot
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m_w + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
                    const float mine = acc[i][j][r] + bme;
                    const float ph = __shfl_xor(mine, 16, 64);
                    if (is_mag && in && m < p.M) {
                        float mag = expf(mine);                         // heads.py:55
                        mag = fminf(mag, 100.f);                         // heads.py:56 clip(max=1e2)
                        Cg[(long)m * p.c_rstride + f] = mag * cosf(ph);  // heads.py:58,65
                        Cg[(long)m * p.c_rstride + p.head_kb + f] = mag * sinf(ph);
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
            float bn_ = 0.f, gm = 1.f;
            if (EPI == EPI_BIAS || EPI == EPI_BIAS_RES || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GAMMA_RES ||
                EPI == EPI_BIAS_RES_ELU)
                bn_ = p.bias ? p.bias[n] : 0.f;
            if (EPI == EPI_BIAS_GAMMA_RES) gm = p.gamma[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_w + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
                if (m >= p.M) continue;
                float v = acc[i][j][r];
                if (EPI == EPI_BIAS) {
                    Cg[(long)m * p.c_rstride + n] = v + bn_;
                } else if (EPI == EPI_BIAS_RES) {
                    Cg[(long)m * p.c_rstride + n] = (v + bn_) + p.R[(long)m * p.r_rstride + n];
                } else if (EPI == EPI_BIAS_RES_ELU) {
                    Cg[(long)m * p.c_rstride + n] = elu1((v + bn_) + p.R[(long)m * p.r_rstride + n]);
                } else if (EPI == EPI_BIAS_GELU) {
                    Cg[(long)m * p.c_rstride + n] = gelu_erf(v + bn_);
                } else if (EPI == EPI_BIAS_GAMMA_RES) {
                    Cg[(long)m * p.c_rstride + n] = p.R[(long)m * p.r_rstride + n] + gm * (v + bn_);
                } else if (EPI == EPI_SCALE) {
                    Cg[(long)m * p.c_rstride + n] = v * p.alpha;
                } else if (EPI == EPI_BIAS_ROW) {
                    Cg[(long)m * p.c_rstride + n] = v + p.bias[m];
                }
            }
        }
}

// ---------------------------------------------------------------------------------- host side
template <int BM, int BN, int WMs, int WNs, int PRO, int EPI>
static int launch_one(const GemmArgs& a, hipStream_t s) {
    static PerDeviceOnce attr_once;
    const size_t smem = 2ull * (BM + BN) * LDS_PITCH * sizeof(float) + (size_t)a.taps * BM * sizeof(unsigned);
    constexpr size_t smem_max = 2ull * (BM + BN) * LDS_PITCH * sizeof(float) + 32ull * BM * sizeof(unsigned);
    auto kern = gemm_kernel<BM, BN, WMs, WNs, PRO, EPI>;
    if (int rc = attr_once.run([&]() -> int {
        WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_max));
        return 0;
    })) return rc;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, 1, a.nz);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, a);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// experiment hook (tools/gemm_bench.py): WT_GEMM_TILE picks another tile for plain bias GEMMs
static int tile_override() {
    const char* e = lab_env("WT_GEMM_TILE");       // LAB builds only
    return e ? atoi(e) : -1;
}

template <int PRO, int EPI>
static int launch_tiled(const GemmArgs& a, hipStream_t s) {
    if constexpr (PRO == PRO_NONE && EPI == EPI_BIAS) {
        switch (tile_override()) {
            case 1: return launch_one<128, 64, 2, 2, PRO, EPI>(a, s);
            case 2: return launch_one<128, 96, 4, 1, PRO, EPI>(a, s);
            case 3: return launch_one<128, 64, 4, 1, PRO, EPI>(a, s);
            case 4: return launch_one<64, 128, 1, 4, PRO, EPI>(a, s);
            case 5: return launch_one<128, 192, 2, 2, PRO, EPI>(a, s);
            default: break;
        }
    }
    if constexpr (EPI == EPI_HEAD || EPI == EPI_ARGMAX) {
        return launch_one<128, 128, 2, 2, PRO, EPI>(a, s);
    } else {
        if (a.N <= 32) return launch_one<128, 32, 4, 1, PRO, EPI>(a, s);
        if (a.N <= 64) return launch_one<128, 64, 2, 2, PRO, EPI>(a, s);
        {
            // 512 workgroup slots (2 per CU).  Cost of a tiling ~ rounds of slots x tile width / tile
            // efficiency; narrower tiles win when 128x128 would leave most of the chip idle in its last
            // round (7680x768: 360 tiles on 512 slots; 7680x512: 240).  Measured: tools/gemm_bench.py.
            const long tm = (a.M + 127) / 128;
            auto cost = [&](int bn, double eff) {
                const long t = tm * ((a.N + bn - 1) / bn) * a.nz;
                return std::ceil((double)t / 512.0) * bn / eff;
            };
            const double c128 = cost(128, 1.0), c96 = cost(96, 0.97), c64 = cost(64, 0.88);
            if (c64 < c96 && c64 < c128) return launch_one<128, 64, 2, 2, PRO, EPI>(a, s);
            if (c96 < c128) return launch_one<128, 96, 4, 1, PRO, EPI>(a, s);
        }
        return launch_one<128, 128, 2, 2, PRO, EPI>(a, s);
    }
}

int gemm_vq_parts(int N) { return ((N + 127) / 128) * 2; }

// row-tiles per scheduling group (see the tile remap in the kernel)
static int pick_group_m(const GemmArgs& a) {
    if (const char* e = lab_env("WT_GEMM_GM")) if (atoi(e) > 0) return atoi(e);      // LAB builds only (tools/gm_sweep.sh)
    // An XCD runs ~64 tiles at once and its 4 MiB L2 only holds what they stream in lockstep, so
    // fabric reads ~ sum over such waves of (distinct A panels + distinct W panels).  With many column
    // tiles an 8 x 8 wave is the minimum; with <= 8 column tiles a full row of tiles already is one
    // (FETCH_SIZE sweep: tools/gm_sweep.sh, profiles/r01_gm_sweep.txt).
    const int bn = a.N <= 32 ? 32 : (a.N <= 64 ? 64 : 96);
    return (a.N + bn - 1) / bn > 8 ? 8 : 1;
}

static int check_args(const GemmArgs& a) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) { set_error("gemm: empty problem"); return -1; }
    if (a.K % 4 != 0 || a.Cin % 4 != 0) { set_error("gemm: K and Cin must be multiples of 4"); return -1; }
    if (a.taps > 1 && a.Cin % BK != 0) { set_error("gemm: multi-tap gather needs Cin % 32 == 0"); return -1; }
    if (a.taps > 32) { set_error("gemm: at most 32 taps"); return -1; }
    {   // 31-bit byte offsets inside one workgroup's window of A and inside W
        const long clips_per_tile = 128 / a.T_out + 2;
        if ((clips_per_tile * a.a_bstride + (long)a.T_in * a.a_rstride) * 4 >= 0x40000000L ||
            (long)a.N * a.w_rstride * 4 >= 0x40000000L) {
            set_error("gemm: operand window exceeds the 1 GiB buffer-offset range"); return -1;
        }
    }
    if (a.K != a.taps * a.Cin) { set_error("gemm: K != taps*Cin"); return -1; }
    if (a.T_out <= 0 || a.M % a.T_out != 0) { set_error("gemm: M must be nclips*T_out"); return -1; }
    if ((a.a_rstride % 4) || (a.a_bstride % 4) || (a.w_rstride % 4) || (a.zA % 4) || (a.zW % 4)) {
        set_error("gemm: operand strides must keep 16-byte alignment"); return -1;
    }
    if ((reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.W) & 15)) {
        set_error("gemm: operand base pointers must be 16-byte aligned"); return -1;
    }
    if (a.pad_mode == PAD_REFLECT && a.Tp < a.T_in) { set_error("gemm: reflect Tp < T_in"); return -1; }
    return 0;
}

int launch_gemm(const GemmArgs& a_in, int pro, int epi, hipStream_t s) {
    if (int rc = check_args(a_in)) return rc;
    GemmArgs a = a_in;
    a.group_m = pick_group_m(a);
#define WT_CASE(P, E) \
    if (pro == P && epi == E) return launch_tiled<P, E>(a, s);
    // the (prologue, epilogue) pairs the plans use
    WT_CASE(PRO_NONE, EPI_BIAS)
    WT_CASE(PRO_ELU, EPI_BIAS)
    WT_CASE(PRO_ELU, EPI_BIAS_RES)
    WT_CASE(PRO_ELU, EPI_BIAS_RES_ELU)
    WT_CASE(PRO_NONE, EPI_BIAS_RES)
    WT_CASE(PRO_NONE, EPI_BIAS_GELU)
    WT_CASE(PRO_NONE, EPI_BIAS_GAMMA_RES)
    WT_CASE(PRO_NONE, EPI_HEAD)
    WT_CASE(PRO_NONE, EPI_ARGMAX)
    WT_CASE(PRO_NONE, EPI_SCALE)
    WT_CASE(PRO_NONE, EPI_BIAS_ROW)
#undef WT_CASE
    set_error("gemm: unsupported prologue/epilogue pair");
    return -1;
}

}  // namespace wt
