// Fused SEANetResnetBlock for the wide-and-shallow encoder stages (C = 32 at 24 kHz, C = 64 at
// 6 kHz), optionally with the first encoder conv folded in.
//
//   y = shortcut(x) + conv1(elu(conv3(elu(x))))          encoder/modules/seanet.py:62-63
//
// Unfused these stages are HBM- and launch-bound GEMMs with K = 16..192 (stage 1 moves 5.3 GB per
// 64-clip step where 0.6 GB is compulsory).  Here a workgroup keeps a 128-frame tile of x in LDS
// (with the +-1 frame halo of the k=3 conv, reflect-resolved at clip edges), each of
// its waves owns 32 frames and runs both contractions on the fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32); conv1 is frame-local, so the hidden activations never leave the
// wave's own LDS rows.  Weights (<= 51 KB) stay resident in LDS; workgroups are persistent over tiles.
// With `wav != nullptr` the x tile is computed from the waveform (SEANetEncoder model[0], k=7,
// seanet.py:107-110), so the 32-channel full-rate tensor is never written.
#include "common.h"

namespace wt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));


// elu(x) = x > 0 ? x : exp(x) - 1 (the form ATen's CPU kernel evaluates); __expf keeps the absolute error at
// ~1e-7, the size of fp32 rounding of the O(1) activations it feeds
__device__ __forceinline__ float rb_elu(float x) { return x > 0.f ? x : __expf(x) - 1.f; }

template <int C, int ROWS>
struct RbLayout {
    static constexpr int H = C / 2;                 // hidden channels
    static constexpr int N1 = H < 32 ? 32 : H;      // conv3 output columns padded to an MFMA tile
    static constexpr int PX = C + 4;                // pitch of x rows
    static constexpr int PH = H + 4;                // pitch of hidden rows
    static constexpr int K1 = 3 * C, P1 = K1 + 4;   // conv3 weights [N1][P1]
    static constexpr int K2 = H + C, P2 = K2 + 4;   // [conv1 | shortcut] weights [C][P2]
    static constexpr int NX = ROWS + 2;
    static constexpr int off_xr = 0;
    static constexpr int off_he = off_xr + NX * PX;
    static constexpr int off_w1 = off_he + ROWS * PH;
    static constexpr int off_w2 = off_w1 + N1 * P1;
    static constexpr int off_b = off_w2 + C * P2;   // b3[N1], b12[C]
    static constexpr int total = off_b + N1 + C;
};

template <int C, int ROWS>
__global__ __launch_bounds__(ROWS * 2) void resblock_kernel(const ResblockArgs a) {
    using L = RbLayout<C, ROWS>;
    constexpr int NT = ROWS * 2;                     // one wave per 32 frames
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xr = smem + L::off_xr;
    float* he = smem + L::off_he;
    float* w1 = smem + L::off_w1;
    float* w2 = smem + L::off_w2;
    float* bb = smem + L::off_b;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- resident weights
    for (int e = tid; e < L::N1 * (L::K1 / 4); e += NT) {
        const int n = e / (L::K1 / 4), k4 = (e - n * (L::K1 / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < L::H) v = *reinterpret_cast<const f32x4*>(a.W3 + (long)n * L::K1 + k4);
        *reinterpret_cast<f32x4*>(w1 + n * L::P1 + k4) = v;
    }
    for (int e = tid; e < C * (L::K2 / 4); e += NT) {
        const int n = e / (L::K2 / 4), k4 = (e - n * (L::K2 / 4)) * 4;
        f32x4 v;
        if (k4 < L::H) v = *reinterpret_cast<const f32x4*>(a.W1 + (long)n * L::H + k4);
        else v = *reinterpret_cast<const f32x4*>(a.Ws + (long)n * C + (k4 - L::H));
        *reinterpret_cast<f32x4*>(w2 + n * L::P2 + k4) = v;
    }
    for (int e = tid; e < L::N1 + C; e += NT)
        bb[e] = e < L::N1 ? (e < L::H ? a.b3[e] : 0.f) : (a.b1[e - L::N1] + a.bs[e - L::N1]);

    const long xbs = a.x_bstride ? a.x_bstride : (long)a.T * C;
    const int tiles_per_clip = (a.T + ROWS - 1) / ROWS;
    const long n_tiles = (long)a.B * tiles_per_clip;
    const int Tp1 = a.T > 1 ? a.T : 2;               // reflect pad 1 (k=3): conv.py:86-91
    const int Tp3 = a.T > 3 ? a.T : 4;               // reflect pad 3 (k=7)

    const int li = lane & 31, lh = lane >> 5;
    const int row0 = wave * 32;                      // this wave's frames inside the tile

    // x tile = frames t0-1 .. t0+ROWS (reflect-resolved).  The global loads of tile i+1 are issued before
    // the contractions of tile i and parked in registers, so their latency hides under the MFMAs.
    constexpr int XITEMS = (L::NX * (C / 4) + NT - 1) / NT;        // float4 per thread per x tile
    constexpr int WAVN = ROWS + 8;                                  // waveform samples per tile (k=7 halo + k=3 halo)
    float* wtile = he;                                              // folded first conv: staged in he (free at this point)
    f32x4 px[XITEMS];
    float pw = 0.f;
    auto prefetch = [&](long tile) {
        const int b = (int)(tile / tiles_per_clip);
        const int t0 = (int)(tile - (long)b * tiles_per_clip) * ROWS;
        if (a.wav) {
            if (tid < WAVN) {                                       // sample index t0 - 4 + tid, k=7 reflect
                int p = t0 - 4 + tid;
                p = p < 0 ? -p : p;
                p = p >= Tp3 ? 2 * (Tp3 - 1) - p : p;
                pw = (p >= 0 && p < a.T) ? a.wav[(long)b * a.T + p] : 0.f;
            }
        } else {
#pragma unroll
            for (int it = 0; it < XITEMS; ++it) {
                const int e = tid + it * NT;
                const int r = e / (C / 4), c4 = (e - r * (C / 4)) * 4;
                int pos = t0 - 1 + r;
                pos = pos < 0 ? -pos : pos;
                pos = pos >= Tp1 ? 2 * (Tp1 - 1) - pos : pos;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (e < L::NX * (C / 4) && pos >= 0 && pos < a.T)
                    v = *reinterpret_cast<const f32x4*>(a.x + (long)b * xbs + (long)pos * C + c4);
                px[it] = v;
            }
        }
    };
    if ((long)blockIdx.x < n_tiles) prefetch(blockIdx.x);

    for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = (int)(tile / tiles_per_clip);
        const int t0 = (int)(tile - (long)b * tiles_per_clip) * ROWS;
        __syncthreads();                             // previous tile fully consumed (and weights landed)
        if (a.wav) {
            // first encoder conv from the staged samples: x[f] = b + sum_j w[j] * wav[refl(f + j - 3)]
            if (tid < WAVN) wtile[tid] = pw;
            __syncthreads();
            for (int e = tid; e < L::NX * (C / 4); e += NT) {
                const int r = e / (C / 4), c4 = (e - r * (C / 4)) * 4;
                int pos = t0 - 1 + r;                // frame of this x row, k=3 reflect
                pos = pos < 0 ? -pos : pos;
                pos = pos >= Tp1 ? 2 * (Tp1 - 1) - pos : pos;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (pos >= 0 && pos < a.T) {
                    v = *reinterpret_cast<const f32x4*>(a.e0_b + c4);
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        // the staged window holds raw positions t0-4 .. t0+ROWS+3; a reflected frame near a clip
                        // edge may need samples outside it, which are re-read from memory (rare)
                        int p = pos + j - 3;
                        p = p < 0 ? -p : p;
                        p = p >= Tp3 ? 2 * (Tp3 - 1) - p : p;
                        float xv = 0.f;
                        if (p >= 0 && p < a.T) {
                            const int wi = p - (t0 - 4);
                            const bool direct = (t0 - 4 >= 0) && (t0 + ROWS + 4 <= a.T);   // window not reflected itself
                            xv = (direct && wi >= 0 && wi < WAVN) ? wtile[wi] : a.wav[(long)b * a.T + p];
                        }
                        v += xv * *reinterpret_cast<const f32x4*>(a.e0_w + j * C + c4);
                    }
                }
                *reinterpret_cast<f32x4*>(xr + r * L::PX + c4) = v;
            }
        } else {
#pragma unroll
            for (int it = 0; it < XITEMS; ++it) {
                const int e = tid + it * NT;
                if (e < L::NX * (C / 4)) {
                    const int r = e / (C / 4), c4 = (e - r * (C / 4)) * 4;
                    *reinterpret_cast<f32x4*>(xr + r * L::PX + c4) = px[it];
                }
            }
        }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) prefetch(tile + gridDim.x);

        // ---- conv3: h[32 frames][N1] = sum over (tap, ci) elu(x)[frame + tap - 1][ci] * W3[n][tap][ci]
        constexpr int TN1 = L::N1 / 32;
        f32x16 acc1[TN1];
#pragma unroll
        for (int j = 0; j < TN1; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
#pragma unroll 4
        for (int q = 0; q < L::K1 / 8; ++q) {
            const int kk = q * 8 + 4 * lh;
            const int tap = kk / C, ci = kk - tap * C;
            f32x4 fa = *reinterpret_cast<const f32x4*>(xr + (row0 + li + tap) * L::PX + ci);
            fa.x = rb_elu(fa.x); fa.y = rb_elu(fa.y); fa.z = rb_elu(fa.z); fa.w = rb_elu(fa.w);
            f32x4 fb[TN1];
#pragma unroll
            for (int j = 0; j < TN1; ++j) fb[j] = *reinterpret_cast<const f32x4*>(w1 + (j * 32 + li) * L::P1 + kk);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < TN1; ++j)
                    acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[j][e], acc1[j], 0, 0, 0);
        }
        // elu(h + b3) into this wave's own rows of he  (C layout: col = lane&31, row = (r&3) + 8(r>>2) + 4(lane>>5))
#pragma unroll
        for (int j = 0; j < TN1; ++j) {
            const int n = j * 32 + li;
            if (n < L::H) {
                const float bn = bb[n];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    he[(row0 + (r & 3) + 8 * (r >> 2) + 4 * lh) * L::PH + n] = rb_elu(acc1[j][r] + bn);
            }
        }
        __syncthreads();

        // ---- y[32 frames][C] = [elu(h) | x] . [W1 | Ws]^T + (b1 + bs)
        constexpr int TN2 = C / 32;
        f32x16 acc2[TN2];
#pragma unroll
        for (int j = 0; j < TN2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
#pragma unroll 4
        for (int q = 0; q < L::K2 / 8; ++q) {
            const int kk = q * 8 + 4 * lh;
            const f32x4 fa = kk < L::H ? *reinterpret_cast<const f32x4*>(he + (row0 + li) * L::PH + kk)
                                       : *reinterpret_cast<const f32x4*>(xr + (row0 + li + 1) * L::PX + (kk - L::H));
            f32x4 fb[TN2];
#pragma unroll
            for (int j = 0; j < TN2; ++j) fb[j] = *reinterpret_cast<const f32x4*>(w2 + (j * 32 + li) * L::P2 + kk);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < TN2; ++j)
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[j][e], acc2[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < TN2; ++j) {
            const int n = j * 32 + li;
            const float bn = bb[L::N1 + n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = t0 + row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (t < a.T) {
                    float v = acc2[j][r] + bn;
                    if (a.elu_out) v = rb_elu(v);
                    a.y[((long)b * a.T + t) * C + n] = v;
                }
            }
        }
    }
}

template <int C, int ROWS>
static int launch_rb(const ResblockArgs& a, hipStream_t s) {
    static PerDeviceOnce attr_once;
    constexpr size_t smem = (size_t)RbLayout<C, ROWS>::total * sizeof(float);
    auto kern = resblock_kernel<C, ROWS>;
    if (int rc = attr_once.run([&]() -> int {
        WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)smem));
        return 0;
    })) return rc;
    const long tiles = (long)a.B * ((a.T + ROWS - 1) / ROWS);
    const int per_cu = (int)(160 * 1024 / smem) < 4 ? (int)(160 * 1024 / smem) : 4;
    const long slots = (long)device_cus() * per_cu;
    const long grid = tiles < slots ? tiles : slots;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(ROWS * 2), smem, s, a);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

bool resblock_fusable(int C) { return C == 32 || C == 64; }

int launch_resblock(const ResblockArgs& a, hipStream_t s) {
    if (a.wav && a.C != 32) { set_error("resblock: the folded first conv needs C == 32"); return -1; }
    if (a.C == 32) return launch_rb<32, 128>(a, s);     // 48 KB LDS: 3 workgroups per CU
    if (a.C == 64) return launch_rb<64, 64>(a, s);      // 78 KB LDS: 2 workgroups (of 2 waves) per CU
    set_error("resblock: fused kernel exists for C = 32 and 64");
    return -1;
}

}  // namespace wt
