// The non-GEMM kernels of the WavTokenizer path (gfx950): all HBM/L2-bound, 64-lane waves,
// 16-byte accesses along the contiguous channel axis of the time-major layout.
#include "common.h"
#include <stdlib.h>

namespace wt {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
typedef _Float16 f16x4s __attribute__((ext_vector_type(4)));
// S32 layout of gemm16s.hip: every 32 fp32 elements of a row become 128 bytes [32 x f16 hi | 32 x f16 lo],
// x = hi + lo * 2^-11; element e of a row -> half index (e >> 5) * 64 + (e & 31), lo 32 halves later
__device__ __forceinline__ void store_s32_1(float* row_base, int e, float v, float& amax) {
    amax = amax1(amax, v);
    _Float16* p = reinterpret_cast<_Float16*>(row_base) + ((e >> 5) * 64 + (e & 31));
    const _Float16 h = (_Float16)v;
    p[0] = h;
    p[32] = (_Float16)((v - (float)h) * 2048.f);
}
__device__ __forceinline__ void store_s32_4(float* row_base, int e, const f32x4 v, float& amax) {   // e % 4 == 0
    amax = amax4(amax, v.x, v.y, v.z, v.w);
    f16x4s hi, lo;
    split4_f16(v.x, v.y, v.z, v.w, hi, lo);
    _Float16* p = reinterpret_cast<_Float16*>(row_base) + ((e >> 5) * 64 + (e & 31));
    *reinterpret_cast<f16x4s*>(p) = hi;
    *reinterpret_cast<f16x4s*>(p + 32) = lo;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int reflect_pos(int pos, int T, int Tp, bool& ok) {
    // encoder/modules/conv.py:79-96: reflect about 0 and Tp-1 (Tp > T only for inputs shorter
    // than the pad, which the reference zero-extends first)
    if (pos < 0) pos = -pos;
    if (pos >= Tp) pos = 2 * (Tp - 1) - pos;
    ok = pos < T;
    return pos;
}

// ------------------------------------------------------------------ first encoder conv (Cin = 1)
// SEANetEncoder model[0]: SConv1d(1, 32, k=7) (encoder/modules/seanet.py:107-110), reflect pad.
// wav [B][T] -> y [B][T][Cout]; one thread = one frame x 4 output channels (16-B store).
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         long BT, int T, int k, int Cout, int Tp) {
    const int c4n = Cout >> 2;
    const long total = BT * c4n;
    const int pl = (k - 1) - (k - 1) / 2;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long m = idx / c4n;
        const int c = (int)(idx - m * c4n) * 4;
        const long b = m / T;
        const int t = (int)(m - b * T);
        f32x4 acc = *reinterpret_cast<const f32x4*>(bias + c);
        const float* x = wav + b * T;
        for (int j = 0; j < k; ++j) {
            bool ok;
            const int pos = reflect_pos(t + j - pl, T, Tp, ok);
            const float xv = ok ? x[pos] : 0.f;
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + j * Cout + c);
            acc += xv * wv;
        }
        *reinterpret_cast<f32x4*>(y + m * Cout + c) = acc;
    }
}

int launch_conv_first(const float* wav, const float* w, const float* bias, float* y, int B, long T, int k, int Cout,
                      hipStream_t s) {
    const long BT = (long)B * T;
    const int pl = (k - 1) - (k - 1) / 2, pr = (k - 1) / 2;
    const int maxpad = pl > pr ? pl : pr;
    const int Tp = T > maxpad ? (int)T : maxpad + 1;
    long total = BT * (Cout / 4);
    int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(conv_first_kernel, dim3(blocks), dim3(256), 0, s, wav, w, bias, y, BT, (int)T, k, Cout, Tp);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ----------------------------------------------------------- last SEANetDecoder conv (Cout = 1)
// SEANetDecoder final SConv1d(32, 1, k=7) (seanet.py:223-226) with ELU on its input (:222).
__global__ __launch_bounds__(256) void conv_last_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        long BT, int T, int Cin, int k, int Tp, int elu_in) {
    // Cin/4 lanes share one output sample (each owns 4 channels: 16-byte loads, a wave instruction reads
    // whole 128-byte frame rows back to back); the lanes' partial sums meet in a shuffle reduction.
    const int lps = Cin >> 2;                       // lanes per sample (8 for the 32-channel last conv)
    const int sps = 256 / lps;                      // samples per workgroup pass
    const int sub = threadIdx.x % lps, sl = threadIdx.x / lps;
    const int pl = (k - 1) - (k - 1) / 2;
    for (long m0 = (long)blockIdx.x * sps; m0 < BT; m0 += (long)gridDim.x * sps) {
        const long m = m0 + sl;
        float acc = 0.f;
        if (m < BT) {
            const long b = m / T;
            const int t = (int)(m - b * T);
            for (int j = 0; j < k; ++j) {
                bool ok;
                const int pos = reflect_pos(t + j - pl, T, Tp, ok);
                if (!ok) continue;
                f32x4 xv = *reinterpret_cast<const f32x4*>(x + (b * T + pos) * Cin + sub * 4);
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + j * Cin + sub * 4);
                if (elu_in) {
                    xv.x = xv.x > 0.f ? xv.x : __expf(xv.x) - 1.f;
                    xv.y = xv.y > 0.f ? xv.y : __expf(xv.y) - 1.f;
                    xv.z = xv.z > 0.f ? xv.z : __expf(xv.z) - 1.f;
                    xv.w = xv.w > 0.f ? xv.w : __expf(xv.w) - 1.f;
                }
                acc += (xv.x * wv.x + xv.y * wv.y) + (xv.z * wv.z + xv.w * wv.w);
            }
        }
        for (int off = lps >> 1; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (sub == 0 && m < BT) y[m] = acc + bias[0];
    }
}

// The shape the model has (32 channels, k = 7): 256 consecutive output samples per workgroup; their 262 input frames
// go through LDS once (coalesced 128-byte rows, reflect / zero padding resolved at the fill), then one thread forms
// one sample from 7 rows x 32 channels (row pitch 36 floats: the 16-byte reads of 16 neighbouring threads cover all
// banks).  Every input byte crosses the vector memory path once instead of seven times.
__global__ __launch_bounds__(256) void conv_last32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          int T, int Tp, int elu_in) {
    constexpr int K = 7, C = 32, PITCH = 36, ROWS = 256 + K - 1;
    __shared__ __attribute__((aligned(16))) float tile[ROWS * PITCH];
    __shared__ __attribute__((aligned(16))) float wl[K * C];
    const int tid = threadIdx.x;
    const long b = blockIdx.y;
    const int t0 = blockIdx.x * 256;
    const int pl = (K - 1) - (K - 1) / 2;
    if (tid < K * C) wl[tid] = w[tid];
    const int rows = T - t0 + K - 1 < ROWS ? T - t0 + K - 1 : ROWS;       // rows that an output t < T reads
    for (int e = tid; e < rows * (C / 4); e += 256) {
        const int r = e >> 3, c4 = e & 7;
        bool ok;
        const int pos = reflect_pos(t0 - pl + r, T, Tp, ok);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok && pos >= 0) {
            v = *reinterpret_cast<const f32x4*>(x + (b * T + pos) * C + c4 * 4);
            if (elu_in) {
                v.x = v.x > 0.f ? v.x : __expf(v.x) - 1.f;
                v.y = v.y > 0.f ? v.y : __expf(v.y) - 1.f;
                v.z = v.z > 0.f ? v.z : __expf(v.z) - 1.f;
                v.w = v.w > 0.f ? v.w : __expf(v.w) - 1.f;
            }
        }
        *reinterpret_cast<f32x4*>(tile + r * PITCH + c4 * 4) = v;
    }
    __syncthreads();
    const int t = t0 + tid;
    if (t >= T) return;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        float aj = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < C / 4; ++c4) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(tile + (tid + j) * PITCH + c4 * 4);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + j * C + c4 * 4);       // same address in every lane: broadcast
            aj += (xv.x * wv.x + xv.y * wv.y) + (xv.z * wv.z + xv.w * wv.w);
        }
        acc += aj;
    }
    y[b * T + t] = acc + bias[0];
}

int launch_conv_last(const float* x, const float* w, const float* bias, float* y, int B, long T, int Cin, int k,
                     int elu_in, hipStream_t s) {
    const long BT = (long)B * T;
    const int pl = (k - 1) - (k - 1) / 2, pr = (k - 1) / 2;
    const int maxpad = pl > pr ? pl : pr;
    const int Tp = T > maxpad ? (int)T : maxpad + 1;
    if (Cin == 32 && k == 7 && B <= 65535) {
        hipLaunchKernelGGL(conv_last32_kernel, dim3((unsigned)((T + 255) / 256), B), dim3(256), 0, s, x, w, bias, y, (int)T, Tp, elu_in);
        WT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (Cin < 4 || Cin > 256 || (Cin & (Cin - 1))) { set_error("conv_last: Cin must be a power of two in [4, 256]"); return -1; }
    const long groups = (BT + (256 / (Cin / 4)) - 1) / (256 / (Cin / 4));
    int blocks = (int)(groups < 16384 ? groups : 16384);
    hipLaunchKernelGGL(conv_last_kernel, dim3(blocks), dim3(256), 0, s, x, w, bias, y, BT, (int)T, Cin, k, Tp, elu_in);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------ transpose
// [B][R][C] -> [B][C][R] through a padded 32x32 LDS tile (coalesced on both sides).
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R,
                                                        int C, int s32, unsigned* status) {
    __shared__ float tile[32][33];
    float amax = 0.f;
    const long boff = (long)blockIdx.z * R * C;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        if (r < R && c < C) tile[i][tx] = in[boff + (long)r * C + c];
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < C) {
            if (s32) store_s32_1(out + boff + (long)c * R, r, tile[tx][i], amax);
            else out[boff + (long)c * R + r] = tile[tx][i];
        }
    }
    range_report(status, amax);
}

int launch_transpose(const float* in, float* out, int B, int R, int C, hipStream_t s, int out_s32) {
    if (out_s32 && (R % 32)) { set_error("transpose: an S32 output needs rows in multiples of 32 elements"); return -1; }
    dim3 grid((C + 31) / 32, (R + 31) / 32, B);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, in, out, R, C, out_s32, g_launch.status);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------- GroupNorm statistics
// decoder/models.py:15-16 Normalize = GroupNorm(32, C, eps=1e-6, affine): per (clip, group) mean and
// biased variance over L x C/32 values, emitted as the per-(clip, channel) scale/shift
//   y = x * (rstd*gamma[c]) + (beta[c] - mean*rstd*gamma[c])
// (consumed by the row-norm pass for pos_net[5]); APPLY > 0 also writes the normalised (and
// swish-activated) tensor once, which the following conv reads as a plain operand.
template <int APPLY>   // 0: scale/shift only; 1: y = x*scale + shift; 2: y = swish(x*scale + shift)
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ scale,
                                                       float* __restrict__ shift, float* __restrict__ y, int L, int C,
                                                       int cg, float eps, int s32, unsigned* status) {
    __shared__ float red[4];
    float amax = 0.f;
    __shared__ float s_mean, s_rstd;
    const int g = blockIdx.x, b = blockIdx.y;
    const float* xb = x + (long)b * L * C + g * cg;
    const int n = L * cg;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float sum = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int t = i / cg, j = i - t * cg;
        sum += xb[(long)t * C + j];
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    if (threadIdx.x == 0) s_mean = (red[0] + red[1] + red[2] + red[3]) / (float)n;
    __syncthreads();
    const float mean = s_mean;
    float sq = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int t = i / cg, j = i - t * cg;
        const float d = xb[(long)t * C + j] - mean;
        sq += d * d;
    }
    sq = wave_sum(sq);
    __syncthreads();
    if (lane == 0) red[wv] = sq;
    __syncthreads();
    if (threadIdx.x == 0) s_rstd = 1.f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)n + eps);
    __syncthreads();
    if (threadIdx.x < cg) {
        const int c = g * cg + threadIdx.x;
        const float sc = s_rstd * gamma[c];
        scale[(long)b * C + c] = sc;
        shift[(long)b * C + c] = beta[c] - mean * sc;
    }
    if (APPLY) {
        float* yb = y + (long)b * L * C + g * cg;
        const float rstd = s_rstd;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int t = i / cg, j = i - t * cg;
            const float sc = rstd * gamma[g * cg + j];
            float v = xb[(long)t * C + j] * sc + (beta[g * cg + j] - mean * sc);
            if (APPLY == 2) v = v / (1.f + expf(-v));
            if (s32) store_s32_1(y + ((long)b * L + t) * C, g * cg + j, v, amax);
            else yb[(long)t * C + j] = v;
        }
        range_report(status, amax);
    }
}

static int launch_gn_chunked(const float*, const float*, const float*, float*, float*, float*, int, int, int, int, int, int, float,
                             hipStream_t, int, float*);
static int gn_slab_groups(int C, int groups);
int launch_gn_stats(const float* x, const float* gamma, const float* beta, float* scale, float* shift, int B, int L,
                    int C, int groups, float eps, hipStream_t s, float* part) {
    const int GBs = gn_slab_groups(C, groups), cgs = C / groups;
    if (part && L > 256 && GBs && (cgs % 4 == 0) && (C % 4 == 0) && GBs <= 8)
        return launch_gn_chunked(x, gamma, beta, scale, shift, nullptr, 0, B, L, C, groups, GBs, eps, s, 0, part);
    hipLaunchKernelGGL(gn_stats_kernel<0>, dim3(groups, B), dim3(256), 0, s, x, gamma, beta, scale, shift,
                       (float*)nullptr, L, C, C / groups, eps, 0, (unsigned*)nullptr);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// GroupNorm apply for short sequences: a block owns GB groups (a 384-byte channel slab for C/32 = 24, GB = 4) of one
// clip, pulls the L x (GB*cg) slab into LDS with full-line loads, takes mean and variance from LDS (two-pass, one
// wave per group), and writes the normalised (swish-activated) slab back once, fp32 or S32: one global read and one
// write per element where gn_stats_kernel makes three strided read passes.
template <int SWISH>
__global__ __launch_bounds__(512) void gn_tile_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ scale,
                                                      float* __restrict__ shift, float* __restrict__ y, int L, int C,
                                                      int cg, int GB, float eps, int s32, unsigned* status) {
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [L][W], W = GB * cg
    float amax = 0.f;
    __shared__ float s_sc[128], s_sh[128], s_red[8];
    const int NT = blockDim.x;                             // 256, or 512 for slabs so large that one workgroup fills the CU
    const int W = GB * cg, W4 = W / 4;
    const int c0 = blockIdx.x * W, b = blockIdx.y;
    const float* xb = x + (long)b * L * C + c0;
    // (row, float4) of element e = threadIdx.x + NT k, advanced without divisions
    const int dt = NT / W4, dq = NT - dt * W4;
    {
        int t = threadIdx.x / W4, q = threadIdx.x - t * W4;
        for (; t < L; t += dt, q += dq) {
            if (q >= W4) { q -= W4; ++t; if (t >= L) break; }
            *reinterpret_cast<f32x4*>(tile + t * W + q * 4) = *reinterpret_cast<const f32x4*>(xb + (long)t * C + q * 4);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = NT >> 6;
    if (nw == 2 * GB) {
        // two waves per group, each over half of the rows; the halves meet in LDS in a fixed order
        const int gl = wv >> 1, part = wv & 1;
        const int r0 = part ? L / 2 : 0, r1 = part ? L : L / 2;
        const int n = L * cg;
        const float* col = tile + gl * cg;
        // a lane reads 4 channels of one row per step: 64 / (cg / 4) rows per wave instruction
        const int cg4 = cg >> 2, rp = 64 / cg4, lr = lane / cg4, lq = lane - lr * cg4;
        float sum = 0.f;
        if (lr < rp)
            for (int t = r0 + lr; t < r1; t += rp) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(col + t * W + lq * 4);
                sum += (v.x + v.y) + (v.z + v.w);
            }
        sum = wave_sum(sum);
        if (lane == 0) s_red[wv] = sum;
        __syncthreads();
        const float mean = (s_red[2 * gl] + s_red[2 * gl + 1]) / (float)n;
        __syncthreads();
        float sq = 0.f;
        if (lr < rp)
            for (int t = r0 + lr; t < r1; t += rp) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(col + t * W + lq * 4);
                const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
                sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        sq = wave_sum(sq);
        if (lane == 0) s_red[wv] = sq;
        __syncthreads();
        const float rstd = 1.f / sqrtf((s_red[2 * gl] + s_red[2 * gl + 1]) / (float)n + eps);
        if (part == 0 && lane < cg) {
            const int c = c0 + gl * cg + lane;
            const float sc = rstd * gamma[c], sh = beta[c] - mean * sc;
            s_sc[gl * cg + lane] = sc; s_sh[gl * cg + lane] = sh;
            scale[(long)b * C + c] = sc; shift[(long)b * C + c] = sh;
        }
    } else
    for (int gl = wv; gl < GB; gl += nw) {                 // one wave per group
        const int n = L * cg;
        const float* col = tile + gl * cg;
        const int cg4 = cg >> 2, rp = 64 / cg4, lr = lane / cg4, lq = lane - lr * cg4;
        float sum = 0.f;
        if (lr < rp)
            for (int t = lr; t < L; t += rp) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(col + t * W + lq * 4);
                sum += (v.x + v.y) + (v.z + v.w);
            }
        const float mean = wave_sum(sum) / (float)n;
        float sq = 0.f;
        if (lr < rp)
            for (int t = lr; t < L; t += rp) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(col + t * W + lq * 4);
                const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
                sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        const float rstd = 1.f / sqrtf(wave_sum(sq) / (float)n + eps);
        if (lane < cg) {
            const int c = c0 + gl * cg + lane;
            const float sc = rstd * gamma[c], sh = beta[c] - mean * sc;
            s_sc[gl * cg + lane] = sc; s_sh[gl * cg + lane] = sh;
            scale[(long)b * C + c] = sc; shift[(long)b * C + c] = sh;
        }
    }
    __syncthreads();
    float* yb = y + (long)b * L * C;
    {
        int t = threadIdx.x / W4, q = threadIdx.x - t * W4;
        for (; t < L; t += dt, q += dq) {
            if (q >= W4) { q -= W4; ++t; if (t >= L) break; }
            const f32x4 v = *reinterpret_cast<const f32x4*>(tile + t * W + q * 4);
            const f32x4 sc = *reinterpret_cast<const f32x4*>(s_sc + q * 4), sh = *reinterpret_cast<const f32x4*>(s_sh + q * 4);
            f32x4 o = v * sc + sh;
            if (SWISH) {     // x * sigmoid(x) on the hardware exp / rcp (relative error ~1e-7)
                o.x *= __builtin_amdgcn_rcpf(1.f + __expf(-o.x)); o.y *= __builtin_amdgcn_rcpf(1.f + __expf(-o.y));
                o.z *= __builtin_amdgcn_rcpf(1.f + __expf(-o.z)); o.w *= __builtin_amdgcn_rcpf(1.f + __expf(-o.w));
            }
            if (s32) store_s32_4(yb + (long)t * C, c0 + q * 4, o, amax);
            else *reinterpret_cast<f32x4*>(yb + (long)t * C + c0 + q * 4) = o;
        }
    }
    range_report(status, amax);
}

// GroupNorm for sequences too long for one LDS slab (30 s clips: L = 1200): the L x 96-channel slab is cut into chunks of
// GN_CH rows.  Pass 1: every (slab, chunk) workgroup pulls its chunk into LDS and leaves, per group, the chunk mean and
// the sum of squared deviations about it.  Pass 2: every workgroup merges the chunk statistics of its groups in chunk
// order (Chan's pairwise update: deterministic, no atomics) and normalises its own chunk straight from global memory.
// Two coalesced reads and one write per element where gn_stats_kernel makes three strided reads.
static constexpr int GN_CH = 128;
__global__ __launch_bounds__(256) void gn_chunk_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int L,
                                                             int C, int cg, int GB, int groups, int nch) {
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [rows][W]
    const int W = GB * cg, W4 = W / 4;
    const int c0 = blockIdx.x * W, k = blockIdx.y, b = blockIdx.z;
    const int t0 = k * GN_CH, rows = L - t0 < GN_CH ? L - t0 : GN_CH;
    const float* xb = x + ((long)b * L + t0) * C + c0;
    for (int e = threadIdx.x; e < rows * W4; e += 256) {
        const int t = e / W4, q = e - t * W4;
        *reinterpret_cast<f32x4*>(tile + t * W + q * 4) = *reinterpret_cast<const f32x4*>(xb + (long)t * C + q * 4);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int gl = wv; gl < GB; gl += 4) {
        const float* col = tile + gl * cg;
        const int cg4 = cg >> 2, rp = 64 / cg4, lr = lane / cg4, lq = lane - lr * cg4;
        float sum = 0.f;
        if (lr < rp)
            for (int t = lr; t < rows; t += rp) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(col + t * W + lq * 4);
                sum += (v.x + v.y) + (v.z + v.w);
            }
        const float mean = wave_sum(sum) / (float)(rows * cg);
        float sq = 0.f;
        if (lr < rp)
            for (int t = lr; t < rows; t += rp) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(col + t * W + lq * 4);
                const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
                sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        sq = wave_sum(sq);
        if (lane == 0) {
            float* o = part + (((long)b * groups + blockIdx.x * GB + gl) * nch + k) * 2;
            o[0] = mean; o[1] = sq;
        }
    }
}

template <int APPLY>   // 0: scale/shift only; 1: y = x*scale + shift; 2: y = swish(x*scale + shift)
__global__ __launch_bounds__(256) void gn_chunk_apply_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ scale, float* __restrict__ shift,
                                                             float* __restrict__ y, int L, int C, int cg, int GB, int groups,
                                                             int nch, float eps, int s32, unsigned* status) {
    __shared__ float s_mean[8], s_rstd[8], s_sc[128], s_sh[128];
    float amax = 0.f;
    const int W = GB * cg, W4 = W / 4;
    const int c0 = blockIdx.x * W, k = blockIdx.y, b = blockIdx.z;
    if (threadIdx.x < GB) {
        const float* pp = part + ((long)b * groups + blockIdx.x * GB + threadIdx.x) * nch * 2;
        float n = 0.f, mean = 0.f, m2 = 0.f;
        for (int q = 0; q < nch; ++q) {
            const int rows = L - q * GN_CH < GN_CH ? L - q * GN_CH : GN_CH;
            const float nq = (float)(rows * cg), d = pp[2 * q] - mean, tot = n + nq;
            mean += d * (nq / tot);
            m2 += pp[2 * q + 1] + d * d * (n * nq / tot);
            n = tot;
        }
        s_mean[threadIdx.x] = mean;
        s_rstd[threadIdx.x] = 1.f / sqrtf(m2 / n + eps);
    }
    __syncthreads();
    if (threadIdx.x < W) {
        const int c = c0 + threadIdx.x, gl = threadIdx.x / cg;
        const float sc = s_rstd[gl] * gamma[c], sh = beta[c] - s_mean[gl] * sc;
        s_sc[threadIdx.x] = sc; s_sh[threadIdx.x] = sh;
        if (k == 0) { scale[(long)b * C + c] = sc; shift[(long)b * C + c] = sh; }
    }
    if (APPLY == 0) return;
    __syncthreads();
    const int t0 = k * GN_CH, rows = L - t0 < GN_CH ? L - t0 : GN_CH;
    const float* xb = x + ((long)b * L + t0) * C + c0;
    float* yb = y + ((long)b * L + t0) * C;
    for (int e = threadIdx.x; e < rows * W4; e += 256) {
        const int t = e / W4, q = e - t * W4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long)t * C + q * 4);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(s_sc + q * 4), sh = *reinterpret_cast<const f32x4*>(s_sh + q * 4);
        f32x4 o = v * sc + sh;
        if (APPLY == 2) {
            o.x *= __builtin_amdgcn_rcpf(1.f + __expf(-o.x)); o.y *= __builtin_amdgcn_rcpf(1.f + __expf(-o.y));
            o.z *= __builtin_amdgcn_rcpf(1.f + __expf(-o.z)); o.w *= __builtin_amdgcn_rcpf(1.f + __expf(-o.w));
        }
        if (s32) store_s32_4(yb + (long)t * C, c0 + q * 4, o, amax);
        else *reinterpret_cast<f32x4*>(yb + (long)t * C + c0 + q * 4) = o;
    }
    range_report(status, amax);
}

size_t gn_part_floats(int B, int L, int groups) { return (size_t)B * groups * ((L + GN_CH - 1) / GN_CH) * 2; }

// the chunked pair of launches; mode as gn_chunk_apply_kernel's APPLY
static int launch_gn_chunked(const float* x, const float* gamma, const float* beta, float* scale, float* shift, float* y,
                             int mode, int B, int L, int C, int groups, int GB, float eps, hipStream_t s, int out_s32,
                             float* part) {
    const int cg = C / groups, nch = (L + GN_CH - 1) / GN_CH;
    const size_t smem = (size_t)GN_CH * GB * cg * sizeof(float);
    static PerDeviceOnce attr_once;
    if (int rc = attr_once.run([&]() -> int { WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gn_chunk_stats_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); return 0; })) return rc;
    hipLaunchKernelGGL(gn_chunk_stats_kernel, dim3(groups / GB, nch, B), dim3(256), smem, s, x, part, L, C, cg, GB, groups, nch);
    const dim3 grid(groups / GB, mode ? nch : 1, B);
    if (mode == 0) hipLaunchKernelGGL(gn_chunk_apply_kernel<0>, grid, dim3(256), 0, s, x, part, gamma, beta, scale, shift, y, L, C, cg, GB, groups, nch, eps, out_s32, g_launch.status);
    else if (mode == 1) hipLaunchKernelGGL(gn_chunk_apply_kernel<1>, grid, dim3(256), 0, s, x, part, gamma, beta, scale, shift, y, L, C, cg, GB, groups, nch, eps, out_s32, g_launch.status);
    else hipLaunchKernelGGL(gn_chunk_apply_kernel<2>, grid, dim3(256), 0, s, x, part, gamma, beta, scale, shift, y, L, C, cg, GB, groups, nch, eps, out_s32, g_launch.status);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}
static int gn_slab_groups(int C, int groups) {      // groups per slab: a multiple of 32 channels, at most 128
    const int cg = C / groups;
    for (int g = 1; g <= groups; ++g)
        if (groups % g == 0 && (g * cg) % 32 == 0 && g * cg <= 128) return g;
    return 0;
}

int launch_gn_apply(const float* x, const float* gamma, const float* beta, float* scale, float* shift, float* y,
                    int swish, int B, int L, int C, int groups, float eps, hipStream_t s, int out_s32, float* part) {
    if (out_s32 && (C % 32)) { set_error("gn_apply: an S32 output needs C % 32 == 0"); return -1; }
    const int cg = C / groups;
    // slab kernel: GB groups = a multiple of 32 channels (whole S32 groups, 16-byte rows), slab within the LDS budget
    int GB = 0;
    for (int g = 1; g <= groups; ++g)
        if (groups % g == 0 && (g * cg) % 32 == 0 && g * cg <= 128) { GB = g; break; }
    if (GB && (cg % 4 == 0) && (size_t)L * GB * cg * 4 <= 96 * 1024 && (C % 4 == 0)) {
        const size_t smem = (size_t)L * GB * cg * sizeof(float);
        static PerDeviceOnce attr_once;
        if (int rc = attr_once.run([&]() -> int {
            WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gn_tile_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gn_tile_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        return 0;
    })) return rc;
        dim3 grid(groups / GB, B);
        // four-group slabs get 8 waves, two per group (r02 did so only for slabs above 48 KB; measured now for the 46 KB slab
        // of 3 s clips: 17.9 -> 15.3 us at B = 64, 14.8 -> 11.0 at B = 1).  The choice depends on the architecture alone, so a
        // clip's statistics are summed in the same order whatever the batch
        const int nt = GB == 4 ? 512 : 256;
        if (swish) hipLaunchKernelGGL(gn_tile_kernel<1>, grid, dim3(nt), smem, s, x, gamma, beta, scale, shift, y, L, C, cg, GB, eps, out_s32, g_launch.status);
        else hipLaunchKernelGGL(gn_tile_kernel<0>, grid, dim3(nt), smem, s, x, gamma, beta, scale, shift, y, L, C, cg, GB, eps, out_s32, g_launch.status);
        WT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (part && GB && (cg % 4 == 0) && (C % 4 == 0) && GB * cg <= 128 && GB <= 8)
        return launch_gn_chunked(x, gamma, beta, scale, shift, y, swish ? 2 : 1, B, L, C, groups, GB, eps, s, out_s32, part);
    if (swish)
        hipLaunchKernelGGL(gn_stats_kernel<2>, dim3(groups, B), dim3(256), 0, s, x, gamma, beta, scale, shift, y, L, C,
                           C / groups, eps, out_s32, g_launch.status);
    else
        hipLaunchKernelGGL(gn_stats_kernel<1>, dim3(groups, B), dim3(256), 0, s, x, gamma, beta, scale, shift, y, L, C,
                           C / groups, eps, out_s32, g_launch.status);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ row LayerNorm (+ dwconv / affine)
// One wave per frame row of C = NV*256 channels:
//   RN_DWCONV   : ConvNeXtBlock dwconv k7 p3 groups=C (decoder/modules.py:28,45) then AdaLayerNorm
//                 (modules.py:81-86): LN(no affine, eps) * scale[id] + shift[id]
//   RN_PLAIN    : final_layer_norm (decoder/models.py:195,234)
//   RN_AFFINE_IN: pos_net[5] GroupNorm apply (models.py:213) then backbone AdaLayerNorm (:228)
template <int NV, int MODE>
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ x, float* __restrict__ y, long M, int L,
                                                      const float* __restrict__ dw_w, const float* __restrict__ dw_b,
                                                      const float* __restrict__ in_scale,
                                                      const float* __restrict__ in_shift,
                                                      const float* __restrict__ out_scale,
                                                      const float* __restrict__ out_shift, float eps, int s32, unsigned* status) {
    constexpr int C = NV * 256;
    const int lane = threadIdx.x & 63;
    float amax = 0.f;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const long b = m / L;
    const int t = (int)(m - b * L);
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (MODE == RN_DWCONV) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(dw_b + c);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int tt = t + j - 3;
                if (tt >= 0 && tt < L) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (b * L + tt) * C + c);
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(dw_w + j * C + c);
                    acc += xv * wv;
                }
            }
            v[i] = acc;
        } else {
            f32x4 xv = *reinterpret_cast<const f32x4*>(x + m * C + c);
            if (MODE == RN_AFFINE_IN) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(in_scale + b * C + c);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(in_shift + b * C + c);
                xv = xv * sc + sh;
            }
            v[i] = xv;
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(sum) * (1.f / C);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const f32x4 d = v[i] - mean;
        sq += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
    const float rstd = 1.f / sqrtf(wave_sum(sq) * (1.f / C) + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        const f32x4 os = *reinterpret_cast<const f32x4*>(out_scale + c);
        const f32x4 oh = *reinterpret_cast<const f32x4*>(out_shift + c);
        const f32x4 o = ((v[i] - mean) * rstd) * os + oh;
        if (s32) store_s32_4(y + m * C, c, o, amax);       // the consumer is the S32 split-f16 GEMM
        else *reinterpret_cast<f32x4*>(y + m * C + c) = o;
    }
    range_report(status, amax);
}

// RN_DWCONV with each wave producing R consecutive frames of one clip: the R + 6 input rows and the 7 tap rows are
// loaded once per 256-channel slice instead of once per output frame (7 row loads + 7 tap loads per frame before).
// Same accumulation order per output as rownorm_kernel, so the results are identical.
template <int NV, int R>
__global__ __launch_bounds__(256) void dwconv_ln_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int L,
                                                        const float* __restrict__ dw_w, const float* __restrict__ dw_b,
                                                        const float* __restrict__ out_scale,
                                                        const float* __restrict__ out_shift, float eps, int s32, unsigned* status) {
    constexpr int C = NV * 256;
    const int lane = threadIdx.x & 63;
    float amax = 0.f;
    const int per_clip = (L + R - 1) / R;
    const long wq = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wq >= (long)B * per_clip) return;
    const int b = (int)(wq / per_clip);
    const int t0 = (int)(wq - (long)b * per_clip) * R;
    const float* xb = x + (long)b * L * C;
    f32x4 v[R][NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        f32x4 xr[R + 6], w[7];
#pragma unroll
        for (int k = 0; k < R + 6; ++k) {
            const int tt = t0 + k - 3;
            xr[k] = (tt >= 0 && tt < L) ? *reinterpret_cast<const f32x4*>(xb + (long)tt * C + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) w[j] = *reinterpret_cast<const f32x4*>(dw_w + j * C + c);
        const f32x4 bias = *reinterpret_cast<const f32x4*>(dw_b + c);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            f32x4 acc = bias;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int tt = t0 + r + j - 3;
                if (tt >= 0 && tt < L) acc += xr[r + j] * w[j];
            }
            v[r][i] = acc;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (t0 + r >= L) break;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) sum += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
        const float mean = wave_sum(sum) * (1.f / C);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f32x4 d = v[r][i] - mean;
            sq += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
        const float rstd = 1.f / sqrtf(wave_sum(sq) * (1.f / C) + eps);
        float* yrow = y + ((long)b * L + t0 + r) * C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            const f32x4 os = *reinterpret_cast<const f32x4*>(out_scale + c);
            const f32x4 oh = *reinterpret_cast<const f32x4*>(out_shift + c);
            const f32x4 o = ((v[r][i] - mean) * rstd) * os + oh;
            if (s32) store_s32_4(yrow, c, o, amax);
            else *reinterpret_cast<f32x4*>(yrow + c) = o;
        }
    }
    range_report(status, amax);
}

template <int NV>
static int launch_rownorm_nv(int mode, const float* x, float* y, long M, int L, const float* dw_w, const float* dw_b,
                             const float* is, const float* ih, const float* os, const float* oh, float eps,
                             hipStream_t s, int s32) {
    dim3 grid((unsigned)((M + 3) / 4));
    if (mode == RN_DWCONV) {
        const int B = (int)(M / L);
        if (M <= 2048) {
            // a few clips: one frame per wave (four frames per wave would be fewer waves than the chip has SIMDs, each walking
            // its rows alone: 9.4 us for 120 frames).  Same accumulation order per output: same bits
            const long waves = (long)B * L;
            hipLaunchKernelGGL((dwconv_ln_kernel<NV, 1>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, y, B, L, dw_w, dw_b,
                               os, oh, eps, s32, g_launch.status);
        } else {
            constexpr int R = 4;
            const long waves = (long)B * ((L + R - 1) / R);
            hipLaunchKernelGGL((dwconv_ln_kernel<NV, R>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, y, B, L, dw_w, dw_b,
                               os, oh, eps, s32, g_launch.status);
        }
    }
    else if (mode == RN_PLAIN)
        hipLaunchKernelGGL((rownorm_kernel<NV, RN_PLAIN>), grid, dim3(256), 0, s, x, y, M, L, dw_w, dw_b, is, ih, os, oh, eps, s32, g_launch.status);
    else
        hipLaunchKernelGGL((rownorm_kernel<NV, RN_AFFINE_IN>), grid, dim3(256), 0, s, x, y, M, L, dw_w, dw_b, is, ih, os, oh, eps, s32, g_launch.status);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_rownorm(int mode, const float* x, float* y, int B, int L, int C, const float* dw_w, const float* dw_b,
                   const float* in_scale, const float* in_shift, const float* out_scale, const float* out_shift,
                   float eps, hipStream_t s, int out_s32) {
    const long M = (long)B * L;
    switch (C) {
        case 256: return launch_rownorm_nv<1>(mode, x, y, M, L, dw_w, dw_b, in_scale, in_shift, out_scale, out_shift, eps, s, out_s32);
        case 512: return launch_rownorm_nv<2>(mode, x, y, M, L, dw_w, dw_b, in_scale, in_shift, out_scale, out_shift, eps, s, out_s32);
        case 768: return launch_rownorm_nv<3>(mode, x, y, M, L, dw_w, dw_b, in_scale, in_shift, out_scale, out_shift, eps, s, out_s32);
        case 1024: return launch_rownorm_nv<4>(mode, x, y, M, L, dw_w, dw_b, in_scale, in_shift, out_scale, out_shift, eps, s, out_s32);
        default: set_error("rownorm: backbone dim must be 256, 512, 768 or 1024"); return -1;
    }
}

// -------------------------------------------------------------------------------------- softmax
// AttnBlock softmax over keys (decoder/models.py:119); one wave per query row; pad columns
// [L, ld) are zero-filled so the P.V contraction can run over the padded length.
__global__ __launch_bounds__(256) void softmax_kernel(float* __restrict__ S, long rows, int L, int ld, float* __restrict__ P_s32) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float* row = S + r * ld;
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) mx = fmaxf(mx, row[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) {
        const float e = expf(row[j] - mx);
        row[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (P_s32) {        // probabilities for a split-f16 GEMM: S32 rows in a separate buffer (pad columns zero)
        float unused = 0.f;                                      // probabilities never leave [0, 1]
        for (int j = lane; j < ld; j += 64) store_s32_1(P_s32 + r * ld, j, j < L ? row[j] / sum : 0.f, unused);
    } else {
        for (int j = lane; j < ld; j += 64) row[j] = j < L ? row[j] / sum : 0.f;
    }
}

// The same with the row held in registers (NV4 float4 per lane, row pitch <= 256 * NV4): the scores are read ONCE with
// 16-byte loads and the probabilities written once (S32: 8 + 8 bytes per four values), instead of read / write-back of the
// exponentials / read / 2-byte stores (30 s clips: 737 MB -> 368 MB per launch).  Same arithmetic per element (max, exp(x - max),
// sum in the same lane order, division by the sum).
template <int NV4>
__global__ __launch_bounds__(256) void softmax_reg_kernel(float* __restrict__ S, long rows, int L, int ld, float* __restrict__ P_s32) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const f32x4* row4 = reinterpret_cast<const f32x4*>(S + r * ld);
    f32x4 v[NV4];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int j = 4 * (lane + 64 * i);
        v[i] = j < ld ? row4[lane + 64 * i] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (j + e >= L) v[i][e] = -INFINITY;
            mx = fmaxf(mx, v[i][e]);
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ex = 4 * (lane + 64 * i) + e < L ? expf(v[i][e] - mx) : 0.f;
            v[i][e] = ex;
            sum += ex;
        }
    sum = wave_sum(sum);
    float unused = 0.f;                                          // probabilities never leave [0, 1]
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int j = 4 * (lane + 64 * i);
        if (j >= ld) continue;
        f32x4 p;
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = j + e < L ? v[i][e] / sum : 0.f;
        if (P_s32) store_s32_4(P_s32 + r * ld, j, p, unused);
        else *reinterpret_cast<f32x4*>(S + r * ld + j) = p;
    }
}

int launch_softmax(float* S, int rows, int L, int ld, hipStream_t s, float* P_s32) {
    if (P_s32 && (ld % 32)) { set_error("softmax: an S32 output needs a row pitch in multiples of 32"); return -1; }
    if (ld % 4 == 0 && ld <= 2048 && !(reinterpret_cast<uintptr_t>(S) & 15)) {
        const dim3 grid((rows + 3) / 4), block(256);
        if (ld <= 256) hipLaunchKernelGGL(softmax_reg_kernel<1>, grid, block, 0, s, S, (long)rows, L, ld, P_s32);
        else if (ld <= 512) hipLaunchKernelGGL(softmax_reg_kernel<2>, grid, block, 0, s, S, (long)rows, L, ld, P_s32);
        else if (ld <= 1280) hipLaunchKernelGGL(softmax_reg_kernel<5>, grid, block, 0, s, S, (long)rows, L, ld, P_s32);
        else hipLaunchKernelGGL(softmax_reg_kernel<8>, grid, block, 0, s, S, (long)rows, L, ld, P_s32);
        WT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, S, (long)rows, L, ld, P_s32);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------- ISTFT tail
// ISTFT.forward (decoder/spectral_ops.py:33-75) after the four quarter transforms
// Ce, Co, Se, So [frame][0..N/4]: rebuild x_t[n] with the two radix-2 butterflies, multiply by the
// window, overlap-add the n_fft/hop frames that cover an output sample (ascending n, like fold),
// trim and divide by the window-square envelope.  One thread per output sample.
// "same" (:46, 56-73): trim (n_fft - hop) / 2 at both ends, L * hop samples.  "center" (:43-45, torch.istft(center=True)):
// trim n_fft / 2 at both ends, (L - 1) * hop samples; the same overlap-add and the same envelope otherwise.
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ parts, const float* __restrict__ win,
                                                        const float* __restrict__ wsq, float* __restrict__ out,
                                                        long total, long Mrows, int L, int N, int hop, int Kq, int pad, long Tout) {
    const int R = N / hop, Q = N / 4, Nh = N / 2;
    const float* Ce = parts;
    const float* Co = parts + Mrows * Kq;
    const float* Se = parts + 2 * Mrows * Kq;
    const float* So = parts + 3 * Mrows * Kq;
    // Workgroup i runs on XCD i % 8 and every spectrum value is read by four output samples up to n_fft apart: with the
    // chunks of 256 samples dealt out round robin each XCD's L2 fetched (nearly) all of `parts` for itself (301 MB of traffic
    // for 93 MB, profiles/r03_pmc_traffic.json).  XCD x takes the x-th contiguous eighth of the chunks instead
    const long nchunk = (long)gridDim.x;                     // a multiple of 8 (host), one chunk per workgroup
    const long chunk = (long)(blockIdx.x & 7) * (nchunk >> 3) + (blockIdx.x >> 3);
    for (long idx = chunk * blockDim.x + threadIdx.x; idx < total; idx += total) {       // (one pass)
        const long b = idx / Tout;
        const long u = idx - b * Tout;
        const long up = u + pad;
        const int jp = (int)(up / hop), r = (int)(up - (long)jp * hop);
        float acc = 0.f, env = 0.f;
        for (int d = 0; d < R; ++d) {
            const int t = jp - d;
            if (t < 0 || t >= L) continue;
            const int n = r + hop * d;
            const int m = n <= Nh ? n : N - n;               // x[N-m] = C[m] + S[m]
            const int mm = m <= Q ? m : Nh - m;              // C[N/2-mm] = Ce - Co, S[N/2-mm] = So - Se
            const long o = (b * L + t) * Kq + mm;
            const float ce = Ce[o], co = Co[o], se = Se[o], so = So[o];
            const float Cv = m <= Q ? ce + co : ce - co;
            const float Sv = m <= Q ? se + so : so - se;
            const float x = n <= Nh ? Cv - Sv : Cv + Sv;
            acc += x * win[n];
            env += wsq[n];
        }
        out[idx] = acc / env;
    }
}

int launch_istft_ola(const float* parts, const float* win, const float* wsq, float* out, int B, int L, int n_fft, int hop,
                     int Kq, int center, hipStream_t s) {
    const int pad = center ? n_fft / 2 : (n_fft - hop) / 2;
    const long Tout = center ? (long)hop * (L - 1) : (long)hop * L;
    const long total = (long)B * Tout;
    if (total <= 0) return 0;
    const long blocks = (((total + 255) / 256) + 7) / 8 * 8;          // one 256-sample chunk per workgroup, a multiple of 8
    if (blocks > 0x7fffffffL) { set_error("istft_ola: too many samples for one launch"); return -1; }
    hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)blocks), dim3(256), 0, s, parts, win, wsq, out, total, (long)B * L, L, n_fft,
                       hop, Kq, pad, Tout);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------- call guard
// The last step of every plan.  status == 0 (the normal case): each workgroup reads one word and leaves.  Otherwise
// (common.h: WT_STATUS_*) the call's outputs are overwritten — codes with -1, float outputs with NaN — so that a
// failed call can never hand out plausible-looking data, and the bits are OR-ed into the plan's host-visible word,
// where the next host call on the plan (or wt_plan_status) finds them.
// The control block holds `nwords` status words: word 0 (plan-level reports) and, from word `site0` on, one word per range
// site (model.h Site: the steps of a site report into the site's word).  Their OR is the call's status; the sites whose word
// carries WT_STATUS_RANGE are published as a 64-bit mask in host_status[2..3], so the host can put exactly those sites on
// fp32 operands instead of the whole model.
__global__ __launch_bounds__(256) void plan_guard_kernel(const unsigned* __restrict__ status, int nwords, int site0, unsigned* host_status,
                                                         unsigned* model_status, int64_t* codes, long n_codes, float* f0, long n0, float* f1, long n1,
                                                         float* f2, long n2) {
    __shared__ unsigned s_or, s_lo, s_hi;
    if (threadIdx.x == 0) { s_or = 0u; s_lo = 0u; s_hi = 0u; }
    __syncthreads();
    if ((int)threadIdx.x < nwords) {
        const unsigned w = status[threadIdx.x];
        if (w) {
            atomicOr(&s_or, w);
            const int site = (int)threadIdx.x - site0;
            if ((w & WT_STATUS_RANGE) && site >= 0 && site < 64) atomicOr(site < 32 ? &s_lo : &s_hi, 1u << (site & 31));
        }
    }
    __syncthreads();
    const unsigned st = s_or;
    if (st == 0u) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (host_status) {
            if (s_lo) __hip_atomic_fetch_or(host_status + 2, s_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (s_hi) __hip_atomic_fetch_or(host_status + 3, s_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_fetch_or(host_status, st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (model_status) __hip_atomic_fetch_or(model_status, st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const long step = (long)gridDim.x * blockDim.x, i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const float qnan = __builtin_nanf("");
    if (codes) for (long i = i0; i < n_codes; i += step) codes[i] = -1;
    if (f0) for (long i = i0; i < n0; i += step) f0[i] = qnan;
    if (f1) for (long i = i0; i < n1; i += step) f1[i] = qnan;
    if (f2) for (long i = i0; i < n2; i += step) f2[i] = qnan;
}

// Fills of plan buffers are kernels, never hipMemsetAsync: replayed from a hipGraph, the memset nodes of ROCm 7.2 were
// seen to leave eight bytes of garbage (a host address) at the start of their destination AFTER later nodes had
// written there (found through the call status word; it also explained codes that differed between direct and
// replayed persistent-LSTM launches).  16-byte stores; p 16-byte aligned, n_bytes a multiple of 16.
__global__ __launch_bounds__(256) void fill_u32_kernel(uint4* __restrict__ p, unsigned v, long n16) {
    const uint4 w = {v, v, v, v};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) p[i] = w;
}
int launch_fill_u32(void* p, unsigned value, size_t n_bytes, hipStream_t s) {
    if ((reinterpret_cast<uintptr_t>(p) & 15) || (n_bytes & 15)) { set_error("fill: buffer must be 16-byte aligned and sized"); return -1; }
    const long n16 = (long)(n_bytes / 16);
    if (n16 == 0) return 0;
    const int blocks = (int)((n16 + 255) / 256 < 2048 ? (n16 + 255) / 256 : 2048);
    hipLaunchKernelGGL(fill_u32_kernel, dim3(blocks), dim3(256), 0, s, static_cast<uint4*>(p), value, n16);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_plan_guard(const unsigned* status, int nwords, int site0, unsigned* host_status, unsigned* model_status, int64_t* codes, long n_codes,
                      float* f0, long n0, float* f1, long n1, float* f2, long n2, hipStream_t s) {
    if (nwords < 1 || nwords > 256) { set_error("plan_guard: control block size"); return -1; }
    hipLaunchKernelGGL(plan_guard_kernel, dim3(256), dim3(256), 0, s, status, nwords, site0, host_status, model_status, codes, n_codes, f0, n0, f1, n1, f2, n2);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// Largest |hi + lo * 2^-11| over an S32 array of `numel` fp32-equivalent elements, as the bit pattern of a non-negative float
// (which orders like an unsigned integer): the range report of a plan (WT_PLAN_FLAG_RANGE_REPORT).  NaNs are skipped, an
// infinite hi half (a value beyond the f16 range) reads as +inf.
__global__ __launch_bounds__(256) void s32_amax_kernel(const _Float16* __restrict__ p, long numel, unsigned* __restrict__ out) {
    float m = 0.f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < numel; e += (long)gridDim.x * blockDim.x) {
        const long g = (e >> 5) * 64 + (e & 31);
        m = fmaxf(m, fabsf((float)p[g] + (float)p[g + 32] * (1.f / 2048.f)));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}
int launch_s32_amax(const void* s32, long numel, unsigned* out_bits, hipStream_t s) {
    if (numel <= 0) return 0;
    const long blocks = (numel + 255) / 256 < 2048 ? (numel + 255) / 256 : 2048;
    hipLaunchKernelGGL(s32_amax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const _Float16*>(s32), numel, out_bits);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ----------------------------------------------------------------------------------------- VQ
__global__ __launch_bounds__(256) void row_sumsq_kernel(const float* __restrict__ x, float* __restrict__ out, long rows,
                                                        int D) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * D + c);
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    s = wave_sum(s);
    if (lane == 0) out[r] = s;
}

int launch_row_sumsq(const float* x, float* out, long rows, int D, hipStream_t s) {
    hipLaunchKernelGGL(row_sumsq_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, out, rows, D);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// Reduce the per-column-slab (value, index) partials in slab order (strict '>' keeps the lowest
// index on ties, like torch.max), write int64 codes and — if feat != null — the dequantised
// embedding in the reference's (B, D, L) layout (core_vq.py:188-190 + rearrange b n d -> b d n).
__global__ __launch_bounds__(256) void vq_finalize_kernel(const float* __restrict__ pval, const int* __restrict__ pidx,
                                                          int nparts, const float* __restrict__ embed,
                                                          int64_t* __restrict__ codes, float* __restrict__ feat, int L,
                                                          int D, int bins) {
    // One workgroup per (clip, 32 frames).  Phase 1: 8 lanes per frame scan the frame's partial candidates (a lane takes
    // parts q = sub, sub + 8, ... in ascending order, the eight lanes are then merged lowest-part-first, so ties keep the
    // lowest index like the serial scan).  Phase 2: the 32 selected codebook rows are copied through LDS (coalesced 16-byte
    // reads of each 2 KB row) and written transposed, 32 consecutive frames per channel.
    __shared__ int s_code[32];
    __shared__ float tile[32][257];                       // 256 channels at a time, padded against bank conflicts
    const int b = blockIdx.y, t0 = blockIdx.x * 32;
    const int fr = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const int t = t0 + fr;
    float best = -INFINITY;
    int bi = 0x7fffffff, bq = 0x7fffffff;
    if (t < L) {
        const long m = (long)b * L + t;
        for (int q = sub; q < nparts; q += 8) {
            const float v = pval[m * nparts + q];
            if (v > best) { best = v; bi = pidx[m * nparts + q]; bq = q; }
        }
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64), oq = __shfl_xor(bq, off, 64);
        // the serial scan keeps the FIRST part that reaches the maximum: prefer the lower part number on equal values
        if (ov > best || (ov == best && oq < bq)) { best = ov; bi = oi; bq = oq; }
    }
    if (sub == 0 && t < L) {
        // a row of NaN distances (NaN audio in) selects nothing in the GEMM epilogue: index 0 then, like torch.max on
        // an all-NaN row, instead of an out-of-range gather below
        if ((unsigned)bi >= (unsigned)bins) bi = 0;
        s_code[fr] = bi;
        codes[(long)b * L + t] = (int64_t)bi;
    }
    if (feat == nullptr) return;
    __syncthreads();
    const int nfr = L - t0 < 32 ? L - t0 : 32;
    for (int c0 = 0; c0 < D; c0 += 256) {
        // 32 rows x 256 channels: thread -> (row = tid / 8, 32 channels as 8 x 16 bytes, all requested before any is used)
        if (fr < nfr) {
            const float* src = embed + (long)s_code[fr] * D + c0 + sub * 4;
            f32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(src + k * 32);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float* d = &tile[fr][sub * 4 + k * 32];
                d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
            }
        }
        __syncthreads();
        // write: thread -> (channel = tid, 32 frames): a wave stores 64 channels x one frame run each
        {
            float* dst = feat + ((long)b * D + c0 + threadIdx.x) * L + t0;
#pragma unroll 8
            for (int k = 0; k < nfr; ++k) dst[k] = tile[k][threadIdx.x];
        }
        __syncthreads();
    }
}

int launch_vq_finalize(const float* pval, const int* pidx, int nparts, const float* embed, int64_t* codes,
                       float* feat_ncl, int B, int L, int D, int bins, hipStream_t s) {
    if (D % 256) { set_error("vq_finalize: codebook width must be a multiple of 256"); return -1; }
    dim3 grid((L + 31) / 32, B);
    hipLaunchKernelGGL(vq_finalize_kernel, grid, dim3(256), 0, s, pval, pidx, nparts, embed, codes, feat_ncl, L, D, bins);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// WavTokenizer.codes_to_features (decoder/pretrained.py:227-237): sum over the K codebooks of
// embed[codes[k] + k*bins], transposed to (B, D, L).  F.embedding (pretrained.py:236) raises on an index outside
// the table; here such a code never reaches the gather (row 0 is read instead), the frame's features become NaN and
// `bad` (optional) is set to 1, which the host side turns into the error.
__global__ __launch_bounds__(256) void codes_to_features_kernel(const int64_t* __restrict__ codes,
                                                                const float* __restrict__ embed, int K, int bins,
                                                                long L, int D, float* __restrict__ feat, int cchunk,
                                                                unsigned* bad) {
    const int b = blockIdx.x, B = gridDim.x;
    const int c0 = blockIdx.y * cchunk;
    const long t0 = (long)blockIdx.z * 1024;
    const long tn = (L - t0) < 1024 ? (L - t0) : 1024;
    const long n = (long)cchunk * tn;
    for (long i = threadIdx.x; i < n; i += 256) {
        const int c = c0 + (int)(i / tn);
        const long t = t0 + i % tn;
        if (c >= D) continue;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) {
            long code = codes[((long)k * B + b) * L + t];
            const bool oob = code < 0 || code >= bins;
            if (oob) { code = 0; if (bad) __hip_atomic_store(bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
            acc += oob ? __builtin_nanf("") : embed[((long)k * bins + code) * D + c];
        }
        feat[((long)b * D + c) * L + t] = acc;
    }
}

int launch_codes_to_features(const int64_t* codes, const float* embed, int K, int bins, int B, long L, int D,
                             float* feat_ncl, hipStream_t s, unsigned* bad) {
    const int cchunk = 64;
    dim3 grid(B, (D + cchunk - 1) / cchunk, (unsigned)((L + 1023) / 1024));
    hipLaunchKernelGGL(codes_to_features_kernel, grid, dim3(256), 0, s, codes, embed, K, bins, L, D, feat_ncl, cchunk, bad);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------- LSTM
// SLSTM (encoder/modules/lstm.py:12-39): nn.LSTM(512, 512, num_layers=2), zero initial state, gate
// order i,f,g,o, plus the skip add.  The recurrence is serial in time, so one launch = one
// time step of BOTH layers, layer 1 running one step behind layer 0 (launch s: layer 0 step s,
// layer 1 step s-1).  A workgroup owns 4 hidden units (their 16 gate rows, packed at load time) for a
// tile of 64 clips; its 16 waves split K, v_mfma_f32_16x16x4_f32 does the recurrent product, LDS adds
// the sixteen K slices, and each thread then updates one (clip, unit) cell.
//
// The step is bound by the address/tag work of re-reading h (every one of the 256 workgroups reads the
// whole hidden state: 48 MB of L2 hits per step), so the state is kept K-MAJOR, h[k][clip] with the clip
// pitch padded to 64: a lane's 16-byte load is 4 consecutive clips of one k, a wave's load instruction is
// 4 full 256-byte rows (8 cache lines, all bytes used; clip-major rows gave 16 half-used lines), and the
// four elements feed four MFMAs whose row r stands for clip 4r + e.  The weights are packed per
// workgroup as [k/16][lane][4] so that a lane's 16-byte load is its B operand for four k-steps.
typedef float f32x4acc __attribute__((ext_vector_type(4)));

// sigmoid and tanh on the hardware exp (absolute error ~1e-7, the size of fp32 rounding of their O(1) results)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    const float ax = fabsf(x);
    const float e = __expf(-2.f * ax);                       // in (0, 1]: no overflow
    const float t = ax < 0.04f ? ax * (1.f - ax * ax * (1.f / 3.f)) : (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);
    return copysignf(t, x);
}

static constexpr int LSTM_WAVES = 16;
static constexpr int LSTM_MAXS = 64 / LSTM_WAVES;     // 16-k groups per wave of the widest (layer-1, K = 1024) slice
typedef _Float16 f16x8l __attribute__((ext_vector_type(8)));

// F16 = false: v_mfma_f32_16x16x4_f32 on the fp32 state (exact fp32 multiply-add chain).
// F16 = true : the recurrent product on the f16 matrix pipe with fp32-equivalent products (x = hi + lo * 2^-11,
//              three v_mfma_f32_16x16x32_f16 per 32-deep block, gemm16s.hip): the state is stored pre-split, K-major,
//              16 bytes per (k, clip quad) = [hi of 4 clips | lo of 4 clips] (same pitch and addresses as the fp32
//              state), so a lane's load is the same 16 bytes; its 8 loads of a 32-deep block are re-packed in
//              registers into the A operands of the four clip groups, and the weights arrive packed per lane as
//              [hi of its 8 k | lo of its 8 k].  The fp32 pipe needs 64 MFMAs x 32 cycles per layer-1 wave and
//              step, this 24 x 16.
template <bool F16>
__global__ __launch_bounds__(64 * LSTM_WAVES) void lstm_step_kernel(const LstmArgs a, int s_in) {
    __shared__ float red[LSTM_WAVES][64][17];
    const int s = s_in & 0xffff;
    constexpr int dbg = 0;                          // (round-1 timing experiments; the switches fold away)
    if (dbg & 1) return;
    const int H = a.H, B = a.B, L = a.L;
    const int Bp = (B + 63) & ~63;                  // clip pitch of the K-major state
    const int nj = H / 4;
    const int layer = blockIdx.x >= nj ? 1 : 0;
    const int bj = blockIdx.x - layer * nj;
    const int t = s - layer;                        // time step this block advances
    if (t < 0 || t >= L) return;                    // uniform per block
    const int b0 = blockIdx.y * 64;
    // the wave index as a scalar: the K-slice bases it selects (weights, state) then live in SGPRs (the kernel is capped at
    // 128 VGPRs by its 1024 threads and spilled 12 with them in vector registers)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lk = lane >> 4;

    const int Ktot = layer ? 2 * H : H;
    const int kw = Ktot / LSTM_WAVES;               // K slice of this wave (16 waves: 32 / 64; 8 waves: 64 / 128)
    const int kbeg = wave * kw;
    // packed weights of this workgroup, 16 bytes per lane and 16 k.  fp32: [Ktot/16][64 lanes][4]: element e of
    // lane (li, lk) in group S is W[gate row bj*16 + li][k = 16 S + 4 e + lk].  f16: [Ktot/32][hi, lo][64 lanes][8]:
    // half p of lane (li, lk) in block P is W[row][k = 32 P + 16 (p >> 2) + 4 (p & 3) + lk]
    const float* W = (layer ? a.W1 : a.W0) + (long)bj * 16 * Ktot + (long)(kbeg / 16) * 256 + lane * 4;
    // source of the K slice: layer 0: h0[t-1]; layer 1: [h0[t] | h1[t-1]]   (each [H][Bp], K-major)
    const float* hsrc;
    int koff;                                       // first k of the slice inside hsrc
    if (!layer) { hsrc = a.h0 + (long)((t + 1) & 1) * H * Bp; koff = kbeg; }
    else if (kbeg < H) { hsrc = a.h0 + (long)(t & 1) * H * Bp; koff = kbeg; }
    else { hsrc = a.h1 + (long)((t + 1) & 1) * H * Bp; koff = kbeg - H; }
    const float* hp = hsrc + (long)(koff + lk) * Bp + b0 + 4 * li;      // k-step q: + 4 q Bp

    // cell-update operands (threads 0..255: one (clip, unit) each) are fetched now, under the MFMA loop
    const int br = (threadIdx.x >> 2) & 63, jj = threadIdx.x & 3;
    const int cb = b0 + br;
    const bool cell = threadIdx.x < 256 && cb < B;
    const int j = bj * 4 + jj;
    float gin[4] = {0.f, 0.f, 0.f, 0.f}, c_prev = 0.f, x_skip = 0.f;
    float* cst = (layer ? a.c1 : a.c0) + (long)(cell ? cb : 0) * H + j;
    if (cell) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            gin[g] = layer ? a.b1[bj * 16 + g * 4 + jj] : a.xg0[((long)t * B + cb) * (4 * H) + bj * 16 + g * 4 + jj];
        c_prev = *cst;
        if (layer) x_skip = a.x[((long)cb * L + t) * H + j];
    }

    // the whole K slice of a wave is one batch of loads (one L2 round trip), then its MFMAs; 16 waves
    // (4 per SIMD) keep the matrix pipe fed while other waves' loads are in flight
    f32x4acc acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = (f32x4acc){0.f, 0.f, 0.f, 0.f};
    f32x4 wv[LSTM_MAXS], hv[4 * LSTM_MAXS];
    const int nS = kw / 16;                         // 16-k groups: 2 or 4; k-steps: 4 per group
    // groups [S0, S1) of the slice into registers.  fp32: everything at once (one L2 round trip).  f16: one 32-k block per
    // batch: the second accumulator set of the split-f16 form leaves no room for 80 operand registers under the 128-VGPR cap
    // of a 1024-thread workgroup (the whole-slice batch spilled 12 registers)
    auto load_groups = [&](int S0, int S1) {
#pragma unroll
        for (int S = 0; S < LSTM_MAXS; ++S) {
            if (S >= S0 && S < S1 && S < nS) {
                if (dbg & 2) {
                    wv[S] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q) hv[4 * S + q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    continue;
                }
                wv[S] = *reinterpret_cast<const f32x4*>(W + S * 256);
#pragma unroll
                for (int q = 0; q < 4; ++q) hv[4 * S + q] = *reinterpret_cast<const f32x4*>(hp + (long)(4 * (4 * S + q)) * Bp);
            }
        }
    };
    if constexpr (!F16) load_groups(0, LSTM_MAXS);
    if constexpr (!F16) {
#pragma unroll
        for (int S = 0; S < LSTM_MAXS; ++S) {
            if (S < nS) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e)         // MFMA row r <-> clip 4 r + e; k = 16 S + 4 q + lk
                        acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[4 * S + q][e], wv[S][q], acc[e], 0, 0, 0);
            }
        }
    } else {
        f32x4acc accc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) accc[e] = (f32x4acc){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int P = 0; P < LSTM_MAXS / 2; ++P) {
            if (2 * P < nS) {
                load_groups(2 * P, 2 * P + 2);
                // weights of block P: wv[2P] = hi halves of the lane's 8 k, wv[2P+1] = lo halves
                const f16x8l wh = __builtin_bit_cast(f16x8l, wv[2 * P]), wl = __builtin_bit_cast(f16x8l, wv[2 * P + 1]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {           // clip group e: half e (hi) / 4 + e (lo) of the 8 loaded quads
                    f16x8l ah, al;
#pragma unroll
                    for (int pq = 0; pq < 8; ++pq) {
                        const f16x8l v = __builtin_bit_cast(f16x8l, hv[8 * P + pq]);
                        ah[pq] = v[e];
                        al[pq] = v[4 + e];
                    }
                    acc[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh, acc[e], 0, 0, 0);
                    accc[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl, accc[e], 0, 0, 0);
                    accc[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh, accc[e], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = acc[e] + accc[e] * (1.f / 2048.f);
    }
    // C layout 16x16: col = lane & 15 (gate row), row = 4*(lane>>4) + reg  ->  clip 4 row + e
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][16 * lk + 4 * r + e][li] = acc[e][r];
    __syncthreads();

    if (!cell || (dbg & 4)) return;
    float g4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = g * 4 + jj;
        float v = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < LSTM_WAVES; w8 += 4)
            v += (red[w8][br][col] + red[w8 + 1][br][col]) + (red[w8 + 2][br][col] + red[w8 + 3][br][col]);
        g4[g] = v + gin[g];
    }
    const float ig = sigmoidf_(g4[0]), fg = sigmoidf_(g4[1]), gg = tanhf_(g4[2]), og = sigmoidf_(g4[3]);
    const float c = fg * c_prev + ig * gg;
    const float h = og * tanhf_(c);
    *cst = c;
    float* hdst = (layer ? a.h1 : a.h0) + (long)(t & 1) * H * Bp;
    if constexpr (F16) {
        // quad-split state: 16 bytes per (k, clip quad) = [hi x 4 | lo x 4]
        _Float16* hq = reinterpret_cast<_Float16*>(hdst) + ((long)j * Bp + (cb & ~3)) * 2 + (cb & 3);
        const _Float16 hh = (_Float16)h;
        hq[0] = hh;
        hq[4] = (_Float16)((h - (float)hh) * 2048.f);
    } else {
        hdst[(long)j * Bp + cb] = h;
    }
    if (layer) {
        const float yv = h + x_skip;                    // lstm.py:37-38 skip
        const float o = a.elu_out ? (yv > 0.f ? yv : __expf(yv) - 1.f) : yv;
        if (a.out_s32) {
            float amax = 0.f;
            store_s32_1(a.y + ((long)cb * L + t) * H, j, o, amax);
            range_report(a.status, amax);
        } else a.y[((long)cb * L + t) * H + j] = o;
    }
}

int launch_lstm_step(const LstmArgs& a, int s, hipStream_t stream) {
    if (a.H != 512) { set_error("lstm: the step kernel is built for hidden size 512 (SEANet dimension)"); return -1; }
    if (s < 0 || s > 0xffff) { set_error("lstm: step index out of range"); return -1; }
    dim3 grid(2 * (a.H / 4), (a.B + 63) / 64);
    const int dbg = 0;
    LstmArgs b = a;
    if (!b.status) b.status = g_launch.status;
    if (a.f16x3) hipLaunchKernelGGL(lstm_step_kernel<true>, grid, dim3(64 * LSTM_WAVES), 0, stream, b, s | (dbg << 16));
    else hipLaunchKernelGGL(lstm_step_kernel<false>, grid, dim3(64 * LSTM_WAVES), 0, stream, b, s | (dbg << 16));
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------- ConvTranspose1d
// SConvTranspose1d (encoder/modules/conv.py:232-253), time-major: y[b][u][co] = bias[co] +
// sum over (t, j) with t*stride + j - trim_left == u of elu(x[b][t][ci]) * w[j][ci][co].
// With k = 2*stride every output sample has exactly two contributing input frames.
__global__ __launch_bounds__(256) void convtr_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y, long total,
                                                     int Tin, int Tout, int Cin, int Cout, int k, int stride, int trim_l,
                                                     int elu_in) {
    const int c4n = Cout >> 2;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long m = idx / c4n;
        const int co = (int)(idx - m * c4n) * 4;
        const long b = m / Tout;
        const int u = (int)(m - b * Tout) + trim_l;     // position in the untrimmed output
        f32x4 acc = *reinterpret_cast<const f32x4*>(bias + co);
        // t ranges over frames with 0 <= u - t*stride < k
        int t_hi = u / stride;
        if (t_hi > Tin - 1) t_hi = Tin - 1;
        int t_lo = (u - k + stride) / stride;
        if (u - k + 1 <= 0) t_lo = 0;
        if (t_lo < 0) t_lo = 0;
        for (int t = t_lo; t <= t_hi; ++t) {
            const int j = u - t * stride;
            if (j < 0 || j >= k) continue;
            const float* xr = x + (b * Tin + t) * Cin;
            const float* wr = w + ((long)j * Cin) * Cout + co;
            for (int ci = 0; ci < Cin; ++ci) {
                float xv = xr[ci];
                if (elu_in) xv = xv > 0.f ? xv : expm1f(xv);
                acc += xv * *reinterpret_cast<const f32x4*>(wr + (long)ci * Cout);
            }
        }
        *reinterpret_cast<f32x4*>(y + m * Cout + co) = acc;
    }
}

int launch_convtr(const float* x, const float* w, const float* bias, float* y, int B, int Tin, int Cin, int Cout,
                  int k, int stride, int elu_in, hipStream_t s) {
    const int pad_total = k - stride;
    const int pr = pad_total / 2, pl = pad_total - pr;
    const int Tout = (Tin - 1) * stride + k - pad_total;    // = Tin*stride
    const long total = (long)B * Tout * (Cout / 4);
    int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(convtr_kernel, dim3(blocks), dim3(256), 0, s, x, w, bias, y, total, Tin, Tout, Cin, Cout, k,
                       stride, pl, elu_in);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace wt
