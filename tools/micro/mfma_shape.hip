// Micro-benchmark (MI355X_MICROARCH.md, DVFS give-back item 7): does the chip hold a higher clock on
// v_mfma_f32_16x16x32_f16 than on v_mfma_f32_32x32x16_f16 for the SAME work?  Both loops compute a 32 x 96 wave tile
// over k = 32 per iteration as split-f16 (3 MFMAs per product term, like gemm16s.hip), with every operand fragment
// re-read from LDS by ds_read_b128 each iteration (random f16 data), 8 waves per workgroup, one workgroup per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip ; run: ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>   // 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(512, 2) void loop_kernel(const _Float16* src, float* out, int iters, unsigned long long* stamps) {
    __shared__ __attribute__((aligned(16))) char lds[(128 + 192) * 128];       // one K stage of the GEMM: 320 rows x 128 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (128 + 192) * 128 / 16; i += 512)
        reinterpret_cast<f32x4*>(lds)[i] = reinterpret_cast<const f32x4*>(src)[(blockIdx.x * 37 + i) % 16384];
    __syncthreads();
    const int wm = wave >> 1, wn = wave & 1;
    const char* sA = lds + wm * 32 * 128;
    const char* sB = lds + (128 + wn * 96) * 128;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if (SHAPE == 0) {
        f32x16 accm[3], accc[3];
        for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) { accm[j][r] = 0.f; accc[j][r] = 0.f; }
        const int frow = (lane & 31) * 128, fsw = ((lane & 31) >> 1) & 7, fh = lane >> 5;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(sA + frow + (((2 * s + fh) ^ fsw) * 16));
                const f16x8 al = *reinterpret_cast<const f16x8*>(sA + frow + (((4 + 2 * s + fh) ^ fsw) * 16));
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const f16x8 bh = *reinterpret_cast<const f16x8*>(sB + j * 32 * 128 + frow + (((2 * s + fh) ^ fsw) * 16));
                    const f16x8 bl = *reinterpret_cast<const f16x8*>(sB + j * 32 * 128 + frow + (((4 + 2 * s + fh) ^ fsw) * 16));
                    accm[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, accm[j], 0, 0, 0);
                    accc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, accc[j], 0, 0, 0);
                    accc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, accc[j], 0, 0, 0);
                }
            }
            asm volatile("" ::: "memory");
        }
        for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) sum += accm[j][r] + accc[j][r];
    } else {
        // 16x16x32: the wave tile is 2 x 6 tiles of 16 x 16, one k = 32 step per iteration; lane (r = lane & 15, q = lane >> 4)
        // holds k = 8 q .. 8 q + 7 of row r: the hi half's chunk q, the lo half's chunk 4 + q
        f32x4 accm[2][6], accc[2][6];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 6; ++j) { accm[i][j] = (f32x4){0, 0, 0, 0}; accc[i][j] = (f32x4){0, 0, 0, 0}; }
        const int r16 = lane & 15, q = lane >> 4;
        for (int it = 0; it < iters; ++it) {
            f16x8 ah[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = i * 16 + r16, sw = (row >> 1) & 7;
                ah[i] = *reinterpret_cast<const f16x8*>(sA + row * 128 + ((q ^ sw) * 16));
                al[i] = *reinterpret_cast<const f16x8*>(sA + row * 128 + (((4 + q) ^ sw) * 16));
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int row = j * 16 + r16, sw = (row >> 1) & 7;
                const f16x8 bh = *reinterpret_cast<const f16x8*>(sB + row * 128 + ((q ^ sw) * 16));
                const f16x8 bl = *reinterpret_cast<const f16x8*>(sB + row * 128 + (((4 + q) ^ sw) * 16));
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    accm[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[i], accm[i][j], 0, 0, 0);
                    accc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[i], accc[i][j], 0, 0, 0);
                    accc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[i], accc[i][j], 0, 0, 0);
                }
            }
            asm volatile("" ::: "memory");
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 6; ++j) for (int r = 0; r < 4; ++r) sum += accm[i][j][r] + accc[i][j][r];
    }
    if (tid == 0) {
        stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[blockIdx.x * 512 + tid] = sum;
}

int main() {
    const int iters = 24 * 3 * 4;                       // four pwconv1-sized launches' worth of K steps per workgroup
    std::vector<_Float16> h(16384 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)(((rand() % 2001) - 1000) / 500.0f);
    _Float16* src; float* out; unsigned long long* st;
    hipMalloc(&src, h.size() * 2); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 256 * 16);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape = 0; shape < 2; ++shape) {
        for (int pass = 0; pass < 2; ++pass) {
            const int reps = pass ? 200 : 400;          // settle the clock first
            hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) {
                if (shape == 0) hipLaunchKernelGGL(loop_kernel<0>, dim3(256), dim3(512), 0, 0, src, out, iters, st);
                else hipLaunchKernelGGL(loop_kernel<1>, dim3(256), dim3(512), 0, 0, src, out, iters, st);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (pass) {
                std::vector<unsigned long long> s(512);
                hipMemcpy(s.data(), st, 256 * 16, hipMemcpyDeviceToHost);
                double ghz = 0; for (int b = 0; b < 256; ++b) ghz += (double)s[2 * b] / (double)s[2 * b + 1] * 0.1;
                const double flop = 256.0 * 8 * iters * (2.0 * 32 * 96 * 32 * 3);
                printf("%s: %.1f us per launch, %.0f TF f16 MFMA (%.0f TF fp32-equivalent), in-kernel clock %.2f GHz\n",
                       shape ? "16x16x32" : "32x32x16", 1e3 * ms / reps, flop / (ms / reps * 1e-3) / 1e12,
                       flop / 3 / (ms / reps * 1e-3) / 1e12, ghz / 256);
            }
        }
    }
    return 0;
}
