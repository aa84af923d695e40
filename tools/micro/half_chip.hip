// Micro-experiment for DESIGN section 8 item 9 / VERDICT r03 #1: can another stream's kernels make progress beside a kernel
// that fills the CUs of HALF the XCDs (a persistent LSTM cut down to 4 XCDs)?
//
// `occupier`: 256 workgroups, each big enough (150 KB of LDS, 768 threads) to own a CU.  Workgroups that find themselves on
// an XCC >= keep leave at once; the others hold their CU for hold_us microseconds (bounded spin on the 100 MHz clock).
// `probe`: a second stream's kernel, launched while the occupier holds XCDs [0, keep): every workgroup records its XCC id
// and its entry / exit clock and works for work_us.  Two shapes: many small workgroups (a non-persistent launch) and 256
// workgroups of 100 KB LDS each (the shape of the persistent gemm16s launches: one per CU).
// What it answers: where the probe's workgroups land, and whether a probe launch can COMPLETE before the occupier ends -
// i.e. whether the dispatcher sends a launch's workgroups only to XCDs with room, or deals them out round-robin and lets
// those dealt to a full XCD wait (in which case the launch ends no earlier than the occupier does).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/half_chip.hip -o tools/micro/half_chip && tools/micro/half_chip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

struct Rec { unsigned long long t0, t1; unsigned xcc, pad; };

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7; }
__device__ __forceinline__ void hold(unsigned long long t_end) {
    for (long spin = 0; spin < (1L << 22); ++spin) {              // bounded: at most a few hundred ms even if the clock stood still
        if (__builtin_amdgcn_s_memrealtime() >= t_end) break;
        __builtin_amdgcn_s_sleep(8);
    }
}

__global__ __launch_bounds__(768) void occupier(Rec* rec, int keep, int hold_us) {
    extern __shared__ char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned xcc = xcc_id();
    if (threadIdx.x == 0) lds[0] = 1;
    if ((int)xcc < keep) hold(t0 + 100ull * hold_us);
    if (threadIdx.x == 0) { rec[blockIdx.x].t0 = t0; rec[blockIdx.x].t1 = __builtin_amdgcn_s_memrealtime(); rec[blockIdx.x].xcc = xcc; }
}

__global__ __launch_bounds__(512) void probe(Rec* rec, int work_us) {
    extern __shared__ char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) lds[0] = 1;
    hold(t0 + 100ull * work_us);
    if (threadIdx.x == 0) { rec[blockIdx.x].t0 = t0; rec[blockIdx.x].t1 = __builtin_amdgcn_s_memrealtime(); rec[blockIdx.x].xcc = xcc_id(); }
}

static void summarize(const char* what, const std::vector<Rec>& r, unsigned long long base, unsigned long long occ_end) {
    unsigned long long a = ~0ull, b = 0;
    int per[8] = {}, early[8] = {};
    for (const Rec& x : r) {
        a = std::min(a, x.t0); b = std::max(b, x.t1);
        per[x.xcc & 7]++;
        if (x.t1 <= occ_end) early[x.xcc & 7]++;
    }
    printf("%-34s first entry %+8.1f us, last exit %+8.1f us (occupier ends at %+8.1f)  wgs per xcc:", what,
           ((double)a - (double)base) / 100.0, ((double)b - (double)base) / 100.0, ((double)occ_end - (double)base) / 100.0);
    for (int i = 0; i < 8; ++i) printf(" %d", per[i]);
    printf("  done before the occupier ended:");
    for (int i = 0; i < 8; ++i) printf(" %d", early[i]);
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const int occ_lds = 150 * 1024, big_lds = 100 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(occupier), hipFuncAttributeMaxDynamicSharedMemorySize, occ_lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, big_lds));
    const int NP = 6;
    Rec *d_occ, *d_pr;
    CHECK(hipMalloc(&d_occ, 256 * sizeof(Rec)));
    CHECK(hipMalloc(&d_pr, (size_t)NP * 4096 * sizeof(Rec)));
    for (int keep : {0, 4, 8}) {
        for (int shape = 0; shape < 2; ++shape) {
            const int nwg = shape ? 256 : 2048, threads = shape ? 512 : 256, lds = shape ? big_lds : 1024, work_us = shape ? 20 : 5;
            CHECK(hipMemset(d_occ, 0, 256 * sizeof(Rec)));
            CHECK(hipMemset(d_pr, 0, (size_t)NP * 4096 * sizeof(Rec)));
            CHECK(hipDeviceSynchronize());
            hipLaunchKernelGGL(occupier, dim3(256), dim3(768), occ_lds, sa, d_occ, keep, 400);
            // give the occupier's workgroups time to take their CUs: a tiny kernel on the probe stream first
            hipLaunchKernelGGL(probe, dim3(8), dim3(64), 1024, sb, d_pr + (size_t)(NP - 1) * 4096, 30);
            for (int k = 0; k < NP - 1; ++k)
                hipLaunchKernelGGL(probe, dim3(nwg), dim3(threads), lds, sb, d_pr + (size_t)k * 4096, work_us);
            CHECK(hipGetLastError());
            CHECK(hipDeviceSynchronize());
            std::vector<Rec> occ(256), pr((size_t)NP * 4096);
            CHECK(hipMemcpy(occ.data(), d_occ, 256 * sizeof(Rec), hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(pr.data(), d_pr, pr.size() * sizeof(Rec), hipMemcpyDeviceToHost));
            unsigned long long base = ~0ull, occ_end = 0;
            for (const Rec& x : occ) { base = std::min(base, x.t0); if ((int)x.xcc < keep) occ_end = std::max(occ_end, x.t1); }
            if (!occ_end) occ_end = base;
            printf("---- occupier keeps XCCs [0, %d) for 400 us; probe shape: %d workgroups x %d threads, %d KB LDS, %d us of work each\n",
                   keep, nwg, threads, lds / 1024, work_us);
            summarize("occupier", occ, base, occ_end);
            for (int k = 0; k < NP - 1; ++k) {
                char nm[64];
                snprintf(nm, sizeof nm, "probe launch %d", k);
                summarize(nm, std::vector<Rec>(pr.begin() + (size_t)k * 4096, pr.begin() + (size_t)k * 4096 + nwg), base, occ_end);
            }
        }
    }
    printf("done\n");
    return 0;
}
