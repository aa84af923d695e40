// Same-process A/B of gemm16s_kernel variants on the ConvNeXt pwconv shapes (no torch, no Python): the kernel source is
// included with WT_GEMM16S_LAB (its product dispatcher is compiled out) and the compile-time experiment masks listed in
// VARIANTS are instantiated side by side.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWT_GEMM16S_LAB -DWT_LAB -o gpurun_out/gemm_lab tools/micro/gemm_lab.hip && gpurun_out/gemm_lab
// Rounds are interleaved (variant 0, 1, 2, ..., 0, 1, ...) and the median launch time per variant is printed together with
// the largest difference of its output from variant 0's (identical arithmetic unless the mask changes it).
#include "../../wavtokenizer_amd/csrc/gemm16s.hip"

#include <algorithm>
#include <cstring>
#include <random>
#include <vector>

namespace wt {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
thread_local LaunchCtx g_launch;
}  // namespace wt
using namespace wt;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Variant { const char* name; int (*launch)(const GemmArgs&, hipStream_t); };

template <int EPI, int OUT, int DBG>
static int run128x192(const GemmArgs& a, hipStream_t s) { return launch16s_one<128, 192, 4, 2, 3, EPI, OUT, 2, DBG>(a, s); }
// 4 waves, one per SIMD, 512 registers (wave tile 64 x 96)
template <int EPI, int OUT, int DBG>
static int run128x128(const GemmArgs& a, hipStream_t s) { return launch16s_one<128, 128, 4, 2, 3, EPI, OUT, 2, DBG>(a, s); }
// the narrow tiles of small batches: one K tile per barrier on three stages, two per barrier on six
template <int EPI, int OUT, int DBG>
static int run128x32(const GemmArgs& a, hipStream_t s) { return launch16s_one<128, 32, 4, 1, 3, EPI, OUT, 2, DBG>(a, s); }
template <int EPI, int OUT, int DBG>
static int run128x32ks2(const GemmArgs& a, hipStream_t s) { return launch16s_one<128, 32, 4, 1, 6, EPI, OUT, 2, DBG, 2>(a, s); }
template <int EPI, int OUT, int DBG>
static int run128x32prod(const GemmArgs& a, hipStream_t s) { return launch16s_one<128, 32, 4, 1, 6, EPI, OUT, 2, DBG, 2, 1>(a, s); }
template <int EPI, int OUT, int DBG>
static int run64x32prod(const GemmArgs& a, hipStream_t s) { return launch16s_one<64, 32, 2, 1, 6, EPI, OUT, 2, DBG, 2, 1>(a, s); }
template <int EPI, int OUT, int DBG>
static int run32x32prod(const GemmArgs& a, hipStream_t s) { return launch16s_one<32, 32, 1, 1, 6, EPI, OUT, 2, DBG, 2, 1>(a, s); }
template <int EPI, int OUT, int DBG>
static int run64x64prod(const GemmArgs& a, hipStream_t s) { return launch16s_one<64, 32, 2, 1, 6, EPI, OUT, 2, DBG, 2, 2>(a, s); }
template <int EPI, int OUT, int DBG>
static int run128x64(const GemmArgs& a, hipStream_t s) { return launch16s_one<128, 64, 4, 1, 3, EPI, OUT, 2, DBG>(a, s); }
template <int EPI, int OUT, int DBG>
static int run128x64prod(const GemmArgs& a, hipStream_t s) { return launch16s_one<128, 64, 4, 1, 6, EPI, OUT, 2, DBG, 2, 1>(a, s); }
template <int EPI, int OUT, int DBG>
static int run64x32ks2(const GemmArgs& a, hipStream_t s) { return launch16s_one<64, 32, 2, 1, 6, EPI, OUT, 2, DBG, 2>(a, s); }
template <int EPI, int OUT, int DBG>
static int run64x64ks2(const GemmArgs& a, hipStream_t s) { return launch16s_one<64, 64, 2, 1, 6, EPI, OUT, 2, DBG, 2>(a, s); }
template <int EPI, int OUT, int DBG>
static int run4w(const GemmArgs& a, hipStream_t s) { return launch16s_one<128, 192, 2, 2, 3, EPI, OUT, 1, DBG>(a, s); }

static void fill_s32(std::vector<uint16_t>& v, long rows, long K, float scale, unsigned seed) {
    std::mt19937 rng(seed);
    std::normal_distribution<float> nd(0.f, scale);
    v.resize((size_t)rows * K * 2);
    for (long r = 0; r < rows; ++r)
        for (long g = 0; g < K / 32; ++g)
            for (int i = 0; i < 32; ++i) {
                const float x = nd(rng);
                const _Float16 h = (_Float16)x;
                const _Float16 l = (_Float16)((x - (float)h) * 2048.f);
                uint16_t hb, lb;
                hb = __builtin_bit_cast(uint16_t, h); lb = __builtin_bit_cast(uint16_t, l);
                v[((size_t)r * (K / 32) + g) * 64 + i] = hb;
                v[((size_t)r * (K / 32) + g) * 64 + 32 + i] = lb;
            }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 30;
    // LAB_FLUSH=1: 1 GB is overwritten before every timed launch, so that operands and output lines start out of the L2s and the
    // 256 MB memory-side cache, as they do inside a plan (the same buffers come round only once per step)
    void* dFlush = nullptr;
    const size_t flush_bytes = 1ull << 30;
    if (const char* e = getenv("LAB_FLUSH")) if (e[0] == '1') CK(hipMalloc(&dFlush, flush_bytes));
    struct Shape { const char* name; int M, N, K, epi; } shapes[] = {
        {"pwconv1 7680x2304x768 gelu->S32", 7680, 2304, 768, EPI_BIAS_GELU},
        {"pwconv2 7680x768x2304 gamma+res->f32", 7680, 768, 2304, EPI_BIAS_GAMMA_RES},
        {"head.out 7680x2432x768 exp/sincos->S32", 7680, 2432, 768, EPI_HEAD},
        {"pwconv2 at B = 1: 120x768x2304 gamma+res->f32", 120, 768, 2304, EPI_BIAS_GAMMA_RES},
        {"pwconv2 at B = 16: 1920x768x2304 gamma+res->f32", 1920, 768, 2304, EPI_BIAS_GAMMA_RES},
    };
    for (const Shape& sh : shapes) {
        std::vector<uint16_t> hA, hW;
        fill_s32(hA, sh.M, sh.K, 1.0f, 1);
        fill_s32(hW, sh.N, sh.K, 1.0f / sqrtf((float)sh.K), 2);
        std::vector<float> hb(sh.N), hg(sh.N), hR((size_t)sh.M * sh.N);
        std::mt19937 rng(3);
        std::normal_distribution<float> nd(0.f, 1.f);
        for (auto& x : hb) x = nd(rng);
        for (auto& x : hg) x = 0.1f + 0.2f * fabsf(nd(rng));
        for (auto& x : hR) x = nd(rng);
        void *dA, *dW, *dC, *dC0, *db, *dg, *dR;
        const size_t cbytes = (size_t)sh.M * sh.N * 4;
        CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dC, cbytes)); CK(hipMalloc(&dC0, cbytes));
        CK(hipMalloc(&db, sh.N * 4)); CK(hipMalloc(&dg, sh.N * 4)); CK(hipMalloc(&dR, cbytes));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), sh.N * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dg, hg.data(), sh.N * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dR, hR.data(), cbytes, hipMemcpyHostToDevice));
        GemmArgs a;
        a.A = static_cast<const float*>(dA); a.a_bstride = 0; a.a_rstride = sh.K; a.T_in = sh.M; a.T_out = sh.M; a.Cin = sh.K; a.taps = 1;
        a.W = static_cast<const float*>(dW); a.W_hi = dW; a.w_rstride = sh.K; a.bias = static_cast<const float*>(db);
        a.M = sh.M; a.N = sh.N; a.K = sh.K; a.C = static_cast<float*>(dC); a.c_rstride = sh.N;
        a.R = static_cast<const float*>(dR); a.r_rstride = sh.N; a.gamma = static_cast<const float*>(dg);
        a.group_m = (sh.N + 191) / 192 > 8 ? 8 : 1;
        if (sh.epi == EPI_HEAD) a.head_kb = sh.N / 2;
        if (const char* e = getenv("LAB_GM")) a.group_m = atoi(e);
        if (const char* e = getenv("LAB_GN")) a.group_n = atoi(e);
        std::vector<Variant> vs;
        // experiment masks (gemm16s.hip): 2048 no DMA spread, 4096 plain (not sc1) staged stores, 8192 s_setprio around the MFMA blocks,
        // 16384 round 2's two-branch GELU, 65536 vector wave id, 262144 sc1 stores in the direct epilogue too, 4 no epilogue, 1 no DMA
        if (sh.epi == EPI_BIAS_GELU) {
            vs = {{"r02 arithmetic and structure", run128x192<EPI_BIAS_GELU, OUT_S32, 16384 + 65536 + 2048 + 4096>},
                  {"shipped r03", run128x192<EPI_BIAS_GELU, OUT_S32, 0>},
                  {"  r03 without DMA spread", run128x192<EPI_BIAS_GELU, OUT_S32, 2048>},
                  {"  r03 without sc1 stores", run128x192<EPI_BIAS_GELU, OUT_S32, 4096>},
                  {"  r03 with the two-branch GELU", run128x192<EPI_BIAS_GELU, OUT_S32, 16384>},
                  {"  r03 with GELU one sub-run at a time", run128x192<EPI_BIAS_GELU, OUT_S32, 524288>},
                  {"  r03 with a vector wave id", run128x192<EPI_BIAS_GELU, OUT_S32, 65536>},
                  {"  r03 + setprio", run128x192<EPI_BIAS_GELU, OUT_S32, 8192>},
                  {"no epilogue + in-loop fake drain", run128x192<EPI_BIAS_GELU, OUT_S32, 4 + 2097152>},
                  {"r03 + in-loop fake drain", run128x192<EPI_BIAS_GELU, OUT_S32, 2097152>},
                  {"no epilogue + fake drain, SIMD partners in opposite halves", run128x192<EPI_BIAS_GELU, OUT_S32, 4 + 2097152 + 16777216>},
                  {"r03 + fake drain, SIMD partners in opposite halves", run128x192<EPI_BIAS_GELU, OUT_S32, 2097152 + 16777216>},
                  {"epilogue without GELU", run128x192<EPI_BIAS_GELU, OUT_S32, 512>},
                  {"epilogue without global stores", run128x192<EPI_BIAS_GELU, OUT_S32, 128>},
                  {"epilogue without GELU and stores", run128x192<EPI_BIAS_GELU, OUT_S32, 512 + 128>},
                  {"no epilogue", run128x192<EPI_BIAS_GELU, OUT_S32, 4>},
                  {"no epilogue, DMA fetches nothing", run128x192<EPI_BIAS_GELU, OUT_S32, 4 + 131072>},
                  {"no epilogue, DMA re-reads one 8 KB window", run128x192<EPI_BIAS_GELU, OUT_S32, 4 + 1048576>},
                  {"no epilogue, no DMA", run128x192<EPI_BIAS_GELU, OUT_S32, 5>}};
        } else if (sh.epi == EPI_BIAS_GAMMA_RES && sh.M == 1920) {
            vs = {{"128x64, one K tile per barrier (shipped for 33-100 tiles of 128x128)", run128x64<EPI_BIAS_GAMMA_RES, OUT_F32, 0>},
                  {"128x64, two K tiles per barrier + 4 loader waves", run128x64prod<EPI_BIAS_GAMMA_RES, OUT_F32, 0>},
                  {"128x32, two K tiles per barrier + 4 loader waves (360 tiles)", run128x32prod<EPI_BIAS_GAMMA_RES, OUT_F32, 0>},
                  {"128x128, 8 waves (the large-problem kernel)", run128x128<EPI_BIAS_GAMMA_RES, OUT_F32, 0>}};
        } else if (sh.epi == EPI_BIAS_GAMMA_RES && sh.M < 1000) {
            vs = {{"narrow 128x32, one K tile per barrier (r02)", run128x32<EPI_BIAS_GAMMA_RES, OUT_F32, 0>},
                  {"narrow 128x32, two K tiles per barrier, six stages (shipped r03)", run128x32ks2<EPI_BIAS_GAMMA_RES, OUT_F32, 0>},
                  {"128x32, two K tiles per barrier + 4 loader waves", run128x32prod<EPI_BIAS_GAMMA_RES, OUT_F32, 0>},
                  {"64x32, two K tiles per barrier + 2 loader waves", run64x32prod<EPI_BIAS_GAMMA_RES, OUT_F32, 0>},
                  {"32x32, two K tiles per barrier + 1 loader wave", run32x32prod<EPI_BIAS_GAMMA_RES, OUT_F32, 0>},
                  {"64x32, two K tiles per barrier + 4 loader waves", run64x64prod<EPI_BIAS_GAMMA_RES, OUT_F32, 0>},
                  {"64x32 tiles, 2 waves, two K tiles per barrier", run64x32ks2<EPI_BIAS_GAMMA_RES, OUT_F32, 0>},
                  {"64x64 tiles, 2 waves, two K tiles per barrier", run64x64ks2<EPI_BIAS_GAMMA_RES, OUT_F32, 0>},
                  {"narrow, one per barrier: no epilogue", run128x32<EPI_BIAS_GAMMA_RES, OUT_F32, 4>},
                  {"narrow, one per barrier: no epilogue, DMA fetches nothing", run128x32<EPI_BIAS_GAMMA_RES, OUT_F32, 4 + 131072>},
                  {"narrow, one per barrier: no epilogue, no DMA", run128x32<EPI_BIAS_GAMMA_RES, OUT_F32, 5>},
                  {"narrow, one per barrier: no epilogue, no DMA, no barrier", run128x32<EPI_BIAS_GAMMA_RES, OUT_F32, 13>},
                  {"narrow, one per barrier: no epilogue, no DMA, no barrier, no LDS reads", run128x32<EPI_BIAS_GAMMA_RES, OUT_F32, 29>}};
        } else if (sh.epi == EPI_HEAD) {
            // 8388608: direct (unstaged) 8-byte stores; 4194304: libm expf / sincosf; 128: no stores; 4: no epilogue
            vs = {{"r02: direct stores, libm expf/sincosf", run128x128<EPI_HEAD, OUT_S32, 4194304 + 8388608>},
                  {"shipped r03: 128x128 staged", run128x128<EPI_HEAD, OUT_S32, 0>},
                  {"  128x128 staged with libm", run128x128<EPI_HEAD, OUT_S32, 4194304>},
                  {"  128x128 direct stores", run128x128<EPI_HEAD, OUT_S32, 8388608>},
                  {"  128x128 staged, plain (not sc1) stores", run128x128<EPI_HEAD, OUT_S32, 4096>},
                  {"  128x128 staged without global stores", run128x128<EPI_HEAD, OUT_S32, 128>},
                  {"  128x128 no epilogue", run128x128<EPI_HEAD, OUT_S32, 4>},
                  {"  128x192 tiles (direct stores)", run128x192<EPI_HEAD, OUT_S32, 0>}};
        } else {
            vs = {{"r02 structure", run128x192<EPI_BIAS_GAMMA_RES, OUT_F32, 65536 + 2048>},
                  {"shipped r03", run128x192<EPI_BIAS_GAMMA_RES, OUT_F32, 0>},
                  {"  r03 + sc1 stores (direct epilogue)", run128x192<EPI_BIAS_GAMMA_RES, OUT_F32, 262144>},
                  {"  r03 without DMA spread", run128x192<EPI_BIAS_GAMMA_RES, OUT_F32, 2048>},
                  {"no epilogue", run128x192<EPI_BIAS_GAMMA_RES, OUT_F32, 4>}};
        }
        if (argc > 2) {          // keep variant 0 (the reference for the output check) and the variants whose name contains argv[2]
            std::vector<Variant> keep;
            for (size_t v = 0; v < vs.size(); ++v) if (v == 0 || strstr(vs[v].name, argv[2])) keep.push_back(vs[v]);
            vs = keep;
        }
        const size_t nv = vs.size();
        std::vector<std::vector<float>> times(nv);
        std::vector<double> maxdiff(nv, 0.0);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        std::vector<float> h0((size_t)sh.M * sh.N), h1((size_t)sh.M * sh.N);
        for (int r = 0; r < rounds + 3; ++r)
            for (size_t v = 0; v < nv; ++v) {
                if (sh.epi == EPI_BIAS_GAMMA_RES) CK(hipMemcpyAsync(dC, dR, cbytes, hipMemcpyDeviceToDevice, nullptr));   // residual in place, as the plan runs it
                GemmArgs b = a;
                if (sh.epi == EPI_BIAS_GAMMA_RES) b.R = b.C;
                if (dFlush) CK(hipMemsetAsync(dFlush, r + (int)v, flush_bytes, nullptr));
                CK(hipEventRecord(e0, nullptr));
                if (vs[v].launch(b, nullptr)) { fprintf(stderr, "launch failed: %s\n", g_err.c_str()); return 1; }
                CK(hipEventRecord(e1, nullptr));
                CK(hipEventSynchronize(e1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 3) times[v].push_back(ms * 1e3f);
                if (r == 0) {
                    CK(hipMemcpy(v == 0 ? h0.data() : h1.data(), dC, cbytes, hipMemcpyDeviceToHost));
                    if (v) {
                        // S32 output: compare decoded values (hi + lo / 2048); fp32: direct
                        double md = 0.0;
                        if (sh.epi == EPI_BIAS_GELU || sh.epi == EPI_HEAD) {
                            const uint16_t* p0 = reinterpret_cast<const uint16_t*>(h0.data());
                            const uint16_t* p1 = reinterpret_cast<const uint16_t*>(h1.data());
                            for (size_t g = 0; g < h0.size() / 32; g += 7)
                                for (int i = 0; i < 32; ++i) {
                                    const _Float16 a0 = __builtin_bit_cast(_Float16, p0[g * 64 + i]), b0 = __builtin_bit_cast(_Float16, p0[g * 64 + 32 + i]);
                                    const _Float16 a1 = __builtin_bit_cast(_Float16, p1[g * 64 + i]), b1 = __builtin_bit_cast(_Float16, p1[g * 64 + 32 + i]);
                                    md = std::max(md, fabs(((double)a0 + (double)b0 / 2048.0) - ((double)a1 + (double)b1 / 2048.0)));
                                }
                        } else {
                            for (size_t i = 0; i < h0.size(); i += 5) md = std::max(md, fabs((double)h0[i] - (double)h1[i]));
                        }
                        maxdiff[v] = md;
                    }
                }
            }
        printf("== %s (%d rounds, interleaved)\n", sh.name, rounds);
        for (size_t v = 0; v < nv; ++v) {
            std::sort(times[v].begin(), times[v].end());
            const float med = times[v][times[v].size() / 2], mn = times[v].front();
            printf("  %-44s median %7.1f us  min %7.1f us  %6.1f TF fp32-equiv   max |diff vs variant 0| %.3g\n", vs[v].name, med, mn,
                   2.0 * sh.M * sh.N * sh.K / med / 1e6, maxdiff[v]);
        }
        fflush(stdout);
        CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dC)); CK(hipFree(dC0)); CK(hipFree(db)); CK(hipFree(dg)); CK(hipFree(dR));
    }
    return 0;
}
