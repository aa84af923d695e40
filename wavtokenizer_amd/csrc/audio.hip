// Helpers on either side of the hot path: channel mix + resample to the codec rate, float -> PCM16,
// and the linear overlap-add of segment outputs.  All memory-bound, one pass over the samples each.
#include "../../include/wavtokenizer_amd.h"
#include "common.h"

#include <cmath>
#include <numeric>
#include <vector>

struct wt_resampler {
    int device = 0;
    int orig = 1, nw = 1, K = 0, width = 0;     // gcd-reduced rates, taps per phase, half width
    float* kern = nullptr;                      // [nw][K]
};

namespace wt {

// convert_audio (encoder/utils.py:79-92): mean over channels, then torchaudio.transforms.Resample(sr, target_sr):
// polyphase windowed sinc, out[n] = sum_k kern[n % new][k] * xpad[(n / new) * orig + k], xpad = x shifted by `width`
// zeros.  A block produces 256 consecutive outputs of one clip from an LDS window of the mixed input.
__global__ __launch_bounds__(256) void resample_mono_kernel(const float* __restrict__ wav, const float* __restrict__ kern,
                                                            float* __restrict__ out, int C, long T, long Tout, int orig,
                                                            int nw, int K, int width) {
    extern __shared__ float win[];
    const long n0 = (long)blockIdx.x * 256;
    const int b = blockIdx.y;
    const long i0 = n0 / nw;
    const long nlast = (n0 + 255 < Tout ? n0 + 255 : Tout - 1);
    const long i1 = nlast / nw;
    const long p0 = i0 * orig - width;                     // input position of win[0]
    const int Lw = (int)((i1 - i0) * orig) + K;
    const float inv = 1.f / (float)C;
    for (int e = threadIdx.x; e < Lw; e += 256) {
        const long p = p0 + e;
        float v = 0.f;
        if (p >= 0 && p < T) {
            for (int c = 0; c < C; ++c) v += wav[((long)b * C + c) * T + p];
            if (C > 1) v *= inv;
        }
        win[e] = v;
    }
    __syncthreads();
    const long n = n0 + threadIdx.x;
    if (n >= Tout) return;
    const long i = n / nw;
    const int ph = (int)(n - i * nw);
    const float* kp = kern + (long)ph * K;
    const float* wp = win + (i - i0) * orig;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(kp[k], wp[k], acc);
    out[(long)b * Tout + n] = acc;
}

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));      // non-negative floats order like their bits
}

// save_audio (encoder/utils.py:95-103) + PCM_S 16: clamp to +-limit (or scale by min(limit / max|x|, 1)), then
// round-half-even of x * 32768 clipped to int16
__global__ __launch_bounds__(256) void pcm16_kernel(const float* __restrict__ x, long n, float limit, int rescale,
                                                    const unsigned* __restrict__ amax, int16_t* __restrict__ out) {
    float scale = 1.f;
    if (rescale) {
        const float mx = __uint_as_float(*amax);
        scale = mx > 0.f ? fminf(limit / mx, 1.f) : 1.f;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v = x[i];
        v = rescale ? v * scale : fminf(fmaxf(v, -limit), limit);
        const float r = rintf(v * 32768.f);
        out[i] = (int16_t)fminf(fmaxf(r, -32768.f), 32767.f);
    }
}

// _linear_overlap_add (encoder/utils.py:17-56): out[r][t] = (sum_f w[t - f stride] * frame_f[r][t - f stride]) / (sum_f
// w[t - f stride]) over the frames covering t, added in frame order with separately rounded products and sums like
// the reference's `out += weight * frame` loop (bit-identical)
__global__ __launch_bounds__(256) void overlap_add_kernel(const float* __restrict__ frames, const float* __restrict__ w,
                                                          int nf, long rows, long flen, long last_len, long stride,
                                                          long total, float* __restrict__ out) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long r = blockIdx.y;
    if (t >= total) return;
    long f0 = t >= flen ? (t - flen) / stride + 1 : 0;      // first frame with f*stride + flen > t
    float acc = 0.f, sw = 0.f;
    for (long f = f0; f < nf && f * stride <= t; ++f) {
        const long o = t - f * stride;
        const long len = f + 1 == nf ? last_len : flen;
        if (o < len) {
            acc = __fadd_rn(acc, __fmul_rn(w[o], frames[((long)f * rows + r) * flen + o]));
            sw = __fadd_rn(sw, w[o]);
        }
    }
    out[r * total + t] = acc / sw;
}

}  // namespace wt

using namespace wt;

extern "C" {

int wt_resampler_create(int32_t orig_sr, int32_t new_sr, int32_t device, wt_resampler** out) {
    if (!out || orig_sr <= 0 || new_sr <= 0) { set_error("wt_resampler_create: bad argument"); return WT_ERR_INVALID; }
    WT_HIP_CHECK(hipSetDevice(device));
    const int g = std::gcd(orig_sr, new_sr);
    const int orig = orig_sr / g, nw = new_sr / g;
    // torchaudio.functional.resample defaults: sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99
    const int lpw = 6;
    const double rolloff = 0.99;
    const double base = std::min(orig, nw) * rolloff;
    const bool same = orig == nw;             // Resample.forward returns the waveform unchanged at equal rates
    const int width = same ? 0 : (int)std::ceil(lpw * orig / base);
    const int K = same ? 1 : 2 * width + orig;
    if (256.0 * orig / nw + K > 15000) { set_error("wt_resampler_create: rate ratio too large for the LDS window"); return WT_ERR_INVALID; }
    std::vector<float> k((size_t)nw * K);
    const double pi = 3.14159265358979323846;
    for (int p = 0; p < nw; ++p)
        for (int j = 0; j < K; ++j) {
            double t = (-(double)p / nw + (double)(j - width) / orig) * base;
            t = std::min(std::max(t, -(double)lpw), (double)lpw);
            const double window = std::pow(std::cos(t * pi / lpw / 2), 2);
            t *= pi;
            const double sinc = t == 0.0 ? 1.0 : std::sin(t) / t;
            k[(size_t)p * K + j] = same ? 1.f : (float)(sinc * window * (base / orig));
        }
    wt_resampler* r = new wt_resampler();
    r->device = device; r->orig = orig; r->nw = nw; r->K = K; r->width = width;
    if (hipMalloc(reinterpret_cast<void**>(&r->kern), k.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(r->kern, k.data(), k.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("wt_resampler_create: device allocation failed");
        delete r;
        return WT_ERR_HIP;
    }
    *out = r;
    return WT_OK;
}

void wt_resampler_destroy(wt_resampler* r) {
    if (!r) return;
    (void)hipFree(r->kern);
    delete r;
}

int64_t wt_resampler_out_length(const wt_resampler* r, int64_t T) {
    return r ? (int64_t)((r->nw * T + r->orig - 1) / r->orig) : 0;       // ceil(new * length / orig)
}

int wt_convert_audio(const wt_resampler* r, const float* wav, int32_t B, int32_t C, int64_t T, float* out, void* stream) {
    if (!r || !wav || !out || B < 1 || T < 1) { set_error("wt_convert_audio: bad argument"); return WT_ERR_INVALID; }
    if (C != 1 && C != 2) { set_error("wt_convert_audio: audio must be mono or stereo (encoder/utils.py:81)"); return WT_ERR_INVALID; }
    WT_HIP_CHECK(hipSetDevice(r->device));
    const int64_t Tout = wt_resampler_out_length(r, T);
    const size_t smem = (size_t)(256 / r->nw + 2) * r->orig * sizeof(float) + (size_t)r->K * sizeof(float);
    static PerDeviceOnce attr_once;
    if (int rc = attr_once.run([&]() -> int {
        WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(resample_mono_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        return 0;
    })) return rc;
    dim3 grid((unsigned)((Tout + 255) / 256), B);
    hipLaunchKernelGGL(resample_mono_kernel, grid, dim3(256), smem, static_cast<hipStream_t>(stream), wav, r->kern, out, C,
                       (long)T, (long)Tout, r->orig, r->nw, r->K, r->width);
    WT_HIP_CHECK(hipGetLastError());
    return WT_OK;
}

int wt_pcm16(const float* x, int64_t n, float limit, int32_t rescale, int16_t* out, void* workspace, void* stream) {
    if (!x || !out || n < 1 || (rescale && !workspace)) { set_error("wt_pcm16: bad argument"); return WT_ERR_INVALID; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned* amax = static_cast<unsigned*>(workspace);
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 2048);
    if (rescale) {
        WT_HIP_CHECK(hipMemsetAsync(amax, 0, sizeof(unsigned), s));
        hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, s, x, (long)n, amax);
    }
    hipLaunchKernelGGL(pcm16_kernel, dim3(blocks), dim3(256), 0, s, x, (long)n, limit, rescale, amax, out);
    WT_HIP_CHECK(hipGetLastError());
    return WT_OK;
}

int wt_linear_overlap_add(const float* frames, const float* weight, int32_t n_frames, int64_t rows, int64_t frame_len,
                          int64_t last_len, int64_t stride, float* out, void* stream) {
    if (!frames || !weight || !out || n_frames < 1 || rows < 1 || frame_len < 1 || last_len < 1 || last_len > frame_len ||
        stride < 1 || rows > 65535) {
        set_error("wt_linear_overlap_add: bad argument"); return WT_ERR_INVALID;
    }
    if (stride > frame_len && n_frames > 1) { set_error("wt_linear_overlap_add: stride beyond the frame length leaves uncovered samples"); return WT_ERR_INVALID; }
    const long total = stride * (n_frames - 1) + last_len;
    dim3 grid((unsigned)((total + 255) / 256), (unsigned)rows);
    hipLaunchKernelGGL(overlap_add_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), frames, weight, n_frames,
                       (long)rows, (long)frame_len, (long)last_len, (long)stride, total, out);
    WT_HIP_CHECK(hipGetLastError());
    return WT_OK;
}

}  // extern "C"
