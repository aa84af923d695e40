// Fused SEANetResnetBlock (encoder/modules/seanet.py:21-63) on the f16 matrix pipe with fp32-equivalent products.
//
//   y = shortcut(x) + conv1(elu(conv3(elu(x))))
//
// Same fusion as resblock.hip (x tile with its k=3 halo in LDS, conv1 frame-local, weights resident in LDS, persistent
// workgroups), but every contraction runs as split-f16 (x = hi + lo * 2^-11, three f16 MFMAs per K step, main + correction
// accumulator: gemm16s.hip) instead of v_mfma_f32_32x32x2_f32.  The split is done ONCE per element when the tile is filled
// (ELU too, not once per tap), the LDS images hold rows of [hi | lo] f16 with XOR-swizzled 16-byte chunks so every
// ds_read_b128 fragment read is conflict-free, and the weights are the MFMA's A operand: the accumulator comes out with the
// frame on the lane and 4-channel runs in the registers, so the epilogue stores 16 bytes (fp32) or 8 + 8 bytes (S32) per lane.
// Three forms (template arguments below):
//   C = 32, FOLD: the first encoder conv (k7, 1 -> 32) is the tile fill, itself an MFMA pass over the waveform;
//   C = 32, FOLD, DOWN = r: ... and ELU + the stage's strided conv run on the tile while it is in LDS: encoder stage 1 in one
//       kernel (the shipped plan; wt_resblock_down);
//   C = 64, FPW = 16: the plain block on 8 waves of 16 frames (encoder stage 2, SEANetDecoder).
#include "common.h"

#include <stdlib.h>

namespace wt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rb16_elu(float x) { return elu_med3(x); }

__device__ __forceinline__ void rb16_split8(const float* v, f16x8& hi, f16x8& lo, float& amax) {
    amax = amax4(amax4(amax, v[0], v[1], v[2], v[3]), v[4], v[5], v[6], v[7]);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) { unsigned a, b; split2_f16(v[2 * i], v[2 * i + 1], a, b); h[i] = a; l[i] = b; }
    hi = __builtin_bit_cast(f16x8, h);
    lo = __builtin_bit_cast(f16x8, l);
}
__device__ __forceinline__ void rb16_split4(const f32x4 v, f16x4& hi, f16x4& lo, float& amax) {
    amax = amax4(amax, v[0], v[1], v[2], v[3]);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 h, l;
#pragma unroll
    for (int i = 0; i < 2; ++i) { unsigned a, b; split2_f16(v[2 * i], v[2 * i + 1], a, b); h[i] = a; l[i] = b; }
    hi = __builtin_bit_cast(f16x4, h);
    lo = __builtin_bit_cast(f16x4, l);
}

// LDS images (bytes).  An activation row of CH channels is CH*4 bytes: per 32 channels a 128-byte group
// [32 x hi | 32 x lo] (CH = 16: one 64-byte group [16 hi | 16 lo]); the 16-byte chunk c of row r is stored at chunk
// c ^ swz(r) with swz chosen per row pitch so that 16 lanes reading the same logical chunk of 16 different rows
// cover all 64 banks:  64-byte rows: (r >> 2) & 3, 128-byte rows: (r >> 1) & 7, 256-byte rows: r & 15.
template <int CH>
struct RbRow {
    static constexpr int bytes = CH * 4;
    static constexpr int chunks = bytes / 16;
    __device__ static __forceinline__ int swz(int r) {
        // 256-byte rows (C = 64): a 4-bit XOR-linear function of r & 7, (r0, r0^r1, r2, r1^r2), found by search with the bank
        // model of tools/lds_banks.py: conflict-free for 16 consecutive rows starting at ANY row (conv3's taps start at
        // offsets 0, 1, 2; round 2's r & 15 was 1.33 LDS cycles per ideal one there) and for the 8-lanes-per-row stores
        return bytes == 64 ? ((r >> 2) & 3) : (bytes == 128 ? ((r >> 1) & 7) : (int)((0x56FC9A30u >> (4 * (r & 7))) & 15u));
    }
    // byte offset of the 8-half chunk holding channels ci .. ci+7 (ci % 8 == 0) of row r; lo = the lo half
    __device__ static __forceinline__ int off(int r, int ci, int lo) {
        constexpr int G = CH < 32 ? CH : 32;                    // channels per [hi | lo] group
        const int c = (ci / G) * (G / 4) + lo * (G / 8) + (ci % G) / 8;
        return r * bytes + ((c ^ swz(r)) * 16);
    }
};

template <int C, int ROWS>
struct Rb16Layout {
    static constexpr int H = C / 2;
    static constexpr int N1 = H < 16 ? 16 : H;       // conv3 output rows: 16 (one 16x16x32 tile) or a multiple of 32
    static constexpr int K1 = 3 * C, K2 = H + C;
    static constexpr int NX = ROWS + 2;
    static constexpr int off_xe = 0;                                 // elu(x), rows -1 .. ROWS
    static constexpr int off_xr = off_xe + NX * C * 4;               // raw x, rows 0 .. ROWS-1 (FOLD: no raw copy exists - the shortcut goes
                                                                     // through the first conv - and the rows only stage elu(y))
    static constexpr int off_he = off_xr + ROWS * C * 4;             // elu(hidden)
    static constexpr int off_w3 = off_he + ROWS * H * 4;             // [K1/16][hi, lo][N1 rows][32 B]
    static constexpr int off_w2 = off_w3 + (K1 / 16) * 2 * N1 * 32;  // [K2/16][hi, lo][C rows][32 B]
    static constexpr int off_b = off_w2 + (K2 / 16) * 2 * C * 32;    // b3[N1], b12[C] fp32
    static constexpr int off_wav = off_b + (N1 + C) * 4;             // folded first conv: ROWS + 8 samples
    static constexpr int total = off_wav + 2 * (ROWS + 8) * 4;       // two windows: the next tile's is written before the barrier that ends this one
};

// weight image: 16-deep k-step ks, part hl, row n: 32 bytes = k 16 ks .. +15 as two 16-byte chunks (k half h),
// stored at chunk h ^ ((n >> 3) & 1): conflict-free for the 16-lane groups of a ds_read_b128
// sw: the image is read as v_mfma_f32_32x32x16_f16 operands (lane = (row of 32, k half)); images read as 16x16x32 operands
// (lane = (row of 16, k quarter)) take no swizzle: there the 16-lane groups of a ds_read_b128 pair rows n and n + 8 with
// OPPOSITE halves already, and the swizzle would put them on the same banks
__device__ __forceinline__ int rb16_woff(int rows, int ks, int hl, int n, int h, bool sw = true) {
    return ((ks * 2 + hl) * rows + n) * 32 + ((h ^ (sw ? ((n >> 3) & 1) : 0)) * 16);
}

// DBG: timing-experiment build (WT_RB16_DBG); the shipped instantiations test nothing at run time.
// DOWN = r > 0 (C = 32 with the folded first conv only): the stage's down conv — ELU, SConv1d(32 -> 64, k = 2r, stride r,
// reflect; seanet.py:123-127) — is computed from the tile's output while it is still in LDS, and only ITS output
// (64 channels at 1/r of the frame rate) goes to HBM: the 590 MB of stage-1 activations are neither written nor read back.
// A tile is the 126-frame window [i*OPT*r - r/2, + 126) that OPT = (126 - 2r)/r + 1 consecutive output frames need
// (30 for r = 4, 62 for r = 2: consecutive windows overlap, 5 % recomputed; the last window of a clip is shifted to end at the
// clip, so lengths that are no multiple of r work too); wave w owns output channels
// [16 w, 16 w + 16) with its 2r x 32 weights resident in registers as v_mfma_f32_16x16x32_f16 operands.
// FPW = frames per wave: 32 (one v_mfma_f32_32x32x16_f16 column block per wave, 4 waves) or 16 (C = 64: 8 waves, everything on
// v_mfma_f32_16x16x32_f16).  The 64-channel tile + weights take 130 KB of LDS, i.e. one workgroup per CU: with 4 waves that is ONE
// wave per SIMD and nothing overlaps its fill, MFMA and store phases; 8 waves of 16 frames give every SIMD a second wave.
template <int C, int ROWS, int FOLD, bool DBG = false, int DOWN = 0, int FPW = 32>
__global__ __launch_bounds__(ROWS / FPW * 64, DOWN ? 2 : 1) void resblock16_kernel(const ResblockArgs a) {
    static_assert(DOWN == 0 || (C == 32 && ROWS == 128 && FOLD == 1 && (DOWN == 2 || DOWN == 4)), "fused down conv: stage 1 only");
    static_assert(FPW == 32 || (FPW == 16 && C == 64 && FOLD == 0 && DOWN == 0), "16 frames per wave: the plain 64-channel block");
    constexpr int DK = 2 * DOWN;                                     // down conv taps
    // FOLD: the tile fill is one MFMA pass of 32 x-rows per wave, so the tile is ROWS x-rows INCLUDING the k=3 halo and
    // yields ROWS - 2 frames of y (the last two MFMA columns of conv3 read two zero rows and are dropped)
    constexpr int VALID = FOLD ? ROWS - 2 : ROWS;
    constexpr int OPT = DOWN ? (VALID - DK) / (DOWN ? DOWN : 1) + 1 : 0;          // output frames per tile
    constexpr int DMT = (OPT + 15) / 16;                             // 16-frame MFMA row tiles of the down conv
    using L = Rb16Layout<C, ROWS>;
    using XR = RbRow<C>;
    using HR = RbRow<L::H>;
    constexpr int NT = ROWS / FPW * 64;              // one wave per FPW frames
    constexpr bool W3_SW = !(L::H == 16 || FPW == 16), W2_SW = FPW != 16;      // which MFMA shape reads each weight image (rb16_woff)
    extern __shared__ __attribute__((aligned(256))) char smem16[];
    char* xe = smem16 + L::off_xe;
    char* xr = smem16 + L::off_xr;
    char* he = smem16 + L::off_he;
    char* w3 = smem16 + L::off_w3;
    char* w2 = smem16 + L::off_w2;
    float* bb = reinterpret_cast<float*>(smem16 + L::off_b);
    float* wtile0 = reinterpret_cast<float*>(smem16 + L::off_wav);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float amax = 0.f, wmax = 0.f;        // largest activation / weight magnitude converted to the split-f16 form
    // Byte offset in xr of the 16-byte chunk `chunk` (0 .. C/4 - 1: hi chunks, then lo, per 32 channels) of frame row r.
    // DOWN: frame r = R o + j sits at row j * (ROWS / R) + o ("planes" of equal tap phase) and its chunks are XORed with
    // (o ^ 5 j) & 7: with it the four non-contiguous 16-lane groups of a ds_read_b128 cover all 64 banks BOTH for 32
    // consecutive frames (shortcut conv) and for 16 frames a stride R apart (one tap of the down conv for 16 output frames);
    // round 2's ((o >> 1) ^ j * (8 / R)) left the tap reads 1.5-way conflicted (tools/lds_banks.py: the bank model and
    // the search that picked this form; the 8-byte stores of an MFMA result are 2-way whatever the swizzle).
    auto xr_chunk_off = [&](int r, int chunk) -> int {
        if constexpr (DOWN > 0) {
            const int o = r / DOWN, j = r % DOWN;
            const int g = (o ^ (5 * j)) & 7;
            return (j * (ROWS / DOWN) + o) * XR::bytes + ((chunk ^ g) * 16);
        } else {
            return r * XR::bytes + ((chunk ^ XR::swz(r)) * 16);
        }
    };
    // xe (elu(x), read by conv3 with a row offset of 0, 1, 2 per tap): C = 32 stores chunk c of row r at c ^ (r & 7); the
    // (r >> 1) & 7 of RbRow is conflict-free only for reads that start at an even row (the taps start at odd ones too:
    // 1.67 LDS cycles per ideal one).  C = 64 keeps RbRow's r & 15.
    auto xe_off = [&](int r, int ci, int lo) -> int {
        if constexpr (C == 32) return r * XR::bytes + (((lo * 4 + (ci % 32) / 8) ^ (r & 7)) * 16);
        else return XR::off(r, ci, lo);
    };
    auto xr_off = [&](int r, int ci, int lo) -> int {              // the 8-half chunk of channels ci .. ci + 7 (cf. RbRow::off)
        constexpr int G = C < 32 ? C : 32;
        return xr_chunk_off(r, (ci / G) * (G / 4) + lo * (G / 8) + (ci % G) / 8);
    };

    // ---- resident split weights.  W3 [H][3][C] -> rows n < H (n >= H: zero padding), k = tap*C + ci;
    //      W2 = [W1 (C x H) | Ws (C x C)] -> rows n < C, k < H from conv1, then the shortcut
    for (int e = tid; e < L::N1 * (L::K1 / 8); e += NT) {
        const int n = e / (L::K1 / 8), k8 = (e - n * (L::K1 / 8)) * 8;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = n < L::H ? a.W3[(long)n * L::K1 + k8 + i] : 0.f;
        f16x8 hi, lo;
        rb16_split8(v, hi, lo, wmax);
        *reinterpret_cast<f16x8*>(w3 + rb16_woff(L::N1, k8 / 16, 0, n, (k8 / 8) & 1, W3_SW)) = hi;
        *reinterpret_cast<f16x8*>(w3 + rb16_woff(L::N1, k8 / 16, 1, n, (k8 / 8) & 1, W3_SW)) = lo;
    }
    for (int e = tid; e < C * (L::K2 / 8); e += NT) {
        const int n = e / (L::K2 / 8), k8 = (e - n * (L::K2 / 8)) * 8;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            v[i] = k8 < L::H ? a.W1[(long)n * L::H + k8 + i] : a.Ws[(long)n * C + (k8 - L::H) + i];
        f16x8 hi, lo;
        rb16_split8(v, hi, lo, wmax);
        *reinterpret_cast<f16x8*>(w2 + rb16_woff(C, k8 / 16, 0, n, (k8 / 8) & 1, W2_SW)) = hi;
        *reinterpret_cast<f16x8*>(w2 + rb16_woff(C, k8 / 16, 1, n, (k8 / 8) & 1, W2_SW)) = lo;
    }
    for (int e = tid; e < L::N1 + C; e += NT)
        bb[e] = e < L::N1 ? (e < L::H ? a.b3[e] : 0.f) : (a.b1[e - L::N1] + a.bs[e - L::N1]);

    const long xbs = a.x_bstride ? a.x_bstride : (long)a.T * C;
    const int Tdown = DOWN ? (a.T + DOWN - 1) / (DOWN ? DOWN : 1) : 0;   // conv.py:54-61: the last window is completed by extra (reflected) padding
    const int tiles_per_clip = DOWN ? (Tdown + OPT - 1) / (OPT ? OPT : 1) : (a.T + VALID - 1) / VALID;
    // DOWN: a window that would run past the clip is shifted left to end at it, so that the frames the last outputs take
    // their reflected taps from are inside it (host: T >= 1024 > VALID)
    auto tile_t0 = [&](int ti) {
        if (!DOWN) return ti * VALID;
        const int nominal = ti * OPT * DOWN - DOWN / 2;
        return nominal + VALID > a.T ? a.T - VALID : nominal;
    };
    const long n_tiles = (long)a.B * tiles_per_clip;
    const int Tp1 = a.T > 1 ? a.T : 2;               // reflect pad 1 (k=3): conv.py:86-91
    const int Tp3 = a.T > 3 ? a.T : 4;               // reflect pad 3 (k=7)

    // ---- tile fill: item = (x row r in -1 .. ROWS, 8-channel chunk); the global loads of tile i+1 are issued before
    // the contractions of tile i and parked in registers
    constexpr int CPR = C / 8;                                       // chunks per row
    constexpr int ITEMS = (L::NX * CPR + NT - 1) / NT;
    constexpr int WAVN = ROWS + 8;                                   // waveform samples per tile (k=7 halo + k=3 halo)
    f32x4 px[FOLD ? 1 : ITEMS][2];
    float pw = 0.f;
    auto prefetch = [&](int b, int ti) {
        const int t0 = tile_t0(ti);
        if (FOLD) {
            if (tid < WAVN) {                                        // sample index t0 - 4 + tid, k=7 reflect
                int p = t0 - 4 + tid;
                p = p < 0 ? -p : p;
                p = p >= Tp3 ? 2 * (Tp3 - 1) - p : p;
                pw = (p >= 0 && p < a.T) ? a.wav[(long)b * a.T + p] : 0.f;
            }
        } else {
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int e = tid + it * NT;
                const int r = e / CPR, c8 = (e - r * CPR) * 8;
                int pos = t0 - 1 + r;
                pos = pos < 0 ? -pos : pos;
                pos = pos >= Tp1 ? 2 * (Tp1 - 1) - pos : pos;
                f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
                if (e < L::NX * CPR && pos >= 0 && pos < a.T) {
                    const float* src = a.x + (long)b * xbs + (long)pos * C + c8;
                    v0 = *reinterpret_cast<const f32x4*>(src);
                    v1 = *reinterpret_cast<const f32x4*>(src + 4);
                }
                px[it][0] = v0; px[it][1] = v1;
            }
        }
    };
    // one item into the LDS images: raw split into xr (rows 0 .. ROWS-1 only), elu split into xe
    auto put_item = [&](int r, int c8, const float* v) {
        f16x8 hi, lo;
        if (r >= 1 && r <= ROWS) {
            rb16_split8(v, hi, lo, amax);
            *reinterpret_cast<f16x8*>(xr + xr_off(r - 1, c8, 0)) = hi;
            *reinterpret_cast<f16x8*>(xr + xr_off(r - 1, c8, 1)) = lo;
        }
        float ev[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ev[i] = (DBG && (a.dbg & 16)) ? v[i] : rb16_elu(v[i]);
        float unused = 0.f;                  // |elu(v)| <= |v|, which the raw split above has covered (or, for the halo rows, a neighbouring tile's)
        rb16_split8(ev, hi, lo, unused);
        *reinterpret_cast<f16x8*>(xe + xe_off(r, c8, 0)) = hi;
        *reinterpret_cast<f16x8*>(xe + xe_off(r, c8, 1)) = lo;
    };
    if ((long)blockIdx.x < n_tiles) prefetch((int)(blockIdx.x / (unsigned)tiles_per_clip), (int)(blockIdx.x % (unsigned)tiles_per_clip));
    // Folded first conv (seanet.py:117: SConv1d(1 -> 32, k = 7)) as a split-f16 MFMA: x[ch][frame] = sum over k of
    // A[ch][k] B[k][frame] with k = 0 .. 6 the taps, k = 7 the bias against a constant 1, k = 8 .. 15 zero.  This lane's A
    // fragment (channel fl, k half fh) stays in registers.
    f16x8 e0h = {0, 0, 0, 0, 0, 0, 0, 0}, e0l = e0h;
    if (FOLD) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if ((lane >> 5) == 0) {
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = a.e0_w[j * C + (lane & 31)];
            v[7] = a.e0_b[lane & 31];
        }
        rb16_split8(v, e0h, e0l, wmax);
        for (int e = tid; e < 2 * XR::bytes / 16; e += NT)         // xe rows ROWS, ROWS + 1: zero, never written again
            *reinterpret_cast<f32x4*>(xe + ROWS * XR::bytes + e * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // FOLD: the shortcut conv (1x1, seanet.py:62) of a block whose input IS the first conv's output is linear in the same seven
    // samples: shortcut(x)[n][f] = sum_k (Ws . E0)[n][k] w[f + k - 3] + (Ws . b0)[n].  Its 32 x 8 matrix is this lane's second A
    // fragment (formed in double, split like e0), and the block's output accumulators START from that product instead of
    // reading a raw copy of x back from LDS: the raw split of x, its two 8-byte stores per 4 channels and the shortcut's two K
    // steps (fragment reads + 6 MFMAs) are gone (r03: stage 1 is vector-issue bound, 27 % of its LDS cycles conflicts from
    // exactly those stores)
    f16x8 esh = {0, 0, 0, 0, 0, 0, 0, 0}, esl = esh;
    if (FOLD) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if ((lane >> 5) == 0) {
            double acc[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
            for (int c = 0; c < C; ++c) {
                const double wsc = (double)a.Ws[(long)(lane & 31) * C + c];
#pragma unroll
                for (int j = 0; j < 7; ++j) acc[j] += wsc * (double)a.e0_w[j * C + c];
                acc[7] += wsc * (double)a.e0_b[c];
            }
            // ... and the block's output bias b1 + bs rides on the same constant-1 column: the output epilogue adds none
            acc[7] += (double)a.b1[lane & 31] + (double)a.bs[lane & 31];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)acc[j];
        }
        rb16_split8(v, esh, esl, wmax);
    }
    // the reflect-padded waveform (conv.py:79-96) at padded position p of clip b; zero outside (clips shorter than the pad)
    auto wav_pad = [&](int b, int p) -> float {
        p = p < 0 ? -p : p;
        p = p >= Tp3 ? 2 * (Tp3 - 1) - p : p;
        return (p >= 0 && p < a.T) ? a.wav[(long)b * a.T + p] : 0.f;
    };
    // DOWN: this wave's down-conv weights, W[16 wave + n16][tap][8 q .. 8 q + 7] as (hi, lo) fragments per tap
    f16x8 wdh[DOWN ? DK : 1], wdl[DOWN ? DK : 1];
    float bd4[4] = {0.f, 0.f, 0.f, 0.f};
    if (DOWN) {
        const int n16 = lane & 15, q = lane >> 4;
#pragma unroll
        for (int j = 0; j < DK; ++j) {
            const float* src = a.Wd + ((long)(16 * wave + n16) * DK + j) * C + 8 * q;
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(src), w1 = *reinterpret_cast<const f32x4*>(src + 4);
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = w0[i]; v[4 + i] = w1[i]; }
            rb16_split8(v, wdh[DOWN ? j : 0], wdl[DOWN ? j : 0], wmax);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) bd4[i] = a.bd[16 * wave + 4 * q + i];
    }
    // The weights' range is reported NOW: left to the end of the kernel, hipcc kept every raw fp32 weight register (about 80 of
    // them with the down conv's) alive across the whole tile loop just to take their maximum there (256 VGPRs + 16 spilled;
    // the asm makes the maximum opaque, so it has to be formed here)
    asm volatile("" : "+v"(wmax));
    range_report(a.status, wmax);

    const int fl = lane & 31, fh = lane >> 5;        // MFMA lane: (row or column fl, k half fh)
    const int row0 = wave * FPW;                     // this wave's frames inside the tile
    const int dbg = DBG ? a.dbg : 0;                 // timing experiments only (WT_RB16_DBG): 1 no tile fill, 2 no MFMA, 4 no store, 8 no down-conv taps, 16 no ELU
    constexpr float LO_SCALE = 1.f / 2048.f;

    // C = 32: the conv3 / conv1 weight fragments of a lane do not depend on the tile: held in registers for the whole tile
    // loop (3 taps x (hi, lo) + conv1's pair = 32 VGPRs) instead of eight ds_read_b128 per wave and tile
    constexpr bool WREG = C == 32 && L::H == 16 && FPW == 32;
    f16x8 w3rh[WREG ? 3 : 1], w3rl[WREG ? 3 : 1], w1rh = {0, 0, 0, 0, 0, 0, 0, 0}, w1rl = w1rh;
    if constexpr (WREG) {
        __syncthreads();                                 // the weight images are complete
        const int m16 = lane & 15, q = lane >> 4;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            w3rh[tap] = *reinterpret_cast<const f16x8*>(w3 + rb16_woff(L::N1, 2 * tap + (q >> 1), 0, m16, q & 1, W3_SW));
            w3rl[tap] = *reinterpret_cast<const f16x8*>(w3 + rb16_woff(L::N1, 2 * tap + (q >> 1), 1, m16, q & 1, W3_SW));
        }
        w1rh = *reinterpret_cast<const f16x8*>(w2 + rb16_woff(C, 0, 0, fl, fh, W2_SW));
        w1rl = *reinterpret_cast<const f16x8*>(w2 + rb16_woff(C, 0, 1, fl, fh, W2_SW));
    }

    int b = (int)(blockIdx.x / (unsigned)tiles_per_clip), ti = (int)(blockIdx.x % (unsigned)tiles_per_clip);
    int wpar = 0;
    for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, wpar ^= 1) {
        const int t0 = tile_t0(ti);
        // FOLD: this tile's waveform window goes into the buffer that the tile before the previous one used (its readers
        // passed the barrier after that tile's fill long ago), so ONE barrier covers "previous tile fully consumed", "weights
        // landed" and "window visible": 424 -> 404 us (a barrier costs this kernel about 20 us).  Tried on top: no top barrier
        // at all for DOWN (x_raw / staged y alternating between two LDS buffers, the next window written before the barrier in
        // front of the down conv): 420 us — the waves drift apart and wait longer at the two barriers that remain.
        float* wtile = wtile0 + wpar * (ROWS + 8);
        if (FOLD && !(dbg & 1) && tid < ROWS + 8) wtile[tid] = pw;
        __syncthreads();
        if (dbg & 1) { __syncthreads(); } else
        if (FOLD) {
            // first encoder conv from the staged samples: xe row r = frame t0 - 1 + r = b + sum_j w[j] * wavpad[frame + j - 3],
            // wtile[i] = wavpad[t0 - 4 + i].  Wave w produces rows 32 w .. 32 w + 31: B fragment = the lane's 7 samples + 1.
            const bool direct = (t0 - 4 >= 0) && (t0 + ROWS + 4 <= a.T);      // no reflected sample or frame in the window
            const int r = row0 + fl;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (direct) {
#pragma unroll
                for (int j = 0; j < 7; ++j) v[j] = wtile[r + j];
                v[7] = 1.f;
            } else {
                int pos = t0 - 1 + r;                // frame of this x row, k=3 reflect
                pos = pos < 0 ? -pos : pos;
                pos = pos >= Tp1 ? 2 * (Tp1 - 1) - pos : pos;
                if (pos >= 0 && pos < a.T) {
#pragma unroll
                    for (int j = 0; j < 7; ++j) {    // a reflected frame near a clip edge may need samples outside the window (rare)
                        const int wi = pos + j - 3 - (t0 - 4);
                        v[j] = (wi >= 0 && wi < WAVN) ? wtile[wi] : wav_pad(b, pos + j - 3);
                    }
                    v[7] = 1.f;
                }
            }
            f16x8 bh, bl;
            rb16_split8(v, bh, bl, amax);
            // (the fh = 1 lanes hold the same samples against k = 8 .. 15, where A is zero)
            f32x16 xm, xc;
#pragma unroll
            for (int i = 0; i < 16; ++i) { xm[i] = 0.f; xc[i] = 0.f; }
            xm = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0h, bh, xm, 0, 0, 0);
            xc = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0l, bh, xc, 0, 0, 0);
            xc = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0h, bl, xc, 0, 0, 0);
            // lane: row r, channels 8 g + 4 fh .. + 3: elu split -> xe row r (no raw copy: the shortcut is folded into esh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 8 * g + 4 * fh;
                f32x4 x4;
#pragma unroll
                for (int i = 0; i < 4; ++i) x4[i] = xm[4 * g + i] + xc[4 * g + i] * LO_SCALE;
                const f32x4 e4 = (DBG && (a.dbg & 16)) ? x4 : elu_med3_x4(x4);
                f16x4 hi, lo;
                // (no raw copy of x: the shortcut is folded into the first conv, see esh; elu(x) > -1, so its split tracks
                // every magnitude that could leave the f16 range)
                rb16_split4(e4, hi, lo, amax);
                *reinterpret_cast<f16x4*>(xe + xe_off(r, n & ~7, 0) + (n & 7) * 2) = hi;
                *reinterpret_cast<f16x4*>(xe + xe_off(r, n & ~7, 1) + (n & 7) * 2) = lo;
            }
        } else {
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int e = tid + it * NT;
                if (e < L::NX * CPR) {
                    const int r = e / CPR, c8 = (e - r * CPR) * 8;
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { v[i] = px[it][0][i]; v[4 + i] = px[it][1][i]; }
                    put_item(r, c8, v);
                }
            }
        }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) {
            int nb = b, nti = ti + (int)gridDim.x;
            while (nti >= tiles_per_clip) { nti -= tiles_per_clip; ++nb; }
            prefetch(nb, nti);
        }

        if constexpr (FPW == 16) {
            // ---- the 64-channel block, 16 frames per wave, on v_mfma_f32_16x16x32_f16: lane (m16, q) = (frame, k quarter);
            // a K step is 32 channels of one operand row; D: frame m16, output rows 16 mt + 4 q .. + 3
            static_assert(L::H == 32 && L::N1 == 32, "two 16-row tiles of hidden channels");
            const int m16 = lane & 15, q = lane >> 4;
            const int frame = row0 + m16;
            f32x4 hm[2], hc[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) { hm[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; hc[mt] = hm[mt]; }
            if (!(dbg & 2))
#pragma unroll
            for (int st = 0; st < 6; ++st) {                        // conv3: K = 3 taps x 64 channels, step = (tap, channel half)
                const int tap = st >> 1, cb = (st & 1) * 32;
                const f16x8 bh = *reinterpret_cast<const f16x8*>(xe + xe_off(frame + tap, cb + 8 * q, 0));
                const f16x8 bl = *reinterpret_cast<const f16x8*>(xe + xe_off(frame + tap, cb + 8 * q, 1));
                f16x8 wh[2], wl[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    wh[mt] = *reinterpret_cast<const f16x8*>(w3 + rb16_woff(L::N1, 2 * st + (q >> 1), 0, 16 * mt + m16, q & 1, W3_SW));
                    wl[mt] = *reinterpret_cast<const f16x8*>(w3 + rb16_woff(L::N1, 2 * st + (q >> 1), 1, 16 * mt + m16, q & 1, W3_SW));
                }
                hm[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[0], bh, hm[0], 0, 0, 0);
                hc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[0], bh, hc[0], 0, 0, 0);
                hm[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[1], bh, hm[1], 0, 0, 0);
                hc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[1], bh, hc[1], 0, 0, 0);
                hc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[0], bl, hc[0], 0, 0, 0);
                hc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[1], bl, hc[1], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {                        // elu(h + b3), split, into this wave's own rows of he
                const int n = 16 * mt + 4 * q;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bb + n);
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = hm[mt][i] + hc[mt][i] * LO_SCALE + b4[i];
                if (!(dbg & 16)) v = elu_med3_x4(v);
                f16x4 hi, lo;
                rb16_split4(v, hi, lo, amax);
                *reinterpret_cast<f16x4*>(he + HR::off(frame, n & ~7, 0) + (n & 7) * 2) = hi;
                *reinterpret_cast<f16x4*>(he + HR::off(frame, n & ~7, 1) + (n & 7) * 2) = lo;
            }
            // y = [W1 | Ws] . [elu(h) | x]: K = 32 + 64 = three steps; 64 output channels = four row tiles
            f32x4 ym[4], yc[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) { ym[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; yc[mt] = ym[mt]; }
            if (!(dbg & 2))
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                f16x8 bh, bl;
                if (st == 0) {
                    bh = *reinterpret_cast<const f16x8*>(he + HR::off(frame, 8 * q, 0));
                    bl = *reinterpret_cast<const f16x8*>(he + HR::off(frame, 8 * q, 1));
                } else {
                    bh = *reinterpret_cast<const f16x8*>(xr + xr_off(frame, (st - 1) * 32 + 8 * q, 0));
                    bl = *reinterpret_cast<const f16x8*>(xr + xr_off(frame, (st - 1) * 32 + 8 * q, 1));
                }
                f16x8 wh[4], wl[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    wh[mt] = *reinterpret_cast<const f16x8*>(w2 + rb16_woff(C, 2 * st + (q >> 1), 0, 16 * mt + m16, q & 1, W2_SW));
                    wl[mt] = *reinterpret_cast<const f16x8*>(w2 + rb16_woff(C, 2 * st + (q >> 1), 1, 16 * mt + m16, q & 1, W2_SW));
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) ym[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[mt], bh, ym[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) yc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[mt], bh, yc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) yc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[mt], bl, yc[mt], 0, 0, 0);
            }
            // stage the wave's 16 frames in its own xr rows (the shortcut was their last reader), then full-line stores
            char* stg = xr + row0 * XR::bytes;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int n = 16 * mt + 4 * q;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bb + L::N1 + n);
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ym[mt][i] + yc[mt][i] * LO_SCALE + b4[i];
                if (a.elu_out && !(dbg & 16)) v = elu_med3_x4(v);
                if (a.out_s32) {
                    f16x4 hi, lo;
                    rb16_split4(v, hi, lo, amax);
                    const int ch = (n >> 5) * 8 + ((n & 31) >> 3);
                    *reinterpret_cast<f16x4*>(stg + m16 * XR::bytes + ((ch ^ XR::swz(m16)) * 16) + (n & 7) * 2) = hi;
                    *reinterpret_cast<f16x4*>(stg + m16 * XR::bytes + (((ch + 4) ^ XR::swz(m16)) * 16) + (n & 7) * 2) = lo;
                } else {
                    *reinterpret_cast<f32x4*>(stg + m16 * XR::bytes + (((n >> 2) ^ XR::swz(m16)) * 16)) = v;
                }
            }
            constexpr int LPR = XR::chunks, RPI = 64 / LPR;         // 16 lanes per 256-byte row, 4 rows per store instruction
            const long tbase = (long)b * a.T + t0 + row0;
#pragma unroll
            for (int it = 0; it < FPW / RPI; ++it) {
                const int r = it * RPI + lane / LPR, ch = lane % LPR;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * XR::bytes + ((ch ^ XR::swz(r)) * 16));
                if (t0 + row0 + r < a.T && !(dbg & 4)) *reinterpret_cast<f32x4*>(a.y + (tbase + r) * C + ch * 4) = v;
            }
        } else {
        // ---- conv3 (transposed): h[n][frame] = sum over (tap, ci) W3[n][tap][ci] * elu(x)[frame + tap - 1][ci]
        if constexpr (L::H == 16) {
            // C = 32: the 16 hidden channels are exactly one v_mfma_f32_16x16x32_f16 row tile and a K step is one tap's 32 input
            // channels (no zero-padded rows as with 32x32x16): lane (m16, q) holds A = W3[m16][tap][8 q ..], B = elu(x) of frame
            // 16 nt + m16 + tap, channels 8 q .. 8 q + 7; D: frame m16, hidden channels 4 q .. 4 q + 3
            static_assert(C == 32, "one K step per tap");
            const int m16 = lane & 15, q = lane >> 4;
            f32x4 hm[2], hc[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) { hm[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; hc[nt] = hm[nt]; }
            if (!(dbg & 2))
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const f16x8 wh = WREG ? w3rh[WREG ? tap : 0] : *reinterpret_cast<const f16x8*>(w3 + rb16_woff(L::N1, 2 * tap + (q >> 1), 0, m16, q & 1, W3_SW));
                const f16x8 wl = WREG ? w3rl[WREG ? tap : 0] : *reinterpret_cast<const f16x8*>(w3 + rb16_woff(L::N1, 2 * tap + (q >> 1), 1, m16, q & 1, W3_SW));
                f16x8 bh[2], bl[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int xrow = row0 + 16 * nt + m16 + tap;
                    bh[nt] = *reinterpret_cast<const f16x8*>(xe + xe_off(xrow, 8 * q, 0));
                    bl[nt] = *reinterpret_cast<const f16x8*>(xe + xe_off(xrow, 8 * q, 1));
                }
                // issue order: no MFMA directly behind the one it depends on
                hm[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh[0], hm[0], 0, 0, 0);
                hc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh[0], hc[0], 0, 0, 0);
                hm[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh[1], hm[1], 0, 0, 0);
                hc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh[1], hc[1], 0, 0, 0);
                hc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl[0], hc[0], 0, 0, 0);
                hc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl[1], hc[1], 0, 0, 0);
            }
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bb + 4 * q);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = hm[nt][i] + hc[nt][i] * LO_SCALE + b4[i];
                if (!(dbg & 16)) v = elu_med3_x4(v);
                f16x4 hi, lo;
                rb16_split4(v, hi, lo, amax);
                const int n = 4 * q, frame = row0 + 16 * nt + m16;
                *reinterpret_cast<f16x4*>(he + HR::off(frame, n & ~7, 0) + (n & 7) * 2) = hi;
                *reinterpret_cast<f16x4*>(he + HR::off(frame, n & ~7, 1) + (n & 7) * 2) = lo;
            }
        } else {
            constexpr int TN1 = L::N1 / 32;
            f32x16 a1m[TN1], a1c[TN1];
    #pragma unroll
            for (int j = 0; j < TN1; ++j)
    #pragma unroll
                for (int r = 0; r < 16; ++r) { a1m[j][r] = 0.f; a1c[j][r] = 0.f; }
            if (!(dbg & 2))
    #pragma unroll
            for (int ks = 0; ks < L::K1 / 16; ++ks) {
                const int tap = (ks * 16) / C, ci = (ks * 16) % C + 8 * fh;
                const int xrow = row0 + fl + tap;                    // xe row index = frame + 1 + (tap - 1)
                const f16x8 bh = *reinterpret_cast<const f16x8*>(xe + xe_off(xrow, ci, 0));
                const f16x8 bl = *reinterpret_cast<const f16x8*>(xe + xe_off(xrow, ci, 1));
    #pragma unroll
                for (int j = 0; j < TN1; ++j) {
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(w3 + rb16_woff(L::N1, ks, 0, j * 32 + fl, fh, W3_SW));
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(w3 + rb16_woff(L::N1, ks, 1, j * 32 + fl, fh, W3_SW));
                    a1m[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, a1m[j], 0, 0, 0);
                    a1c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, a1c[j], 0, 0, 0);
                    a1c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, a1c[j], 0, 0, 0);
                }
            }
            // elu(h + b3), split, into this wave's own rows of he.  D layout: lane -> frame fl, register r -> row
            // n = 32 j + (r & 3) + 8 (r >> 2) + 4 fh
    #pragma unroll
            for (int j = 0; j < TN1; ++j)
    #pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = j * 32 + 8 * g + 4 * fh;
                    if (n < L::H) {
                        f32x4 v;
    #pragma unroll
                        for (int i = 0; i < 4; ++i)
                            v[i] = a1m[j][4 * g + i] + a1c[j][4 * g + i] * LO_SCALE + bb[n + i];
                        if (!(dbg & 16)) v = elu_med3_x4(v);
                        f16x4 hi, lo;
                        rb16_split4(v, hi, lo, amax);
                        *reinterpret_cast<f16x4*>(he + HR::off(row0 + fl, n & ~7, 0) + (n & 7) * 2) = hi;
                        *reinterpret_cast<f16x4*>(he + HR::off(row0 + fl, n & ~7, 1) + (n & 7) * 2) = lo;
                    }
                }
        }
        // no barrier: a wave reads back only its own he rows, and a wave's LDS operations execute in order

        // ---- y[n][frame] = [W1 | Ws][n] . [elu(h) | x][frame] + (b1 + bs)
        constexpr int TN2 = C / 32;
        f32x16 a2m[TN2], a2c[TN2];
#pragma unroll
        for (int j = 0; j < TN2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { a2m[j][r] = 0.f; a2c[j][r] = 0.f; }
        if constexpr (FOLD != 0) {
            // shortcut(x) of output frame row0 + fl (= x row r + 1) straight from the tile's waveform window: B fragment = the
            // frame's seven samples + 1 against (Ws . E0 | Ws . b0)
            const int r = row0 + fl, f = t0 + r;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // wtile[i] = the reflect-padded waveform at padded position t0 - 4 + i, so frame f's tap j is wtile[r + 1 + j]
            // (r + 7 <= ROWS + 6 < WAVN); an output frame is never a reflected one: rows beyond the clip are dropped by the store /
            // never read by the down conv, and stay zero here
            if (f >= 0 && f < a.T) {
#pragma unroll
                for (int j = 0; j < 7; ++j) v[j] = wtile[r + 1 + j];
                v[7] = 1.f;
            }
            f16x8 sh, sl;
            rb16_split8(v, sh, sl, amax);
            if (!(dbg & 2)) {
                a2m[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(esh, sh, a2m[0], 0, 0, 0);
                a2c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(esl, sh, a2c[0], 0, 0, 0);
                a2c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(esh, sl, a2c[0], 0, 0, 0);
            }
        }
        if (!(dbg & 2))
#pragma unroll
        for (int ks = 0; ks < (FOLD ? L::H / 16 : L::K2 / 16); ++ks) {
            f16x8 bh, bl;
            if (ks < L::H / 16) {
                bh = *reinterpret_cast<const f16x8*>(he + HR::off(row0 + fl, ks * 16 + 8 * fh, 0));
                bl = *reinterpret_cast<const f16x8*>(he + HR::off(row0 + fl, ks * 16 + 8 * fh, 1));
            } else {
                const int ci = ks * 16 - L::H + 8 * fh;
                bh = *reinterpret_cast<const f16x8*>(xr + xr_off(row0 + fl, ci, 0));
                bl = *reinterpret_cast<const f16x8*>(xr + xr_off(row0 + fl, ci, 1));
            }
#pragma unroll
            for (int j = 0; j < TN2; ++j) {
                const bool reg = WREG && ks == 0 && j == 0;           // (compile-time after unrolling)
                const f16x8 wh = reg ? w1rh : *reinterpret_cast<const f16x8*>(w2 + rb16_woff(C, ks, 0, j * 32 + fl, fh, W2_SW));
                const f16x8 wl = reg ? w1rl : *reinterpret_cast<const f16x8*>(w2 + rb16_woff(C, ks, 1, j * 32 + fl, fh, W2_SW));
                a2m[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, a2m[j], 0, 0, 0);
                a2c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, a2c[j], 0, 0, 0);
                a2c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, a2c[j], 0, 0, 0);
            }
        }
        // The wave's 32 output frames are one contiguous block of y.  Each lane holds 4-channel runs of ONE frame, so
        // the tile is staged in the wave's own xr rows (free now: the shortcut was their last reader, and they are
        // private to the wave) and written out with every lane storing 16 bytes of a full 128-byte line.
        {
            char* st = xr + row0 * XR::bytes;                       // 32 rows x C*4 bytes, chunk-swizzled like xr (DOWN: the wave's rows of the plane layout)
#pragma unroll
            for (int j = 0; j < TN2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = j * 32 + 8 * g + 4 * fh;
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = a2m[j][4 * g + i] + a2c[j][4 * g + i] * LO_SCALE;
                        if constexpr (FOLD == 0) v[i] += bb[L::N1 + n + i];       // (FOLD: b1 + bs came in through the folded shortcut's bias column)
                    }
                    if ((DOWN || a.elu_out) && !(dbg & 16)) v = elu_med3_x4(v);
                    if (DOWN || a.out_s32) {                        // S32: chunk (n/32)*8 + n%32/8 holds hi, + 4 holds lo
                        f16x4 hi, lo;
                        rb16_split4(v, hi, lo, amax);
                        const int ch = (n >> 5) * 8 + ((n & 31) >> 3);
                        *reinterpret_cast<f16x4*>(xr + xr_chunk_off(row0 + fl, ch) + (n & 7) * 2) = hi;
                        *reinterpret_cast<f16x4*>(xr + xr_chunk_off(row0 + fl, ch + 4) + (n & 7) * 2) = lo;
                    } else {                                        // fp32: chunk n/4
                        *reinterpret_cast<f32x4*>(st + fl * XR::bytes + (((n >> 2) ^ XR::swz(fl)) * 16)) = v;
                    }
                }
            if constexpr (DOWN == 0) {
            constexpr int LPR = XR::chunks;                         // lanes per row: 8 (C = 32) or 16 (C = 64)
            constexpr int RPI = 64 / LPR;                           // rows per store instruction
            const long tbase = (long)b * a.T + t0 + row0;
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int r = it * RPI + lane / LPR, ch = lane % LPR;
                const f32x4 v = *reinterpret_cast<const f32x4*>(st + r * XR::bytes + ((ch ^ XR::swz(r)) * 16));
                if (row0 + r < VALID && t0 + row0 + r < a.T && !(dbg & 4)) *reinterpret_cast<f32x4*>(a.y + (tbase + r) * C + ch * 4) = v;
            }
            }
        }
        if constexpr (DOWN > 0) {
            // ---- down conv on the tile: out[n][o] = bd[n] + sum over (tap j, ci) Wd[n][j][ci] * elu(y)[o * r - r/2 + j][ci], the
            // S32(elu(y)) image of all 128 frames sitting in xr.  Reflect padding (conv.py:79-96; T % r == 0 and T > 2r: host)
            // maps the few positions beyond a clip edge back into this tile's window.
            __syncthreads();                                        // every wave's rows of y are staged
            const int m16 = lane & 15, q = lane >> 4;
            // a tile whose taps all lie inside the clip reads frame o * r + j for output o, tap j: with the plane layout
            // that is a per-lane base (two of them: j < r, j >= r) XOR a constant, plus an immediate
            const int last_o = (ti * OPT + OPT <= Tdown ? ti * OPT + OPT : Tdown) - 1;
            const bool inner = ti > 0 && t0 == ti * OPT * DOWN - DOWN / 2 && last_o * DOWN - DOWN / 2 + DK - 1 < a.T;
#pragma unroll 1                                                    // (unrolled, hipcc hoists every tap's fragment reads: registers)
            for (int mt = 0; mt < DMT; ++mt) {
                const int o = 16 * mt + m16;
                const int t_out = ti * OPT + o;
                const bool valid = o < OPT && t_out < Tdown;
                f32x4 dm = {0.f, 0.f, 0.f, 0.f}, dc = dm, dc2 = dm;     // two correction chains: no MFMA directly behind its producer
                if (dbg & 8) {
                } else if (inner) {
#pragma unroll
                    for (int j = 0; j < DK; ++j) {
                        const char* src = xr + mt * (16 * XR::bytes) + (j % DOWN) * (ROWS / DOWN) * XR::bytes;
                        const int base = (m16 + j / DOWN) * XR::bytes + ((q ^ ((m16 + j / DOWN) & 7)) * 16);      // o & 7 = (m16 + j / R) & 7
                        const f16x8 yh = *reinterpret_cast<const f16x8*>(src + (base ^ (((5 * (j % DOWN)) & 7) * 16)));
                        const f16x8 yl = *reinterpret_cast<const f16x8*>(src + (base ^ (((5 * (j % DOWN)) & 7) * 16) ^ 64));
                        dm = __builtin_amdgcn_mfma_f32_16x16x32_f16(wdh[DOWN ? j : 0], yh, dm, 0, 0, 0);
                        dc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wdl[DOWN ? j : 0], yh, dc, 0, 0, 0);
                        dc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wdh[DOWN ? j : 0], yl, dc2, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < DK; ++j) {
                        int pos = t_out * DOWN - DOWN / 2 + j;
                        pos = pos < 0 ? -pos : pos;
                        pos = pos >= a.T ? 2 * (a.T - 1) - pos : pos;
                        int row = valid ? pos - t0 : 0;
                        row = row < 0 ? 0 : (row > ROWS - 1 ? ROWS - 1 : row);
                        const f16x8 yh = *reinterpret_cast<const f16x8*>(xr + xr_off(row, 8 * q, 0));
                        const f16x8 yl = *reinterpret_cast<const f16x8*>(xr + xr_off(row, 8 * q, 1));
                        dm = __builtin_amdgcn_mfma_f32_16x16x32_f16(wdh[DOWN ? j : 0], yh, dm, 0, 0, 0);
                        dc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wdl[DOWN ? j : 0], yh, dc, 0, 0, 0);
                        dc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wdh[DOWN ? j : 0], yl, dc2, 0, 0, 0);
                    }
                }
                if (valid && !(dbg & 4)) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = dm[i] + (dc[i] + dc2[i]) * LO_SCALE + bd4[i];
                    *reinterpret_cast<f32x4*>(a.y_down + ((long)b * Tdown + t_out) * 64 + 16 * wave + 4 * q) = v;
                }
            }
        }
        }   // FPW == 32
        ti += (int)gridDim.x;
        while (ti >= tiles_per_clip) { ti -= tiles_per_clip; ++b; }
    }
    range_report(a.status, amax);
}

template <int C, int ROWS, int FOLD, int FPW = 32>
static int launch_rb16(const ResblockArgs& a, hipStream_t s) {
    static PerDeviceOnce attr_once;
    constexpr size_t smem = (size_t)Rb16Layout<C, ROWS>::total;
    static_assert(smem <= 160 * 1024, "LDS budget");
    auto kern = resblock16_kernel<C, ROWS, FOLD, false, 0, FPW>;
    int dbg_req = 0;
#ifdef WT_LAB       // the phase-ablation instantiations (tools/rb16_bench.py, WT_RB16_DBG) exist in LAB builds only
    if (const char* e = lab_env("WT_RB16_DBG")) dbg_req = atoi(e);
    if (dbg_req) kern = resblock16_kernel<C, ROWS, FOLD, true, 0, FPW>;
#endif
    if (int rc = attr_once.run([&]() -> int {
        WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(resblock16_kernel<C, ROWS, FOLD, false, 0, FPW>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
#ifdef WT_LAB
        WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(resblock16_kernel<C, ROWS, FOLD, true, 0, FPW>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
#endif
        return 0;
    })) return rc;
    constexpr int VALID = FOLD ? ROWS - 2 : ROWS;
    const long tiles = (long)a.B * ((a.T + VALID - 1) / VALID);
    const int per_cu = (int)(160 * 1024 / smem) < 4 ? (int)(160 * 1024 / smem) : 4;
    const long slots = (long)device_cus() * per_cu;
    const long grid = tiles < slots ? tiles : slots;
    ResblockArgs b = a;
    b.dbg = dbg_req;
    if (!b.status) b.status = g_launch.status;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(ROWS / FPW * 64), smem, s, b);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

// stage 1 with the down conv fused in: wav [B][T] -> y_down [B][ceil(T / r)][64] fp32 (r = a.R = 2 or 4)
template <int R>
static int launch_rb16_down(const ResblockArgs& a, hipStream_t s) {
    static PerDeviceOnce attr_once;
    constexpr size_t smem = (size_t)Rb16Layout<32, 128>::total;
    constexpr int OPT = (126 - 2 * R) / R + 1;
    auto kern = resblock16_kernel<32, 128, 1, false, R>;
    int dbg_req = 0;
#ifdef WT_LAB
    if (const char* e = lab_env("WT_RB16_DBG")) dbg_req = atoi(e);
    if (dbg_req) kern = resblock16_kernel<32, 128, 1, true, R>;
#endif
    if (int rc = attr_once.run([&]() -> int {
        WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(resblock16_kernel<32, 128, 1, false, R>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
#ifdef WT_LAB
        WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(resblock16_kernel<32, 128, 1, true, R>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
#endif
        return 0;
    })) return rc;
    const long tiles = (long)a.B * (((a.T + R - 1) / R + OPT - 1) / OPT);
    const int per_cu = (int)(160 * 1024 / smem) < 4 ? (int)(160 * 1024 / smem) : 4;
    const long slots = (long)device_cus() * per_cu;
    const long grid = tiles < slots ? tiles : slots;
    ResblockArgs b = a;
    b.dbg = dbg_req;
    if (!b.status) b.status = g_launch.status;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), smem, s, b);
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

bool resblock16_down_fusable(int C, long T, int r, int k) { return C == 32 && (r == 2 || r == 4) && k == 2 * r && T >= 1024; }

int launch_resblock16_down(const ResblockArgs& a, hipStream_t s) {
    if (!a.wav || !a.Wd || !a.bd || !a.y_down || !resblock16_down_fusable(a.C, a.T, a.R, 2 * a.R)) {
        set_error("resblock16_down: needs the waveform, C = 32, stride 2 or 4 with k = 2 * stride, T >= 1024"); return -1;
    }
    return a.R == 4 ? launch_rb16_down<4>(a, s) : launch_rb16_down<2>(a, s);
}

int launch_resblock16(const ResblockArgs& a, hipStream_t s) {
    if (a.wav && a.C != 32) { set_error("resblock16: the folded first conv needs C == 32"); return -1; }
    if (!a.wav && ((reinterpret_cast<uintptr_t>(a.x) & 15) || (a.x_bstride % 4))) {
        set_error("resblock16: x must be 16-byte aligned"); return -1;
    }
    if (a.C == 32) return a.wav ? launch_rb16<32, 128, 1>(a, s) : launch_rb16<32, 128, 0>(a, s);
    if (a.C == 64) {
        // A/B timing: WT_RB16_FPW=32 is the 4-wave form.  (Tried: 48-row tiles of 3 waves, 75 KB, i.e. two independent
        // workgroups per CU: 252 us against 223 for the 8-wave form and 268 for the 4-wave form, kernel alone, one box.)
#ifdef WT_LAB
        if (const char* e = lab_env("WT_RB16_FPW")) if (atoi(e) == 32) return launch_rb16<64, 128, 0, 32>(a, s);
#endif
        return launch_rb16<64, 128, 0, 16>(a, s);
    }
    set_error("resblock16: fused kernel exists for C = 32 and 64");
    return -1;
}

}  // namespace wt
