// SLSTM (encoder/modules/lstm.py:12-39) as ONE persistent launch: the recurrence of all L steps of both layers.
//
// The step-per-launch kernel (ops.hip) spends 3.2-3.5 of its 8 us per step in the dependent launch itself.  A
// persistent kernel needs a barrier between steps instead, and a device-wide one is no cheaper (the eight XCDs' L2s
// are not coherent: agent-scope release/acquire costs 18 us per round, tools/micro/xcd_barrier.hip).  So the work is
// cut so that no data ever crosses an XCD:  XCD x owns clips [x*Bx, (x+1)*Bx) — clips are independent — and runs the
// whole network for them on its 32 CUs; workgroup w of the XCD owns hidden units [16w, 16w+16) of both layers, keeps
// their recurrent weights resident in registers for the entire sequence (three wave roles x four 16-row gate tiles =
// 12 waves: W_hh_l0, W_ih_l1, W_hh_l1, as split-f16 hi/lo operands of v_mfma_f32_16x16x32_f16), and the 32
// workgroups exchange the new hidden state once per step through the XCD's own L2: plain stores (write-through),
// loads that skip the CU's L1 (sc1), and an arrival counter bumped with an atomic that is performed in that L2.
// Measured round trip of such a barrier: 2.7 us.
//
// Placement: the hardware dispatches workgroup i to XCD i % 8 (measured 256/256, 32 per XCD); the kernel does not
// rely on the order, it reads HW_REG_XCC_ID and takes a ticket per XCD, and it never spins unboundedly: a round that
// does not complete (an XCD with fewer than 32 resident workgroups) sets `err` in the control block and every
// workgroup leaves.
#include "common.h"

#include <stdlib.h>

namespace wt {

typedef float f32x4p __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8p __attribute__((ext_vector_type(8)));

static constexpr int PW = 12;                      // waves per workgroup: 3 roles x 4 gate tiles
static constexpr int HROW = 2 * 2048;              // staged state row: [layer 0 | layer 1], each 512 k in S32 (2048 B)
static constexpr int HPITCH = HROW;                // LDS row pitch: no padding, the 16-byte chunks of a row are XOR-swizzled instead
// Chunk swizzle of the staged state.  Every product wave reads the whole state tile once per step (12 waves x 16 KB), which
// paces the product phase, and a ds_read_b128 is served in four NON-contiguous 16-lane groups (lanes 0-3, 12-15, 20-27 ...):
// with the padded pitch of round 2 each group hit every bank twice (36 % of the LDS cycles were conflicts).  Chunk c of clip
// row r is stored at c ^ lp_swz(r): SMALL (rows 0-7; a group reads hi chunks of four rows and lo chunks of the other four)
// 2 r, else (16 rows, one chunk per group) r: every group then covers all 64 banks (tools/lds_banks.py)
template <bool SMALL> __device__ __forceinline__ int lp_swz(int row) { return (SMALL ? 2 * row : row) & 15; }
static constexpr long SPIN_LIMIT_DF = 1L << 20;    // polls of the state itself (about a microsecond each)
// lo-weight blocks kept in LDS instead of registers, and staged clip rows.  SMALL (at most 8 clips per XCD, e.g. B = 64): the
// state tile needs 8 rows and one 4-register A operand per block (hi and lo rows packed, see the product phase), which leaves
// room for a double-buffered operand; 7 blocks in LDS measured the same as 8 or 9
template <bool SMALL> struct LpCfg { static constexpr int NLDS = SMALL ? 8 : 6, HROWS = SMALL ? 8 : 16, POLL = SMALL ? 3 : 1; };

__device__ __forceinline__ float sigm_p(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_p(float x) {
    const float ax = fabsf(x);
    const float e = __expf(-2.f * ax);
    const float t = ax < 0.04f ? ax * (1.f - ax * ax * (1.f / 3.f)) : (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);
    return copysignf(t, x);
}

// a 32-bit word holds two f16 halves; nonzero iff one of them is the "not written yet" mark 0xFFFF
__device__ __forceinline__ unsigned has_mark(unsigned d) { return (~d - 0x00010001u) & d & 0x80008000u; }
// a state half that would collide with the mark (a NaN with an all-ones payload) is stored as the canonical NaN
__device__ __forceinline__ _Float16 no_mark(_Float16 v) {
    const unsigned short b = __builtin_bit_cast(unsigned short, v);
    return __builtin_bit_cast(_Float16, (unsigned short)(b == 0xFFFFu ? 0x7E00u : b));
}

// The state exchange carries its own readiness.  Three buffers per XCD; every (layer, clip, unit) half is
// either the mark 0xFFFF or data: the step-s reader polls the DATA of step s-1 (buffer (s-1) % 3) until no mark is left,
// a writer stores step s into buffer s % 3 and, one step ahead of need, re-marks its own slice of buffer (s+1) % 3 (last
// read during step s-1, which every workgroup has finished once this one holds all of step s-1; the re-mark is acknowledged
// by L2 before the step's data is stored, so whoever sees that data can never see the older contents again).  Per step
// that is ONE L2 round trip on each side (store; load) instead of store -> acknowledge -> counter atomic -> poll -> load.
// phase timestamps of workgroup 0 of XCD 0, steps 64..71 (WT_LSTM_TRACE=1, tools/lstm_trace.py): 100 MHz ticks into ctl[520..]
#define LP_TRACE(ph) do { if (TRACE && tr_on && s >= 64 && s < 72) { \
    const unsigned long long tt = __builtin_amdgcn_s_memrealtime(); \
    if (lane == 0) { a.ctl[520 + ((s - 64) * 6 + (ph)) * 2] = (unsigned)tt; a.ctl[521 + ((s - 64) * 6 + (ph)) * 2] = (unsigned)(tt >> 32); } } } while (0)
// TRACE: the phase-timestamp build (WT_LSTM_TRACE=1); the shipped instantiations carry no run-time test for it
template <bool SMALL, bool TRACE = false>
__global__ __launch_bounds__(64 * PW) void lstm_persist_kernel(const LstmPersistArgs a) {
    constexpr int NBUF = 3;
    constexpr int NLDS = LpCfg<SMALL>::NLDS, HROWS = LpCfg<SMALL>::HROWS;
    extern __shared__ __attribute__((aligned(16))) char sm_p[];
    char* hst = sm_p;                                                   // [HROWS clips][HPITCH]
    float* gbuf = reinterpret_cast<float*>(sm_p + HROWS * HPITCH);      // [3 roles][4 tiles][16 clips][17]
    // the register file holds 16 hi + 13 lo weight blocks per lane at three waves per SIMD; the last NLDS lo blocks live here
    f32x4p* wlds = reinterpret_cast<f32x4p*>(sm_p + HROWS * HPITCH + 3 * 4 * 16 * 17 * sizeof(float)) + (threadIdx.x >> 6) * NLDS * 64 + (threadIdx.x & 63);
    __shared__ unsigned s_xcc, s_w, s_stop;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int role = wave >> 2, ntile = wave & 3;
    const int li = lane & 15, lk = lane >> 4;
    if (tid == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7;          // HW_REG_XCC_ID[3:0]
        s_xcc = xcc;
        // the control block starts as all-ones like the state buffers (one memset), so the first ticket is ~0u + 1
        s_w = __hip_atomic_fetch_add(a.ctl + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
        s_stop = 0;
    }
    __syncthreads();
    const int xcc = (int)s_xcc, w = (int)s_w;
    if (w >= 32) return;                                 // more than 32 workgroups on this XCD: the extra ones have no role
    const int H = 512, B = a.B, L = a.L;
    const int c0g = xcc * a.Bx;
    const int nb = B - c0g < a.Bx ? B - c0g : a.Bx;      // clips of this XCD
    if (nb <= 0) return;                                 // uniform over the XCD's workgroups
    unsigned* err = a.ctl + 512;
    char* hx = reinterpret_cast<char*>(a.hx) + (size_t)xcc * (NBUF * 2 * 16 * 2048);    // [buffer][layer][clip][2048 B]
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(hx, 0, NBUF * 2 * 16 * 2048, 0x00020000);

    // ---- resident weights: [role][wg][tile][blk 16][hi, lo][lane][8 halves]
    f16x8p wh[16], wl[16 - NLDS];
    {
        const f32x4p* wp = reinterpret_cast<const f32x4p*>(a.Wp) + ((((size_t)role * 32 + w) * 4 + ntile) * 16 * 2) * 64 + lane;
#pragma unroll
        for (int blk = 0; blk < 16; ++blk) {
            wh[blk] = __builtin_bit_cast(f16x8p, wp[(blk * 2 + 0) * 64]);
            if (blk < 16 - NLDS) wl[blk] = __builtin_bit_cast(f16x8p, wp[(blk * 2 + 1) * 64]);
            else wlds[(blk - (16 - NLDS)) * 64] = wp[(blk * 2 + 1) * 64];
        }
    }
    // ---- cell threads: (layer, clip, unit) fixed for the whole sequence, cell state in a register
    const int c_layer = tid >> 8, c_clip = (tid >> 4) & 15, c_unit = tid & 15;
    const bool c_thr = tid < 512 && c_clip < nb;
    const int c_nt = c_unit >> 2, c_u4 = c_unit & 3;
    const int j = 16 * w + c_unit;                       // hidden unit
    const long gclip = c0g + c_clip;
    float cst = 0.f;

    for (int e = tid; e < HROWS * HPITCH / 16; e += 64 * PW) reinterpret_cast<f32x4p*>(hst)[e] = (f32x4p){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    // the spin bounds; a test hook (LstmPersistArgs::dbg_spin_shift) shortens them so that a forced loss of co-residency
    // is reported within milliseconds
    const long spin_limit_df = SPIN_LIMIT_DF >> a.dbg_spin_shift;
    const bool tr_on = TRACE && xcc == 0 && w == 0 && __builtin_amdgcn_readfirstlane(wave) == 0;
    for (int s = 0; s <= L; ++s) {
        LP_TRACE(0);
        // layer-0 input projection of this step (independent of the recurrence): requested before the wait
        float xg[4] = {0.f, 0.f, 0.f, 0.f};
        float xs = 0.f;
        if (c_thr && c_layer == 0 && s < L) {
#pragma unroll
            for (int g = 0; g < 4; ++g) xg[g] = a.xg0[((long)s * B + gclip) * (4 * H) + 64 * w + 16 * c_nt + g * 4 + c_u4];
        }
        if (c_thr && c_layer == 1 && s >= 1) {
            xs = a.x[(gclip * L + (s - 1)) * H + j];
#pragma unroll
            for (int g = 0; g < 4; ++g) xg[g] = a.b1[64 * w + 16 * c_nt + g * 4 + c_u4];     // layer 1: its bias (an L1 hit; no registers held across steps)
        }

        {
            // 1+2. poll the state of step s-1 itself (L2 loads that skip the CU's L1) until every half is data, then
            //      stage it in LDS; h0[-1] = h1[-1] = 0 are not loaded (layer 1 runs one step behind: its first
            //      state appears in buffer 1)
            const int rb = (s + 2) % 3, tot = 2 * nb * 128;
            // up to 16 clips: 4096 chunks = 6 per thread, polled POLL at a time (what the resident weights leave free)
            constexpr int POLL = LpCfg<SMALL>::POLL;
            for (int e0 = tid; e0 < tot; e0 += POLL * 64 * PW) {
                f32x4p v[POLL];
                unsigned need = 0;
#pragma unroll
                for (int it = 0; it < POLL; ++it) {
                    const int e = e0 + it * 64 * PW;
                    if (e < tot && s >= 1 + (e >> 7) / nb) need |= 1u << it;
                }
                unsigned pend = need;
                long spin = 0;
                while (pend) {
#pragma unroll
                    for (int it = 0; it < POLL; ++it)
                        if (pend >> it & 1) {
                            const int e = e0 + it * 64 * PW;
                            const int c16 = e & 127, lc = e >> 7, layer = lc / nb, clip = lc - layer * nb;
                            v[it] = __builtin_bit_cast(f32x4p, __builtin_amdgcn_raw_buffer_load_b128(
                                        rsH, ((rb * 2 + layer) * 16 + clip) * 2048 + c16 * 16, 0, 16 /* sc1 */));
                        }
                    unsigned still = 0;
#pragma unroll
                    for (int it = 0; it < POLL; ++it)
                        if (pend >> it & 1) {
                            const unsigned m = has_mark(__builtin_bit_cast(unsigned, v[it].x)) | has_mark(__builtin_bit_cast(unsigned, v[it].y)) |
                                               has_mark(__builtin_bit_cast(unsigned, v[it].z)) | has_mark(__builtin_bit_cast(unsigned, v[it].w));
                            if (m) still |= 1u << it;
                        }
                    pend = still;
                    if (pend) {
                        if (++spin > spin_limit_df || ((spin & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u)) {
                            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (a.status) __hip_atomic_fetch_or(a.status, (unsigned)WT_STATUS_LSTM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            s_stop = 1;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
#pragma unroll
                for (int it = 0; it < POLL; ++it) {
                    const int e = e0 + it * 64 * PW;
                    if (e < tot) {
                        const int c16 = e & 127, lc = e >> 7, layer = lc / nb, clip = lc - layer * nb;
                        *reinterpret_cast<f32x4p*>(hst + clip * HPITCH + layer * 2048 + ((c16 ^ lp_swz<SMALL>(clip)) * 16)) =
                            (need >> it & 1) ? v[it] : (f32x4p){0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            LP_TRACE(1);
            __syncthreads();
            LP_TRACE(2);
            if (s_stop) return;
            // this workgroup now holds all of step s-1, so every workgroup is past its reads of step s-2: re-mark the
            // own slice of the buffer that held it (it is written again at step s+1)
            if (c_thr) {
                unsigned short* mrow = reinterpret_cast<unsigned short*>(hx + ((((s + 1) % 3) * 2 + c_layer) * 16 + c_clip) * 2048);
                mrow[(j >> 5) * 64 + (j & 31)] = 0xFFFFu;
                mrow[(j >> 5) * 64 + 32 + (j & 31)] = 0xFFFFu;
            }
        }
        // 3. recurrent products: role 0: W_hh_l0 . h0[s-1] (layer 0, t = s); role 1: W_ih_l1 . h0[s-1]; role 2:
        //    W_hh_l1 . h1[s-2] (layer 1, t = s-1)
        {
            const bool act = role == 0 ? (s < L) : (s >= 1);
            if (act && SMALL) {
                // At most 8 clips: the 16-row A operand holds the hi halves of the 8 clip rows in rows 0-7 and their lo
                // halves in rows 8-15, so ONE MFMA against W_hi yields the main term (rows 0-7) and the lo.W_hi correction
                // (rows 8-15); a second one against W_lo yields the hi.W_lo correction in rows 0-7 (its rows 8-15 are the
                // lo.lo term nobody needs): two MFMAs per block instead of three, one 16-byte LDS read per lane per block.
                // chunk (blk * 8 + 4 * lo + lk) ^ swizzle: the swizzle only touches the low four bits, blk * 8 toggles bit 3 and up
                // (the swizzle touches the low four bits of the chunk index and blk * 8 toggles bit 3 and up, so the sixteen reads are
                // two per-lane bases, even and odd blocks, plus immediates)
                const int ach = (lk + ((li & 8) ? 4 : 0)) ^ lp_swz<SMALL>(li & 7);
                const char* ap0 = hst + (li & 7) * HPITCH + (role == 2 ? 2048 : 0) + (ach & 7) * 16;
                const char* ape = ap0 + (ach >> 3) * 128;           // blocks 0, 2, 4, ...: + (blk >> 1) * 256
                const char* apo = ap0 + (1 ^ (ach >> 3)) * 128;     // blocks 1, 3, 5, ...
                const bool arow = (li & 7) < nb;                   // absent clip rows stay zero and are not read
                f32x4p accm = {0.f, 0.f, 0.f, 0.f}, accc = accm;
                f16x8p a1b[2] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
                if (arow) a1b[0] = *reinterpret_cast<const f16x8p*>(ape);
#pragma unroll
                for (int blk = 0; blk < 16; ++blk) {
                    if (blk + 1 < 16 && arow) a1b[(blk + 1) & 1] = *reinterpret_cast<const f16x8p*>((((blk + 1) & 1) ? apo : ape) + ((blk + 1) >> 1) * 256);
                    const f16x8p a1 = a1b[blk & 1];
                    const f16x8p wlb = blk < 16 - NLDS ? wl[blk < 16 - NLDS ? blk : 0]
                                                       : __builtin_bit_cast(f16x8p, wlds[(blk - (16 - NLDS)) * 64]);
                    accm = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, wh[blk], accm, 0, 0, 0);
                    accc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, wlb, accc, 0, 0, 0);
                }
                // D: col = lane & 15 (gate row), row = 4 (lane >> 4) + reg: clip r's main term sits in lanes lk < 2, its
                // lo.W_hi correction 32 lanes further up
                float* gb = gbuf + ((role * 4 + ntile) * 16) * 17;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float c2 = __shfl_down(accm[r], 32, 64);
                    if (lk < 2) gb[(4 * lk + r) * 17 + li] = accm[r] + (accc[r] + c2) * (1.f / 2048.f);
                }
            } else if (act) {
                // hi chunk lk, lo chunk lk + 4 of every 8-chunk block, swizzled: the lo chunk is the hi chunk's address with bit 6
                // flipped, an odd block's is an even block's with bit 7 flipped (rows are 4096-byte aligned)
                const int ch_h = lk ^ lp_swz<SMALL>(li);
                const char* ap0 = hst + li * HPITCH + (role == 2 ? 2048 : 0) + (ch_h & 7) * 16;
                const char* ape = ap0 + (ch_h >> 3) * 128;
                const char* apo = ap0 + (1 ^ (ch_h >> 3)) * 128;
                const int lo_d = (ch_h & 4) ? -64 : 64;             // (chunk ^ 4) - chunk, in bytes
                f32x4p accm = {0.f, 0.f, 0.f, 0.f}, accc = accm;
                // the state reads of the twelve waves pace this phase: the next block's operand is requested before this
                // block's MFMAs are issued
                f16x8p ahb[2], alb[2];
                ahb[0] = *reinterpret_cast<const f16x8p*>(ape);
                alb[0] = *reinterpret_cast<const f16x8p*>(ape + lo_d);
#pragma unroll
                for (int blk = 0; blk < 16; ++blk) {
                    if (blk + 1 < 16) {
                        const char* pb = (((blk + 1) & 1) ? apo : ape) + ((blk + 1) >> 1) * 256;
                        ahb[(blk + 1) & 1] = *reinterpret_cast<const f16x8p*>(pb);
                        alb[(blk + 1) & 1] = *reinterpret_cast<const f16x8p*>(pb + lo_d);
                    }
                    const f16x8p ah = ahb[blk & 1], al = alb[blk & 1];
                    const f16x8p wlb = blk < 16 - NLDS ? wl[blk < 16 - NLDS ? blk : 0]
                                                       : __builtin_bit_cast(f16x8p, wlds[(blk - (16 - NLDS)) * 64]);
                    accm = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[blk], accm, 0, 0, 0);
                    accc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wlb, accc, 0, 0, 0);
                    accc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[blk], accc, 0, 0, 0);
                }
                // D: col = lane & 15 (gate row), row = 4 (lane >> 4) + reg (clip)
                float* gb = gbuf + ((role * 4 + ntile) * 16) * 17;
#pragma unroll
                for (int r = 0; r < 4; ++r) gb[(4 * lk + r) * 17 + li] = accm[r] + accc[r] * (1.f / 2048.f);
            }
        }
        LP_TRACE(3);
        __syncthreads();
        LP_TRACE(4);
        // 4. cell update; the new state goes to the exchange buffer of parity s (S32 rows) and, for layer 1, to y
        if (c_thr && (c_layer == 0 ? (s < L) : (s >= 1))) {
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = g * 4 + c_u4;
                if (c_layer == 0) pre[g] = gbuf[((0 * 4 + c_nt) * 16 + c_clip) * 17 + col] + xg[g];
                else pre[g] = (gbuf[((1 * 4 + c_nt) * 16 + c_clip) * 17 + col] + gbuf[((2 * 4 + c_nt) * 16 + c_clip) * 17 + col]) + xg[g];
            }
            const float ig = sigm_p(pre[0]), fg = sigm_p(pre[1]), gg = tanh_p(pre[2]), og = sigm_p(pre[3]);
            cst = fg * cst + ig * gg;
            const float h = og * tanh_p(cst);
            _Float16* hrow = reinterpret_cast<_Float16*>(hx + (((s % 3) * 2 + c_layer) * 16 + c_clip) * 2048);
            const _Float16 hh = (_Float16)h;
            const _Float16 hl = (_Float16)((h - (float)hh) * 2048.f);
            __builtin_amdgcn_s_waitcnt(0);                      // the re-mark of this step is in L2
            hrow[(j >> 5) * 64 + (j & 31)] = no_mark(hh);
            hrow[(j >> 5) * 64 + 32 + (j & 31)] = no_mark(hl);
            if (c_layer == 1) {
                const int t = s - 1;
                const float yv = h + xs;                                // lstm.py:37-38 skip
                const float o = a.elu_out ? (yv > 0.f ? yv : __expf(yv) - 1.f) : yv;
                float* yrow = a.y + (gclip * L + t) * H;
                if (a.out_s32) {
                    if (fabsf(o) >= 65504.f && a.status)       // beyond the f16 range of the split form (common.h: WT_STATUS_RANGE)
                        __hip_atomic_fetch_or(a.status, (unsigned)WT_STATUS_RANGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    _Float16* yp = reinterpret_cast<_Float16*>(yrow) + ((j >> 5) * 64 + (j & 31));
                    const _Float16 oh = (_Float16)o;
                    yp[0] = oh;
                    yp[32] = (_Float16)((o - (float)oh) * 2048.f);
                } else {
                    yrow[j] = o;
                }
            }
        }
        LP_TRACE(5);
    }
}

size_t lstm_persist_hx_bytes() { return (size_t)8 * 3 * 2 * 16 * 2048; }
size_t lstm_persist_ctl_bytes() { return 1024 * sizeof(unsigned); }

int launch_lstm_persist(const LstmPersistArgs& a, hipStream_t stream) {
    if (a.H != 512) { set_error("lstm_persist: built for hidden size 512"); return -1; }
    if (a.B < 1 || a.Bx < 1 || a.Bx > 16 || 8 * a.Bx < a.B) { set_error("lstm_persist: at most 16 clips per XCD (B <= 128)"); return -1; }
    static PerDeviceOnce attr_once;
    constexpr size_t smem_big = (size_t)16 * HPITCH + (size_t)3 * 4 * 16 * 17 * sizeof(float) + (size_t)PW * LpCfg<false>::NLDS * 1024;
    constexpr size_t smem_small = (size_t)8 * HPITCH + (size_t)3 * 4 * 16 * 17 * sizeof(float) + (size_t)PW * LpCfg<true>::NLDS * 1024;
    static_assert(smem_big + 64 <= 160 * 1024 && smem_small + 64 <= 160 * 1024, "LDS budget");
    if (int rc = attr_once.run([&]() -> int {
        WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_persist_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_big));
        WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_persist_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_small));
#ifdef WT_LAB
        WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_persist_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_big));
        WT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_persist_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_small));
#endif
        return 0;
    })) return rc;
    // the caller has filled hx and ctl with 0xFF bytes (the marks of the data-flag exchange)
    const bool small = a.Bx <= 8;
    LstmPersistArgs b = a;
    if (!b.status) b.status = g_launch.status;
    // Fault injection, LAB builds only (tests/lab/test_lab_faults.py, run on tools/lib/libwavtok_hip_lab.so): WT_LSTM_PERSIST_FAULT=1
    // launches 8 workgroups too few, so every XCD waits for a 32nd workgroup that never comes, and shortens the spin bounds;
    // the kernel must then report WT_STATUS_LSTM instead of handing out a half-written sequence.  The product library has
    // neither the hook nor the phase-timestamp instantiations (tools/lstm_trace.py)
    const char* fault = lab_env("WT_LSTM_PERSIST_FAULT");
    const bool forced = fault && fault[0] == '1';
    if (forced) b.dbg_spin_shift = 8;
    const dim3 grid(forced ? 248 : 256), block(64 * PW);
#ifdef WT_LAB
    if (a.data_flag & 4) {          // phase timestamps (tools/lstm_trace.py)
        if (small) hipLaunchKernelGGL((lstm_persist_kernel<true, true>), grid, block, smem_small, stream, b);
        else hipLaunchKernelGGL((lstm_persist_kernel<false, true>), grid, block, smem_big, stream, b);
    } else
#endif
    {
        if (small) hipLaunchKernelGGL((lstm_persist_kernel<true>), grid, block, smem_small, stream, b);
        else hipLaunchKernelGGL((lstm_persist_kernel<false>), grid, block, smem_big, stream, b);
    }
    WT_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace wt
