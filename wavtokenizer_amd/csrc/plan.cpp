// Launch plans: per (kind, B, len) a flat list of kernel launches over a lifetime-packed workspace.  No compute
// happens on the host at call time.
#include "model.h"

namespace wt {

// The dense chain runs on S32 operands (the shipped path) unless the unfused debug plan or fp32 GEMMs are asked for, or
// the model's weights do not fit the split-f16 range (wt_model::s32_ok)
static bool site_fp32(const wt_plan* P, int site) { return (P->fp32_sites >> site) & 1u; }
static bool plan_fp32(const wt_plan* P) {
    if ((P->flags & WT_PLAN_FLAG_FP32_GEMM) || !P->model->s32_ok) return true;
    // plans that are one range site as a whole (model.h Site); the decode plan decides site by site (build_decode)
    if (P->kind == WT_PLAN_SEANET_DECODER) return !P->model->sd_s32_ok || site_fp32(P, SITE_SEADEC);
    if (P->kind == WT_PLAN_ENCODE || P->kind == WT_PLAN_UNIT_LSTM) return site_fp32(P, SITE_ENC);
    if (P->kind == WT_PLAN_HEAD) return site_fp32(P, SITE_HEAD);
    return false;
}
static bool plan_unfused(const wt_plan* P) { return P->flags & WT_PLAN_FLAG_UNFUSED; }
static bool plan_s32(const wt_plan* P) { return !plan_unfused(P) && !plan_fp32(P); }

// Everything the S32 chain does not cover (the fp32 plans: WT_PLAN_FLAG_FP32_GEMM, the unfused debug twin, a model whose
// weights do not fit the split-f16 range) runs on the fp32 MFMA chain of gemm.hip
static int gemm_auto(const wt_plan* P, const GemmArgs& a, int pro, int epi, hipStream_t s) {
    // safety net for a model that came from a packed image (its fp32 GEMM weights are rebuilt on demand: wt_plan_create does
    // it for the plans it can tell will need them; any other fp32 GEMM finds them here, in the plan's first, eager call)
    if (P->model->f32_stale.load()) if (int rc = ensure_f32_weights(P->model)) return rc;
    return launch_gemm(a, pro, epi, s);
}

// Both operands pre-split (S32): the activations were written in S32 by their producer, the weight has an S32 copy
static int gemm_s32(const wt_plan* P, const GemmArgs& a, int epi, int out, hipStream_t s) {
    auto it = P->model->s32.find(a.W);
    if (it == P->model->s32.end()) { set_error("internal: no S32 copy of this weight"); return WT_ERR_INVALID; }
    GemmArgs b = a;
    b.W_hi = it->second;
    b.tap_pair = P->model->s32_tap_pair.count(a.W) ? 1 : 0;
    auto sc = P->model->s32_acc_scale.find(a.W);
    if (sc != P->model->s32_acc_scale.end()) b.acc_scale = sc->second;
    return launch_gemm16s(b, epi, out, s);
}

// SConv1d geometry (encoder/modules/conv.py:195-211, 54-61), non-causal.

SConvGeom sconv_geom(long T, int k, int stride, int dil) {
    const int keff = (k - 1) * dil + 1;
    const int pt = keff - stride;
    const long nfr_num = T - keff + pt;                 // n_frames = nfr_num/stride + 1
    const long nfr = (nfr_num + stride - 1) / stride + 1;   // ceil (nfr_num >= 0 here since pt = keff - stride)
    const long ideal = (nfr - 1) * stride + (keff - pt);
    const int extra = (int)(ideal - T);
    SConvGeom g;
    const int pr = pt / 2;
    g.pl = pt - pr;
    g.pr_total = pr + extra;
    g.Tout = (int)((T + pt + extra - keff) / stride + 1);
    const int maxpad = std::max(g.pl, g.pr_total);
    g.Tp = T > maxpad ? (int)T : maxpad + 1;
    return g;
}

// x [B][T][cin] (time-major) -> y [B][Tout][cout]; reflect-padded SConv1d as one implicit GEMM
GemmArgs sconv_args(const ConvW& w, int B, long T, int stride, int dil) {
    const SConvGeom g = sconv_geom(T, w.k, stride, dil);
    GemmArgs a;
    a.a_bstride = T * w.cin; a.a_rstride = w.cin;
    a.T_in = (int)T; a.T_out = g.Tout; a.Cin = w.cin; a.taps = w.k; a.stride = stride; a.dil = dil;
    a.pad_left = g.pl; a.pad_mode = PAD_REFLECT; a.Tp = g.Tp;
    a.W = w.w; a.w_rstride = (long)w.k * w.cin; a.bias = w.b;
    a.M = B * g.Tout; a.N = w.cout; a.K = w.k * w.cin;
    a.c_rstride = w.cout;
    return a;
}
// zero-padded 'same' Conv1d (decoder/models.py:29-43,177): k odd, padding (k-1)/2
GemmArgs zconv_args(const ConvW& w, int B, int L) {
    GemmArgs a;
    a.a_bstride = (long)L * w.cin; a.a_rstride = w.cin;
    a.T_in = L; a.T_out = L; a.Cin = w.cin; a.taps = w.k; a.pad_left = (w.k - 1) / 2; a.pad_mode = PAD_ZERO;
    a.W = w.w; a.w_rstride = (long)w.k * w.cin; a.bias = w.b;
    a.M = B * L; a.N = w.cout; a.K = w.k * w.cin; a.c_rstride = w.cout;
    return a;
}
// plain X[M][K] . W[N][K]^T
GemmArgs linear_args(const float* W, const float* bias, long M, int N, int K) {
    GemmArgs a;
    a.a_bstride = 0; a.a_rstride = K; a.T_in = (int)M; a.T_out = (int)M; a.Cin = K; a.taps = 1;
    a.W = W; a.w_rstride = K; a.bias = bias; a.M = (int)M; a.N = N; a.K = K; a.c_rstride = N;
    return a;
}

// SEANetResnetBlock (seanet.py:62-63): y = shortcut(x) + conv1(elu(conv3(elu(x)))); returns y's buffer
static int plan_resblock(wt_plan* P, const ConvW& c3, const ConvW& c1, const ConvW& sc, int B, long T, int xin,
                         const std::string& name, bool elu_out = false, const wt_model* e0 = nullptr, long x_off = 0,
                         long x_bstride = 0, bool out_s32 = false) {
    const int C = sc.cout;
    if (resblock_fusable(C) && !plan_unfused(P)) {
        // one fused kernel (resblock.hip); with e0 set, xin is unused and the tile is built from the waveform
        const int y = P->buf(name, (size_t)B * T * C, (out_s32 && !plan_fp32(P) ? BUF_S32 : BUF_F32) | (elu_out ? BUF_ELU : 0));
        P->step({e0 ? -1 : xin, y}, [=](const RunCtx& c) {
            ResblockArgs a{};
            a.x = e0 ? nullptr : P->ptr(c, xin) + x_off;
            a.x_bstride = x_bstride;
            a.wav = e0 ? c.in_f : nullptr;
            a.e0_w = e0 ? e0->e0_w : nullptr; a.e0_b = e0 ? e0->e0_b : nullptr;
            a.W3 = c3.w; a.b3 = c3.b; a.W1 = c1.w; a.b1 = c1.b; a.Ws = sc.w; a.bs = sc.b;
            a.y = P->ptr(c, y); a.B = B; a.T = (int)T; a.C = C; a.elu_out = elu_out ? 1 : 0;
            if (plan_fp32(P)) return launch_resblock(a, c.stream);
            a.out_s32 = out_s32 ? 1 : 0;
            return launch_resblock16(a, c.stream);
        }, 1, "resblock.fused");
        return y;
    }
    const int h = P->buf(name + ".h", (size_t)B * T * (C / 2));
    const int y = P->buf(name, (size_t)B * T * C, elu_out ? BUF_ELU : BUF_F32);
    GemmArgs a3 = sconv_args(c3, B, T, 1, 1);
    P->step({xin, h}, [=](const RunCtx& c) {
        GemmArgs a = a3; a.A = P->ptr(c, xin) + x_off; a.C = P->ptr(c, h);
        if (x_bstride) a.a_bstride = x_bstride;
        return gemm_auto(P, a, PRO_ELU, EPI_BIAS, c.stream);
    });
    GemmArgs as = sconv_args(sc, B, T, 1, 1);
    P->step({xin, y}, [=](const RunCtx& c) {
        GemmArgs a = as; a.A = P->ptr(c, xin) + x_off; a.C = P->ptr(c, y);
        if (x_bstride) a.a_bstride = x_bstride;
        return gemm_auto(P, a, PRO_NONE, EPI_BIAS, c.stream);
    });
    GemmArgs a1 = sconv_args(c1, B, T, 1, 1);
    P->step({h, y}, [=](const RunCtx& c) {
        GemmArgs a = a1; a.A = P->ptr(c, h); a.C = P->ptr(c, y); a.R = P->ptr(c, y); a.r_rstride = C;
        return gemm_auto(P, a, PRO_ELU, elu_out ? EPI_BIAS_RES_ELU : EPI_BIAS_RES, c.stream);
    });
    return y;
}

// SLSTM (lstm.py:31-39) on x [B][L][H]; returns y = lstm(x) + x.  xin_s32 >= 0: an S32 copy of x for the input
// projection (split-f16 GEMM); y_s32: write y in S32 (its only consumer is a split-f16 conv).
static int plan_lstm(wt_plan* P, const LstmW& w, int B, int L, int H, int xin, const std::string& name,
                     bool elu_out = false, int xin_s32 = -1, bool y_s32 = false) {
    const int xg = P->buf(name + ".xg", (size_t)B * L * 4 * H);
    const int Bp = (B + 63) / 64 * 64;                            // clip pitch of the K-major hidden state
    const size_t st_numel = (size_t)4 * H * Bp + (size_t)2 * B * H;
    const int st = P->buf(name + ".state", st_numel);            // h0[2][H][Bp], h1[2][H][Bp], c0[B][H], c1[B][H]
    const int y = P->buf(name, (size_t)B * L * H, (y_s32 ? BUF_S32 : BUF_F32) | (elu_out ? BUF_ELU : 0));
    // input projection written time-major ([L][B][4H]) so each recurrent step reads one contiguous
    // slab: the gather treats a time step as the "clip" (stride H) and the clip as the row (stride L*H)
    GemmArgs ax = linear_args(w.Wih0, w.b0, (long)B * L, 4 * H, H);
    ax.T_in = B; ax.T_out = B; ax.a_bstride = H; ax.a_rstride = (long)L * H;
    const int xsrc = xin_s32 >= 0 ? xin_s32 : xin;
    P->step({xsrc, xg}, [=](const RunCtx& c) {
        GemmArgs a = ax; a.A = P->ptr(c, xsrc); a.C = P->ptr(c, xg);
        if (xin_s32 >= 0) return gemm_s32(P, a, EPI_BIAS, OUT_F32, c.stream);
        return gemm_auto(P, a, PRO_NONE, EPI_BIAS, c.stream);
    });
    // one persistent launch for the whole recurrence (lstm_persist.hip) when the batch fits its per-XCD clip groups and
    // the device is a full MI355X (256 CUs: one resident workgroup per CU, 32 per XCD)
    const bool persist_env = [] { const char* e = lab_env("WT_LSTM_PERSIST"); return !e || e[0] != '0'; }();       // (LAB builds; callers use WT_PLAN_FLAG_STEP_LSTM)
    bool persist = persist_env && !plan_fp32(P) && !(P->flags & WT_PLAN_FLAG_STEP_LSTM) && w.Wp && H == 512 && B <= 128 && L < 65536;
    if (persist) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, P->model->device) != hipSuccess || cus != 256) persist = false;
    }
    if (persist) P->uses_persist = true;
    const size_t hxn = lstm_persist_hx_bytes() / sizeof(float), ctn = lstm_persist_ctl_bytes() / sizeof(float);
    const int hx = persist ? P->buf(name + ".hx", hxn + ctn) : -1;
    P->step({xin, xg, st, hx, y}, [=](const RunCtx& c) {
        if (persist && P->model->persist_ok.load()) {
            float* hb = P->ptr(c, hx);
            if (int rc = launch_fill_u32(hb, 0xFFFFFFFFu, (hxn + ctn) * sizeof(float), c.stream)) return rc;
            LstmPersistArgs pa;
            const int df_trace = [] { const char* e = lab_env("WT_LSTM_TRACE"); return e ? atoi(e) : 0; }();
            pa.data_flag = 1 | (df_trace ? 4 : 0);
            pa.xg0 = P->ptr(c, xg); pa.Wp = w.Wp; pa.b1 = w.b1; pa.x = P->ptr(c, xin); pa.y = P->ptr(c, y);
            pa.hx = hb; pa.ctl = reinterpret_cast<unsigned*>(hb + hxn);
            pa.B = B; pa.L = L; pa.H = H; pa.Bx = (B + 7) / 8; pa.elu_out = elu_out ? 1 : 0; pa.out_s32 = y_s32 ? 1 : 0;
            return launch_lstm_persist(pa, c.stream);
        }
        float* s = P->ptr(c, st);
        if (int rc = launch_fill_u32(s, 0u, (st_numel * sizeof(float) + 15) / 16 * 16, c.stream)) return rc;
        LstmArgs la;
        la.f16x3 = plan_fp32(P) ? 0 : 1;     // recurrent product on split-f16 MFMAs unless fp32 is forced
        la.xg0 = P->ptr(c, xg); la.W0 = la.f16x3 ? w.W0h : w.W0; la.W1 = la.f16x3 ? w.W1h : w.W1; la.b1 = w.b1;
        la.h0 = s; la.h1 = s + (size_t)2 * H * Bp; la.c0 = s + (size_t)4 * H * Bp; la.c1 = la.c0 + (size_t)B * H;
        la.x = P->ptr(c, xin); la.y = P->ptr(c, y); la.B = B; la.L = L; la.H = H; la.elu_out = elu_out ? 1 : 0;
        la.out_s32 = y_s32 ? 1 : 0;
        for (int t = 0; t <= L; ++t)
            if (int rc = launch_lstm_step(la, t, c.stream)) return rc;
        return 0;
    }, L + 2);
    return y;
}

// Unfused SEANetResnetBlock with every operand pre-split: x arrives as S32(x) (shortcut) and S32(elu(x)) (conv3),
// the hidden activation and the output are written as S32(elu(.)); returns the output buffer
static int plan_resblock_s32(wt_plan* P, const ConvW& c3, const ConvW& c1, const ConvW& sc, int B, long T, int x_raw,
                             int x_elu, const std::string& name, long x_off = 0, long x_bstride = 0, const ConvW* cat = nullptr) {
    const int C = sc.cout;
    const int h = P->buf(name + ".h", (size_t)B * T * (C / 2), BUF_S32 | BUF_ELU);
    GemmArgs a3 = sconv_args(c3, B, T, 1, 1);
    P->step({x_elu, h}, [=](const RunCtx& c) {
        GemmArgs a = a3; a.A = P->ptr(c, x_elu) + x_off; a.C = P->ptr(c, h);
        if (x_bstride) a.a_bstride = x_bstride;
        return gemm_s32(P, a, EPI_BIAS_ELU, OUT_S32, c.stream);
    });
    const bool cat_env = [] { const char* e = lab_env("WT_RESBLOCK_CAT"); return !e || e[0] != '0'; }();      // A/B timing (LAB builds)
    if (cat && cat->w && cat_env && P->model->s32.count(cat->w)) {
        // shortcut + conv1 as one GEMM over K = [x (C) | elu(h) (C/2)] (GemmArgs::A2): the fp32 shortcut tensor is neither
        // written nor read back, and the output goes through the staged full-line epilogue
        const int o = P->buf(name, (size_t)B * T * C, BUF_S32 | BUF_ELU);
        GemmArgs ac = sconv_args(sc, B, T, 1, 1);
        ac.W = cat->w; ac.w_rstride = cat->cin; ac.bias = cat->b; ac.K = cat->cin; ac.Cin = cat->cin;
        ac.K1 = C; ac.a2_bstride = T * (C / 2); ac.a2_rstride = C / 2;
        P->step({x_raw, h, o}, [=](const RunCtx& c) {
            GemmArgs a = ac; a.A = P->ptr(c, x_raw) + x_off; a.A2 = P->ptr(c, h); a.C = P->ptr(c, o);
            if (x_bstride) a.a_bstride = x_bstride;
            return gemm_s32(P, a, EPI_BIAS_ELU, OUT_S32, c.stream);
        });
        return o;
    }
    const int y = P->buf(name + ".sc", (size_t)B * T * C);
    const int o = P->buf(name, (size_t)B * T * C, BUF_S32 | BUF_ELU);
    GemmArgs as = sconv_args(sc, B, T, 1, 1);
    P->step({x_raw, y}, [=](const RunCtx& c) {
        GemmArgs a = as; a.A = P->ptr(c, x_raw) + x_off; a.C = P->ptr(c, y);
        if (x_bstride) a.a_bstride = x_bstride;
        return gemm_s32(P, a, EPI_BIAS, OUT_F32, c.stream);
    });
    GemmArgs a1 = sconv_args(c1, B, T, 1, 1);
    P->step({h, y, o}, [=](const RunCtx& c) {
        GemmArgs a = a1; a.A = P->ptr(c, h); a.C = P->ptr(c, o); a.R = P->ptr(c, y); a.r_rstride = C;
        return gemm_s32(P, a, EPI_BIAS_RES_ELU, OUT_S32, c.stream);
    });
    return o;
}

int build_encode(wt_plan* P) {
    const wt_model* M = P->model;
    const int B = P->B;
    const long T = P->T;
    // the first conv is folded into the fused stage-1 resblock unless stage taps are kept
    const bool fold_e0 = !plan_unfused(P) && !M->stages.empty() && M->stages[0].C == 32 &&
                         M->e0_k == 7 && resblock_fusable(32);
    int x = -1;
    if (!fold_e0) {
        x = P->buf("enc.0", (size_t)B * T * M->e0_c);
        const int x0 = x;
        P->step({x0}, [=](const RunCtx& c) {
            return launch_conv_first(c.in_f, M->e0_w, M->e0_b, P->ptr(c, x0), B, T, M->e0_k, M->e0_c, c.stream);
        });
    }
    // ELU is applied once by the producer wherever its only consumer is "ELU -> conv" (resblock
    // output -> down conv, LSTM output -> last conv); the unfused debug plan keeps raw tensors instead
    const bool fuse_elu = !plan_unfused(P);
    long Tc = T;
    int idx = 1;
    // S32 mode (default): from the first fused stage on, every GEMM operand of the encoder is written pre-split by
    // its producer and multiplied by gemm16s.hip; tensors that fp32 kernels read too (fused resblock input, LSTM
    // skip, embeddings) are written in both forms by the producing GEMM
    const bool s32 = plan_s32(P);
    int x_raw = -1, x_elu = -1;          // current stage input as S32(x) and S32(elu(x)) (unfused S32 stages)
    int x_s32 = -1;                      // S32 copy of the last down conv output (LSTM input projection)
    for (size_t si = 0; si < M->stages.size(); ++si) {
        const ResStage& st = M->stages[si];
        const bool fused = resblock_fusable(st.C) && !plan_unfused(P);
        // a fused stage only needs the S32 down-conv weights (its own convs run inside resblock16); an unfused one
        // needs S32 copies of all four
        const bool ws32 = s32 && (st.C % 32 == 0) && M->s32.count(st.down.w) &&
                          (fused || (M->s32.count(st.c3.w) && M->s32.count(st.c1.w) && M->s32.count(st.sc.w)));
        bool x_is_s32;                   // the resblock output (elu'd) is S32
        // stage 1 of the shipped plan: first conv + resblock + ELU + down conv in ONE kernel, the stage's activations never
        // leave LDS (resblock16.hip, DOWN).  WT_RB16_DOWN=0 keeps the two launches (A/B timing).
        const bool down_env = [] { const char* e = lab_env("WT_RB16_DOWN"); return !e || e[0] != '0'; }();
        const bool fuse_down = down_env && fused && idx == 1 && fold_e0 && ws32 &&
                               si + 1 < M->stages.size() && resblock_fusable(M->stages[si + 1].C) && st.down.cin == 32 &&
                               st.down.cout == 64 && resblock16_down_fusable(st.C, Tc, st.r, st.down.k);
        if (fuse_down) {
            const long Td = sconv_args(st.down, B, Tc, st.r, 1).T_out;      // ceil(Tc / r)
            const int y = P->buf("enc." + std::to_string(idx + 2), (size_t)B * Td * st.down.cout);
            P->step({-1, y}, [=](const RunCtx& c) {
                ResblockArgs a{};
                a.wav = c.in_f; a.e0_w = M->e0_w; a.e0_b = M->e0_b;
                a.W3 = st.c3.w; a.b3 = st.c3.b; a.W1 = st.c1.w; a.b1 = st.c1.b; a.Ws = st.sc.w; a.bs = st.sc.b;
                a.Wd = st.down.w; a.bd = st.down.b; a.y_down = P->ptr(c, y); a.R = st.r;
                a.B = B; a.T = (int)Tc; a.C = st.C;
                return launch_resblock16_down(a, c.stream);
            }, 1, "resblock.fused_down");
            x_raw = x_elu = -1;
            x = y; Tc = Td; idx += 3;
            continue;
        }
        if (fused) {
            x = plan_resblock(P, st.c3, st.c1, st.sc, B, Tc, x, "enc." + std::to_string(idx), fuse_elu,
                              (idx == 1 && fold_e0) ? M : nullptr, 0, 0, ws32);
            x_is_s32 = ws32;
        } else if (ws32 && x_raw >= 0) {
            x = plan_resblock_s32(P, st.c3, st.c1, st.sc, B, Tc, x_raw, x_elu, "enc." + std::to_string(idx), 0, 0, &st.cat);
            x_is_s32 = true;
        } else {
            x = plan_resblock(P, st.c3, st.c1, st.sc, B, Tc, x, "enc." + std::to_string(idx), fuse_elu, nullptr);
            x_is_s32 = false;
        }
        x_raw = x_elu = -1;
        GemmArgs ad = sconv_args(st.down, B, Tc, st.r, 1);
        const size_t ynum = (size_t)B * ad.T_out * st.down.cout;
        const int xin = x;
        const bool last = si + 1 == M->stages.size();
        // what the next consumer wants: a fused resblock reads fp32; an unfused S32 stage reads S32 raw + S32 elu; after
        // the last stage the LSTM reads fp32 (skip) and its input projection S32
        const bool next_s32_stage = !last && x_is_s32 && (st.down.cout % 32 == 0) &&
                                    !(resblock_fusable(M->stages[si + 1].C) && !plan_unfused(P)) &&
                                    M->s32.count(M->stages[si + 1].c3.w) && M->s32.count(M->stages[si + 1].sc.w) &&
                                    M->s32.count(M->stages[si + 1].c1.w) && M->s32.count(M->stages[si + 1].down.w);
        const bool lstm_s32 = last && x_is_s32 && M->s32.count(M->enc_lstm.Wih0);
        const int y = P->buf("enc." + std::to_string(idx + 2), ynum, next_s32_stage ? BUF_S32 : BUF_F32);
        const int y2 = (next_s32_stage || lstm_s32) ? P->buf("enc." + std::to_string(idx + 2) + ".s32", ynum, BUF_S32 | (next_s32_stage ? BUF_ELU : 0)) : -1;
        P->step({xin, y, y2}, [=](const RunCtx& c) {
            GemmArgs a = ad; a.A = P->ptr(c, xin); a.C = P->ptr(c, y);
            if (y2 >= 0) a.C2 = P->ptr(c, y2);
            if (x_is_s32)
                return gemm_s32(P, a, EPI_BIAS, next_s32_stage ? OUT_S32_DUAL_ELU : (lstm_s32 ? OUT_F32_AND_S32 : OUT_F32), c.stream);
            return gemm_auto(P, a, fuse_elu ? PRO_NONE : PRO_ELU, EPI_BIAS, c.stream);
        });
        if (next_s32_stage) { x_raw = y; x_elu = y2; }
        if (lstm_s32) x_s32 = y2;
        x = y; Tc = ad.T_out; idx += 3;
    }
    const int L = (int)Tc;
    if (L != P->L) { set_error("internal: frame count mismatch"); return WT_ERR_INVALID; }
    const int H = M->H;
    const bool tail_s32 = s32 && x_s32 >= 0 && M->s32.count(M->enc_final.w) && M->s32.count(M->embed);
    x = plan_lstm(P, M->enc_lstm, B, L, H, x, "enc." + std::to_string(idx), fuse_elu, x_s32, tail_s32);
    GemmArgs af = sconv_args(M->enc_final, B, L, 1, 1);
    const int emb = P->buf("enc." + std::to_string(idx + 2), (size_t)B * L * 512);
    const int emb_s32 = tail_s32 ? P->buf("enc." + std::to_string(idx + 2) + ".s32", (size_t)B * L * 512, BUF_S32) : -1;
    {
        const int xin = x;
        P->step({xin, emb, emb_s32}, [=](const RunCtx& c) {
            GemmArgs a = af; a.A = P->ptr(c, xin); a.C = P->ptr(c, emb);
            if (tail_s32) { a.C2 = P->ptr(c, emb_s32); return gemm_s32(P, a, EPI_BIAS, OUT_F32_AND_S32, c.stream); }
            return gemm_auto(P, a, fuse_elu ? PRO_NONE : PRO_ELU, EPI_BIAS, c.stream);
        });
    }
    // ---- VQ (core_vq.py:175-183, 206-231)
    const int bins = M->arch.vq_bins;
    GemmArgs av = linear_args(M->embed, nullptr, (long)B * L, bins, 512);
    // the argmax epilogue leaves one (value, index) candidate per wave column slab; their number depends on
    // which kernel the distance GEMM runs on
    const int np = tail_s32 ? gemm16s_vq_parts(bins) : gemm_vq_parts(bins);
    const int xx = P->buf("vq.xx", (size_t)B * L);
    const int pv = P->buf("vq.pval", (size_t)B * L * np);
    const int pi = P->buf("vq.pidx", (size_t)B * L * np);
    P->step({emb, xx}, [=](const RunCtx& c) { return launch_row_sumsq(P->ptr(c, emb), P->ptr(c, xx), (long)B * L, 512, c.stream); });
    P->step({emb, emb_s32, xx, pv, pi}, [=](const RunCtx& c) {
        GemmArgs a = av; a.A = P->ptr(c, tail_s32 ? emb_s32 : emb);
        a.vq_xx = P->ptr(c, xx); a.vq_ee = M->ee; a.vq_pval = P->ptr(c, pv);
        a.vq_pidx = reinterpret_cast<int*>(P->ptr(c, pi)); a.vq_nparts = np;
        if (tail_s32) return gemm_s32(P, a, EPI_ARGMAX, OUT_F32, c.stream);
        return gemm_auto(P, a, PRO_NONE, EPI_ARGMAX, c.stream);
    }, 1, "vq.argmin");
    P->step({pv, pi, emb}, [=](const RunCtx& c) {
        if (int rc = launch_vq_finalize(P->ptr(c, pv), reinterpret_cast<int*>(P->ptr(c, pi)), np, M->embed, c.codes,
                                        c.out_f, B, L, 512, bins, c.stream)) return rc;
        if (c.aux) return launch_transpose(P->ptr(c, emb), c.aux, B, L, 512, c.stream);
        return 0;
    }, 2);
    return 0;
}

// ISTFTHead (heads.py:53-66) on the backbone output xo [M][dim] (S32 when s32): Linear + exp/clip/cos/sin fused ->
// spectrum rows [re | im]; ISTFT (spectral_ops.py:56-73) as four quarter-size real transforms (one batched GEMM), then
// the butterflies + window + overlap-add + trim + envelope divide in one pass into the caller's audio buffer
static void plan_head(wt_plan* P, int xo, bool s32) {
    const wt_model* M = P->model;
    const wt_arch& ar = M->arch;
    const int B = P->B, L = (int)P->L, D = ar.dim;
    const long Mrows = (long)B * L;
    const int Kb = M->Kb, hop = ar.hop_length;
    const int spec = P->buf("head.spec", (size_t)Mrows * 2 * Kb, s32 ? BUF_S32 : BUF_F32);
    GemmArgs ah = linear_args(M->head_W, M->head_b, Mrows, 2 * Kb, D);
    P->step({xo, spec}, [=](const RunCtx& c) {
        GemmArgs a = ah; a.A = P->ptr(c, xo); a.C = P->ptr(c, spec); a.c_rstride = 2 * Kb; a.head_kb = Kb;
        if (s32) return gemm_s32(P, a, EPI_HEAD, 1, c.stream);                // spectrum pre-split for the ISTFT GEMM
        return gemm_auto(P, a, PRO_NONE, EPI_HEAD, c.stream);
    }, 1, "head.out");
    const int Kq = M->Kq;
    const int parts = P->buf("head.parts", (size_t)4 * Mrows * Kq);       // Ce, Co, Se, So: [4][M][Kq]
    P->step({spec, parts}, [=](const RunCtx& c) {
        GemmArgs a = linear_args(M->istft_W, nullptr, Mrows, Kq, Kq);
        a.A = P->ptr(c, spec); a.a_rstride = 2 * Kb; a.zA = Kq;              // z picks the spectrum quarter
        a.zW = (long)Kq * Kq; a.nz = 4;
        a.C = P->ptr(c, parts); a.c_rstride = Kq; a.zC = (long)Mrows * Kq;
        if (s32) return gemm_s32(P, a, EPI_BIAS, 0, c.stream);
        return gemm_auto(P, a, PRO_NONE, EPI_BIAS, c.stream);
    }, 1, "head.istft");
    P->step({parts}, [=](const RunCtx& c) {
        return launch_istft_ola(P->ptr(c, parts), M->win, M->wsq, c.out_f, B, L, ar.n_fft, hop, Kq, ar.padding_same ? 0 : 1, c.stream);
    }, 1, "head.ola");
}

int build_decode(wt_plan* P) {
    const wt_model* M = P->model;
    const wt_arch& ar = M->arch;
    const int B = P->B, L = (int)P->L, D = ar.dim, I = ar.intermediate_dim, Cin = ar.input_channels;
    const long Mrows = (long)B * L;
    const int Lp = ((L + 31) / 32) * 32;
    // S32 mode: every operand of the dense chain is written pre-split by its producer (transpose, norm kernels,
    // GELU / head epilogues) and multiplied by gemm16s.hip; the residual stream and the norm inputs stay fp32
    const bool s32_plan = plan_s32(P) && (Cin % 32 == 0) && (D % 32 == 0) && (I % 32 == 0);
    // ... site by site (model.h Site): a site listed in fp32_sites keeps fp32 operands and runs its GEMMs on gemm.hip
    auto s32_at = [&](int site) { return s32_plan && !site_fp32(P, site); };
    P->cur_site = SITE_BB_EMBED;
    const bool s32_e = s32_at(SITE_BB_EMBED);
    const int x0 = P->buf("bb.in", (size_t)Mrows * Cin, s32_e ? BUF_S32 : BUF_F32);
    P->step({x0}, [=](const RunCtx& c) { return launch_transpose(c.in_f, P->ptr(c, x0), B, Cin, L, c.stream, s32_e); });
    const int x = P->buf("bb.x", (size_t)Mrows * D);       // residual stream, updated in place
    GemmArgs ae = zconv_args(M->bb_embed, B, L);
    P->step({x0, x}, [=](const RunCtx& c) {
        GemmArgs a = ae; a.A = P->ptr(c, x0); a.C = P->ptr(c, x);
        if (s32_e) return gemm_s32(P, a, EPI_BIAS, 0, c.stream);
        return gemm_auto(P, a, PRO_NONE, EPI_BIAS, c.stream);
    });
    const bool keep = P->flags & WT_PLAN_FLAG_KEEP_STAGES;
    auto snapshot = [&](const std::string& name) {   // debug taps of the in-place residual stream
        if (!keep) return;
        const int s = P->buf(name, (size_t)Mrows * D);
        P->step({x, s}, [=](const RunCtx& c) {
            WT_HIP_CHECK(hipMemcpyAsync(P->ptr(c, s), P->ptr(c, x), (size_t)Mrows * D * sizeof(float), hipMemcpyDeviceToDevice, c.stream));
            return 0;
        });
    };
    snapshot("bb.embed");
    const int sc = P->buf("bb.gn_scale", (size_t)B * D), sh = P->buf("bb.gn_shift", (size_t)B * D);
    const int gp = P->buf("bb.gn_part", gn_part_floats(B, L, 32));       // chunk statistics (long clips)
    const int h1 = P->buf("bb.h1", (size_t)Mrows * D, s32_plan ? BUF_S32 : BUF_F32);
    const int h2 = P->buf("bb.h2", (size_t)Mrows * D);

    // ResnetBlock (models.py:58-78).  GroupNorm+swish is applied ONCE per element by the statistics
    // kernel (a second pass over its own L x 24 slab) instead of in the conv's operand staging,
    // where every element would be re-normalised by each of the 18 (tap, column-tile) re-reads.
    auto resnet = [&](const PosRes& r, const std::string& name, int site) {
        P->cur_site = site;
        const bool s32 = s32_at(site);
        P->bufs[h1].fmt = s32 ? BUF_S32 : BUF_F32;         // (h1 is shared by the sites; the range report reads the format per step)
        P->step({x, sc, sh, h1, gp}, [=](const RunCtx& c) {
            return launch_gn_apply(P->ptr(c, x), r.n1w, r.n1b, P->ptr(c, sc), P->ptr(c, sh), P->ptr(c, h1), 1, B, L, D, 32, 1e-6f, c.stream, s32, P->ptr(c, gp));
        }, 1, "res.gn1");
        GemmArgs a1 = zconv_args(r.c1, B, L);
        P->step({h1, h2}, [=](const RunCtx& c) {
            GemmArgs a = a1; a.A = P->ptr(c, h1); a.C = P->ptr(c, h2);
            if (s32) return gemm_s32(P, a, EPI_BIAS, 0, c.stream);
            return gemm_auto(P, a, PRO_NONE, EPI_BIAS, c.stream);
        }, 1, "res.conv1");
        P->step({h2, sc, sh, h1, gp}, [=](const RunCtx& c) {
            return launch_gn_apply(P->ptr(c, h2), r.n2w, r.n2b, P->ptr(c, sc), P->ptr(c, sh), P->ptr(c, h1), 1, B, L, D, 32, 1e-6f, c.stream, s32, P->ptr(c, gp));
        }, 1, "res.gn2");
        GemmArgs a2 = zconv_args(r.c2, B, L);
        P->step({h1, x}, [=](const RunCtx& c) {
            GemmArgs a = a2; a.A = P->ptr(c, h1); a.C = P->ptr(c, x); a.R = P->ptr(c, x); a.r_rstride = D;
            if (s32) return gemm_s32(P, a, EPI_BIAS_RES, 0, c.stream);
            return gemm_auto(P, a, PRO_NONE, EPI_BIAS_RES, c.stream);
        }, 1, "res.conv2");
        snapshot(name);
    };
    resnet(M->res[0], "bb.pos_net.0", SITE_RES0);
    resnet(M->res[1], "bb.pos_net.1", SITE_RES1);
    P->cur_site = SITE_ATTN;
    if (s32_at(SITE_ATTN) && M->s32.count(M->at_Wqk) && M->s32.count(M->at_Wv) && M->s32.count(M->at_Wp)) {
        P->bufs[h1].fmt = BUF_S32;
        // AttnBlock (models.py:107-127), single head of width D, every product on split-f16 MFMAs: the normalised
        // input, q | k, V^T, the probabilities and the attention output are all written pre-split by their producers
        const int qk = P->buf("bb.attn.qk", (size_t)Mrows * 2 * D, BUF_S32);          // S32 [M][q | k]
        const int vt = P->buf("bb.attn.vt", (size_t)B * D * Lp, BUF_S32);              // S32 [B][D][Lp]
        const int S = P->buf("bb.attn.s", (size_t)Mrows * Lp);                // fp32 scores
        const int Ps = P->buf("bb.attn.p", (size_t)Mrows * Lp, BUF_S32);               // S32 probabilities
        const int o = P->buf("bb.attn.o", (size_t)Mrows * D, BUF_S32);                 // S32
        P->step({x, sc, sh, gp, h1}, [=](const RunCtx& c) {
            return launch_gn_apply(P->ptr(c, x), M->at_nw, M->at_nb, P->ptr(c, sc), P->ptr(c, sh), P->ptr(c, h1), 0, B, L, D, 32, 1e-6f, c.stream, 1, P->ptr(c, gp));
        }, 1, "attn.gn");
        GemmArgs aqk = linear_args(M->at_Wqk, M->at_bqk, Mrows, 2 * D, D);
        P->step({h1, qk}, [=](const RunCtx& c) {
            GemmArgs a = aqk; a.A = P->ptr(c, h1); a.C = P->ptr(c, qk);
            return gemm_s32(P, a, EPI_BIAS, OUT_S32, c.stream);
        }, 1, "attn.qk");
        P->step({h1, vt}, [=](const RunCtx& c) {     // V^T[b] = Wv . hn[b]^T + bv   (D x L, pitch Lp; pad columns stay zero)
            if (int rc = launch_fill_u32(P->ptr(c, vt), 0u, (size_t)B * D * Lp * sizeof(float), c.stream)) return rc;
            GemmArgs a = linear_args(P->ptr(c, h1), M->at_bv, D, L, D);
            a.A = reinterpret_cast<const float*>(M->s32.at(M->at_Wv)); a.zA = 0;
            if (M->s32_acc_scale.count(M->at_Wv)) a.acc_scale = M->s32_acc_scale.at(M->at_Wv);
            a.W_hi = P->ptr(c, h1); a.zW = (long)L * D; a.nz = B;
            a.C = P->ptr(c, vt); a.c_rstride = Lp; a.zC = (long)D * Lp;
            return launch_gemm16s(a, EPI_BIAS_ROW, OUT_S32, c.stream);
        }, 2, "attn.vt");
        P->step({qk, S}, [=](const RunCtx& c) {      // S[b] = q[b] . k[b]^T * D^-0.5
            GemmArgs a = linear_args(nullptr, nullptr, L, L, D);
            a.A = P->ptr(c, qk); a.a_rstride = 2 * D; a.zA = (long)L * 2 * D;
            a.W_hi = P->ptr(c, qk) + D; a.w_rstride = 2 * D; a.zW = (long)L * 2 * D; a.nz = B;
            a.C = P->ptr(c, S); a.c_rstride = Lp; a.zC = (long)L * Lp;
            a.alpha = (float)std::pow((double)D, -0.5);
            return launch_gemm16s(a, EPI_SCALE, OUT_F32, c.stream);
        }, 1, "attn.s");
        P->step({S, Ps}, [=](const RunCtx& c) { return launch_softmax(P->ptr(c, S), (int)Mrows, L, Lp, c.stream, P->ptr(c, Ps)); });
        P->step({Ps, vt, o}, [=](const RunCtx& c) {  // O[b] = P[b] . V[b]
            GemmArgs a = linear_args(nullptr, nullptr, L, D, Lp);
            a.A = P->ptr(c, Ps); a.zA = (long)L * Lp;
            a.W_hi = P->ptr(c, vt); a.zW = (long)D * Lp; a.nz = B;
            a.C = P->ptr(c, o); a.c_rstride = D; a.zC = (long)L * D;
            return launch_gemm16s(a, EPI_BIAS, OUT_S32, c.stream);
        }, 1, "attn.o");
        GemmArgs ap = linear_args(M->at_Wp, M->at_bp, Mrows, D, D);
        P->step({o, x}, [=](const RunCtx& c) {
            GemmArgs a = ap; a.A = P->ptr(c, o); a.C = P->ptr(c, x); a.R = P->ptr(c, x); a.r_rstride = D;
            return gemm_s32(P, a, EPI_BIAS_RES, OUT_F32, c.stream);
        }, 1, "attn.proj");
        snapshot("bb.pos_net.2");
    } else
    {   // AttnBlock (models.py:107-127), single head of width D
        P->bufs[h1].fmt = BUF_F32;
        const int qk = P->buf("bb.attn.qk", (size_t)Mrows * 2 * D);
        const int vt = P->buf("bb.attn.vt", (size_t)B * D * Lp);
        const int S = P->buf("bb.attn.s", (size_t)Mrows * Lp);
        const int o = P->buf("bb.attn.o", (size_t)Mrows * D);
        P->step({x, sc, sh, gp, h1}, [=](const RunCtx& c) {
            return launch_gn_apply(P->ptr(c, x), M->at_nw, M->at_nb, P->ptr(c, sc), P->ptr(c, sh), P->ptr(c, h1), 0, B, L, D, 32, 1e-6f, c.stream, 0, P->ptr(c, gp));
        }, 1, "attn.gn");
        GemmArgs aqk = linear_args(M->at_Wqk, M->at_bqk, Mrows, 2 * D, D);
        P->step({h1, qk}, [=](const RunCtx& c) {
            GemmArgs a = aqk; a.A = P->ptr(c, h1); a.C = P->ptr(c, qk);
            return gemm_auto(P, a, PRO_NONE, EPI_BIAS, c.stream);
        });
        P->step({h1, vt}, [=](const RunCtx& c) {     // V^T[b] = Wv . hn[b]^T + bv   (D x L, pitch Lp)
            if (int rc = launch_fill_u32(P->ptr(c, vt), 0u, (size_t)B * D * Lp * sizeof(float), c.stream)) return rc;
            GemmArgs a = linear_args(P->ptr(c, h1), M->at_bv, D, L, D);
            a.A = M->at_Wv; a.zA = 0; a.zW = (long)L * D; a.nz = B;
            a.C = P->ptr(c, vt); a.c_rstride = Lp; a.zC = (long)D * Lp;
            return gemm_auto(P, a, PRO_NONE, EPI_BIAS_ROW, c.stream);
        }, 2);
        P->step({qk, S}, [=](const RunCtx& c) {      // S[b] = q[b] . k[b]^T * D^-0.5
            GemmArgs a = linear_args(P->ptr(c, qk) + D, nullptr, L, L, D);
            a.A = P->ptr(c, qk); a.a_rstride = 2 * D; a.zA = (long)L * 2 * D;
            a.w_rstride = 2 * D; a.zW = (long)L * 2 * D; a.nz = B;
            a.C = P->ptr(c, S); a.c_rstride = Lp; a.zC = (long)L * Lp;
            a.alpha = (float)std::pow((double)D, -0.5);
            return gemm_auto(P, a, PRO_NONE, EPI_SCALE, c.stream);
        });
        P->step({S}, [=](const RunCtx& c) { return launch_softmax(P->ptr(c, S), (int)Mrows, L, Lp, c.stream); });
        P->step({S, vt, o}, [=](const RunCtx& c) {   // O[b] = P[b] . V[b]
            GemmArgs a = linear_args(P->ptr(c, vt), nullptr, L, D, Lp);
            a.A = P->ptr(c, S); a.zA = (long)L * Lp; a.zW = (long)D * Lp; a.nz = B;
            a.C = P->ptr(c, o); a.c_rstride = D; a.zC = (long)L * D;
            return gemm_auto(P, a, PRO_NONE, EPI_BIAS, c.stream);
        });
        GemmArgs ap = linear_args(M->at_Wp, M->at_bp, Mrows, D, D);
        P->step({o, x}, [=](const RunCtx& c) {
            GemmArgs a = ap; a.A = P->ptr(c, o); a.C = P->ptr(c, x); a.R = P->ptr(c, x); a.r_rstride = D;
            return gemm_auto(P, a, PRO_NONE, EPI_BIAS_RES, c.stream);
        });
        snapshot("bb.pos_net.2");
    }
    resnet(M->res[2], "bb.pos_net.3", SITE_RES2);
    resnet(M->res[3], "bb.pos_net.4", SITE_RES3);
    P->cur_site = SITE_CNX0;
    // pos_net[5] GroupNorm + backbone.norm AdaLayerNorm (models.py:213,228), fused into one row pass
    const int xc = P->buf(keep ? "bb.x2" : "bb.norm", (size_t)Mrows * D);
    P->step({x, gp, sc, sh}, [=](const RunCtx& c) {
        return launch_gn_stats(P->ptr(c, x), M->gn5w, M->gn5b, P->ptr(c, sc), P->ptr(c, sh), B, L, D, 32, 1e-6f, c.stream, P->ptr(c, gp));
    });
    P->step({x, sc, sh, xc}, [=](const RunCtx& c) {
        return launch_rownorm(RN_AFFINE_IN, P->ptr(c, x), P->ptr(c, xc), B, L, D, nullptr, nullptr, P->ptr(c, sc),
                              P->ptr(c, sh), M->ada_s + (size_t)c.bw_id * D, M->ada_h + (size_t)c.bw_id * D, 1e-6f, c.stream);
    });
    if (keep) {
        const int sn = P->buf("bb.norm", (size_t)Mrows * D);
        P->step({xc, sn}, [=](const RunCtx& c) {
            WT_HIP_CHECK(hipMemcpyAsync(P->ptr(c, sn), P->ptr(c, xc), (size_t)Mrows * D * sizeof(float), hipMemcpyDeviceToDevice, c.stream));
            return 0;
        });
    }
    // ConvNeXt blocks (modules.py:43-60); xc is the residual stream from here on
    const int nrm = P->buf("bb.cnx.norm", (size_t)Mrows * D, s32_plan ? BUF_S32 : BUF_F32);
    const int mid = P->buf("bb.cnx.mid", (size_t)Mrows * I, s32_plan ? BUF_S32 : BUF_F32);
    for (int i = 0; i < ar.num_layers; ++i) {
        const CnxBlock cb = M->cnx[i];
        P->cur_site = SITE_CNX0 + i;
        const bool s32 = s32_at(SITE_CNX0 + i);
        P->bufs[nrm].fmt = P->bufs[mid].fmt = s32 ? BUF_S32 : BUF_F32;
        P->step({xc, nrm}, [=](const RunCtx& c) {
            return launch_rownorm(RN_DWCONV, P->ptr(c, xc), P->ptr(c, nrm), B, L, D, cb.dw_w, cb.dw_b, nullptr, nullptr,
                                  cb.ada_s + (size_t)c.bw_id * D, cb.ada_h + (size_t)c.bw_id * D, 1e-6f, c.stream, s32);
        });
        GemmArgs a1 = linear_args(cb.W1, cb.b1, Mrows, I, D);
        P->step({nrm, mid}, [=](const RunCtx& c) {
            GemmArgs a = a1; a.A = P->ptr(c, nrm); a.C = P->ptr(c, mid);
            if (s32) return gemm_s32(P, a, EPI_BIAS_GELU, 1, c.stream);       // GELU output pre-split for pwconv2
            return gemm_auto(P, a, PRO_NONE, EPI_BIAS_GELU, c.stream);
        }, 1, "cnx.pwconv1");
        GemmArgs a2 = linear_args(cb.W2, cb.b2, Mrows, D, I);
        P->step({mid, xc}, [=](const RunCtx& c) {
            GemmArgs a = a2; a.A = P->ptr(c, mid); a.C = P->ptr(c, xc); a.R = P->ptr(c, xc); a.r_rstride = D; a.gamma = cb.gamma;
            if (s32) return gemm_s32(P, a, EPI_BIAS_GAMMA_RES, 0, c.stream);
            return gemm_auto(P, a, PRO_NONE, EPI_BIAS_GAMMA_RES, c.stream);
        }, 1, "cnx.pwconv2");
        if (keep && (i == 0 || i == ar.num_layers / 2 - 1 || i == ar.num_layers - 1)) {
            const int s = P->buf("bb.convnext." + std::to_string(i), (size_t)Mrows * D);
            P->step({xc, s}, [=](const RunCtx& c) {
                WT_HIP_CHECK(hipMemcpyAsync(P->ptr(c, s), P->ptr(c, xc), (size_t)Mrows * D * sizeof(float), hipMemcpyDeviceToDevice, c.stream));
                return 0;
            });
        }
    }
    P->cur_site = SITE_HEAD;
    const bool s32 = s32_at(SITE_HEAD);
    const int xo = P->buf("bb.out", (size_t)Mrows * D, s32 ? BUF_S32 : BUF_F32);
    P->step({xc, xo}, [=](const RunCtx& c) {
        if (int rc = launch_rownorm(RN_PLAIN, P->ptr(c, xc), P->ptr(c, xo), B, L, D, nullptr, nullptr, nullptr, nullptr,
                                    M->fln_w, M->fln_b, 1e-6f, c.stream, s32)) return rc;
        if (c.aux && s32)      // the caller wants the backbone output: a second, fp32 pass straight into its buffer
            return launch_rownorm(RN_PLAIN, P->ptr(c, xc), c.aux, B, L, D, nullptr, nullptr, nullptr, nullptr, M->fln_w,
                                  M->fln_b, 1e-6f, c.stream, 0);
        if (c.aux) WT_HIP_CHECK(hipMemcpyAsync(c.aux, P->ptr(c, xo), (size_t)Mrows * D * sizeof(float), hipMemcpyDeviceToDevice, c.stream));
        return 0;
    });
    plan_head(P, xo, s32);
    return 0;
}

// ISTFTHead alone (decoder/heads.py:42-67 + spectral_ops.py:33-75): x [B][L][dim] fp32 -> audio [B][L*hop]
int build_head(wt_plan* P) {
    const wt_model* M = P->model;
    const int B = P->B, L = (int)P->L, D = M->arch.dim;
    const long Mrows = (long)B * L;
    P->cur_site = SITE_HEAD;
    const bool s32 = plan_s32(P) && (D % 32 == 0) && M->s32.count(M->head_W) && M->s32.count(M->istft_W);
    const int xo = P->buf("head.in", (size_t)Mrows * D, s32 ? BUF_S32 : BUF_F32);
    P->step({xo}, [=](const RunCtx& c) {
        if (s32) return launch_split_s32(c.in_f, P->ptr(c, xo), Mrows * D, c.stream);
        WT_HIP_CHECK(hipMemcpyAsync(P->ptr(c, xo), c.in_f, (size_t)Mrows * D * sizeof(float), hipMemcpyDeviceToDevice, c.stream));
        return 0;
    });
    plan_head(P, xo, s32);
    return 0;
}

// SEANetDecoder on S32 operands (the default): every GEMM operand is written pre-split by its producer, as in
// build_encode.  z -> S32 -> conv k7 (fp32 for the LSTM skip + S32 for its input projection) -> LSTM (S32(elu) out)
// -> per stage: transposed conv as r phase GEMMs over (x[t-1], x[t]) -> resblock (fused resblock16 reads fp32 and
// writes S32(elu); an unfused one reads S32 raw + S32 elu, both written by the phase GEMM) -> ... -> last conv.
static bool seadec_s32_ok(const wt_plan* P) {
    const wt_model* M = P->model;
    if (!plan_s32(P) || M->sd_stages.empty() || !M->s32.count(M->sd_first.w) || !M->s32.count(M->sd_lstm.Wih0)) return false;
    for (const SeaDecStage& st : M->sd_stages) {
        if (!st.tr_wp || !M->s32.count(st.tr_wp) || st.cout % 32) return false;
        if (!resblock_fusable(st.cout) && !(M->s32.count(st.c3.w) && M->s32.count(st.c1.w) && M->s32.count(st.sc.w))) return false;
    }
    return resblock_fusable(M->sd_stages.back().cout);      // the last conv reads fp32
}

static int build_seanet_decoder_s32(wt_plan* P) {
    const wt_model* M = P->model;
    const int B = P->B, L = (int)P->L, H = M->H;
    const int x0 = P->buf("sdec.in", (size_t)B * L * 512);
    P->step({x0}, [=](const RunCtx& c) { return launch_transpose(c.in_f, P->ptr(c, x0), B, 512, L, c.stream, 1); });
    const int xf = P->buf("sdec.0", (size_t)B * L * H);
    const int xs = P->buf("sdec.0.s32", (size_t)B * L * H);
    GemmArgs a0 = sconv_args(M->sd_first, B, L, 1, 1);
    P->step({x0, xf, xs}, [=](const RunCtx& c) {
        GemmArgs a = a0; a.A = P->ptr(c, x0); a.C = P->ptr(c, xf); a.C2 = P->ptr(c, xs);
        return gemm_s32(P, a, EPI_BIAS, OUT_F32_AND_S32, c.stream);
    });
    int x = plan_lstm(P, M->sd_lstm, B, L, H, xf, "sdec.1", true, xs, true);        // S32(elu(lstm(x) + x))
    long Tc = L;
    int di = 2;
    for (size_t si = 0; si < M->sd_stages.size(); ++si) {
        const SeaDecStage st = M->sd_stages[si];
        const long To = Tc * st.r;
        const int xin = x;
        const int Tin = (int)Tc;
        const bool fused = resblock_fusable(st.cout);
        const bool last = si + 1 == M->sd_stages.size();
        // SConvTranspose1d (conv.py:232-253), k = 2*stride: see build_seanet_decoder
        const int trim_l = (st.k - st.r) - (st.k - st.r) / 2;
        const size_t ynum = (size_t)B * (Tin + 1) * st.r * st.cout;
        const int y = P->buf("sdec." + std::to_string(di + 1), ynum);
        const int y2 = fused ? -1 : P->buf("sdec." + std::to_string(di + 1) + ".elu", ynum);
        const long y_off = (long)trim_l * st.cout, y_bs = (long)(Tin + 1) * st.r * st.cout;
        P->step({xin, y, y2}, [=](const RunCtx& c) {
            GemmArgs a;
            a.A = P->ptr(c, xin); a.a_bstride = (long)Tin * st.cin; a.a_rstride = st.cin;
            a.T_in = Tin; a.T_out = Tin + 1; a.Cin = st.cin; a.taps = 2; a.pad_left = 1; a.pad_mode = PAD_ZERO;
            a.W = st.tr_wp; a.w_rstride = 2L * st.cin; a.zW = (long)st.cout * 2 * st.cin; a.bias = st.tr_b;
            a.M = B * (Tin + 1); a.N = st.cout; a.K = 2 * st.cin;
            a.C = P->ptr(c, y); a.c_rstride = (long)st.r * st.cout; a.zC = st.cout; a.nz = st.r;
            if (y2 >= 0) a.C2 = P->ptr(c, y2);
            return gemm_s32(P, a, EPI_BIAS, fused ? OUT_F32 : OUT_S32_DUAL_ELU, c.stream);
        }, 1, "sdec.convtr");
        if (fused)
            x = plan_resblock(P, st.c3, st.c1, st.sc, B, To, y, "sdec." + std::to_string(di + 2), true, nullptr, y_off, y_bs, !last);
        else
            x = plan_resblock_s32(P, st.c3, st.c1, st.sc, B, To, y, y2, "sdec." + std::to_string(di + 2), y_off, y_bs, &st.cat);
        Tc = To; di += 3;
    }
    const int xin = x;
    const long Tf = Tc;
    P->step({xin}, [=](const RunCtx& c) {
        return launch_conv_last(P->ptr(c, xin), M->sd_last_w, M->sd_last_b, c.out_f, B, Tf, 32, 7, 0, c.stream);
    }, 1, "sdec.last");
    return 0;
}

int build_seanet_decoder(wt_plan* P) {
    const wt_model* M = P->model;
    if (!M->has_seadec) { set_error("checkpoint holds no SEANetDecoder weights"); return WT_ERR_MISSING_TENSOR; }
    const int B = P->B, L = (int)P->L, H = M->H;
    P->cur_site = SITE_SEADEC;
    if (seadec_s32_ok(P)) return build_seanet_decoder_s32(P);
    const int x0 = P->buf("sdec.in", (size_t)B * L * 512);
    P->step({x0}, [=](const RunCtx& c) { return launch_transpose(c.in_f, P->ptr(c, x0), B, 512, L, c.stream); });
    int x = P->buf("sdec.0", (size_t)B * L * H);
    GemmArgs a0 = sconv_args(M->sd_first, B, L, 1, 1);
    {
        const int y = x;
        P->step({x0, y}, [=](const RunCtx& c) {
            GemmArgs a = a0; a.A = P->ptr(c, x0); a.C = P->ptr(c, y);
            return gemm_auto(P, a, PRO_NONE, EPI_BIAS, c.stream);
        });
    }
    const bool fuse_elu = !plan_unfused(P);      // producers store elu(.) for "ELU -> conv" consumers
    x = plan_lstm(P, M->sd_lstm, B, L, H, x, "sdec.1", fuse_elu);
    long Tc = L;
    int di = 2;
    for (size_t si = 0; si < M->sd_stages.size(); ++si) {
        const SeaDecStage st = M->sd_stages[si];
        const long To = Tc * st.r;
        const int xin = x;
        const int Tin = (int)Tc;
        int y;
        long y_off = 0, y_bs = 0;
        if (st.tr_wp) {
            // SConvTranspose1d (conv.py:232-253) with k = 2*stride: output sample u' = t*stride + r gets
            // x[t].W[r] + x[t-1].W[r+stride], i.e. per phase r one GEMM over rows t = 0..Tin with the two
            // frames as K (zero beyond the clip); the phases are the batch dimension and interleave in the
            // untrimmed output, of which the following resblock reads the trimmed view.
            const int trim_l = (st.k - st.r) - (st.k - st.r) / 2;
            y = P->buf("sdec." + std::to_string(di + 1), (size_t)B * (Tin + 1) * st.r * st.cout);
            y_off = (long)trim_l * st.cout;
            y_bs = (long)(Tin + 1) * st.r * st.cout;
            P->step({xin, y}, [=](const RunCtx& c) {
                GemmArgs a;
                a.A = P->ptr(c, xin); a.a_bstride = (long)Tin * st.cin; a.a_rstride = st.cin;
                a.T_in = Tin; a.T_out = Tin + 1; a.Cin = st.cin; a.taps = 2; a.pad_left = 1; a.pad_mode = PAD_ZERO;
                a.W = st.tr_wp; a.w_rstride = 2L * st.cin; a.zW = (long)st.cout * 2 * st.cin; a.bias = st.tr_b;
                a.M = B * (Tin + 1); a.N = st.cout; a.K = 2 * st.cin;
                a.C = P->ptr(c, y); a.c_rstride = (long)st.r * st.cout; a.zC = st.cout; a.nz = st.r;
                return gemm_auto(P, a, fuse_elu ? PRO_NONE : PRO_ELU, EPI_BIAS, c.stream);
            }, 1, "sdec.convtr");
        } else {
            y = P->buf("sdec." + std::to_string(di + 1), (size_t)B * To * st.cout);
            P->step({xin, y}, [=](const RunCtx& c) {
                return launch_convtr(P->ptr(c, xin), st.tr_w, st.tr_b, P->ptr(c, y), B, Tin, st.cin, st.cout, st.k, st.r,
                                     fuse_elu ? 0 : 1, c.stream);
            }, 1, "sdec.convtr");
        }
        x = plan_resblock(P, st.c3, st.c1, st.sc, B, To, y, "sdec." + std::to_string(di + 2), fuse_elu, nullptr, y_off, y_bs);
        Tc = To; di += 3;
    }
    const int xin = x;
    const long Tf = Tc;
    P->step({xin}, [=](const RunCtx& c) {
        return launch_conv_last(P->ptr(c, xin), M->sd_last_w, M->sd_last_b, c.out_f, B, Tf, 32, 7, fuse_elu ? 0 : 1, c.stream);
    }, 1, "sdec.last");
    return 0;
}


// Every plan starts by zeroing its control block (word 0 = the call's status, common.h) and ends with the guard step
void plan_begin(wt_plan* P) {
    P->ctl = P->buf("ctl", CTL_WORDS);
    const int ctl = P->ctl;
    P->step({ctl}, [=](const RunCtx& c) {
        return launch_fill_u32(P->ptr(c, ctl), 0u, CTL_WORDS * sizeof(unsigned), c.stream);
    }, 1, "ctl.clear");
}

void plan_end(wt_plan* P) {
    const wt_model* M = P->model;
    const int ctl = P->ctl, kind = P->kind;
    const long B = P->B, L = P->L, hop = M->hop, D = M->arch.dim;
    P->step({ctl}, [=](const RunCtx& c) {
        int64_t* codes = nullptr;
        long nc = 0, n0 = 0, n1 = 0;
        float* f1 = c.aux;
        if (kind == WT_PLAN_ENCODE) { codes = c.codes; nc = B * L; n0 = B * 512 * L; n1 = n0; }
        else if (kind == WT_PLAN_DECODE) { n0 = B * wave_samples(M, L); n1 = B * L * D; }
        else if (kind == WT_PLAN_UNIT_LSTM) { n0 = B * L * 512; f1 = nullptr; }
        else if (kind == WT_PLAN_HEAD) { n0 = B * wave_samples(M, L); f1 = nullptr; }
        else { n0 = B * L * hop; f1 = nullptr; }
        return launch_plan_guard(reinterpret_cast<const unsigned*>(P->ptr(c, ctl)), CTL_WORDS, CTL_SITE0, P->status_dev, M->status_dev, codes,
                                 nc, c.out_f, n0, f1, n1, nullptr, 0, c.stream);
    }, 1, "guard");
}

// SLSTM alone (unit parity tests: lstm_persist_kernel / lstm_step_kernel against the oracle): x [B][L][512] fp32,
// time-major -> y = lstm(x) + x, through exactly the steps the encoder plan uses (S32 copy of x for the input
// projection GEMM, then the recurrence).  WT_PLAN_FLAG_STEP_LSTM selects the launch-per-step kernel.
int build_unit_lstm(wt_plan* P) {
    const wt_model* M = P->model;
    const int B = P->B, L = (int)P->L, H = M->H;
    const bool s32 = plan_s32(P) && M->s32.count(M->enc_lstm.Wih0);
    const int x = P->buf("lstm.in", (size_t)B * L * H);
    const int xs = s32 ? P->buf("lstm.in.s32", (size_t)B * L * H, BUF_S32) : -1;
    P->step({x, xs}, [=](const RunCtx& c) {
        WT_HIP_CHECK(hipMemcpyAsync(P->ptr(c, x), c.in_f, (size_t)B * L * H * sizeof(float), hipMemcpyDeviceToDevice, c.stream));
        if (s32) return launch_split_s32(c.in_f, P->ptr(c, xs), (long)B * L * H, c.stream);
        return 0;
    }, 2, "lstm.in");
    const int y = plan_lstm(P, M->enc_lstm, B, L, H, x, "lstm.out", false, xs, false);
    P->step({y}, [=](const RunCtx& c) {
        WT_HIP_CHECK(hipMemcpyAsync(c.out_f, P->ptr(c, y), (size_t)B * L * H * sizeof(float), hipMemcpyDeviceToDevice, c.stream));
        return 0;
    }, 1, "lstm.copy");
    return 0;
}

}  // namespace wt
