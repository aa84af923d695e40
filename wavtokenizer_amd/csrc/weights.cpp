// Weight folding / packing, done once at wt_model_create: weight-norm fold (fp64), conv repacking to [Cout][tap][Cin],
// LSTM gate-row packings (fp32 and split-f16 per-lane forms), the packed ISTFT head and inverse-DFT basis, and the
// split-f16 (S32 / f16x2) copies of every GEMM weight.
#include "model.h"

namespace wt {

static int upload(wt_model* M, const std::vector<float>& h, float** out) {
    void* d = nullptr;
    size_t bytes = std::max<size_t>(h.size(), 4) * sizeof(float);
    WT_HIP_CHECK(hipMalloc(&d, bytes));
    M->allocs.push_back(d);
    M->alloc_bytes.push_back(bytes);
    WT_HIP_CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    M->weight_bytes += (int64_t)h.size() * sizeof(float);
    *out = static_cast<float*>(d);
    return 0;
}
static int upload_raw(wt_model* M, const float* src, int64_t n, float** out) {
    std::vector<float> h(src, src + n);
    return upload(M, h, out);
}

// weight_norm fold (torch.nn.utils.weight_norm, dim=0; conv.py:25-34): w[o] = g[o] * v[o] / ||v[o]||
static std::vector<float> fold_wn(const float* g, const float* v, int d0, int64_t inner) {
    std::vector<float> w((size_t)d0 * inner);
    for (int o = 0; o < d0; ++o) {
        double ss = 0.0;
        for (int64_t i = 0; i < inner; ++i) { double x = v[o * inner + i]; ss += x * x; }
        const double sc = (double)g[o] / std::sqrt(ss);
        for (int64_t i = 0; i < inner; ++i) w[o * inner + i] = (float)(sc * (double)v[o * inner + i]);
    }
    return w;
}

// [cout][cin][k] -> [cout][k][cin]
static std::vector<float> repack_ock(const std::vector<float>& w, int cout, int cin, int k) {
    std::vector<float> o(w.size());
    for (int a = 0; a < cout; ++a)
        for (int c = 0; c < cin; ++c)
            for (int j = 0; j < k; ++j) o[((size_t)a * k + j) * cin + c] = w[((size_t)a * cin + c) * k + j];
    return o;
}

static int load_wn_conv(wt_model* M, TensorMap& tm, const std::string& prefix, int cout, int cin, int k, ConvW* out) {
    const float* g = tm.get(prefix + ".weight_g", cout);
    const float* v = tm.get(prefix + ".weight_v", (int64_t)cout * cin * k);
    const float* b = tm.get(prefix + ".bias", cout);
    if (!g || !v || !b) return WT_ERR_MISSING_TENSOR;
    std::vector<float> w = repack_ock(fold_wn(g, v, cout, (int64_t)cin * k), cout, cin, k);
    out->cout = cout; out->cin = cin; out->k = k;
    if (int rc = upload(M, w, &out->w)) return rc;
    return upload_raw(M, b, cout, &out->b);
}

static int load_plain_conv(wt_model* M, TensorMap& tm, const std::string& prefix, int cout, int cin, int k, ConvW* out) {
    const float* w = tm.get(prefix + ".weight", (int64_t)cout * cin * k);
    const float* b = tm.get(prefix + ".bias", cout);
    if (!w || !b) return WT_ERR_MISSING_TENSOR;
    std::vector<float> wv(w, w + (int64_t)cout * cin * k);
    std::vector<float> p = repack_ock(wv, cout, cin, k);
    out->cout = cout; out->cin = cin; out->k = k;
    if (int rc = upload(M, p, &out->w)) return rc;
    return upload_raw(M, b, cout, &out->b);
}

static int load_vec(wt_model* M, TensorMap& tm, const std::string& key, int64_t n, float** out) {
    const float* p = tm.get(key, n);
    if (!p) return WT_ERR_MISSING_TENSOR;
    return upload_raw(M, p, n, out);
}

// nn.LSTM weights -> packed gate order: packed row (j/4)*16 + g*4 + j%4  <-  row g*H + j
static int load_lstm(wt_model* M, TensorMap& tm, const std::string& prefix, int H, LstmW* out, bool* split_ok) {
    const float* wih0 = tm.get(prefix + ".lstm.weight_ih_l0", 4LL * H * H);
    const float* whh0 = tm.get(prefix + ".lstm.weight_hh_l0", 4LL * H * H);
    const float* bih0 = tm.get(prefix + ".lstm.bias_ih_l0", 4LL * H);
    const float* bhh0 = tm.get(prefix + ".lstm.bias_hh_l0", 4LL * H);
    const float* wih1 = tm.get(prefix + ".lstm.weight_ih_l1", 4LL * H * H);
    const float* whh1 = tm.get(prefix + ".lstm.weight_hh_l1", 4LL * H * H);
    const float* bih1 = tm.get(prefix + ".lstm.bias_ih_l1", 4LL * H);
    const float* bhh1 = tm.get(prefix + ".lstm.bias_hh_l1", 4LL * H);
    if (!wih0 || !whh0 || !bih0 || !bhh0 || !wih1 || !whh1 || !bih1 || !bhh1) return WT_ERR_MISSING_TENSOR;
    std::vector<float> Wih0((size_t)4 * H * H), W0((size_t)4 * H * H), W1((size_t)4 * H * 2 * H), b0(4 * H), b1(4 * H);
    // recurrent weights: per 16 packed gate rows (one workgroup), [K/16][64 lanes][4] with lane = lk*16 + li:
    // element e of group S is W[row li][k = 16 S + 4 e + lk]  (the B operand of four k-steps in one 16-byte load)
    auto put = [&](std::vector<float>& dst, size_t prow, int Ktot, int k, float v) {
        const size_t blk = prow / 16, li = prow % 16;
        const int S = k / 16, e = (k % 16) / 4, lk = k % 4;
        dst[blk * 16 * Ktot + (size_t)S * 256 + (size_t)(lk * 16 + li) * 4 + e] = v;
    };
    // split-f16 packing: per 16 gate rows [K/32][hi, lo][64 lanes][8 halves]: half p of lane (li, lk) in block P is
    // W[row li][k = 32 P + 16 (p >> 2) + 4 (p & 3) + lk], as hi = f16(w) and lo = f16((w - hi) * 2^11)
    std::vector<_Float16> W0h((size_t)4 * H * H * 2), W1h((size_t)4 * H * 2 * H * 2);
    auto put16 = [&](std::vector<_Float16>& dst, size_t prow, int Ktot, int k, float v) {
        const size_t blk = prow / 16, li = prow % 16;
        const int P = k / 32, r = k % 32, pp = (r / 16) * 4 + (r % 16) / 4, lk = r % 4;
        const size_t base = blk * 16 * Ktot * 2 + (size_t)P * 1024 + (size_t)(lk * 16 + li) * 8 + pp;
        const _Float16 h = (_Float16)v;
        dst[base] = h;
        dst[base + 512] = (_Float16)((v - (float)h) * 2048.f);
    };
    // persistent-kernel packing (H = 512): packed gate row -> (workgroup = row / 64, tile = row % 64 / 16, li = row % 16);
    // half p of lane (li, lk) in block blk is W[row][k = 32 blk + 8 lk + p]
    std::vector<_Float16> Wp(H == 512 ? (size_t)3 * 2048 * 512 * 2 : 0);
    auto putp = [&](int role, size_t prow, int k, float v) {
        if (Wp.empty()) return;
        const size_t wg = prow / 64, tile = (prow % 64) / 16, li = prow % 16;
        const int blk = k / 32, lk = (k % 32) / 8, pp = k % 8;
        const size_t base = ((((size_t)role * 32 + wg) * 4 + tile) * 16 + blk) * 2 * 64 * 8 + (size_t)(lk * 16 + li) * 8 + pp;
        const _Float16 h = (_Float16)v;
        Wp[base] = h;
        Wp[base + 64 * 8] = (_Float16)((v - (float)h) * 2048.f);
    };
    // the recurrent weights are split here, unscaled (hi = f16(w)): beyond the f16 range the model runs on fp32 kernels
    for (const float* wsrc : {whh0, wih1, whh1})
        for (size_t i = 0; i < (size_t)4 * H * H; ++i)
            if (!(std::fabs(wsrc[i]) < 65504.f)) *split_ok = false;
    for (int g = 0; g < 4; ++g)
        for (int j = 0; j < H; ++j) {
            const size_t src = (size_t)g * H + j;
            const size_t dst = (size_t)(j / 4) * 16 + g * 4 + (j % 4);
            std::memcpy(&Wih0[dst * H], &wih0[src * H], H * sizeof(float));
            for (int k = 0; k < H; ++k) {
                put(W0, dst, H, k, whh0[src * H + k]);
                put(W1, dst, 2 * H, k, wih1[src * H + k]);
                put(W1, dst, 2 * H, H + k, whh1[src * H + k]);
                put16(W0h, dst, H, k, whh0[src * H + k]);
                put16(W1h, dst, 2 * H, k, wih1[src * H + k]);
                put16(W1h, dst, 2 * H, H + k, whh1[src * H + k]);
                putp(0, dst, k, whh0[src * H + k]);
                putp(1, dst, k, wih1[src * H + k]);
                putp(2, dst, k, whh1[src * H + k]);
            }
            b0[dst] = bih0[src] + bhh0[src];
            b1[dst] = bih1[src] + bhh1[src];
        }
    if (int rc = upload(M, Wih0, &out->Wih0)) return rc;
    if (int rc = upload(M, b0, &out->b0)) return rc;
    if (int rc = upload(M, W0, &out->W0)) return rc;
    if (int rc = upload(M, W1, &out->W1)) return rc;
    {
        std::vector<float> t0(W0.size()), t1(W1.size());         // same byte counts: 2 halves per weight
        std::memcpy(t0.data(), W0h.data(), t0.size() * sizeof(float));
        std::memcpy(t1.data(), W1h.data(), t1.size() * sizeof(float));
        if (int rc = upload(M, t0, &out->W0h)) return rc;
        if (int rc = upload(M, t1, &out->W1h)) return rc;
        if (!Wp.empty()) {
            std::vector<float> tp(Wp.size() / 2);
            std::memcpy(tp.data(), Wp.data(), tp.size() * sizeof(float));
            if (int rc = upload(M, tp, &out->Wp)) return rc;
        }
    }
    return upload(M, b1, &out->b1);
}

// SEANetResnetBlock tail (seanet.py:62-63): shortcut(x) + conv1(elu(h)) = [Ws | W1] . [x | elu(h)] + (bs + b1): both are 1x1
// convs, so the unfused stages run them as ONE GEMM whose K columns come from two tensors (gemm16s, GemmArgs::A2)
static int build_cat(wt_model* M, const ConvW& sc, const ConvW& c1, ConvW* cat) {
    if (sc.k != 1 || c1.k != 1 || sc.cout != c1.cout) return 0;
    const int C = sc.cout, K1 = sc.cin, K2 = c1.cin;
    std::vector<float> ws((size_t)C * K1), w1((size_t)C * K2), bs(C), b1(C), w((size_t)C * (K1 + K2)), b(C);
    WT_HIP_CHECK(hipMemcpy(ws.data(), sc.w, ws.size() * sizeof(float), hipMemcpyDeviceToHost));
    WT_HIP_CHECK(hipMemcpy(w1.data(), c1.w, w1.size() * sizeof(float), hipMemcpyDeviceToHost));
    WT_HIP_CHECK(hipMemcpy(bs.data(), sc.b, bs.size() * sizeof(float), hipMemcpyDeviceToHost));
    WT_HIP_CHECK(hipMemcpy(b1.data(), c1.b, b1.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int n = 0; n < C; ++n) {
        std::memcpy(&w[(size_t)n * (K1 + K2)], &ws[(size_t)n * K1], K1 * sizeof(float));
        std::memcpy(&w[(size_t)n * (K1 + K2) + K1], &w1[(size_t)n * K2], K2 * sizeof(float));
        b[n] = bs[n] + b1[n];
    }
    if (int rc = upload(M, w, &cat->w)) return rc;
    if (int rc = upload(M, b, &cat->b)) return rc;
    cat->cout = C; cat->cin = K1 + K2; cat->k = 1;
    return 0;
}

static const char* ENC = "feature_extractor.encodec.encoder.model.";
static const char* DEC = "feature_extractor.encodec.decoder.model.";
static const char* VQK = "feature_extractor.encodec.quantizer.vq.layers.0._codebook.";

int build_model(wt_model* M, TensorMap& tm) {
    const wt_arch& a = M->arch;
    const int nf = 32, H = 512;
    M->H = H;
    // ---- encoder (encoder/modules/seanet.py:66-144)
    {
        const float* g = tm.get(std::string(ENC) + "0.conv.conv.weight_g", nf);
        const float* v = tm.get(std::string(ENC) + "0.conv.conv.weight_v", (int64_t)nf * 7);
        const float* b = tm.get(std::string(ENC) + "0.conv.conv.bias", nf);
        if (!g || !v || !b) return WT_ERR_MISSING_TENSOR;
        std::vector<float> w = fold_wn(g, v, nf, 7);       // [32][1][7]
        std::vector<float> p((size_t)7 * nf);
        for (int c = 0; c < nf; ++c)
            for (int j = 0; j < 7; ++j) p[(size_t)j * nf + c] = w[(size_t)c * 7 + j];
        if (int rc = upload(M, p, &M->e0_w)) return rc;
        if (int rc = upload_raw(M, b, nf, &M->e0_b)) return rc;
    }
    int idx = 1, mult = 1;
    for (int r : M->enc_ratios) {
        ResStage st;
        st.C = mult * nf; st.r = r;
        const std::string p = std::string(ENC) + std::to_string(idx);
        if (int rc = load_wn_conv(M, tm, p + ".block.1.conv.conv", st.C / 2, st.C, 3, &st.c3)) return rc;
        if (int rc = load_wn_conv(M, tm, p + ".block.3.conv.conv", st.C, st.C / 2, 1, &st.c1)) return rc;
        if (int rc = load_wn_conv(M, tm, p + ".shortcut.conv.conv", st.C, st.C, 1, &st.sc)) return rc;
        if (int rc = load_wn_conv(M, tm, std::string(ENC) + std::to_string(idx + 2) + ".conv.conv", 2 * st.C, st.C, 2 * r, &st.down)) return rc;
        if (!resblock_fusable(st.C)) if (int rc = build_cat(M, st.sc, st.c1, &st.cat)) return rc;
        M->stages.push_back(st);
        idx += 3; mult *= 2;
    }
    if (mult * nf != H) { set_error("encoder width after the last ratio must be 512"); return WT_ERR_INVALID; }
    if (int rc = load_lstm(M, tm, std::string(ENC) + std::to_string(idx), H, &M->enc_lstm, &M->s32_ok)) return rc;
    if (int rc = load_wn_conv(M, tm, std::string(ENC) + std::to_string(idx + 2) + ".conv.conv", 512, H, 7, &M->enc_final)) return rc;

    // ---- codebook (encoder/quantization/core_vq.py:122-138)
    {
        const float* inited = tm.get(std::string(VQK) + "inited", 1);
        const float* e = tm.get(std::string(VQK) + "embed", (int64_t)a.vq_bins * 512);
        if (!inited || !e) return WT_ERR_MISSING_TENSOR;
        if (inited[0] != 1.0f) {
            set_error("codebook buffer `inited` is not 1: the reference would run k-means on the first forward (core_vq.py:140-151)");
            return WT_ERR_NOT_INITED;
        }
        // num_quantizers > 1 (residual codebooks the checkpoint holds beside the first): encode_infer uses layer 0 only
        // (vq.py:137 forces n_q = 1), codes_to_features sums the rows of the first K <= num_quantizers codebooks of the
        // concatenated table (pretrained.py:230-237).  The table is stored concatenated; layer 0 comes first, so everything that
        // addresses "the codebook" (VQ distances, its S32 copy, the encode plan's gather) is unchanged
        std::vector<float> all((size_t)a.num_quantizers * a.vq_bins * 512);
        std::memcpy(all.data(), e, (size_t)a.vq_bins * 512 * sizeof(float));
        for (int q = 1; q < a.num_quantizers; ++q) {
            const std::string pq = "feature_extractor.encodec.quantizer.vq.layers." + std::to_string(q) + "._codebook.";
            const float* eq = tm.get(pq + "embed", (int64_t)a.vq_bins * 512);
            if (!eq) return WT_ERR_MISSING_TENSOR;
            std::memcpy(all.data() + (size_t)q * a.vq_bins * 512, eq, (size_t)a.vq_bins * 512 * sizeof(float));
        }
        if (int rc = upload(M, all, &M->embed)) return rc;
        std::vector<float> ee(a.vq_bins);
        for (int n = 0; n < a.vq_bins; ++n) {   // embed.pow(2).sum(0)
            float s = 0.f;
            for (int c = 0; c < 512; ++c) s += e[(size_t)n * 512 + c] * e[(size_t)n * 512 + c];
            ee[n] = s;
        }
        if (int rc = upload(M, ee, &M->ee)) return rc;
    }

    // ---- backbone (decoder/models.py:166-216)
    const int D = a.dim, I = a.intermediate_dim, A = a.adanorm_num_embeddings;
    if (int rc = load_plain_conv(M, tm, "backbone.embed", D, a.input_channels, 7, &M->bb_embed)) return rc;
    const int ridx[4] = {0, 1, 3, 4};
    for (int i = 0; i < 4; ++i) {
        const std::string p = "backbone.pos_net." + std::to_string(ridx[i]);
        PosRes& r = M->res[i];
        if (int rc = load_vec(M, tm, p + ".norm1.weight", D, &r.n1w)) return rc;
        if (int rc = load_vec(M, tm, p + ".norm1.bias", D, &r.n1b)) return rc;
        if (int rc = load_vec(M, tm, p + ".norm2.weight", D, &r.n2w)) return rc;
        if (int rc = load_vec(M, tm, p + ".norm2.bias", D, &r.n2b)) return rc;
        if (int rc = load_plain_conv(M, tm, p + ".conv1", D, D, 3, &r.c1)) return rc;
        if (int rc = load_plain_conv(M, tm, p + ".conv2", D, D, 3, &r.c2)) return rc;
    }
    {
        const std::string p = "backbone.pos_net.2";
        if (int rc = load_vec(M, tm, p + ".norm.weight", D, &M->at_nw)) return rc;
        if (int rc = load_vec(M, tm, p + ".norm.bias", D, &M->at_nb)) return rc;
        const float* wq = tm.get(p + ".q.weight", (int64_t)D * D);
        const float* wk = tm.get(p + ".k.weight", (int64_t)D * D);
        const float* bq = tm.get(p + ".q.bias", D);
        const float* bk = tm.get(p + ".k.bias", D);
        if (!wq || !wk || !bq || !bk) return WT_ERR_MISSING_TENSOR;
        std::vector<float> wqk((size_t)2 * D * D), bqk(2 * D);
        std::memcpy(&wqk[0], wq, (size_t)D * D * sizeof(float));
        std::memcpy(&wqk[(size_t)D * D], wk, (size_t)D * D * sizeof(float));
        std::memcpy(&bqk[0], bq, D * sizeof(float));
        std::memcpy(&bqk[D], bk, D * sizeof(float));
        if (int rc = upload(M, wqk, &M->at_Wqk)) return rc;
        if (int rc = upload(M, bqk, &M->at_bqk)) return rc;
        if (int rc = load_vec(M, tm, p + ".v.weight", (int64_t)D * D, &M->at_Wv)) return rc;
        if (int rc = load_vec(M, tm, p + ".v.bias", D, &M->at_bv)) return rc;
        if (int rc = load_vec(M, tm, p + ".proj_out.weight", (int64_t)D * D, &M->at_Wp)) return rc;
        if (int rc = load_vec(M, tm, p + ".proj_out.bias", D, &M->at_bp)) return rc;
    }
    if (int rc = load_vec(M, tm, "backbone.pos_net.5.weight", D, &M->gn5w)) return rc;
    if (int rc = load_vec(M, tm, "backbone.pos_net.5.bias", D, &M->gn5b)) return rc;
    if (A <= 0) { set_error("only the AdaLayerNorm (adanorm_num_embeddings > 0) backbone is implemented"); return WT_ERR_INVALID; }
    if (int rc = load_vec(M, tm, "backbone.norm.scale.weight", (int64_t)A * D, &M->ada_s)) return rc;
    if (int rc = load_vec(M, tm, "backbone.norm.shift.weight", (int64_t)A * D, &M->ada_h)) return rc;
    for (int i = 0; i < a.num_layers; ++i) {
        const std::string p = "backbone.convnext." + std::to_string(i);
        CnxBlock c;
        const float* dw = tm.get(p + ".dwconv.weight", (int64_t)D * 7);
        if (!dw) return WT_ERR_MISSING_TENSOR;
        std::vector<float> dwp((size_t)7 * D);
        for (int ch = 0; ch < D; ++ch)
            for (int j = 0; j < 7; ++j) dwp[(size_t)j * D + ch] = dw[(size_t)ch * 7 + j];
        if (int rc = upload(M, dwp, &c.dw_w)) return rc;
        if (int rc = load_vec(M, tm, p + ".dwconv.bias", D, &c.dw_b)) return rc;
        if (int rc = load_vec(M, tm, p + ".norm.scale.weight", (int64_t)A * D, &c.ada_s)) return rc;
        if (int rc = load_vec(M, tm, p + ".norm.shift.weight", (int64_t)A * D, &c.ada_h)) return rc;
        if (int rc = load_vec(M, tm, p + ".pwconv1.weight", (int64_t)I * D, &c.W1)) return rc;
        if (int rc = load_vec(M, tm, p + ".pwconv1.bias", I, &c.b1)) return rc;
        if (int rc = load_vec(M, tm, p + ".pwconv2.weight", (int64_t)D * I, &c.W2)) return rc;
        if (int rc = load_vec(M, tm, p + ".pwconv2.bias", D, &c.b2)) return rc;
        if (int rc = load_vec(M, tm, p + ".gamma", D, &c.gamma)) return rc;
        M->cnx.push_back(c);
    }
    if (int rc = load_vec(M, tm, "backbone.final_layer_norm.weight", D, &M->fln_w)) return rc;
    if (int rc = load_vec(M, tm, "backbone.final_layer_norm.bias", D, &M->fln_b)) return rc;

    // ---- head (decoder/heads.py:36-67, decoder/spectral_ops.py:33-75)
    {
        const int N = a.n_fft, hop = a.hop_length;
        if (N % hop != 0 || N % 2 != 0 || (N - hop) % 2 != 0) {
            set_error("ISTFT kernel needs n_fft to be an even multiple of hop_length"); return WT_ERR_INVALID;
        }
        if (N % 4 != 0) { set_error("ISTFT kernel needs n_fft % 4 == 0"); return WT_ERR_INVALID; }
        const int bins = N / 2 + 1;
        const int Q = N / 4;
        const int Kq = ((Q + 1 + 31) / 32) * 32;          // padded count of even (Q+1) / odd (Q) bins
        const int Kb = 2 * Kq;                            // spectrum half-row: [even bins | odd bins]
        const int R = N / hop;
        M->Kb = Kb; M->Kq = Kq; M->bins_f = bins; M->R = R;
        const float* w = tm.get("head.out.weight", (int64_t)(N + 2) * D);
        const float* b = tm.get("head.out.bias", N + 2);
        const float* win = tm.get("head.istft.window", N);
        if (!w || !b || !win) return WT_ERR_MISSING_TENSOR;
        // spectrum slot s -> frequency bin: s < Kq: even bin 2s; else odd bin 2(s-Kq)+1 (-1 = padding)
        auto slot_bin = [&](int s) { int f = s < Kq ? 2 * s : 2 * (s - Kq) + 1; return f <= N / 2 && (s < Kq || s - Kq < Q) ? f : -1; };
        // packed head rows: 32-row groups = 16 log-magnitude rows then the 16 phase rows of the same slots (a 32 x 32 block
        // of any GEMM tile then holds both halves of 16 slots: no constraint on the tile width)
        std::vector<float> wp((size_t)2 * Kb * D, 0.f), bp((size_t)2 * Kb, 0.f);
        for (int sl = 0; sl < Kb; ++sl) {
            const int f = slot_bin(sl);
            if (f < 0) continue;
            const size_t pm = (size_t)(sl / 16) * 32 + (sl % 16), pp = pm + 16;
            std::memcpy(&wp[pm * D], &w[(size_t)f * D], D * sizeof(float));
            std::memcpy(&wp[pp * D], &w[(size_t)(bins + f) * D], D * sizeof(float));
            bp[pm] = b[f];
            bp[pp] = b[bins + f];
        }
        if (int rc = upload(M, wp, &M->head_W)) return rc;
        if (int rc = upload(M, bp, &M->head_b)) return rc;
        // Inverse real DFT, two radix-2 splits then dense: with theta = 2 pi f n / N,
        //   x[n] = C[n] - S[n], x[N-n] = C[n] + S[n]            (n <= N/2;  C = sum c_f Re cos, S = sum c_f Im sin, /N)
        //   C[n] = Ce[n] + Co[n], C[N/2-n] = Ce[n] - Co[n]      (n <= N/4;  even / odd bins)
        //   S[n] = Se[n] + So[n], S[N/2-n] = So[n] - Se[n]
        // so four (N/4+1) x (N/4+1) bases replace the N x (N/2+1) complex one: 1/4 of the multiply-adds.
        std::vector<float> basis((size_t)4 * Kq * Kq, 0.f);
        const double two_pi = 6.283185307179586476925286766559;
        for (int n = 0; n <= Q; ++n)
            for (int g = 0; g <= Q; ++g) {
                const int fe = 2 * g, fo = 2 * g + 1;
                const bool edge = (fe == 0) || (fe == N / 2);
                const double ce = (edge ? 1.0 : 2.0) / (double)N, co = 2.0 / (double)N;
                const double the = two_pi * (double)(((long)fe * n) % N) / (double)N;
                const double tho = two_pi * (double)(((long)fo * n) % N) / (double)N;
                basis[((size_t)0 * Kq + n) * Kq + g] = (float)(ce * std::cos(the));
                basis[((size_t)2 * Kq + n) * Kq + g] = edge ? 0.f : (float)(ce * std::sin(the));   // C2R ignores Im of DC/Nyquist
                if (g < Q) {
                    basis[((size_t)1 * Kq + n) * Kq + g] = (float)(co * std::cos(tho));
                    basis[((size_t)3 * Kq + n) * Kq + g] = (float)(co * std::sin(tho));
                }
            }
        if (int rc = upload(M, basis, &M->istft_W)) return rc;
        if (int rc = upload_raw(M, win, N, &M->win)) return rc;
        std::vector<float> wsq(N);
        for (int n = 0; n < N; ++n) wsq[n] = win[n] * win[n];
        if (int rc = upload(M, wsq, &M->wsq)) return rc;
    }

    // ---- optional SEANetDecoder (encoder/modules/seanet.py:147-238)
    M->has_seadec = tm.has(std::string(DEC) + "0.conv.conv.weight_v");
    if (M->has_seadec) {
        int m2 = 1 << a.n_ratios;
        if (int rc = load_wn_conv(M, tm, std::string(DEC) + "0.conv.conv", m2 * nf, 512, 7, &M->sd_first)) return rc;
        if (int rc = load_lstm(M, tm, std::string(DEC) + "1", H, &M->sd_lstm, &M->sd_s32_ok)) return rc;
        int di = 2;
        for (int i = 0; i < a.n_ratios; ++i) {
            const int r = a.ratios[i];
            SeaDecStage st;
            st.cin = m2 * nf; st.cout = st.cin / 2; st.k = 2 * r; st.r = r;
            const std::string p = std::string(DEC) + std::to_string(di + 1) + ".convtr.convtr";
            const float* g = tm.get(p + ".weight_g", st.cin);
            const float* v = tm.get(p + ".weight_v", (int64_t)st.cin * st.cout * st.k);
            const float* b = tm.get(p + ".bias", st.cout);
            if (!g || !v || !b) return WT_ERR_MISSING_TENSOR;
            std::vector<float> w = fold_wn(g, v, st.cin, (int64_t)st.cout * st.k);   // [cin][cout][k], g per cin
            std::vector<float> pk((size_t)st.k * st.cin * st.cout);
            for (int ci = 0; ci < st.cin; ++ci)
                for (int co = 0; co < st.cout; ++co)
                    for (int j = 0; j < st.k; ++j)
                        pk[((size_t)j * st.cin + ci) * st.cout + co] = w[((size_t)ci * st.cout + co) * st.k + j];
            if (int rc = upload(M, pk, &st.tr_w)) return rc;
            if (st.k == 2 * r) {      // every output sample has exactly two contributing frames -> one GEMM per phase
                std::vector<float> pp((size_t)r * st.cout * 2 * st.cin);
                for (int ph = 0; ph < r; ++ph)
                    for (int co = 0; co < st.cout; ++co)
                        for (int ci = 0; ci < st.cin; ++ci) {
                            pp[(((size_t)ph * st.cout + co) * 2 + 0) * st.cin + ci] = w[((size_t)ci * st.cout + co) * st.k + ph + r];
                            pp[(((size_t)ph * st.cout + co) * 2 + 1) * st.cin + ci] = w[((size_t)ci * st.cout + co) * st.k + ph];
                        }
                if (int rc = upload(M, pp, &st.tr_wp)) return rc;
            }
            if (int rc = upload_raw(M, b, st.cout, &st.tr_b)) return rc;
            const std::string rp = std::string(DEC) + std::to_string(di + 2);
            const int h = st.cout;
            if (int rc = load_wn_conv(M, tm, rp + ".block.1.conv.conv", h / 2, h, 3, &st.c3)) return rc;
            if (int rc = load_wn_conv(M, tm, rp + ".block.3.conv.conv", h, h / 2, 1, &st.c1)) return rc;
            if (int rc = load_wn_conv(M, tm, rp + ".shortcut.conv.conv", h, h, 1, &st.sc)) return rc;
            if (!resblock_fusable(h)) if (int rc = build_cat(M, st.sc, st.c1, &st.cat)) return rc;
            M->sd_stages.push_back(st);
            di += 3; m2 /= 2;
        }
        const std::string p = std::string(DEC) + std::to_string(di + 1) + ".conv.conv";
        const float* g = tm.get(p + ".weight_g", 1);
        const float* v = tm.get(p + ".weight_v", (int64_t)nf * 7);
        const float* b = tm.get(p + ".bias", 1);
        if (!g || !v || !b) return WT_ERR_MISSING_TENSOR;
        std::vector<float> w = fold_wn(g, v, 1, (int64_t)nf * 7);   // [1][32][7]
        std::vector<float> pk((size_t)7 * nf);
        for (int c = 0; c < nf; ++c)
            for (int j = 0; j < 7; ++j) pk[(size_t)j * nf + c] = w[(size_t)c * 7 + j];
        if (int rc = upload(M, pk, &M->sd_last_w)) return rc;
        if (int rc = upload_raw(M, b, 1, &M->sd_last_b)) return rc;
    }
    return 0;
}

// S32 copy of a GEMM weight.  hi = f16(w) covers |w| < 65504 with fp32-equivalent products only while the tensor's
// largest magnitude is not far from 1 (x = hi + lo * 2^-11 has an absolute floor of 2^-36), so a tensor whose maximum
// lies outside [2^-6, 2^12] is stored as w * 2^e (maximum brought into [1, 2)) and its GEMMs multiply their
// accumulators by 2^-e (GemmArgs::acc_scale; powers of two: exact).  A non-finite weight cannot be split at all:
// the model then runs on the fp32 MFMA chain (wt_model::s32_ok).
static int add_s32(wt_model* M, const float* w, long n, bool* split_ok = nullptr, bool gemm_only = false) {
    if (!w || n <= 0 || (n % 32)) return 0;
    if (!split_ok) split_ok = &M->s32_ok;
    std::vector<float> h((size_t)n);
    WT_HIP_CHECK(hipMemcpy(h.data(), w, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    float amax = 0.f;
    bool finite = true;
    for (long i = 0; i < n; ++i) {
        const float v = std::fabs(h[i]);
        if (!(v <= 3.0e38f)) finite = false;
        else amax = std::max(amax, v);
    }
    if (!finite) *split_ok = false;
    M->w_amax = std::max(M->w_amax, amax);
    float scale = 1.f;
    if (finite && amax > 0.f && (amax < 0x1p-6f || amax >= 0x1p12f)) {
        int e = 0;
        (void)std::frexp(amax, &e);                 // amax = m * 2^e, m in [0.5, 1)
        scale = std::ldexp(1.f, 1 - e);             // amax * scale in [1, 2)
    }
    float* scale_dev = nullptr;
    if (scale != 1.f) {
        if (int rc = upload(M, std::vector<float>{scale}, &scale_dev)) return rc;
        M->s32_acc_scale[w] = 1.f / scale;
    }
    void* d = nullptr;
    WT_HIP_CHECK(hipMalloc(&d, (size_t)n * 4));
    M->allocs.push_back(d);
    M->alloc_bytes.push_back((size_t)n * 4);
    M->weight_bytes += n * 4;
    if (int rc = launch_split_s32(w, d, n, nullptr, scale_dev)) return rc;
    M->s32[w] = d;
    // gemm_only: nothing but a GEMM reads this fp32 array, and the default plans multiply by the S32 copy (wt_model::lazy_f32)
    if (gemm_only && finite) M->lazy_f32.push_back({w, d, (int64_t)n, scale});
    return 0;
}

int build_splits(wt_model* M) {
    const wt_arch& a = M->arch;
    const int D = a.dim, I = a.intermediate_dim;
    auto conv32 = [&](const ConvW& c, bool gemm_only = true) { return (c.cin % 32) ? 0 : add_s32(M, c.w, (long)c.cout * c.k * c.cin, nullptr, gemm_only); };
    // the SEANetDecoder's weights are all zeros -> NaN after the weight-norm fold when a checkpoint without them was
    // loaded into the full module tree: they only decide how the SEANetDecoder plan runs
    auto conv32sd = [&](const ConvW& c) { return (c.cin % 32) ? 0 : add_s32(M, c.w, (long)c.cout * c.k * c.cin, &M->sd_s32_ok, true); };
    for (const ResStage& st : M->stages) {      // encoder chain on S32 operands (build_encode)
        if (st.down.cin % 32 == 0 && st.down.k == 2 * st.r && st.down.k <= 32) {
            // k = 2 * stride: every input frame feeds two output frames (taps j and j + stride).  Packing the taps as
            // (0, r, 1, r+1, ...) puts those two reads in adjacent K steps
            const long n = (long)st.down.cout * st.down.k * st.down.cin;
            std::vector<float> h((size_t)n), pk((size_t)n);
            WT_HIP_CHECK(hipMemcpy(h.data(), st.down.w, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
            for (int co = 0; co < st.down.cout; ++co)
                for (int q = 0; q < st.down.k; ++q) {
                    const int tap = (q >> 1) + (q & 1) * st.r;
                    std::memcpy(&pk[((size_t)co * st.down.k + q) * st.down.cin], &h[((size_t)co * st.down.k + tap) * st.down.cin],
                                st.down.cin * sizeof(float));
                }
            float* dpk = nullptr;
            if (int rc = upload(M, pk, &dpk)) return rc;
            if (int rc = add_s32(M, dpk, n)) return rc;
            M->lazy_f32.push_back({dpk, nullptr, (int64_t)n, 1.f});      // only the split above ever read the repacked copy
            M->s32[st.down.w] = M->s32.at(dpk);
            if (M->s32_acc_scale.count(dpk)) M->s32_acc_scale[st.down.w] = M->s32_acc_scale.at(dpk);
            M->s32_tap_pair[st.down.w] = true;
        } else
        if (int rc = conv32(st.down, false)) return rc;
        // the fused resblock kernels (C = 32, 64) read their conv weights in fp32 and split them themselves
        const bool fused = resblock_fusable(st.C);
        if (int rc = conv32(st.c3, !fused)) return rc;
        if (int rc = conv32(st.c1, !fused)) return rc;
        if (int rc = conv32(st.sc, !fused)) return rc;
        if (st.cat.w) if (int rc = conv32(st.cat)) return rc;
    }
    if (int rc = add_s32(M, M->enc_lstm.Wih0, 4L * M->H * M->H, nullptr, true)) return rc;
    if (int rc = conv32(M->enc_final)) return rc;
    if (int rc = add_s32(M, M->embed, (long)a.vq_bins * 512)) return rc;        // (the gathers read the fp32 codebook)
    if (int rc = conv32(M->bb_embed)) return rc;
    for (int i = 0; i < 4; ++i) {
        if (int rc = conv32(M->res[i].c1)) return rc;
        if (int rc = conv32(M->res[i].c2)) return rc;
    }
    for (const CnxBlock& c : M->cnx) {
        if (int rc = add_s32(M, c.W1, (long)I * D, nullptr, true)) return rc;
        if (int rc = add_s32(M, c.W2, (long)D * I, nullptr, true)) return rc;
    }
    if (int rc = add_s32(M, M->head_W, 2L * M->Kb * D, nullptr, true)) return rc;
    if (int rc = add_s32(M, M->istft_W, 4L * M->Kq * M->Kq, nullptr, true)) return rc;
    if (int rc = add_s32(M, M->at_Wqk, 2L * D * D, nullptr, true)) return rc;
    if (int rc = add_s32(M, M->at_Wv, (long)D * D, nullptr, true)) return rc;
    if (int rc = add_s32(M, M->at_Wp, (long)D * D, nullptr, true)) return rc;
    if (M->has_seadec) {
        if (int rc = conv32sd(M->sd_first)) return rc;
        if (int rc = add_s32(M, M->sd_lstm.Wih0, 4L * M->H * M->H, &M->sd_s32_ok, true)) return rc;
        for (const SeaDecStage& st : M->sd_stages) {
            if (st.tr_wp && st.cin % 16 == 0) if (int rc = add_s32(M, st.tr_wp, (long)st.r * st.cout * 2 * st.cin, &M->sd_s32_ok, true)) return rc;
            if (resblock_fusable(st.cout)) continue;            // its convs run inside resblock16
            if (int rc = conv32sd(st.sc)) return rc;
            if (int rc = conv32sd(st.c3)) return rc;
            if (int rc = conv32sd(st.c1)) return rc;
            if (st.cat.w) if (int rc = conv32sd(st.cat)) return rc;
        }
    }
    WT_HIP_CHECK(hipDeviceSynchronize());
    return 0;
}


// ------------------------------------------------------------------------------------ packed image
// a packed on-disk weight format "folded, pre-tiled".  The image is what wt_model_create leaves in HBM —
// folded conv weights in [Cout][tap][Cin], LSTM gate-row packings, the packed head and inverse-DFT basis, the S32 and
// f16x2 split copies with their per-tensor scales — as the list of device allocations in creation order, preceded by a
// header (magic, version, the wt_arch and a hash of it) and the model struct with every pointer written as
// (allocation index).  Loading it is allocate + upload + fix up pointers: nothing is folded, packed or split again.
static constexpr uint32_t PACK_MAGIC = 0x4b505457u;     // "WTPK"
static constexpr int32_t PACK_VERSION = 6;

static uint64_t arch_hash_of(const wt_arch& a) {        // FNV-1a over the architecture struct and the layout version
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) { for (size_t i = 0; i < n; ++i) { h ^= static_cast<const unsigned char*>(p)[i]; h *= 1099511628211ull; } };
    mix(&a, sizeof(a));
    mix(&PACK_VERSION, sizeof(PACK_VERSION));
    return h;
}

struct Archive {
    bool saving;
    std::vector<char>* out = nullptr;          // saving
    const char* in = nullptr;                  // loading
    size_t pos = 0, n = 0;
    bool ok = true;
    const std::map<const void*, int>* index = nullptr;      // saving: allocation base -> index
    const std::vector<void*>* allocs = nullptr;             // loading: index -> allocation base
    const std::vector<size_t>* alloc_bytes = nullptr;       // loading: index -> allocation size (pointer offsets are checked)
    void raw(void* p, size_t bytes) {
        if (saving) { const char* c = static_cast<const char*>(p); out->insert(out->end(), c, c + bytes); }
        else { if (!ok || bytes > n - pos) { ok = false; std::memset(p, 0, bytes); return; } std::memcpy(p, in + pos, bytes); pos += bytes; }
    }
    // an element count read from the file: bounded before anything is resized by it
    int32_t count(size_t have, int32_t limit) {
        int32_t c = (int32_t)have;
        pod(c);
        if (!saving && (c < 0 || c > limit)) { ok = false; c = 0; }
        return c;
    }
    template <class T> void pod(T& v) { raw(&v, sizeof(T)); }
    template <class T> void ptr(T*& p) {       // a device pointer = (allocation index, byte offset); -1 = null
        int32_t idx = -1;
        int64_t off = 0;
        if (saving && p) {
            auto it = index->upper_bound(static_cast<const void*>(p));
            if (it == index->begin()) { ok = false; }
            else { --it; idx = it->second; off = reinterpret_cast<const char*>(p) - static_cast<const char*>(it->first); }
        }
        pod(idx); pod(off);
        if (!saving) {
            if (idx < 0) p = nullptr;
            else if (!ok || idx >= (int)allocs->size() || off < 0 || (size_t)off >= (*alloc_bytes)[idx]) { ok = false; p = nullptr; }
            else p = reinterpret_cast<T*>(static_cast<char*>((*allocs)[idx]) + off);
        }
    }
    void conv(ConvW& c) { ptr(c.w); ptr(c.b); pod(c.cout); pod(c.cin); pod(c.k); }
    void lstm(LstmW& l) { ptr(l.Wih0); ptr(l.b0); ptr(l.W0); ptr(l.W1); ptr(l.b1); ptr(l.W0h); ptr(l.W1h); ptr(l.Wp); }
};

static void archive_model(Archive& ar, wt_model* M) {
    ar.pod(M->hop); ar.pod(M->H); ar.pod(M->weight_bytes); ar.pod(M->s32_ok); ar.pod(M->sd_s32_ok); ar.pod(M->w_amax);
    int32_t n = ar.count(M->enc_ratios.size(), 8);
    M->enc_ratios.resize(n);
    for (int& r : M->enc_ratios) ar.pod(r);
    ar.ptr(M->e0_w); ar.ptr(M->e0_b); ar.pod(M->e0_k); ar.pod(M->e0_c);
    n = ar.count(M->stages.size(), 8); M->stages.resize(n);
    for (ResStage& st : M->stages) { ar.conv(st.c3); ar.conv(st.c1); ar.conv(st.sc); ar.conv(st.down); ar.conv(st.cat); ar.pod(st.C); ar.pod(st.r); }
    ar.lstm(M->enc_lstm); ar.conv(M->enc_final); ar.ptr(M->embed); ar.ptr(M->ee);
    ar.conv(M->bb_embed);
    for (PosRes& r : M->res) { ar.ptr(r.n1w); ar.ptr(r.n1b); ar.ptr(r.n2w); ar.ptr(r.n2b); ar.conv(r.c1); ar.conv(r.c2); }
    ar.ptr(M->at_nw); ar.ptr(M->at_nb); ar.ptr(M->at_Wqk); ar.ptr(M->at_bqk); ar.ptr(M->at_Wv); ar.ptr(M->at_bv); ar.ptr(M->at_Wp); ar.ptr(M->at_bp);
    ar.ptr(M->gn5w); ar.ptr(M->gn5b); ar.ptr(M->ada_s); ar.ptr(M->ada_h);
    n = ar.count(M->cnx.size(), 64); M->cnx.resize(n);
    for (CnxBlock& c : M->cnx) { ar.ptr(c.dw_w); ar.ptr(c.dw_b); ar.ptr(c.ada_s); ar.ptr(c.ada_h); ar.ptr(c.W1); ar.ptr(c.b1); ar.ptr(c.W2); ar.ptr(c.b2); ar.ptr(c.gamma); }
    ar.ptr(M->fln_w); ar.ptr(M->fln_b); ar.ptr(M->head_W); ar.ptr(M->head_b);
    ar.pod(M->Kb); ar.pod(M->Kq); ar.pod(M->bins_f); ar.pod(M->R);
    ar.ptr(M->istft_W); ar.ptr(M->wsq); ar.ptr(M->win);
    ar.pod(M->has_seadec); ar.conv(M->sd_first); ar.lstm(M->sd_lstm);
    n = ar.count(M->sd_stages.size(), 8); M->sd_stages.resize(n);
    for (SeaDecStage& st : M->sd_stages) {
        ar.ptr(st.tr_w); ar.ptr(st.tr_wp); ar.ptr(st.tr_b); ar.pod(st.cin); ar.pod(st.cout); ar.pod(st.k); ar.pod(st.r);
        ar.conv(st.c3); ar.conv(st.c1); ar.conv(st.sc); ar.conv(st.cat);
    }
    ar.ptr(M->sd_last_w); ar.ptr(M->sd_last_b);
    // the maps keyed by the fp32 weight pointer
    auto map_ptr = [&](auto& m, auto value_io) {
        const int32_t cnt = ar.count(m.size(), 1 << 16);
        if (ar.saving) {
            // in allocation order, not in address order: the image of a model does not depend on where hipMalloc put it
            std::vector<const float*> keys;
            for (auto& kv : m) keys.push_back(kv.first);
            auto alloc_of = [&](const float* k) { auto it = ar.index->upper_bound(static_cast<const void*>(k)); return it == ar.index->begin() ? -1 : std::prev(it)->second; };
            std::sort(keys.begin(), keys.end(), [&](const float* a, const float* b) { return alloc_of(a) < alloc_of(b); });
            for (const float* k0 : keys) { const float* k = k0; ar.ptr(k); value_io(m.at(k0)); }
        } else {
            for (int i = 0; i < cnt && ar.ok; ++i) { const float* k = nullptr; ar.ptr(k); auto& v = m[k]; value_io(v); }
        }
    };
    map_ptr(M->s32, [&](void*& v) { ar.ptr(v); });
    n = ar.count(M->lazy_f32.size(), 1 << 12); M->lazy_f32.resize(n);
    for (wt_model::LazyF32& z : M->lazy_f32) { ar.ptr(z.w); ar.ptr(z.s32); ar.pod(z.n); ar.pod(z.scale); }
    map_ptr(M->s32_tap_pair, [&](bool& v) { ar.pod(v); });
    map_ptr(M->s32_acc_scale, [&](float& v) { ar.pod(v); });
}

struct PackHeader {
    uint32_t magic;
    int32_t version;
    wt_arch arch;
    uint64_t arch_hash;
    uint64_t n_allocs, struct_bytes, payload_bytes;
    uint64_t body_hash;        // over the allocation table, the model section and the payload (everything behind the header)
};

// 64-bit hash of a byte range, four interleaved multiply-xor lanes over 8-byte words (a serial FNV over ~0.7 GB would
// take longer than the upload itself); the tail bytes and the length go into lane 0
static uint64_t body_hash_of(const char* p, size_t n) {
    uint64_t h[4] = {0x9e3779b97f4a7c15ull, 0xc2b2ae3d27d4eb4full, 0x165667b19e3779f9ull, 0x27d4eb2f165667c5ull};
    const uint64_t K = 0x100000001b3ull * 0x9e3779b1ull | 1ull;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        std::memcpy(w, p + i, 32);
        for (int l = 0; l < 4; ++l) { h[l] = (h[l] ^ w[l]) * K; h[l] ^= h[l] >> 29; }
    }
    for (; i < n; ++i) { h[0] = (h[0] ^ (unsigned char)p[i]) * K; h[0] ^= h[0] >> 29; }
    h[0] = (h[0] ^ (uint64_t)n) * K;
    return (h[0] ^ (h[1] * 3) ^ (h[2] * 5) ^ (h[3] * 7)) * K;
}

// allocation table entry of an array the image does not store (wt_model::lazy_f32): its size with this bit set
static constexpr uint64_t PACK_NOT_STORED = 1ull << 63;

static std::vector<bool> lazy_allocs(const wt_model* M) {
    std::vector<bool> lazy(M->allocs.size(), false);
    for (const wt_model::LazyF32& z : M->lazy_f32)
        for (size_t i = 0; i < M->allocs.size(); ++i)
            if (M->allocs[i] == static_cast<const void*>(z.w)) lazy[i] = true;       // lazy arrays are whole allocations
    return lazy;
}

size_t model_export_bytes(const wt_model* M) {
    size_t total = sizeof(PackHeader) + M->allocs.size() * sizeof(uint64_t) + (1u << 17);      // struct section: generous bound
    const std::vector<bool> lazy = lazy_allocs(M);
    for (size_t i = 0; i < M->allocs.size(); ++i) if (!lazy[i]) total += (M->alloc_bytes[i] + 255) / 256 * 256;
    return total + 256;
}

int model_export(const wt_model* Mc, void* buf, size_t n) {
    wt_model* M = const_cast<wt_model*>(Mc);         // archive_model is symmetric; saving does not modify
    std::map<const void*, int> index;
    for (size_t i = 0; i < M->allocs.size(); ++i) index[M->allocs[i]] = (int)i;
    std::vector<char> st;
    Archive ar;
    ar.saving = true; ar.out = &st; ar.index = &index;
    archive_model(ar, M);
    if (!ar.ok) { set_error("wt_model_export: a model pointer lies outside the model's allocations"); return WT_ERR_INVALID; }
    PackHeader h{};
    h.magic = PACK_MAGIC; h.version = PACK_VERSION; h.arch = M->arch; h.arch_hash = arch_hash_of(M->arch);
    h.n_allocs = M->allocs.size(); h.struct_bytes = st.size();
    if (M->f32_stale.load()) if (int rc = ensure_f32_weights(M)) return rc;       // (nothing lazy is exported, but keep the model whole)
    const std::vector<bool> lazy = lazy_allocs(M);
    size_t pos = sizeof(PackHeader) + M->allocs.size() * sizeof(uint64_t) + st.size();
    pos = (pos + 255) / 256 * 256;
    const size_t payload0 = pos;
    for (size_t i = 0; i < M->allocs.size(); ++i) if (!lazy[i]) pos += (M->alloc_bytes[i] + 255) / 256 * 256;
    h.payload_bytes = pos - payload0;
    if (pos > n) { set_error("wt_model_export: buffer too small (wt_model_export_bytes)"); return WT_ERR_INVALID; }
    char* o = static_cast<char*>(buf);
    std::memset(o, 0, payload0);
    for (size_t i = 0; i < M->allocs.size(); ++i) { uint64_t b = M->alloc_bytes[i] | (lazy[i] ? PACK_NOT_STORED : 0); std::memcpy(o + sizeof(h) + i * 8, &b, 8); }
    std::memcpy(o + sizeof(h) + M->allocs.size() * 8, st.data(), st.size());
    pos = payload0;
    WT_HIP_CHECK(hipDeviceSynchronize());
    for (size_t i = 0; i < M->allocs.size(); ++i) {
        if (lazy[i]) continue;
        const size_t b = M->alloc_bytes[i], padded = (b + 255) / 256 * 256;
        WT_HIP_CHECK(hipMemcpy(o + pos, M->allocs[i], b, hipMemcpyDeviceToHost));
        std::memset(o + pos + b, 0, padded - b);
        pos += padded;
    }
    h.body_hash = body_hash_of(o + sizeof(h), pos - sizeof(h));
    std::memcpy(o, &h, sizeof(h));
    return (int)0;
}

int packed_info(const void* buf, size_t n, wt_arch* arch, int32_t* version, uint64_t* arch_hash) {
    if (!buf || n < sizeof(PackHeader)) { set_error("packed image: too short"); return WT_ERR_INVALID; }
    PackHeader h;
    std::memcpy(&h, buf, sizeof(h));
    if (h.magic != PACK_MAGIC) { set_error("packed image: bad magic (not a wavtokenizer_amd packed file)"); return WT_ERR_INVALID; }
    if (version) *version = h.version;
    if (arch) *arch = h.arch;
    if (arch_hash) *arch_hash = h.arch_hash;
    if (h.version != PACK_VERSION) { set_error("packed image: layout version " + std::to_string(h.version) + ", this library reads " + std::to_string(PACK_VERSION)); return WT_ERR_INVALID; }
    if (h.arch_hash != arch_hash_of(h.arch)) { set_error("packed image: architecture hash mismatch (corrupt header)"); return WT_ERR_INVALID; }
    // every term is checked against what is left of the file before it is added: nothing here can wrap
    size_t left = n - sizeof(PackHeader);
    if (h.n_allocs > (1u << 16) || h.n_allocs * 8 > left) { set_error("packed image: truncated (allocation table)"); return WT_ERR_INVALID; }
    left -= h.n_allocs * 8;
    if (h.struct_bytes > (1u << 24) || h.struct_bytes > left) { set_error("packed image: truncated (model section)"); return WT_ERR_INVALID; }
    const size_t head = sizeof(PackHeader) + h.n_allocs * 8 + h.struct_bytes, head_pad = (head + 255) / 256 * 256;
    if (head_pad > n || h.payload_bytes > n - head_pad) { set_error("packed image: truncated (payload)"); return WT_ERR_INVALID; }
    return WT_OK;
}

// length of the image that starts at buf (wt_model_export_bytes is an upper bound taken before the model section is written)
size_t packed_bytes(const void* buf, size_t n) {
    if (packed_info(buf, n, nullptr, nullptr, nullptr)) return 0;
    PackHeader h;
    std::memcpy(&h, buf, sizeof(h));
    return (sizeof(PackHeader) + h.n_allocs * 8 + h.struct_bytes + 255) / 256 * 256 + h.payload_bytes;
}

// full check of an image that needs no GPU: header, table and section bounds, and the hash of everything behind the header
int packed_verify(const void* buf, size_t n) {
    if (int rc = packed_info(buf, n, nullptr, nullptr, nullptr)) return rc;
    PackHeader h;
    std::memcpy(&h, buf, sizeof(h));
    const char* in = static_cast<const char*>(buf);
    const size_t head_pad = (sizeof(PackHeader) + h.n_allocs * 8 + h.struct_bytes + 255) / 256 * 256;
    size_t pos = head_pad;
    for (size_t i = 0; i < h.n_allocs; ++i) {
        uint64_t b;
        std::memcpy(&b, in + sizeof(h) + i * 8, 8);
        if (b & PACK_NOT_STORED) {               // allocated on import, filled in later (wt_model::lazy_f32): bounded, no payload
            if ((b & ~PACK_NOT_STORED) == 0 || (b & ~PACK_NOT_STORED) > (1ull << 32)) { set_error("packed image: bad size of an unstored array"); return WT_ERR_INVALID; }
            continue;
        }
        if (b == 0 || b > n - pos || (b + 255) / 256 * 256 > n - pos) { set_error("packed image: allocation table does not fit the payload"); return WT_ERR_INVALID; }
        pos += (b + 255) / 256 * 256;
    }
    if (pos - head_pad != h.payload_bytes) { set_error("packed image: allocation table and payload size disagree"); return WT_ERR_INVALID; }
    if (body_hash_of(in + sizeof(h), pos - sizeof(h)) != h.body_hash) { set_error("packed image: content hash mismatch (corrupt file)"); return WT_ERR_INVALID; }
    return WT_OK;
}

// The model section of a packed image is data from a file: after archive_model every dimension is recomputed from the
// header's wt_arch (what build_model would have set) and compared, and every pointer's EXTENT - not only its start - must lie
// inside the allocation it points into.  The body hash only detects accidental corruption (it is not cryptographic): a
// crafted file must not be able to make a plan size its launches past an allocation.
static int validate_imported(const wt_model* M) {
    const wt_arch& a = M->arch;
    auto fits = [&](const void* p, size_t bytes) {
        if (!p) return false;
        const char* c = static_cast<const char*>(p);
        for (size_t i = 0; i < M->allocs.size(); ++i) {
            const char* base = static_cast<const char*>(M->allocs[i]);
            if (c >= base && c < base + M->alloc_bytes[i]) return bytes <= (size_t)(base + M->alloc_bytes[i] - c);
        }
        return false;
    };
    auto bad = [](const char* what) { set_error(std::string("packed image: ") + what + " does not match the architecture in its header"); return WT_ERR_INVALID; };
    auto conv_ok = [&](const ConvW& c, int cout, int cin, int k) {
        return c.cout == cout && c.cin == cin && c.k == k && fits(c.w, (size_t)cout * cin * k * 4) && fits(c.b, (size_t)cout * 4);
    };
    auto lstm_ok = [&](const LstmW& l, int H) {
        const size_t hh = (size_t)4 * H * H * 4;
        return fits(l.Wih0, hh) && fits(l.b0, (size_t)16 * H) && fits(l.W0, hh) && fits(l.W1, 2 * hh) && fits(l.b1, (size_t)16 * H) &&
               (!l.W0h || fits(l.W0h, hh)) && (!l.W1h || fits(l.W1h, 2 * hh)) &&
               (!l.Wp || fits(l.Wp, (size_t)3 * 32 * 4 * 16 * 2 * 64 * 16));
    };
    const int nf = 32, H = 512, D = a.dim, I = a.intermediate_dim;
    if (a.n_ratios < 1 || a.n_ratios > 8 || a.num_quantizers < 1 || a.num_quantizers > 32 || a.input_channels != 512 || D % 256 || I % 32 || a.num_layers < 1 || a.num_layers > 32 ||
        a.adanorm_num_embeddings < 1 || a.vq_bins < 1 || a.n_fft < 4 || a.hop_length < 1) return bad("the architecture itself");
    int hop = 1;
    for (int i = 0; i < a.n_ratios; ++i) { if (a.ratios[i] < 1 || a.ratios[i] > 64) return bad("a ratio"); hop *= a.ratios[i]; }
    if (M->hop != hop || M->H != H || (int)M->enc_ratios.size() != a.n_ratios || (int)M->stages.size() != a.n_ratios) return bad("the encoder's shape");
    if (M->e0_k != 7 || M->e0_c != nf || !fits(M->e0_w, 7 * nf * 4) || !fits(M->e0_b, nf * 4)) return bad("the first conv");
    int mult = 1;
    for (int i = 0; i < a.n_ratios; ++i) {
        const ResStage& st = M->stages[i];
        const int r = a.ratios[a.n_ratios - 1 - i], C = mult * nf;
        if (M->enc_ratios[i] != r || st.C != C || st.r != r || !conv_ok(st.c3, C / 2, C, 3) || !conv_ok(st.c1, C, C / 2, 1) || !conv_ok(st.sc, C, C, 1) ||
            !conv_ok(st.down, 2 * C, C, 2 * r)) return bad("an encoder stage");
        if (st.cat.w && !(st.cat.cout == C && st.cat.cin == C + C / 2 && st.cat.k == 1 && fits(st.cat.w, (size_t)C * (C + C / 2) * 4) && fits(st.cat.b, (size_t)C * 4)))
            return bad("a fused shortcut weight");
        mult *= 2;
    }
    if (mult * nf != H || !lstm_ok(M->enc_lstm, H) || !conv_ok(M->enc_final, 512, H, 7)) return bad("the encoder tail");
    if (!fits(M->embed, (size_t)a.num_quantizers * a.vq_bins * 512 * 4) || !fits(M->ee, (size_t)a.vq_bins * 4)) return bad("the codebook");
    if (!conv_ok(M->bb_embed, D, 512, 7)) return bad("backbone.embed");
    for (const PosRes& r : M->res)
        if (!fits(r.n1w, D * 4) || !fits(r.n1b, D * 4) || !fits(r.n2w, D * 4) || !fits(r.n2b, D * 4) || !conv_ok(r.c1, D, D, 3) || !conv_ok(r.c2, D, D, 3)) return bad("a pos_net block");
    if (!fits(M->at_nw, D * 4) || !fits(M->at_nb, D * 4) || !fits(M->at_Wqk, (size_t)2 * D * D * 4) || !fits(M->at_bqk, 2 * D * 4) || !fits(M->at_Wv, (size_t)D * D * 4) ||
        !fits(M->at_bv, D * 4) || !fits(M->at_Wp, (size_t)D * D * 4) || !fits(M->at_bp, D * 4)) return bad("the attention block");
    const size_t ada = (size_t)a.adanorm_num_embeddings * D * 4;
    if (!fits(M->gn5w, D * 4) || !fits(M->gn5b, D * 4) || !fits(M->ada_s, ada) || !fits(M->ada_h, ada)) return bad("backbone.norm");
    if ((int)M->cnx.size() != a.num_layers) return bad("the ConvNeXt block count");
    for (const CnxBlock& c : M->cnx)
        if (!fits(c.dw_w, (size_t)7 * D * 4) || !fits(c.dw_b, D * 4) || !fits(c.ada_s, ada) || !fits(c.ada_h, ada) || !fits(c.W1, (size_t)I * D * 4) || !fits(c.b1, I * 4) ||
            !fits(c.W2, (size_t)D * I * 4) || !fits(c.b2, D * 4) || !fits(c.gamma, D * 4)) return bad("a ConvNeXt block");
    const int N = a.n_fft;
    if (N % 4 || N % a.hop_length) return bad("n_fft / hop_length");
    const int Kq = ((N / 4 + 1 + 31) / 32) * 32, Kb = 2 * Kq;
    if (M->Kq != Kq || M->Kb != Kb || M->bins_f != N / 2 + 1 || M->R != N / a.hop_length) return bad("the ISTFT head's shape");
    if (!fits(M->fln_w, D * 4) || !fits(M->fln_b, D * 4) || !fits(M->head_W, (size_t)2 * Kb * D * 4) || !fits(M->head_b, (size_t)2 * Kb * 4) ||
        !fits(M->istft_W, (size_t)4 * Kq * Kq * 4) || !fits(M->win, (size_t)N * 4) || !fits(M->wsq, (size_t)N * 4)) return bad("the ISTFT head");
    if (M->has_seadec) {
        if ((int)M->sd_stages.size() != a.n_ratios || !conv_ok(M->sd_first, (1 << a.n_ratios) * nf, 512, 7) || !lstm_ok(M->sd_lstm, H)) return bad("the SEANetDecoder");
        int m2 = 1 << a.n_ratios;
        for (int i = 0; i < a.n_ratios; ++i) {
            const SeaDecStage& st = M->sd_stages[i];
            const int r = a.ratios[i], cin = m2 * nf, h = cin / 2;
            if (st.cin != cin || st.cout != h || st.k != 2 * r || st.r != r || !fits(st.tr_w, (size_t)2 * r * cin * h * 4) || (st.tr_wp && !fits(st.tr_wp, (size_t)2 * r * cin * h * 4)) ||
                !fits(st.tr_b, h * 4) || !conv_ok(st.c3, h / 2, h, 3) || !conv_ok(st.c1, h, h / 2, 1) || !conv_ok(st.sc, h, h, 1)) return bad("a SEANetDecoder stage");
            if (st.cat.w && !(st.cat.cout == h && st.cat.cin == h + h / 2 && fits(st.cat.w, (size_t)h * (h + h / 2) * 4) && fits(st.cat.b, (size_t)h * 4))) return bad("a SEANetDecoder fused shortcut");
            m2 /= 2;
        }
        if (!fits(M->sd_last_w, 7 * nf * 4) || !fits(M->sd_last_b, 4)) return bad("the SEANetDecoder's last conv");
    }
    // S32 copies: same footprint as the fp32 array they stand for; keys must be weights the plans look up (checked by extent)
    for (const auto& kv : M->s32) if (!kv.first || !kv.second || !fits(kv.first, 128) || !fits(kv.second, 128)) return bad("an S32 copy");
    for (const auto& kv : M->s32_acc_scale) if (!(kv.second > 0.f) || !std::isfinite(kv.second)) return bad("an S32 scale");
    return WT_OK;
}

int model_import(wt_model* M, const void* buf, size_t n) {
    PackHeader h;
    if (int rc = packed_verify(buf, n)) return rc;
    std::memcpy(&h, buf, sizeof(h));
    M->arch = h.arch;
    const char* in = static_cast<const char*>(buf);
    size_t pos = (sizeof(PackHeader) + h.n_allocs * 8 + h.struct_bytes + 255) / 256 * 256;
    for (size_t i = 0; i < h.n_allocs; ++i) {
        uint64_t b;
        std::memcpy(&b, in + sizeof(h) + i * 8, 8);
        const bool stored = !(b & PACK_NOT_STORED);
        b &= ~PACK_NOT_STORED;
        const size_t padded = (b + 255) / 256 * 256;
        if (stored && pos + padded > n) { set_error("packed image: truncated payload"); return WT_ERR_INVALID; }
        void* d = nullptr;
        WT_HIP_CHECK(hipMalloc(&d, b));
        M->allocs.push_back(d);
        M->alloc_bytes.push_back(b);
        if (stored) {
            WT_HIP_CHECK(hipMemcpy(d, in + pos, b, hipMemcpyHostToDevice));
            pos += padded;
        } else {
            M->f32_stale.store(true);
        }
    }
    Archive ar;
    ar.saving = false; ar.in = in + sizeof(h) + h.n_allocs * 8; ar.n = h.struct_bytes; ar.allocs = &M->allocs; ar.alloc_bytes = &M->alloc_bytes;
    archive_model(ar, M);
    if (!ar.ok || ar.pos != ar.n) { set_error("packed image: model section does not match this library's layout"); return WT_ERR_INVALID; }
    // every lazy entry must name whole arrays of plausible size (the file is not trusted)
    for (const wt_model::LazyF32& z : M->lazy_f32) {
        size_t iw = M->allocs.size(), is = M->allocs.size();
        for (size_t i = 0; i < M->allocs.size(); ++i) {
            if (M->allocs[i] == static_cast<const void*>(z.w)) iw = i;
            if (z.s32 && M->allocs[i] == z.s32) is = i;
        }
        if (iw == M->allocs.size() || z.n <= 0 || (z.n % 32) || (size_t)z.n * 4 > M->alloc_bytes[iw] ||
            (z.s32 && (is == M->allocs.size() || (size_t)z.n * 4 > M->alloc_bytes[is])) || !(z.scale > 0.f) || !std::isfinite(z.scale)) {
            set_error("packed image: bad entry in the list of unstored fp32 arrays"); return WT_ERR_INVALID;
        }
    }
    return validate_imported(M);
}

// The fp32 GEMM weights a packed image does not store, rebuilt from their S32 copies: w = (hi + lo * 2^-11) / scale.  Called
// when the first plan on the fp32 chain is created for a model that came from a packed image (and before such a model is
// exported again).  22 of fp32's 24 significant bits survive; a model made from a state dict keeps the exact arrays.
int ensure_f32_weights(const wt_model* M) {
    std::lock_guard<std::mutex> lock(M->f32_mu);
    if (!M->f32_stale.load()) return WT_OK;
    for (const wt_model::LazyF32& z : M->lazy_f32) {
        if (!z.s32) continue;                      // nothing reads it
        if (int rc = launch_unsplit_s32(z.s32, const_cast<float*>(z.w), (long)z.n, 1.f / z.scale, nullptr)) return rc;
    }
    WT_HIP_CHECK(hipDeviceSynchronize());
    M->f32_stale.store(false);
    return WT_OK;
}

}  // namespace wt
