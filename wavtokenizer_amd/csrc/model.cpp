// Host side of libwavtok_hip.so: weight folding/packing (once, at wt_model_create), launch plans
// (per (kind, B, len)) and the extern "C" entry points declared in include/wavtokenizer_amd.h.
// No compute happens on the host at call time; a plan is a flat list of kernel launches over a
// caller-owned workspace.
#include "../../include/wavtokenizer_amd.h"
#include "common.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace wt {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

// ------------------------------------------------------------------------------------- model
struct ConvW {
    float* w = nullptr;   // [cout][k][cin]
    float* b = nullptr;   // [cout]
    int cout = 0, cin = 0, k = 0;
};
struct LstmW {
    float* Wih0 = nullptr;  // [4H][H] rows in packed gate order
    float* b0 = nullptr;    // [4H]  b_ih_l0 + b_hh_l0, packed
    float* W0 = nullptr;    // [4H][H]  W_hh_l0 packed
    float* W1 = nullptr;    // [4H][2H] [W_ih_l1 | W_hh_l1] packed
    float* b1 = nullptr;    // [4H]
    float* W0h = nullptr;   // W0 / W1 as f16 (hi, lo) per-lane packings for the split-f16 step kernel (same bytes)
    float* W1h = nullptr;
    float* Wp = nullptr;    // persistent kernel: [3 roles][32 wg][4 tiles][16 blk][hi, lo][64 lanes][8 halves] (lstm_persist.hip)
};
struct ResStage {
    ConvW c3, c1, sc, down;
    int C = 0, r = 0;
};
struct PosRes {
    float *n1w, *n1b, *n2w, *n2b;
    ConvW c1, c2;
};
struct CnxBlock {
    float *dw_w, *dw_b, *ada_s, *ada_h, *W1, *b1, *W2, *b2, *gamma;
};
struct SeaDecStage {
    float* tr_w = nullptr;  // [k][cin][cout]
    float* tr_wp = nullptr; // [stride phases][cout][2 taps][cin]: tap 0 <-> frame t-1 (kernel index r+stride), tap 1 <-> frame t (index r)
    float* tr_b = nullptr;
    int cin = 0, cout = 0, k = 0, r = 0;
    ConvW c3, c1, sc;
};

}  // namespace wt

struct wt_model {
    wt_arch arch{};
    int device = 0;
    int hop = 1;
    int H = 512;
    std::vector<int> enc_ratios;
    std::vector<void*> allocs;
    int64_t weight_bytes = 0;
    // f16 (hi, lo) copies of the weight matrices the split-precision GEMM (gemm16.hip) reads, keyed by the
    // fp32 device pointer the plans already use; `lo_off` = elements between the hi and the lo array
    struct Split16 { void* hi; long lo_off; };
    std::map<const float*, Split16> split16;
    // S32 copies (gemm16s.hip: 128-byte groups [32 x f16 hi | 32 x f16 lo], same footprint as fp32) of the weights
    // whose GEMMs take pre-split activations, keyed the same way
    std::map<const float*, void*> s32;
    std::map<const float*, bool> s32_tap_pair;       // that S32 copy holds its taps in paired order (GemmArgs::tap_pair)
    // encoder
    float *e0_w = nullptr, *e0_b = nullptr;   // [7][32], [32]
    int e0_k = 7, e0_c = 32;
    std::vector<wt::ResStage> stages;
    wt::LstmW enc_lstm;
    wt::ConvW enc_final;
    float *embed = nullptr, *ee = nullptr;
    // backbone
    wt::ConvW bb_embed;
    wt::PosRes res[4];
    float *at_nw, *at_nb, *at_Wqk, *at_bqk, *at_Wv, *at_bv, *at_Wp, *at_bp;
    float *gn5w, *gn5b, *ada_s, *ada_h;
    std::vector<wt::CnxBlock> cnx;
    float *fln_w, *fln_b;
    float *head_W = nullptr, *head_b = nullptr;
    int Kb = 0, Kq = 0, bins_f = 0, R = 0;
    float *istft_W = nullptr, *wsq = nullptr, *win = nullptr;
    // SEANetDecoder (present iff the checkpoint holds it)
    bool has_seadec = false;
    wt::ConvW sd_first;
    wt::LstmW sd_lstm;
    std::vector<wt::SeaDecStage> sd_stages;
    float *sd_last_w = nullptr, *sd_last_b = nullptr;   // [7][32], [1]
};

namespace wt {

struct TensorMap {
    std::map<std::string, std::pair<const float*, int64_t>> m;
    std::string missing;
    const float* get(const std::string& k, int64_t numel) {
        auto it = m.find(k);
        if (it == m.end()) { if (missing.empty()) missing = k; return nullptr; }
        if (it->second.second != numel) {
            if (missing.empty()) missing = k + " (numel " + std::to_string(it->second.second) + ", expected " + std::to_string(numel) + ")";
            return nullptr;
        }
        return it->second.first;
    }
    bool has(const std::string& k) const { return m.count(k) != 0; }
};

static int upload(wt_model* M, const std::vector<float>& h, float** out) {
    void* d = nullptr;
    size_t bytes = std::max<size_t>(h.size(), 4) * sizeof(float);
    WT_HIP_CHECK(hipMalloc(&d, bytes));
    M->allocs.push_back(d);
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
static int load_lstm(wt_model* M, TensorMap& tm, const std::string& prefix, int H, LstmW* out) {
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

static const char* ENC = "feature_extractor.encodec.encoder.model.";
static const char* DEC = "feature_extractor.encodec.decoder.model.";
static const char* VQK = "feature_extractor.encodec.quantizer.vq.layers.0._codebook.";

static int build_model(wt_model* M, TensorMap& tm) {
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
        M->stages.push_back(st);
        idx += 3; mult *= 2;
    }
    if (mult * nf != H) { set_error("encoder width after the last ratio must be 512"); return WT_ERR_INVALID; }
    if (int rc = load_lstm(M, tm, std::string(ENC) + std::to_string(idx), H, &M->enc_lstm)) return rc;
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
        if (int rc = upload_raw(M, e, (int64_t)a.vq_bins * 512, &M->embed)) return rc;
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
        // packed head rows: 64-row groups = 32 log-magnitude rows then the 32 phase rows of the same slots
        std::vector<float> wp((size_t)2 * Kb * D, 0.f), bp((size_t)2 * Kb, 0.f);
        for (int sl = 0; sl < Kb; ++sl) {
            const int f = slot_bin(sl);
            if (f < 0) continue;
            const size_t pm = (size_t)(sl / 32) * 64 + (sl % 32), pp = pm + 32;
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
        if (int rc = load_lstm(M, tm, std::string(DEC) + "1", H, &M->sd_lstm)) return rc;
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

static int add_split(wt_model* M, const float* w, long n) {
    if (!w || n <= 0 || (n % 8)) return 0;
    void* d = nullptr;
    WT_HIP_CHECK(hipMalloc(&d, (size_t)n * 4));
    M->allocs.push_back(d);
    M->weight_bytes += n * 4;
    if (int rc = launch_split_f16x2(w, d, static_cast<char*>(d) + (size_t)n * 2, n, nullptr)) return rc;
    M->split16[w] = {d, n};
    return 0;
}

static int add_s32(wt_model* M, const float* w, long n) {
    if (!w || n <= 0 || (n % 32)) return 0;
    void* d = nullptr;
    WT_HIP_CHECK(hipMalloc(&d, (size_t)n * 4));
    M->allocs.push_back(d);
    M->weight_bytes += n * 4;
    if (int rc = launch_split_s32(w, d, n, nullptr)) return rc;
    M->s32[w] = d;
    return 0;
}

static int build_splits(wt_model* M) {
    const wt_arch& a = M->arch;
    const int D = a.dim, I = a.intermediate_dim;
    auto conv = [&](const ConvW& c) { return add_split(M, c.w, (long)c.cout * c.k * c.cin); };
    auto conv32 = [&](const ConvW& c) { return (c.cin % 32) ? 0 : add_s32(M, c.w, (long)c.cout * c.k * c.cin); };
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
            M->s32[st.down.w] = M->s32.at(dpk);
            M->s32_tap_pair[st.down.w] = true;
        } else
        if (int rc = conv32(st.down)) return rc;
        if (int rc = conv32(st.c3)) return rc;
        if (int rc = conv32(st.c1)) return rc;
        if (int rc = conv32(st.sc)) return rc;
    }
    if (int rc = add_s32(M, M->enc_lstm.Wih0, 4L * M->H * M->H)) return rc;
    if (int rc = conv32(M->enc_final)) return rc;
    if (int rc = add_s32(M, M->embed, (long)a.vq_bins * 512)) return rc;
    if (int rc = conv32(M->bb_embed)) return rc;
    for (int i = 0; i < 4; ++i) {
        if (int rc = conv32(M->res[i].c1)) return rc;
        if (int rc = conv32(M->res[i].c2)) return rc;
    }
    for (const CnxBlock& c : M->cnx) {
        if (int rc = add_s32(M, c.W1, (long)I * D)) return rc;
        if (int rc = add_s32(M, c.W2, (long)D * I)) return rc;
    }
    if (int rc = add_s32(M, M->head_W, 2L * M->Kb * D)) return rc;
    if (int rc = add_s32(M, M->istft_W, 4L * M->Kq * M->Kq)) return rc;
    if (int rc = add_s32(M, M->at_Wqk, 2L * D * D)) return rc;
    if (int rc = add_s32(M, M->at_Wv, (long)D * D)) return rc;
    if (int rc = add_s32(M, M->at_Wp, (long)D * D)) return rc;
    if (M->has_seadec) {
        if (int rc = conv32(M->sd_first)) return rc;
        if (int rc = add_s32(M, M->sd_lstm.Wih0, 4L * M->H * M->H)) return rc;
        for (const SeaDecStage& st : M->sd_stages) {
            if (st.tr_wp && st.cin % 16 == 0) if (int rc = add_s32(M, st.tr_wp, (long)st.r * st.cout * 2 * st.cin)) return rc;
            if (resblock_fusable(st.cout)) continue;            // its convs run inside resblock16
            if (int rc = conv32(st.sc)) return rc;
            if (int rc = conv32(st.c3)) return rc;
            if (int rc = conv32(st.c1)) return rc;
        }
    }
    for (const ResStage& st : M->stages) {
        if (int rc = conv(st.down)) return rc;
        if (int rc = conv(st.sc)) return rc;
        if (int rc = conv(st.c3)) return rc;
        if (int rc = conv(st.c1)) return rc;
    }
    if (int rc = add_split(M, M->embed, (long)a.vq_bins * 512)) return rc;
    if (int rc = add_split(M, M->head_W, 2L * M->Kb * D)) return rc;
    if (int rc = conv(M->enc_final)) return rc;
    if (int rc = add_split(M, M->enc_lstm.Wih0, 4L * M->H * M->H)) return rc;
    if (int rc = conv(M->bb_embed)) return rc;
    for (int i = 0; i < 4; ++i) {
        if (int rc = conv(M->res[i].c1)) return rc;
        if (int rc = conv(M->res[i].c2)) return rc;
    }
    if (int rc = add_split(M, M->at_Wqk, 2L * D * D)) return rc;
    if (int rc = add_split(M, M->at_Wp, (long)D * D)) return rc;
    for (const CnxBlock& c : M->cnx) {
        if (int rc = add_split(M, c.W1, (long)I * D)) return rc;
        if (int rc = add_split(M, c.W2, (long)D * I)) return rc;
    }
    if (int rc = add_split(M, M->istft_W, 4L * M->Kq * M->Kq)) return rc;
    if (M->has_seadec) {
        if (int rc = conv(M->sd_first)) return rc;
        if (int rc = add_split(M, M->sd_lstm.Wih0, 4L * M->H * M->H)) return rc;
        for (const SeaDecStage& st : M->sd_stages) {
            if (st.tr_wp) if (int rc = add_split(M, st.tr_wp, (long)st.r * st.cout * 2 * st.cin)) return rc;
            if (int rc = conv(st.sc)) return rc;
            if (int rc = conv(st.c3)) return rc;
            if (int rc = conv(st.c1)) return rc;
        }
    }
    WT_HIP_CHECK(hipDeviceSynchronize());
    return 0;
}

// --------------------------------------------------------------------------------------- plan
struct RunCtx {
    char* ws;
    hipStream_t stream;
    const float* in_f;       // wav (encode) / features (decode)
    float* out_f;            // features (encode) / wav (decode)
    int64_t* codes;
    float* aux;              // emb_out (encode) / backbone_out (decode)
    int bw_id;
};
struct BufSpec {
    std::string name;
    size_t bytes = 0, numel = 0, off = 0;
    int first = INT_MAX, last = -1;
};

}  // namespace wt

struct wt_plan {
    const wt_model* model = nullptr;
    int kind = 0, B = 0, flags = 0;
    int64_t len = 0, L = 0, T = 0;
    std::vector<wt::BufSpec> bufs;
    std::vector<std::function<int(const wt::RunCtx&)>> steps;
    std::vector<std::string> step_names;
    size_t ws_bytes = 0;
    int n_launches = 0;
    // optional HIP-event timing of the steps whose name contains `timing_filter` (bench.py roofline
    // leg); mutable profiling state, not thread-safe, off by default
    mutable std::string timing_filter;
    mutable std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending, ev_free;
    mutable double timing_ms = 0.0;
    mutable long timing_n = 0;
    // persistent LSTM (lstm_persist.hip): a host-mapped word the kernel sets if a step barrier times out (lost
    // co-residency); checked at the next call, which then reports the failure and falls back to one launch per step
    unsigned* persist_err_host = nullptr;
    unsigned* persist_err_dev = nullptr;
    mutable bool persist_ok = true;
    // WT_PLAN_FLAG_GRAPH: the launch sequence of a call, captured once and replayed with hipGraphLaunch while the
    // caller passes the same buffers (small batches are bound by the host's launch rate, not by the GPU)
    struct GraphKey {
        const void *ws = nullptr, *in = nullptr, *out = nullptr, *codes = nullptr, *aux = nullptr;
        int bw = -1;
        bool operator==(const GraphKey& o) const {
            return ws == o.ws && in == o.in && out == o.out && codes == o.codes && aux == o.aux && bw == o.bw;
        }
    };
    mutable GraphKey graph_key, last_key;
    mutable bool graph_failed = false;
    mutable hipGraphExec_t graph_exec = nullptr;
    mutable hipStream_t cap_stream = nullptr;
    mutable long graph_replays = 0;

    int buf(const std::string& name, size_t numel) {
        wt::BufSpec b;
        b.name = name; b.numel = numel; b.bytes = (numel * sizeof(float) + 255) / 256 * 256;
        bufs.push_back(b);
        return (int)bufs.size() - 1;
    }
    void step(std::initializer_list<int> used, std::function<int(const wt::RunCtx&)> fn, int launches = 1,
              const std::string& name = "") {
        const int s = (int)steps.size();
        std::string nm = name;
        for (int id : used) {
            if (id < 0) continue;
            bufs[id].first = std::min(bufs[id].first, s);
            bufs[id].last = std::max(bufs[id].last, s);
            if (name.empty()) nm = bufs[id].name;      // default: the last buffer the step touches
        }
        steps.push_back(std::move(fn));
        step_names.push_back(nm);
        n_launches += launches;
    }
    float* ptr(const wt::RunCtx& c, int id) const { return reinterpret_cast<float*>(c.ws + bufs[id].off); }
    void layout() {
        const bool keep = flags & WT_PLAN_FLAG_KEEP_STAGES;
        std::vector<int> order(bufs.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return bufs[x].first < bufs[y].first; });
        std::vector<int> placed;
        ws_bytes = 0;
        for (int id : order) {
            wt::BufSpec& b = bufs[id];
            if (b.last < 0) { b.off = 0; continue; }
            if (keep) b.last = INT_MAX;
            size_t off = 0;
            bool moved = true;
            while (moved) {
                moved = false;
                for (int pid : placed) {
                    const wt::BufSpec& q = bufs[pid];
                    const bool live = !(q.last < b.first || b.last < q.first);
                    const bool overlap = off < q.off + q.bytes && q.off < off + b.bytes;
                    if (live && overlap) { off = q.off + q.bytes; moved = true; }
                }
            }
            b.off = off;
            placed.push_back(id);
            ws_bytes = std::max(ws_bytes, off + b.bytes);
        }
        ws_bytes = std::max<size_t>(ws_bytes, 256);
    }
};

namespace wt {

// Dense layers run on the split-f16 kernel (fp32-equivalent, gemm16.hip) when the weight has a split copy and
// the shape/epilogue is covered; everything else (ELU prologue, argmax, head, activation x activation
// products, K % 32 != 0, WT_PLAN_FLAG_FP32_GEMM) on the fp32 MFMA chain of gemm.hip.
static bool gemm16_covers(const wt_plan* P, const GemmArgs& a, int pro, int epi) {
    if (P->flags & WT_PLAN_FLAG_FP32_GEMM) return false;
    if (a.N < 64 || a.K % 32 || a.Cin % 8 || (a.taps > 1 && a.Cin % 32) || a.w_rstride % 8 || a.zW % 8) return false;
    const bool ok_pair = (pro == PRO_NONE && (epi == EPI_BIAS || epi == EPI_BIAS_RES || epi == EPI_BIAS_GELU ||
                                               epi == EPI_BIAS_GAMMA_RES || epi == EPI_HEAD || epi == EPI_ARGMAX)) ||
                         (pro == PRO_ELU && (epi == EPI_BIAS || epi == EPI_BIAS_RES || epi == EPI_BIAS_RES_ELU));
    return ok_pair && P->model->split16.count(a.W) != 0;
}
static int gemm_auto(const wt_plan* P, const GemmArgs& a, int pro, int epi, hipStream_t s) {
    if (gemm16_covers(P, a, pro, epi)) {
        const auto& sp = P->model->split16.at(a.W);
        GemmArgs b = a;
        b.W_hi = sp.hi;
        b.w_lo_off = sp.lo_off;
        return launch_gemm16(b, pro, epi, s);
    }
    return launch_gemm(a, pro, epi, s);
}

// Both operands pre-split (S32): the activations were written in S32 by their producer, the weight has an S32 copy
static int gemm_s32(const wt_plan* P, const GemmArgs& a, int epi, int out, hipStream_t s) {
    auto it = P->model->s32.find(a.W);
    if (it == P->model->s32.end()) { set_error("internal: no S32 copy of this weight"); return WT_ERR_INVALID; }
    GemmArgs b = a;
    b.W_hi = it->second;
    b.tap_pair = P->model->s32_tap_pair.count(a.W) ? 1 : 0;
    return launch_gemm16s(b, epi, out, s);
}
// The decoder's dense chain runs on S32 operands unless stage taps are kept (fp32 taps) or fp32 GEMMs are forced
static bool plan_s32(const wt_plan* P) { return !(P->flags & (WT_PLAN_FLAG_KEEP_STAGES | WT_PLAN_FLAG_FP32_GEMM)); }

// SConv1d geometry (encoder/modules/conv.py:195-211, 54-61), non-causal.
struct SConvGeom { int pl, pr_total, Tout, Tp; };
static SConvGeom sconv_geom(long T, int k, int stride, int dil) {
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
static GemmArgs sconv_args(const ConvW& w, int B, long T, int stride, int dil) {
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
static GemmArgs zconv_args(const ConvW& w, int B, int L) {
    GemmArgs a;
    a.a_bstride = (long)L * w.cin; a.a_rstride = w.cin;
    a.T_in = L; a.T_out = L; a.Cin = w.cin; a.taps = w.k; a.pad_left = (w.k - 1) / 2; a.pad_mode = PAD_ZERO;
    a.W = w.w; a.w_rstride = (long)w.k * w.cin; a.bias = w.b;
    a.M = B * L; a.N = w.cout; a.K = w.k * w.cin; a.c_rstride = w.cout;
    return a;
}
// plain X[M][K] . W[N][K]^T
static GemmArgs linear_args(const float* W, const float* bias, long M, int N, int K) {
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
    if (resblock_fusable(C) && !(P->flags & WT_PLAN_FLAG_KEEP_STAGES)) {
        // one fused kernel (resblock.hip); with e0 set, xin is unused and the tile is built from the waveform
        const int y = P->buf(name, (size_t)B * T * C);
        P->step({e0 ? -1 : xin, y}, [=](const RunCtx& c) {
            ResblockArgs a{};
            a.x = e0 ? nullptr : P->ptr(c, xin) + x_off;
            a.x_bstride = x_bstride;
            a.wav = e0 ? c.in_f : nullptr;
            a.e0_w = e0 ? e0->e0_w : nullptr; a.e0_b = e0 ? e0->e0_b : nullptr;
            a.W3 = c3.w; a.b3 = c3.b; a.W1 = c1.w; a.b1 = c1.b; a.Ws = sc.w; a.bs = sc.b;
            a.y = P->ptr(c, y); a.B = B; a.T = (int)T; a.C = C; a.elu_out = elu_out ? 1 : 0;
            if (P->flags & WT_PLAN_FLAG_FP32_GEMM) return launch_resblock(a, c.stream);
            a.out_s32 = out_s32 ? 1 : 0;
            return launch_resblock16(a, c.stream);
        }, 1, "resblock.fused");
        return y;
    }
    const int h = P->buf(name + ".h", (size_t)B * T * (C / 2));
    const int y = P->buf(name, (size_t)B * T * C);
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
    const int y = P->buf(name, (size_t)B * L * H);
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
    static const bool persist_env = [] { const char* e = getenv("WT_LSTM_PERSIST"); return !e || e[0] != '0'; }();
    bool persist = persist_env && !(P->flags & (WT_PLAN_FLAG_FP32_GEMM | WT_PLAN_FLAG_STEP_LSTM)) && w.Wp && H == 512 && B <= 128 && L < 65536;
    if (persist) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, P->model->device) != hipSuccess || cus != 256) persist = false;
    }
    if (persist && !P->persist_err_host) {
        void* hp = nullptr;
        void* dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
            P->persist_err_host = static_cast<unsigned*>(hp);
            P->persist_err_dev = static_cast<unsigned*>(dp);
            *P->persist_err_host = 0;
        } else {
            if (hp) (void)hipHostFree(hp);
            persist = false;
        }
    }
    const size_t hxn = lstm_persist_hx_bytes() / sizeof(float), ctn = lstm_persist_ctl_bytes() / sizeof(float);
    const int hx = persist ? P->buf(name + ".hx", hxn + ctn) : -1;
    P->step({xin, xg, st, hx, y}, [=](const RunCtx& c) {
        if (persist && P->persist_ok) {
            float* hb = P->ptr(c, hx);
            static const bool counter_form = [] { const char* e = getenv("WT_LSTM_PERSIST"); return e && e[0] == '2'; }();
            WT_HIP_CHECK(hipMemsetAsync(hb, counter_form ? 0 : 0xFF, (hxn + ctn) * sizeof(float), c.stream));
            LstmPersistArgs pa;
            static const int df_trace = [] { const char* e = getenv("WT_LSTM_TRACE"); return e ? atoi(e) : 0; }();
            pa.data_flag = counter_form ? 0 : (1 | (df_trace ? 4 : 0));
            pa.xg0 = P->ptr(c, xg); pa.Wp = w.Wp; pa.b1 = w.b1; pa.x = P->ptr(c, xin); pa.y = P->ptr(c, y);
            pa.hx = hb; pa.ctl = reinterpret_cast<unsigned*>(hb + hxn); pa.host_err = P->persist_err_dev;
            pa.B = B; pa.L = L; pa.H = H; pa.Bx = (B + 7) / 8; pa.elu_out = elu_out ? 1 : 0; pa.out_s32 = y_s32 ? 1 : 0;
            return launch_lstm_persist(pa, c.stream);
        }
        float* s = P->ptr(c, st);
        WT_HIP_CHECK(hipMemsetAsync(s, 0, st_numel * sizeof(float), c.stream));
        LstmArgs la;
        la.f16x3 = (P->flags & WT_PLAN_FLAG_FP32_GEMM) ? 0 : 1;     // recurrent product on split-f16 MFMAs unless fp32 is forced
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
                             int x_elu, const std::string& name, long x_off = 0, long x_bstride = 0) {
    const int C = sc.cout;
    const int h = P->buf(name + ".h", (size_t)B * T * (C / 2));
    const int y = P->buf(name + ".sc", (size_t)B * T * C);
    const int o = P->buf(name, (size_t)B * T * C);
    GemmArgs a3 = sconv_args(c3, B, T, 1, 1);
    P->step({x_elu, h}, [=](const RunCtx& c) {
        GemmArgs a = a3; a.A = P->ptr(c, x_elu) + x_off; a.C = P->ptr(c, h);
        if (x_bstride) a.a_bstride = x_bstride;
        return gemm_s32(P, a, EPI_BIAS_ELU, OUT_S32, c.stream);
    });
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

static int build_encode(wt_plan* P) {
    const wt_model* M = P->model;
    const int B = P->B;
    const long T = P->T;
    // the first conv is folded into the fused stage-1 resblock unless stage taps are kept
    const bool fold_e0 = !(P->flags & WT_PLAN_FLAG_KEEP_STAGES) && !M->stages.empty() && M->stages[0].C == 32 &&
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
    // output -> down conv, LSTM output -> last conv); with KEEP_STAGES the taps stay raw instead
    const bool fuse_elu = !(P->flags & WT_PLAN_FLAG_KEEP_STAGES);
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
        const bool fused = resblock_fusable(st.C) && !(P->flags & WT_PLAN_FLAG_KEEP_STAGES);
        // a fused stage only needs the S32 down-conv weights (its own convs run inside resblock16); an unfused one
        // needs S32 copies of all four
        const bool ws32 = s32 && (st.C % 32 == 0) && M->s32.count(st.down.w) &&
                          (fused || (M->s32.count(st.c3.w) && M->s32.count(st.c1.w) && M->s32.count(st.sc.w)));
        bool x_is_s32;                   // the resblock output (elu'd) is S32
        if (fused) {
            x = plan_resblock(P, st.c3, st.c1, st.sc, B, Tc, x, "enc." + std::to_string(idx), fuse_elu,
                              (idx == 1 && fold_e0) ? M : nullptr, 0, 0, ws32);
            x_is_s32 = ws32;
        } else if (ws32 && x_raw >= 0) {
            x = plan_resblock_s32(P, st.c3, st.c1, st.sc, B, Tc, x_raw, x_elu, "enc." + std::to_string(idx));
            x_is_s32 = true;
        } else {
            x = plan_resblock(P, st.c3, st.c1, st.sc, B, Tc, x, "enc." + std::to_string(idx), fuse_elu, nullptr);
            x_is_s32 = false;
        }
        x_raw = x_elu = -1;
        GemmArgs ad = sconv_args(st.down, B, Tc, st.r, 1);
        const size_t ynum = (size_t)B * ad.T_out * st.down.cout;
        const int y = P->buf("enc." + std::to_string(idx + 2), ynum);
        const int xin = x;
        const bool last = si + 1 == M->stages.size();
        // what the next consumer wants: a fused resblock reads fp32; an unfused S32 stage reads S32 raw + S32 elu; after
        // the last stage the LSTM reads fp32 (skip) and its input projection S32
        const bool next_s32_stage = !last && x_is_s32 && (st.down.cout % 32 == 0) &&
                                    !(resblock_fusable(M->stages[si + 1].C) && !(P->flags & WT_PLAN_FLAG_KEEP_STAGES)) &&
                                    M->s32.count(M->stages[si + 1].c3.w) && M->s32.count(M->stages[si + 1].sc.w) &&
                                    M->s32.count(M->stages[si + 1].c1.w) && M->s32.count(M->stages[si + 1].down.w);
        const bool lstm_s32 = last && x_is_s32 && M->s32.count(M->enc_lstm.Wih0);
        const int y2 = (next_s32_stage || lstm_s32) ? P->buf("enc." + std::to_string(idx + 2) + ".s32", ynum) : -1;
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
    const int emb_s32 = tail_s32 ? P->buf("enc." + std::to_string(idx + 2) + ".s32", (size_t)B * L * 512) : -1;
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
    const int np = tail_s32 ? gemm16s_vq_parts(bins)
                            : (gemm16_covers(P, av, PRO_NONE, EPI_ARGMAX) ? gemm16_vq_parts(bins) : gemm_vq_parts(bins));
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
    const int spec = P->buf("head.spec", (size_t)Mrows * 2 * Kb);
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
        return launch_istft_ola(P->ptr(c, parts), M->win, M->wsq, c.out_f, B, L, ar.n_fft, hop, Kq, c.stream);
    }, 1, "head.ola");
}

static int build_decode(wt_plan* P) {
    const wt_model* M = P->model;
    const wt_arch& ar = M->arch;
    const int B = P->B, L = (int)P->L, D = ar.dim, I = ar.intermediate_dim, Cin = ar.input_channels;
    const long Mrows = (long)B * L;
    const int Lp = ((L + 31) / 32) * 32;
    // S32 mode: every operand of the dense chain is written pre-split by its producer (transpose, norm kernels,
    // GELU / head epilogues) and multiplied by gemm16s.hip; the residual stream and the norm inputs stay fp32
    const bool s32 = plan_s32(P) && (Cin % 32 == 0) && (D % 32 == 0) && (I % 32 == 0);
    const int x0 = P->buf("bb.in", (size_t)Mrows * Cin);
    P->step({x0}, [=](const RunCtx& c) { return launch_transpose(c.in_f, P->ptr(c, x0), B, Cin, L, c.stream, s32); });
    const int x = P->buf("bb.x", (size_t)Mrows * D);       // residual stream, updated in place
    GemmArgs ae = zconv_args(M->bb_embed, B, L);
    P->step({x0, x}, [=](const RunCtx& c) {
        GemmArgs a = ae; a.A = P->ptr(c, x0); a.C = P->ptr(c, x);
        if (s32) return gemm_s32(P, a, EPI_BIAS, 0, c.stream);
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
    const int h1 = P->buf("bb.h1", (size_t)Mrows * D);
    const int h2 = P->buf("bb.h2", (size_t)Mrows * D);

    // ResnetBlock (models.py:58-78).  GroupNorm+swish is applied ONCE per element by the statistics
    // kernel (a second pass over its own L x 24 slab) instead of in the conv's operand staging,
    // where every element would be re-normalised by each of the 18 (tap, column-tile) re-reads.
    auto resnet = [&](const PosRes& r, const std::string& name) {
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
    resnet(M->res[0], "bb.pos_net.0");
    resnet(M->res[1], "bb.pos_net.1");
    if (s32 && M->s32.count(M->at_Wqk) && M->s32.count(M->at_Wv) && M->s32.count(M->at_Wp)) {
        // AttnBlock (models.py:107-127), single head of width D, every product on split-f16 MFMAs: the normalised
        // input, q | k, V^T, the probabilities and the attention output are all written pre-split by their producers
        const int qk = P->buf("bb.attn.qk", (size_t)Mrows * 2 * D);          // S32 [M][q | k]
        const int vt = P->buf("bb.attn.vt", (size_t)B * D * Lp);              // S32 [B][D][Lp]
        const int S = P->buf("bb.attn.s", (size_t)Mrows * Lp);                // fp32 scores
        const int Ps = P->buf("bb.attn.p", (size_t)Mrows * Lp);               // S32 probabilities
        const int o = P->buf("bb.attn.o", (size_t)Mrows * D);                 // S32
        P->step({x, sc, sh, gp, h1}, [=](const RunCtx& c) {
            return launch_gn_apply(P->ptr(c, x), M->at_nw, M->at_nb, P->ptr(c, sc), P->ptr(c, sh), P->ptr(c, h1), 0, B, L, D, 32, 1e-6f, c.stream, 1, P->ptr(c, gp));
        }, 1, "attn.gn");
        GemmArgs aqk = linear_args(M->at_Wqk, M->at_bqk, Mrows, 2 * D, D);
        P->step({h1, qk}, [=](const RunCtx& c) {
            GemmArgs a = aqk; a.A = P->ptr(c, h1); a.C = P->ptr(c, qk);
            return gemm_s32(P, a, EPI_BIAS, OUT_S32, c.stream);
        }, 1, "attn.qk");
        P->step({h1, vt}, [=](const RunCtx& c) {     // V^T[b] = Wv . hn[b]^T + bv   (D x L, pitch Lp; pad columns stay zero)
            WT_HIP_CHECK(hipMemsetAsync(P->ptr(c, vt), 0, (size_t)B * D * Lp * sizeof(float), c.stream));
            GemmArgs a = linear_args(P->ptr(c, h1), M->at_bv, D, L, D);
            a.A = reinterpret_cast<const float*>(M->s32.at(M->at_Wv)); a.zA = 0;
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
    } else
    {   // AttnBlock (models.py:107-127), single head of width D
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
            WT_HIP_CHECK(hipMemsetAsync(P->ptr(c, vt), 0, (size_t)B * D * Lp * sizeof(float), c.stream));
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
    resnet(M->res[2], "bb.pos_net.3");
    resnet(M->res[3], "bb.pos_net.4");
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
    const int nrm = P->buf("bb.cnx.norm", (size_t)Mrows * D);
    const int mid = P->buf("bb.cnx.mid", (size_t)Mrows * I);
    for (int i = 0; i < ar.num_layers; ++i) {
        const CnxBlock cb = M->cnx[i];
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
    const int xo = P->buf("bb.out", (size_t)Mrows * D);
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
static int build_head(wt_plan* P) {
    const wt_model* M = P->model;
    const int B = P->B, L = (int)P->L, D = M->arch.dim;
    const long Mrows = (long)B * L;
    const bool s32 = plan_s32(P) && (D % 32 == 0) && M->s32.count(M->head_W) && M->s32.count(M->istft_W);
    const int xo = P->buf("head.in", (size_t)Mrows * D);
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
            x = plan_resblock_s32(P, st.c3, st.c1, st.sc, B, To, y, y2, "sdec." + std::to_string(di + 2), y_off, y_bs);
        Tc = To; di += 3;
    }
    const int xin = x;
    const long Tf = Tc;
    P->step({xin}, [=](const RunCtx& c) {
        return launch_conv_last(P->ptr(c, xin), M->sd_last_w, M->sd_last_b, c.out_f, B, Tf, 32, 7, 0, c.stream);
    }, 1, "sdec.last");
    return 0;
}

static int build_seanet_decoder(wt_plan* P) {
    const wt_model* M = P->model;
    if (!M->has_seadec) { set_error("checkpoint holds no SEANetDecoder weights"); return WT_ERR_MISSING_TENSOR; }
    const int B = P->B, L = (int)P->L, H = M->H;
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
    const bool fuse_elu = !(P->flags & WT_PLAN_FLAG_KEEP_STAGES);      // producers store elu(.) for "ELU -> conv" consumers
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

}  // namespace wt

// ================================================================================== C ABI
using namespace wt;

extern "C" {

const char* wt_last_error(void) { return g_err.c_str(); }
const char* wt_version(void) { return "wavtokenizer_amd 0.1 (gfx950, split-f16 MFMA products, fp32 accumulation)"; }

int wt_model_create(const wt_arch* arch, const wt_tensor* tensors, int32_t n_tensors, int32_t device, wt_model** out) {
    if (!arch || !tensors || !out) { set_error("wt_model_create: null argument"); return WT_ERR_INVALID; }
    if (arch->n_ratios < 1 || arch->n_ratios > 8) { set_error("n_ratios out of range"); return WT_ERR_INVALID; }
    if (!arch->padding_same) { set_error("only ISTFT padding='same' is implemented (the mode every reference YAML selects)"); return WT_ERR_INVALID; }
    if (arch->num_quantizers != 1) { set_error("only num_quantizers=1 is implemented (vq.py:137 forces n_q=1 at inference)"); return WT_ERR_INVALID; }
    if (arch->input_channels != 512) { set_error("input_channels must be 512 (SEANet dimension)"); return WT_ERR_INVALID; }
    if (arch->dim % 256 || arch->intermediate_dim % 32) { set_error("dim must be a multiple of 256, intermediate_dim of 32"); return WT_ERR_INVALID; }
    if (arch->dim % 32 || (arch->dim / 32) % 4) { set_error("dim/32 (GroupNorm group width) must be a multiple of 4"); return WT_ERR_INVALID; }
    WT_HIP_CHECK(hipSetDevice(device));
    std::unique_ptr<wt_model> M(new wt_model());
    M->arch = *arch;
    M->device = device;
    M->hop = 1;
    for (int i = 0; i < arch->n_ratios; ++i) M->hop *= arch->ratios[i];
    for (int i = arch->n_ratios - 1; i >= 0; --i) M->enc_ratios.push_back(arch->ratios[i]);   // seanet.py:100
    TensorMap tm;
    for (int i = 0; i < n_tensors; ++i) tm.m[tensors[i].name] = {tensors[i].data, tensors[i].numel};
    int rc = build_model(M.get(), tm);
    if (!rc) rc = build_splits(M.get());
    if (rc) {
        if (rc == WT_ERR_MISSING_TENSOR) set_error("state_dict tensor missing or mis-shaped: " + tm.missing);
        for (void* p : M->allocs) (void)hipFree(p);
        return rc;
    }
    *out = M.release();
    return WT_OK;
}

void wt_model_destroy(wt_model* m) {
    if (!m) return;
    for (void* p : m->allocs) (void)hipFree(p);
    delete m;
}
int wt_model_hop(const wt_model* m) { return m ? m->hop : 0; }
int64_t wt_model_weight_bytes(const wt_model* m) { return m ? m->weight_bytes : 0; }

int wt_plan_create(const wt_model* m, int32_t kind, int32_t B, int64_t len, int32_t flags, wt_plan** out) {
    if (!m || !out) { set_error("wt_plan_create: null argument"); return WT_ERR_INVALID; }
    if (B < 1 || len < 1) { set_error("wt_plan_create: B and len must be >= 1"); return WT_ERR_INVALID; }
    if ((len + (kind == WT_PLAN_ENCODE ? m->hop - 1 : 0)) / (kind == WT_PLAN_ENCODE ? m->hop : 1) > 12000) {
        set_error("clips longer than 12000 frames are not supported by one plan; split the clip"); return WT_ERR_INVALID;
    }
    std::unique_ptr<wt_plan> P(new wt_plan());
    P->model = m; P->kind = kind; P->B = B; P->len = len; P->flags = flags;
    int rc;
    if (kind == WT_PLAN_ENCODE) {
        P->T = len;
        P->L = (len + m->hop - 1) / m->hop;
        if ((long)B * len >= (long)INT_MAX) { set_error("batch too large for one plan (32-bit row index)"); return WT_ERR_INVALID; }
        rc = build_encode(P.get());
    } else if (kind == WT_PLAN_DECODE) {
        P->L = len; P->T = len * m->hop;
        rc = build_decode(P.get());
    } else if (kind == WT_PLAN_SEANET_DECODER) {
        P->L = len; P->T = len * m->hop;
        rc = build_seanet_decoder(P.get());
    } else if (kind == WT_PLAN_HEAD) {
        P->L = len; P->T = len * m->hop;
        rc = build_head(P.get());
    } else {
        set_error("unknown plan kind"); return WT_ERR_INVALID;
    }
    if (rc) return rc;
    P->layout();
    *out = P.release();
    return WT_OK;
}
void wt_plan_destroy(wt_plan* p) {
    if (!p) return;
    if (p->persist_err_host) (void)hipHostFree(p->persist_err_host);
    if (p->graph_exec) (void)hipGraphExecDestroy(p->graph_exec);
    if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
    for (auto& ev : p->ev_pending) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto& ev : p->ev_free) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    delete p;
}
size_t wt_plan_workspace_bytes(const wt_plan* p) { return p ? p->ws_bytes : 0; }
int64_t wt_plan_frames(const wt_plan* p) { return p ? p->L : 0; }
int wt_plan_num_launches(const wt_plan* p) { return p ? p->n_launches : 0; }

int wt_plan_find_buffer(const wt_plan* p, const char* name, size_t* offset, size_t* numel) {
    if (!p || !name) return WT_ERR_INVALID;
    for (const BufSpec& b : p->bufs)
        if (b.name == name) {
            if (offset) *offset = b.off;
            if (numel) *numel = b.numel;
            return WT_OK;
        }
    set_error(std::string("no stage buffer named ") + name);
    return WT_ERR_INVALID;
}
int wt_plan_buffer_name(const wt_plan* p, int32_t index, const char** name) {
    if (!p || index < 0 || index >= (int)p->bufs.size()) return WT_ERR_INVALID;
    *name = p->bufs[index].name.c_str();
    return WT_OK;
}

static int run_plan(const wt_plan* p, const RunCtx& c) {
    WT_HIP_CHECK(hipSetDevice(p->model->device));
    if (p->persist_err_host && *p->persist_err_host && p->persist_ok) {
        p->persist_ok = false;            // this and every later call run the LSTM one launch per step
        *p->persist_err_host = 0;
        if (p->graph_exec) { (void)hipGraphExecDestroy(p->graph_exec); p->graph_exec = nullptr; }      // it holds the persistent launch
        p->last_key = wt_plan::GraphKey{};
        set_error("the previous persistent LSTM launch on this plan lost co-residency (a step barrier timed out) and its "
                  "outputs are invalid; the plan now runs the LSTM one launch per step");
        return WT_ERR_HIP;
    }
    const bool timing = !p->timing_filter.empty();
    if ((p->flags & WT_PLAN_FLAG_GRAPH) && !timing && !p->graph_failed) {
        const wt_plan::GraphKey key{c.ws, c.in_f, c.out_f, c.codes, c.aux, c.bw_id};
        if (p->graph_exec && key == p->graph_key) {
            WT_HIP_CHECK(hipGraphLaunch(p->graph_exec, c.stream));
            ++p->graph_replays;
            return WT_OK;
        }
        if (key == p->last_key) {
            // second call in a row with these buffers (the first ran eagerly: every one-time kernel attribute is
            // set): record the launches on a capture stream, then replay them on the caller's stream
            if (p->graph_exec) { (void)hipGraphExecDestroy(p->graph_exec); p->graph_exec = nullptr; }
            if (!p->cap_stream) WT_HIP_CHECK(hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking));
            RunCtx cc = c;
            cc.stream = p->cap_stream;
            WT_HIP_CHECK(hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeRelaxed));
            int rc = WT_OK;
            for (size_t i = 0; i < p->steps.size() && !rc; ++i) rc = p->steps[i](cc);
            hipGraph_t g = nullptr;
            const hipError_t ce = hipStreamEndCapture(p->cap_stream, &g);
            if (rc || ce != hipSuccess || !g) {
                if (g) (void)hipGraphDestroy(g);
                (void)hipGetLastError();
                p->graph_failed = true;             // this plan stays on direct launches
                if (rc) return rc;
            } else {
                const hipError_t ie = hipGraphInstantiate(&p->graph_exec, g, nullptr, nullptr, 0);
                (void)hipGraphDestroy(g);
                if (ie != hipSuccess) { p->graph_exec = nullptr; p->graph_failed = true; (void)hipGetLastError(); }
                else {
                    p->graph_key = key;
                    WT_HIP_CHECK(hipGraphLaunch(p->graph_exec, c.stream));
                    ++p->graph_replays;
                    return WT_OK;
                }
            }
        }
        p->last_key = key;
    }
    for (size_t i = 0; i < p->steps.size(); ++i) {
        const bool timed = timing && p->step_names[i].find(p->timing_filter) != std::string::npos;
        std::pair<hipEvent_t, hipEvent_t> ev;
        if (timed) {
            if (!p->ev_free.empty()) { ev = p->ev_free.back(); p->ev_free.pop_back(); }
            else { WT_HIP_CHECK(hipEventCreate(&ev.first)); WT_HIP_CHECK(hipEventCreate(&ev.second)); }
            WT_HIP_CHECK(hipEventRecord(ev.first, c.stream));
        }
        if (int rc = p->steps[i](c)) return rc;
        if (timed) {
            WT_HIP_CHECK(hipEventRecord(ev.second, c.stream));
            p->ev_pending.push_back(ev);
        }
    }
    return WT_OK;
}

int64_t wt_plan_graph_replays(const wt_plan* p) { return p ? p->graph_replays : 0; }
int wt_plan_num_steps(const wt_plan* p) { return p ? (int)p->steps.size() : 0; }
int wt_plan_step_name(const wt_plan* p, int32_t index, const char** name) {
    if (!p || index < 0 || index >= (int)p->step_names.size()) return WT_ERR_INVALID;
    *name = p->step_names[index].c_str();
    return WT_OK;
}
int wt_plan_set_timing(const wt_plan* p, const char* name_substr) {
    if (!p) return WT_ERR_INVALID;
    p->timing_filter = name_substr ? name_substr : "";
    return WT_OK;
}
int wt_plan_read_timing(const wt_plan* p, double* total_ms, int64_t* launches, int32_t reset) {
    if (!p) return WT_ERR_INVALID;
    for (auto& ev : p->ev_pending) {
        WT_HIP_CHECK(hipEventSynchronize(ev.second));
        float ms = 0.f;
        WT_HIP_CHECK(hipEventElapsedTime(&ms, ev.first, ev.second));
        p->timing_ms += ms;
        p->timing_n += 1;
        p->ev_free.push_back(ev);
    }
    p->ev_pending.clear();
    if (total_ms) *total_ms = p->timing_ms;
    if (launches) *launches = p->timing_n;
    if (reset) { p->timing_ms = 0.0; p->timing_n = 0; }
    return WT_OK;
}

int wt_encode(const wt_plan* p, const float* wav, float* features, int64_t* codes, float* emb_out, void* workspace,
              void* stream) {
    if (!p || p->kind != WT_PLAN_ENCODE) { set_error("wt_encode: not an encode plan"); return WT_ERR_INVALID; }
    if (!wav || !codes || !workspace) { set_error("wt_encode: null buffer"); return WT_ERR_INVALID; }
    RunCtx c{static_cast<char*>(workspace), static_cast<hipStream_t>(stream), wav, features, codes, emb_out, 0};
    return run_plan(p, c);
}

int wt_decode(const wt_plan* p, const float* features, int32_t bandwidth_id, float* wav_out, float* backbone_out,
              void* workspace, void* stream) {
    if (!p || p->kind != WT_PLAN_DECODE) { set_error("wt_decode: not a decode plan"); return WT_ERR_INVALID; }
    if (!features || !wav_out || !workspace) { set_error("wt_decode: null buffer"); return WT_ERR_INVALID; }
    if (bandwidth_id < 0 || bandwidth_id >= p->model->arch.adanorm_num_embeddings) {
        set_error("wt_decode: bandwidth_id out of range"); return WT_ERR_INVALID;
    }
    RunCtx c{static_cast<char*>(workspace), static_cast<hipStream_t>(stream), features, wav_out, nullptr, backbone_out, bandwidth_id};
    return run_plan(p, c);
}

int wt_head(const wt_plan* p, const float* x, float* wav_out, void* workspace, void* stream) {
    if (!p || p->kind != WT_PLAN_HEAD) { set_error("wt_head: wrong plan kind"); return WT_ERR_INVALID; }
    if (!x || !wav_out || !workspace) { set_error("wt_head: null buffer"); return WT_ERR_INVALID; }
    RunCtx c{static_cast<char*>(workspace), static_cast<hipStream_t>(stream), x, wav_out, nullptr, nullptr, 0};
    return run_plan(p, c);
}

int wt_seanet_decode(const wt_plan* p, const float* features, float* wav_out, void* workspace, void* stream) {
    if (!p || p->kind != WT_PLAN_SEANET_DECODER) { set_error("wt_seanet_decode: wrong plan kind"); return WT_ERR_INVALID; }
    if (!features || !wav_out || !workspace) { set_error("wt_seanet_decode: null buffer"); return WT_ERR_INVALID; }
    RunCtx c{static_cast<char*>(workspace), static_cast<hipStream_t>(stream), features, wav_out, nullptr, nullptr, 0};
    return run_plan(p, c);
}

int wt_codes_to_features(const wt_model* m, const int64_t* codes, int32_t K, int32_t B, int64_t L, float* features,
                         void* stream) {
    if (!m || !codes || !features) { set_error("wt_codes_to_features: null argument"); return WT_ERR_INVALID; }
    if (K < 1 || K > m->arch.num_quantizers) { set_error("wt_codes_to_features: K exceeds the number of codebooks"); return WT_ERR_INVALID; }
    WT_HIP_CHECK(hipSetDevice(m->device));
    return launch_codes_to_features(codes, m->embed, K, m->arch.vq_bins, B, L, 512, features, static_cast<hipStream_t>(stream));
}

int wt_sconv1d(const float* x, const float* w, const float* bias, float* y, int32_t B, int64_t T, int32_t Cin,
               int32_t Cout, int32_t k, int32_t stride, int32_t dilation, int32_t elu_input, void* stream) {
    ConvW cw; cw.w = const_cast<float*>(w); cw.b = const_cast<float*>(bias); cw.cout = Cout; cw.cin = Cin; cw.k = k;
    GemmArgs a = sconv_args(cw, B, T, stride, dilation);
    a.A = x; a.C = y;
    return launch_gemm(a, elu_input ? PRO_ELU : PRO_NONE, EPI_BIAS, static_cast<hipStream_t>(stream));
}

int wt_linear(const float* x, const float* w, const float* bias, float* y, int64_t M, int32_t N, int32_t K,
              int32_t f16x3, void* workspace, void* stream) {
    if (!x || !w || !y) { set_error("wt_linear: null argument"); return WT_ERR_INVALID; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    GemmArgs a = linear_args(w, bias, M, N, K);
    a.A = x; a.C = y;
    if (!f16x3) return launch_gemm(a, PRO_NONE, EPI_BIAS, s);
    if (!workspace) { set_error("wt_linear: the f16x3 modes need a workspace"); return WT_ERR_INVALID; }
    char* hi = static_cast<char*>(workspace);
    if (f16x3 == 1) {
        if (int rc = launch_split_f16x2(w, hi, hi + (size_t)N * K * 2, (long)N * K, s)) return rc;
        a.W_hi = hi; a.w_lo_off = (long)N * K;
        return launch_gemm16(a, PRO_NONE, EPI_BIAS, s);
    }
    char* xs = hi + (size_t)N * K * 4;
    if (int rc = launch_split_s32(w, hi, (long)N * K, s)) return rc;
    if (int rc = launch_split_s32(x, xs, (long)M * K, s)) return rc;
    a.W_hi = hi; a.A = reinterpret_cast<const float*>(xs);
    return launch_gemm16s(a, EPI_BIAS, f16x3 == 3 ? 1 : 0, s);
}

int wt_conv1d_s32(const float* x, const float* w, const float* bias, float* y, int32_t B, int64_t T, int32_t Cin,
                  int32_t Cout, int32_t k, int32_t stride, int32_t zero_same, void* workspace, void* stream) {
    if (!x || !w || !y || !workspace) { set_error("wt_conv1d_s32: null argument"); return WT_ERR_INVALID; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    ConvW cw; cw.w = const_cast<float*>(w); cw.b = const_cast<float*>(bias); cw.cout = Cout; cw.cin = Cin; cw.k = k;
    GemmArgs a = zero_same ? zconv_args(cw, B, (int)T) : sconv_args(cw, B, T, stride, 1);
    char* ws = static_cast<char*>(workspace);
    char* xs = ws + (size_t)Cout * k * Cin * 4;
    if (int rc = launch_split_s32(w, ws, (long)Cout * k * Cin, s)) return rc;
    if (int rc = launch_split_s32(x, xs, (long)B * T * Cin, s)) return rc;
    a.W_hi = ws; a.A = reinterpret_cast<const float*>(xs); a.C = y;
    return launch_gemm16s(a, EPI_BIAS, 0, s);
}

size_t wt_vq_workspace_bytes(int64_t N, int32_t bins) {
    const size_t np = gemm_vq_parts(bins);
    return ((size_t)N * sizeof(float) + 255) / 256 * 256 + 2 * (((size_t)N * np * sizeof(float) + 255) / 256 * 256) +
           (size_t)bins * sizeof(float) + 256;
}

// the ee[] table here is rebuilt per call on the device by row_sumsq (same kernel as |x|^2)
int wt_vq_nearest(const float* x, const float* embed, int64_t N, int32_t D, int32_t bins, int64_t* codes_out,
                  void* workspace, void* stream) {
    if (!x || !embed || !codes_out || !workspace) { set_error("wt_vq_nearest: null argument"); return WT_ERR_INVALID; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int np = gemm_vq_parts(bins);
    char* ws = static_cast<char*>(workspace);
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    float* xx = reinterpret_cast<float*>(ws); ws += al((size_t)N * sizeof(float));
    float* pv = reinterpret_cast<float*>(ws); ws += al((size_t)N * np * sizeof(float));
    int* pi = reinterpret_cast<int*>(ws); ws += al((size_t)N * np * sizeof(float));
    float* ee = reinterpret_cast<float*>(ws);
    if (int rc = launch_row_sumsq(x, xx, N, D, s)) return rc;
    if (int rc = launch_row_sumsq(embed, ee, bins, D, s)) return rc;
    GemmArgs a = linear_args(embed, nullptr, N, bins, D);
    a.A = x; a.vq_xx = xx; a.vq_ee = ee; a.vq_pval = pv; a.vq_pidx = pi; a.vq_nparts = np;
    if (int rc = launch_gemm(a, PRO_NONE, EPI_ARGMAX, s)) return rc;
    for (int64_t r0 = 0; r0 < N; r0 += 8192) {      // finalize keeps one chunk of codes in LDS
        const int n = (int)std::min<int64_t>(8192, N - r0);
        if (int rc = launch_vq_finalize(pv + r0 * np, pi + r0 * np, np, embed, codes_out + r0, nullptr, 1, n, D, bins, s)) return rc;
    }
    return WT_OK;
}

}  // extern "C"
