// Internal types of the host side of libwavtok_hip.so: the packed model (weights.cpp), launch plans (plan.cpp)
// and what the extern "C" entry points (capi.cpp) share with them.
#pragma once
#include "../../include/wavtokenizer_amd.h"
#include "common.h"

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace wt {

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
    ConvW cat;             // unfused stages (C >= 128): [Ws | W1] [C][C + C/2] and bs + b1, shortcut + conv1 as one contraction
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
    ConvW cat;             // as ResStage::cat
};

}  // namespace wt

struct wt_model {
    wt_arch arch{};
    int device = 0;
    int hop = 1;
    int H = 512;
    std::vector<int> enc_ratios;
    std::vector<void*> allocs;
    std::vector<size_t> alloc_bytes;     // size of each allocation (the packed image stores them in this order)
    int64_t weight_bytes = 0;
    // S32 copies (gemm16s.hip: 128-byte groups [32 x f16 hi | 32 x f16 lo], same footprint as fp32) of the weights
    // whose GEMMs take pre-split activations, keyed by the fp32 device pointer the plans already use
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
    // Range of the split-f16 form (hi = f16(w)): a GEMM weight with |w| >= 65504 or a non-finite value cannot be split;
    // the model then runs every plan on the fp32 MFMA chain (as if WT_PLAN_FLAG_FP32_GEMM were always set)
    bool s32_ok = true;
    bool sd_s32_ok = true;                               // the same for the SEANetDecoder's weights (its plan only)
    float w_amax = 0.f;                                  // largest finite |w| over the GEMM weights
    // fp32 GEMM weights that the default (S32) plans never read: only plans on the fp32 MFMA chain do (WT_PLAN_FLAG_FP32_GEMM,
    // the unfused debug twin).  The packed image does not store them (they are 230 of its 720 MB); a model made from one
    // allocates them and fills them in from the S32 copy when the first such plan is created (w = (hi + lo * 2^-11) /
    // scale: 22 of fp32's 24 significant bits; a model made from a state dict keeps the exact arrays).  s32 = null:
    // an array nothing reads after the split (the tap-paired repacking of a down conv)
    struct LazyF32 { const float* w; const void* s32; int64_t n; float scale; };
    std::vector<LazyF32> lazy_f32;
    mutable std::atomic<bool> f32_stale{false};          // the arrays of lazy_f32 hold nothing yet (packed import)
    mutable std::mutex f32_mu;
    // per-weight power-of-two scale of the S32 copy (tensors whose largest magnitude is far from 1 are stored as
    // w * 2^e; the GEMM brings its accumulators back with acc_scale = 2^-e); absent = 1
    std::map<const float*, float> s32_acc_scale;
    // host-mapped word for wt_codes_to_features: set by the kernel when it meets an index outside the codebook
    unsigned* bad_codes_host = nullptr;
    unsigned* bad_codes_dev = nullptr;
    // Model-level call status: every plan's guard step ORs the failure bits of its call into this host-mapped word too, and
    // the NEXT call on ANY plan of the model consumes it (a file-by-file caller meets a new length, hence a new plan, per
    // file: a per-plan word alone would never be looked at again).  The consequences are sticky for the model:
    // persist_ok = false after a lost-co-residency report sends every plan's LSTM to the launch-per-step kernel.
    unsigned* status_host = nullptr;
    unsigned* status_dev = nullptr;
    mutable std::atomic<bool> persist_ok{true};
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
// Range sites (wt_plan_create_ex, wt_plan_range_sites): the units in which a plan can leave the split-f16 (S32) form.  Every
// S32 tensor is produced and consumed inside ONE site, so a site can run on fp32 operands (gemm.hip) on its own while the
// rest of the plan stays on gemm16s.hip: an activation beyond the f16 range in one ConvNeXt block costs that block's two
// GEMMs on the fp32 pipe, not the whole model.  The steps of a site report WT_STATUS_RANGE into the
// site's own word of the call's control block (ctl[CTL_SITE0 + site]); the guard step publishes the mask of sites that did.
enum Site : int {
    SITE_ENC = 0,            // the whole encode plan (and WT_PLAN_UNIT_LSTM)
    SITE_BB_EMBED = 1,       // transpose + backbone.embed
    SITE_RES0 = 2, SITE_RES1 = 3, SITE_ATTN = 4, SITE_RES2 = 5, SITE_RES3 = 6,
    SITE_CNX0 = 7,           // ConvNeXt block i: SITE_CNX0 + i (num_layers <= 32)
    SITE_HEAD = 40,          // final LayerNorm output + ISTFT head (also the WT_PLAN_HEAD plan)
    SITE_SEADEC = 41,        // the SEANetDecoder plan
    SITE_COUNT = 64
};
constexpr int CTL_SITE0 = 16;                 // first site word of the control block
constexpr int CTL_WORDS = CTL_SITE0 + SITE_COUNT + 48;      // 128 words = 512 bytes

// what a stage buffer holds (wt_plan_buffer_info): fp32 values, or the S32 split-f16 encoding of them; and whether
// the values are the reference tensor's or elu() of it (producers apply ELU once where the only consumer is ELU -> conv)
enum BufFmt : int { BUF_F32 = 0, BUF_S32 = 1, BUF_ELU = 2 };
struct BufSpec {
    std::string name;
    size_t bytes = 0, numel = 0, off = 0;
    int first = INT_MAX, last = -1;
    int fmt = BUF_F32;
};

// current device switched for the duration of a call, restored on every exit path
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess; else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace wt

struct wt_plan {
    const wt_model* model = nullptr;
    int kind = 0, B = 0, flags = 0;
    int64_t len = 0, L = 0, T = 0;
    uint64_t fp32_sites = 0;              // sites (wt::Site) that run on fp32 operands (wt_plan_create_ex)
    int cur_site = 0;                     // site of the steps being added (plan.cpp)
    std::vector<wt::BufSpec> bufs;
    std::vector<std::function<int(const wt::RunCtx&)>> steps;
    std::vector<std::string> step_names;
    std::vector<int> step_sites;          // site of every step: its kernels report WT_STATUS_RANGE into ctl[CTL_SITE0 + site]
    // WT_PLAN_FLAG_RANGE_REPORT: after every step, the largest magnitude in each S32 buffer the step touches (capi.cpp)
    struct RangeEntry { int step, buf; };
    std::vector<RangeEntry> range_entries;
    unsigned* range_dev = nullptr;        // one word (fp32 bit pattern of the maximum) per entry
    mutable std::vector<float> range_host;
    mutable bool range_fresh = false;
    size_t ws_bytes = 0;
    int n_launches = 0;
    // optional HIP-event timing of the steps whose name contains `timing_filter` (bench.py roofline
    // leg); mutable profiling state, not thread-safe, off by default
    mutable std::string timing_filter;
    mutable std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending, ev_free;
    // "@name" timing: device-side stamps of the gemm16s launches (capi.cpp): [0, STAMP_SLOTS) entry clocks, then exit clocks
    static constexpr int STAMP_SLOTS = 8192;
    mutable unsigned long long* stamps = nullptr;
    mutable int stamp_next = 0;
    mutable double timing_ms = 0.0;
    mutable long timing_n = 0;
    // Call status (common.h WT_STATUS_*): the first 256 bytes of the workspace are the call's control block, whose word 0
    // kernels OR failure bits into; the guard step that ends the plan poisons the outputs when it is non-zero and ORs it
    // into this host-mapped word, which the next host call on the plan (or wt_plan_status) consumes
    unsigned* status_host = nullptr;
    unsigned* status_dev = nullptr;       // device address of status_host
    int ctl = -1;                         // buffer id of the control block
    bool uses_persist = false;            // some step launches lstm_persist_kernel (while wt_model::persist_ok holds)
    mutable bool graph_persist = false;   // the recorded graph holds a persistent LSTM launch
    // one host call at a time per plan (graph capture state, timing events and persist_ok are per plan)
    mutable std::mutex mu;
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

    wt_plan() = default;
    wt_plan(const wt_plan&) = delete;
    ~wt_plan() {
        if (status_host) (void)hipHostFree(status_host);
        if (range_dev) (void)hipFree(range_dev);
        if (stamps) (void)hipFree(stamps);
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        for (auto& ev : ev_pending) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
        for (auto& ev : ev_free) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    }

    int buf(const std::string& name, size_t numel, int fmt = wt::BUF_F32) {
        wt::BufSpec b;
        b.name = name; b.numel = numel; b.bytes = (numel * sizeof(float) + 255) / 256 * 256; b.fmt = fmt;
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
        if (flags & WT_PLAN_FLAG_RANGE_REPORT)
            for (int id : used)
                if (id >= 0 && (bufs[id].fmt & wt::BUF_S32)) range_entries.push_back({s, id});
        steps.push_back(std::move(fn));
        step_names.push_back(nm);
        step_sites.push_back(cur_site);
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

// weights.cpp
int build_model(wt_model* M, TensorMap& tm);
int build_splits(wt_model* M);
// packed image of a model (weights.cpp): everything wt_model_create computes and uploads, ready to upload again
size_t model_export_bytes(const wt_model* M);
int model_export(const wt_model* M, void* buf, size_t n);
int model_import(wt_model* M, const void* buf, size_t n);        // M->device set; allocates and uploads
int packed_info(const void* buf, size_t n, wt_arch* arch, int32_t* version, uint64_t* arch_hash);
int ensure_f32_weights(const wt_model* M);                        // fills the lazy fp32 arrays of a packed model (first fp32 plan)
size_t packed_bytes(const void* buf, size_t n);                  // exact length of the image at buf (0: bad header)
int packed_verify(const void* buf, size_t n);                    // header + bounds + content hash; needs no GPU
// plan.cpp
struct SConvGeom { int pl, pr_total, Tout, Tp; };
SConvGeom sconv_geom(long T, int k, int stride, int dil);
GemmArgs sconv_args(const ConvW& w, int B, long T, int stride, int dil);
GemmArgs zconv_args(const ConvW& w, int B, int L);
GemmArgs linear_args(const float* W, const float* bias, long M, int N, int K);
int build_encode(wt_plan* P);
int build_decode(wt_plan* P);
int build_head(wt_plan* P);
int build_seanet_decoder(wt_plan* P);
int build_unit_lstm(wt_plan* P);
// the control block and the closing guard step that every plan kind gets (plan.cpp)
void plan_begin(wt_plan* P);
void plan_end(wt_plan* P);
// samples per clip that the ISTFT head writes for L frames: "same" L * hop, "center" (torch.istft(center=True)) (L - 1) * hop
inline long wave_samples(const wt_model* M, long L) {
    return M->arch.padding_same ? L * (long)M->arch.hop_length : (L - 1) * (long)M->arch.hop_length;
}

}  // namespace wt
