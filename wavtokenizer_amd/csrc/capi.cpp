// The extern "C" entry points declared in include/wavtokenizer_amd.h.
#include "model.h"

namespace wt {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
thread_local LaunchCtx g_launch;
int device_cus() {
    static std::atomic<int> cache[64];
    int d = 0;
    (void)hipGetDevice(&d);
    d &= 63;
    int v = cache[d].load(std::memory_order_relaxed);
    if (v <= 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256;
        cache[d].store(v, std::memory_order_relaxed);
    }
    return v;
}
}  // namespace wt

// ================================================================================== C ABI
using namespace wt;

// one host-mapped block per model: word 0 = wt_codes_to_features' bad-index flag, word 16 = the model-level call status
static int alloc_host_words(wt_model* M, const char* who) {
    void* hp = nullptr;
    void* dp = nullptr;
    if (hipHostMalloc(&hp, 256, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
        if (hp) (void)hipHostFree(hp);
        set_error(std::string(who) + ": no host-mapped memory for the status words");
        return WT_ERR_HIP;
    }
    memset(hp, 0, 256);
    M->bad_codes_host = static_cast<unsigned*>(hp);
    M->bad_codes_dev = static_cast<unsigned*>(dp);
    M->status_host = static_cast<unsigned*>(hp) + 16;
    M->status_dev = static_cast<unsigned*>(dp) + 16;
    return WT_OK;
}

// The library holds gfx950 code objects only.  Launch geometry follows the device's CU count (device_cus()); what is tied to
// the full 256-CU / 8-XCD MI355X is the persistent LSTM (plan.cpp: any other CU count runs the launch-per-step kernel) and
// the tuning of the tile orders (speed only).  A device of another architecture is refused here instead of at the first launch.
static int check_device(int device, const char* who) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); set_error(std::string(who) + ": no such device"); return WT_ERR_HIP; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error(std::string(who) + ": device is " + prop.gcnArchName + "; this library is built for gfx950 (MI355X) only");
        return WT_ERR_INVALID;
    }
    if (prop.multiProcessorCount < 8) { set_error(std::string(who) + ": fewer than 8 compute units"); return WT_ERR_INVALID; }
    return WT_OK;
}

extern "C" {

const char* wt_last_error(void) { return g_err.c_str(); }
#ifdef WT_LAB
const char* wt_version(void) { return "wavtokenizer_amd 0.1 LAB build (gfx950, split-f16 MFMA products, fp32 accumulation; timing experiments and fault hooks compiled in)"; }
#else
const char* wt_version(void) { return "wavtokenizer_amd 0.1 (gfx950, split-f16 MFMA products, fp32 accumulation)"; }
#endif

int wt_model_create(const wt_arch* arch, const wt_tensor* tensors, int32_t n_tensors, int32_t device, wt_model** out) {
    if (!arch || !tensors || !out) { set_error("wt_model_create: null argument"); return WT_ERR_INVALID; }
    if (arch->n_ratios < 1 || arch->n_ratios > 8) { set_error("n_ratios out of range"); return WT_ERR_INVALID; }
    if (arch->num_quantizers < 1 || arch->num_quantizers > 32) { set_error("num_quantizers must be 1 .. 32 (encode_infer uses the first codebook: vq.py:137 forces n_q = 1; codes_to_features sums up to num_quantizers of them)"); return WT_ERR_INVALID; }
    if (arch->input_channels != 512) { set_error("input_channels must be 512 (SEANet dimension)"); return WT_ERR_INVALID; }
    if (arch->dim % 256 || arch->intermediate_dim % 32) { set_error("dim must be a multiple of 256, intermediate_dim of 32"); return WT_ERR_INVALID; }
    if (arch->dim % 32 || (arch->dim / 32) % 4) { set_error("dim/32 (GroupNorm group width) must be a multiple of 4"); return WT_ERR_INVALID; }
    DeviceGuard dg(device);
    if (!dg.ok) { set_error("wt_model_create: hipSetDevice failed"); return WT_ERR_HIP; }
    if (int rc = check_device(device, "wt_model_create")) return rc;
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
    if (!rc) rc = alloc_host_words(M.get(), "wt_model_create");
    if (rc) {
        if (rc == WT_ERR_MISSING_TENSOR) set_error("state_dict tensor missing or mis-shaped: " + tm.missing);
        for (void* p : M->allocs) (void)hipFree(p);
        return rc;
    }
    *out = M.release();
    return WT_OK;
}

size_t wt_model_export_bytes(const wt_model* m) { return m ? model_export_bytes(m) : 0; }
int wt_model_export(const wt_model* m, void* buf, size_t n) {
    if (!m || !buf) { set_error("wt_model_export: null argument"); return WT_ERR_INVALID; }
    DeviceGuard dg(m->device);
    if (!dg.ok) { set_error("hipSetDevice failed"); return WT_ERR_HIP; }
    return model_export(m, buf, n);
}
int wt_packed_info(const void* buf, size_t n, wt_arch* arch, int32_t* version, uint64_t* arch_hash) {
    return packed_info(buf, n, arch, version, arch_hash);
}
size_t wt_packed_bytes(const void* buf, size_t n) { return buf ? packed_bytes(buf, n) : 0; }
int wt_packed_verify(const void* buf, size_t n) {
    try { return packed_verify(buf, n); }
    catch (const std::exception& e) { set_error(std::string("wt_packed_verify: ") + e.what()); return WT_ERR_INVALID; }
}
int wt_model_create_packed(const void* buf, size_t n, int32_t device, wt_model** out) {
    if (!buf || !out) { set_error("wt_model_create_packed: null argument"); return WT_ERR_INVALID; }
    if (int rc = packed_info(buf, n, nullptr, nullptr, nullptr)) return rc;
    DeviceGuard dg(device);
    if (!dg.ok) { set_error("wt_model_create_packed: hipSetDevice failed"); return WT_ERR_HIP; }
    if (int rc = check_device(device, "wt_model_create_packed")) return rc;
    std::unique_ptr<wt_model> M(new wt_model());
    M->device = device;
    int rc;
    try { rc = model_import(M.get(), buf, n); }      // nothing may throw across the C boundary (a bad file must not end the process)
    catch (const std::exception& e) { set_error(std::string("wt_model_create_packed: ") + e.what()); rc = WT_ERR_INVALID; }
    if (!rc) rc = alloc_host_words(M.get(), "wt_model_create_packed");
    if (rc) {
        for (void* p : M->allocs) (void)hipFree(p);
        return rc;
    }
    *out = M.release();
    return WT_OK;
}

void wt_model_destroy(wt_model* m) {
    if (!m) return;
    DeviceGuard dg(m->device);
    for (void* p : m->allocs) (void)hipFree(p);
    if (m->bad_codes_host) (void)hipHostFree(m->bad_codes_host);
    delete m;
}
int wt_model_split_ok(const wt_model* m) { return m && m->s32_ok ? 1 : 0; }
int wt_model_take_bad_codes(const wt_model* m) {
    if (!m || !m->bad_codes_host) return 0;
    const unsigned v = __atomic_exchange_n(m->bad_codes_host, 0u, __ATOMIC_RELAXED);
    return v ? 1 : 0;
}
int wt_model_status(const wt_model* m, int32_t* bits, int32_t clear) {
    if (!m) return WT_ERR_INVALID;
    unsigned b = 0;
    if (m->status_host) b = clear ? __atomic_exchange_n(m->status_host, 0u, __ATOMIC_ACQUIRE) : __atomic_load_n(m->status_host, __ATOMIC_ACQUIRE);
    if (b & WT_STATUS_LSTM) m->persist_ok.store(false);
    if (bits) *bits = (int32_t)b;
    return WT_OK;
}
int wt_model_persistent_lstm(const wt_model* m) {
    if (!m || !m->persist_ok.load()) return 0;
    int cus = 0;
    return hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device) == hipSuccess && cus == 256 ? 1 : 0;
}
int wt_device_info(int32_t device, int32_t* compute_units, int32_t* is_gfx950, int32_t* persistent_lstm) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); set_error("wt_device_info: no such device"); return WT_ERR_HIP; }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (is_gfx950) *is_gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
    if (persistent_lstm) *persistent_lstm = prop.multiProcessorCount == 256 ? 1 : 0;
    return WT_OK;
}
int wt_model_hop(const wt_model* m) { return m ? m->hop : 0; }
int64_t wt_model_weight_bytes(const wt_model* m) { return m ? m->weight_bytes : 0; }

int wt_plan_create(const wt_model* m, int32_t kind, int32_t B, int64_t len, int32_t flags, wt_plan** out) {
    return wt_plan_create_ex(m, kind, B, len, flags, 0, out);
}

int wt_plan_create_ex(const wt_model* m, int32_t kind, int32_t B, int64_t len, int32_t flags, uint64_t fp32_sites, wt_plan** out) {
    if (!m || !out) { set_error("wt_plan_create: null argument"); return WT_ERR_INVALID; }
    if (m->arch.num_layers > SITE_HEAD - SITE_CNX0) { set_error("wt_plan_create: more ConvNeXt blocks than range sites"); return WT_ERR_INVALID; }
    if (B < 1 || len < 1) { set_error("wt_plan_create: B and len must be >= 1"); return WT_ERR_INVALID; }
    if ((len + (kind == WT_PLAN_ENCODE ? m->hop - 1 : 0)) / (kind == WT_PLAN_ENCODE ? m->hop : 1) > 12000) {
        set_error("clips longer than 12000 frames are not supported by one plan; split the clip"); return WT_ERR_INVALID;
    }
    DeviceGuard dg(m->device);
    if (!dg.ok) { set_error("wt_plan_create: hipSetDevice failed"); return WT_ERR_HIP; }
    std::unique_ptr<wt_plan> P(new wt_plan());
    P->model = m; P->kind = kind; P->B = B; P->len = len; P->flags = flags; P->fp32_sites = fp32_sites;
    {
        void* hp = nullptr;
        void* dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
            if (hp) (void)hipHostFree(hp);
            set_error("wt_plan_create: no host-mapped memory for the status word"); return WT_ERR_HIP;
        }
        P->status_host = static_cast<unsigned*>(hp);       // word 0: status bits; words 2, 3: mask of the range sites that reported
        P->status_dev = static_cast<unsigned*>(dp);
        memset(hp, 0, 64);
    }
    // a model that came from a packed image holds no fp32 copies of its GEMM weights until a plan needs them
    if (m->f32_stale.load() && ((flags & (WT_PLAN_FLAG_FP32_GEMM | WT_PLAN_FLAG_UNFUSED)) || fp32_sites || !m->s32_ok ||
                                (kind == WT_PLAN_SEANET_DECODER && !m->sd_s32_ok)))
        if (int rc0 = ensure_f32_weights(m)) return rc0;
    plan_begin(P.get());
    int rc;
    if (kind == WT_PLAN_ENCODE) {
        P->T = len;
        P->L = (len + m->hop - 1) / m->hop;
        if ((long)B * len >= (long)INT_MAX) { set_error("batch too large for one plan (32-bit row index)"); return WT_ERR_INVALID; }
        rc = build_encode(P.get());
    } else if (kind == WT_PLAN_DECODE) {
        P->L = len; P->T = len * m->hop;
        if (!m->arch.padding_same && len < 2) { set_error("ISTFT padding='center' needs at least two frames"); return WT_ERR_INVALID; }
        rc = build_decode(P.get());
    } else if (kind == WT_PLAN_SEANET_DECODER) {
        P->L = len; P->T = len * m->hop;
        rc = build_seanet_decoder(P.get());
    } else if (kind == WT_PLAN_HEAD) {
        P->L = len; P->T = len * m->hop;
        if (!m->arch.padding_same && len < 2) { set_error("ISTFT padding='center' needs at least two frames"); return WT_ERR_INVALID; }
        rc = build_head(P.get());
    } else if (kind == WT_PLAN_UNIT_LSTM) {
        P->L = len; P->T = len * m->hop;
        rc = build_unit_lstm(P.get());
    } else {
        set_error("unknown plan kind"); rc = WT_ERR_INVALID;
    }
    if (rc) return rc;              // ~wt_plan releases the host-mapped word
    plan_end(P.get());
    P->layout();
    if (!P->range_entries.empty()) {
        if (hipMalloc(reinterpret_cast<void**>(&P->range_dev), (P->range_entries.size() * sizeof(unsigned) + 15) / 16 * 16) != hipSuccess) {
            P->range_dev = nullptr; set_error("wt_plan_create: no memory for the range report"); return WT_ERR_HIP;
        }
        P->range_host.assign(P->range_entries.size(), 0.f);
    }
    *out = P.release();
    return WT_OK;
}
void wt_plan_destroy(wt_plan* p) {
    if (!p) return;
    DeviceGuard dg(p->model->device);
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
int wt_plan_buffer_info(const wt_plan* p, const char* name, size_t* offset, size_t* numel, int32_t* format) {
    if (!p || !name) return WT_ERR_INVALID;
    for (const BufSpec& b : p->bufs)
        if (b.name == name) {
            if (offset) *offset = b.off;
            if (numel) *numel = b.numel;
            if (format) *format = b.fmt;
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

// Consumes the failure bits that earlier calls left behind (the plan's lock is held).  Every plan's guard step reports
// into two host-mapped words: the plan's own (attribution: wt_plan_status) and the MODEL's, which is the one that makes
// the next call fail: a caller who never uses a plan twice (one new length per file) still meets the error on its next
// call, and a failure that another plan's call has already consumed and answered is not reported a second time when
// this plan runs again (its own word is then stale and is just cleared).  After a lost-co-residency report every plan
// of the model runs the LSTM one launch per step from now on (wt_model::persist_ok), and a recorded graph that holds a
// persistent launch is dropped.  Returns the model's bits; *own receives this plan's.
static unsigned consume_status(const wt_plan* p, unsigned* own = nullptr) {
    const wt_model* M = p->model;
    const unsigned pb = p->status_host ? __atomic_exchange_n(p->status_host, 0u, __ATOMIC_ACQUIRE) : 0u;
    const unsigned mb = M->status_host ? __atomic_exchange_n(M->status_host, 0u, __ATOMIC_ACQUIRE) : 0u;
    if ((mb | pb) & WT_STATUS_LSTM) M->persist_ok.store(false);
    if (p->graph_exec && p->graph_persist && !M->persist_ok.load()) {
        (void)hipGraphExecDestroy(p->graph_exec);
        p->graph_exec = nullptr; p->graph_persist = false;
        p->last_key = wt_plan::GraphKey{};
    }
    if (own) *own = pb;
    return mb;
}

// Two persistent LSTM launches must never share the GPU: lstm_persist_kernel spins until all of its workgroups are resident
// (one per CU), so two of them enqueued on different streams could each hold part of the CUs and wait for the rest until
// their spin bounds expire (both calls then fail with WT_ERR_LSTM_SYNC).  Calls that carry one are therefore chained per
// device: a call on another stream than the previous one first waits (on the GPU, hipStreamWaitEvent) for the event recorded
// behind that previous call.  The lock is held from the wait to the record, so concurrent host threads are ordered too.
// Kernels of other plans may run beside a persistent launch: they finish on their own and its workgroups then take their CUs.
struct LstmChain {
    std::mutex mu;
    hipEvent_t ev = nullptr;
    hipStream_t last = nullptr;
    bool any = false;       // a call has been made
    bool multi = false;     // calls have come from more than one stream: from then on every call records the event
    bool have = false;      // ev marks the end of the previous call
};
static LstmChain g_lstm_chain[64];
struct LstmChainScope {
    LstmChain* ch = nullptr;
    hipStream_t stream = nullptr;
    int rc = WT_OK;
    LstmChainScope(const wt_plan* p, hipStream_t s) {
        if (!p->uses_persist || p->model->device < 0 || p->model->device >= 64) return;
        ch = &g_lstm_chain[p->model->device];
        stream = s;
        ch->mu.lock();
        if (!ch->any || ch->last == s) return;
        if (!ch->ev && hipEventCreateWithFlags(&ch->ev, hipEventDisableTiming) != hipSuccess) { ch->ev = nullptr; return; }
        if (!ch->multi) {
            // first call from a second stream.  Single-stream callers never pay for an event record (it is a packet of its own
            // in the stream: 4-7 us), so there is none behind the previous call: order this one behind everything that stream
            // holds right now instead (conservative, once), and record from here on
            ch->multi = true;
            ch->have = hipEventRecord(ch->ev, ch->last) == hipSuccess;
            if (!ch->have) (void)hipGetLastError();      // (the stream may be gone: then so is its work)
        }
        if (ch->have && hipStreamWaitEvent(s, ch->ev, 0) != hipSuccess) { set_error("hipStreamWaitEvent failed"); rc = WT_ERR_HIP; }
    }
    ~LstmChainScope() {
        if (!ch) return;
        if (ch->multi) {
            if (!ch->ev && hipEventCreateWithFlags(&ch->ev, hipEventDisableTiming) != hipSuccess) ch->ev = nullptr;
            ch->have = ch->ev && hipEventRecord(ch->ev, stream) == hipSuccess;
        }
        ch->last = stream;
        ch->any = true;
        ch->mu.unlock();
    }
};

static int run_plan_locked(const wt_plan* p, const RunCtx& c);
static int run_plan(const wt_plan* p, const RunCtx& c) {
    std::lock_guard<std::mutex> lock(p->mu);
    DeviceGuard dg(p->model->device);
    if (!dg.ok) { set_error("hipSetDevice failed"); return WT_ERR_HIP; }
    if (const unsigned bits = consume_status(p)) {
        if (bits & WT_STATUS_LSTM) {
            set_error("an earlier persistent LSTM launch of this model lost co-residency (a step barrier timed out); that call's "
                      "outputs were overwritten (codes = -1, NaN); the model's plans now run the LSTM one launch per step: repeat the call");
            return WT_ERR_LSTM_SYNC;
        }
        set_error("an earlier call on this model met a value outside the f16 range of the split-f16 (S32) form (|v| >= 65504); "
                  "that call's outputs were overwritten (codes = -1, NaN); re-plan with WT_PLAN_FLAG_FP32_GEMM and repeat the call");
        return WT_ERR_RANGE;
    }
    struct CtxScope {        // the launch functions take the status word from this thread's context while the steps run
        explicit CtxScope(unsigned* s) { g_launch.status = s; }
        ~CtxScope() { g_launch.status = nullptr; }
    } scope(reinterpret_cast<unsigned*>(c.ws + p->bufs[p->ctl].off));
    LstmChainScope chain(p, c.stream);
    if (chain.rc) return chain.rc;
    return run_plan_locked(p, c);
}

static int run_plan_locked(const wt_plan* p, const RunCtx& c) {
    const bool timing = !p->timing_filter.empty();
    if ((p->flags & WT_PLAN_FLAG_GRAPH) && !timing && !p->graph_failed && !p->range_dev) {
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
            unsigned* const cap_base = reinterpret_cast<unsigned*>(cc.ws + p->bufs[p->ctl].off);
            for (size_t i = 0; i < p->steps.size() && !rc; ++i) {
                g_launch.status = cap_base + CTL_SITE0 + p->step_sites[i];
                rc = p->steps[i](cc);
            }
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
                    // the recording holds a persistent launch only if the model still allowed one when it was made: after a
                    // lost-co-residency fallback the steps record the launch-per-step kernel, and such a graph must survive
                    // consume_status (it used to be destroyed and re-captured on every other call for the rest of the model's life)
                    p->graph_persist = p->uses_persist && p->model->persist_ok.load();
                    WT_HIP_CHECK(hipGraphLaunch(p->graph_exec, c.stream));
                    ++p->graph_replays;
                    return WT_OK;
                }
            }
        }
        p->last_key = key;
    }
    // "@name": no events - the step's gemm16s launch stamps its own entry / exit on the device (LaunchCtx).  An event record
    // is a packet of its own: bracketing a launch puts 4-7 us between it and its neighbours and counts them in (measured:
    // pwconv1 94.6 us between bracketing events, 88.7 us in the rocprofv3 trace of the same run; hipExtLaunchKernel's
    // start / stop events behave the same: 95.1 vs 90.3)
    const bool stamp = timing && p->timing_filter[0] == '@';
    const std::string filt = stamp ? p->timing_filter.substr(1) : p->timing_filter;
    unsigned* const ctl_base = reinterpret_cast<unsigned*>(c.ws + p->bufs[p->ctl].off);
    size_t next_range = 0;
    if (p->range_dev) {
        if (int rc = launch_fill_u32(p->range_dev, 0u, (p->range_entries.size() * sizeof(unsigned) + 15) / 16 * 16, c.stream)) return rc;
        p->range_fresh = false;
    }
    // WT_PLAN_FLAG_RANGE_REPORT: behind step i, the largest magnitude in every S32 buffer the step touches
    auto measure_ranges = [&](size_t i) -> int {
        for (; next_range < p->range_entries.size() && p->range_entries[next_range].step == (int)i; ++next_range) {
            const BufSpec& b = p->bufs[p->range_entries[next_range].buf];
            if (int rc = launch_s32_amax(c.ws + b.off, (long)b.numel, p->range_dev + next_range, c.stream)) return rc;
        }
        return 0;
    };
    for (size_t i = 0; i < p->steps.size(); ++i) {
        // the step's kernels report into their site's word of the control block (model.h Site)
        g_launch.status = ctl_base + CTL_SITE0 + p->step_sites[i];
        const bool timed = timing && p->step_names[i].find(filt) != std::string::npos;
        std::pair<hipEvent_t, hipEvent_t> ev;
        if (timed && stamp) {
            if (p->stamps && p->stamp_next < wt_plan::STAMP_SLOTS) {
                g_launch.stamp_start = p->stamps + p->stamp_next;
                g_launch.stamp_end = p->stamps + wt_plan::STAMP_SLOTS + p->stamp_next;
                g_launch.stamp_used = false;
            }
            const int step_rc = p->steps[i](c);
            if (!step_rc) if (int rc = measure_ranges(i)) return rc;
            const bool armed = g_launch.stamp_start != nullptr, used = g_launch.stamp_used;
            g_launch.stamp_start = g_launch.stamp_end = nullptr; g_launch.stamp_used = false;
            if (step_rc) return step_rc;
            if (armed && !used) {
                set_error("wt_plan_set_timing(\"@...\"): step '" + p->step_names[i] + "' does not launch a gemm16s kernel");
                return WT_ERR_INVALID;
            }
            if (armed) ++p->stamp_next;
            continue;
        }
        if (timed) {
            if (!p->ev_free.empty()) { ev = p->ev_free.back(); p->ev_free.pop_back(); }
            else { WT_HIP_CHECK(hipEventCreate(&ev.first)); WT_HIP_CHECK(hipEventCreate(&ev.second)); }
            WT_HIP_CHECK(hipEventRecord(ev.first, c.stream));
        }
        const int step_rc = p->steps[i](c);
        if (int rc = step_rc) return rc;
        if (int rc = measure_ranges(i)) return rc;
        const bool dbg_status = lab_env("WT_DEBUG_STATUS") != nullptr;      // LAB builds only
        if (dbg_status) {        // debugging aid: which step left a non-zero status word (synchronises after every step)
            unsigned st = 0;
            WT_HIP_CHECK(hipStreamSynchronize(c.stream));
            WT_HIP_CHECK(hipMemcpy(&st, c.ws + p->bufs[p->ctl].off, sizeof(st), hipMemcpyDeviceToHost));
            if (st) fprintf(stderr, "[wt status] plan kind %d B %d len %ld: step %zu (%s) -> status 0x%08x\n", p->kind, p->B, (long)p->len, i, p->step_names[i].c_str(), st);
        }
        if (timed) {
            WT_HIP_CHECK(hipEventRecord(ev.second, c.stream));
            p->ev_pending.push_back(ev);
        }
    }
    return WT_OK;
}

int wt_plan_status(const wt_plan* p, int32_t* bits, int32_t clear) {
    if (!p) return WT_ERR_INVALID;
    std::lock_guard<std::mutex> lock(p->mu);
    unsigned b;
    if (clear) {
        DeviceGuard dg(p->model->device);
        unsigned own = 0;
        b = consume_status(p, &own) | own;      // this plan's failures and whatever the model's word still held
    } else {
        b = p->status_host ? __atomic_load_n(p->status_host, __ATOMIC_ACQUIRE) : 0u;
    }
    if (bits) *bits = (int32_t)b;
    return WT_OK;
}

int wt_plan_range_sites(const wt_plan* p, uint64_t* sites, int32_t clear) {
    if (!p || !sites) return WT_ERR_INVALID;
    std::lock_guard<std::mutex> lock(p->mu);
    unsigned lo = 0, hi = 0;
    if (p->status_host) {
        lo = clear ? __atomic_exchange_n(p->status_host + 2, 0u, __ATOMIC_ACQUIRE) : __atomic_load_n(p->status_host + 2, __ATOMIC_ACQUIRE);
        hi = clear ? __atomic_exchange_n(p->status_host + 3, 0u, __ATOMIC_ACQUIRE) : __atomic_load_n(p->status_host + 3, __ATOMIC_ACQUIRE);
    }
    *sites = ((uint64_t)hi << 32) | lo;
    return WT_OK;
}

int wt_plan_range_report(const wt_plan* p, int32_t index, const char** step, const char** buffer, float* amax) {
    if (!p) return WT_ERR_INVALID;
    std::lock_guard<std::mutex> lock(p->mu);
    if (!p->range_dev) { set_error("wt_plan_range_report: the plan was not created with WT_PLAN_FLAG_RANGE_REPORT"); return WT_ERR_INVALID; }
    if (index < 0 || index >= (int)p->range_entries.size()) return WT_ERR_INVALID;       // past the end (no message: callers iterate)
    if (!p->range_fresh) {
        DeviceGuard dg(p->model->device);
        WT_HIP_CHECK(hipDeviceSynchronize());
        static_assert(sizeof(float) == sizeof(unsigned), "bit patterns");
        WT_HIP_CHECK(hipMemcpy(p->range_host.data(), p->range_dev, p->range_entries.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        p->range_fresh = true;
    }
    const wt_plan::RangeEntry& e = p->range_entries[index];
    if (step) *step = p->step_names[e.step].c_str();
    if (buffer) *buffer = p->bufs[e.buf].name.c_str();
    if (amax) *amax = p->range_host[index];
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
    if (!p->timing_filter.empty() && p->timing_filter[0] == '@') {
        // device stamps: entry clocks start as all-ones (atomic min), exit clocks as zero (atomic max); one slot per timed launch
        DeviceGuard dg(p->model->device);
        const size_t half = (size_t)wt_plan::STAMP_SLOTS * sizeof(unsigned long long);
        if (!p->stamps) WT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p->stamps), 2 * half));
        WT_HIP_CHECK(hipDeviceSynchronize());
        WT_HIP_CHECK(hipMemset(p->stamps, 0xFF, half));
        WT_HIP_CHECK(hipMemset(p->stamps + wt_plan::STAMP_SLOTS, 0, half));
        WT_HIP_CHECK(hipDeviceSynchronize());
        p->stamp_next = 0;
    }
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
    if (p->stamps && p->stamp_next > 0) {
        DeviceGuard dg(p->model->device);
        WT_HIP_CHECK(hipDeviceSynchronize());
        const int n = p->stamp_next;
        std::vector<unsigned long long> t0(n), t1(n);
        WT_HIP_CHECK(hipMemcpy(t0.data(), p->stamps, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        WT_HIP_CHECK(hipMemcpy(t1.data(), p->stamps + wt_plan::STAMP_SLOTS, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i)
            if (t1[i] > t0[i]) { p->timing_ms += (double)(t1[i] - t0[i]) * 1e-5; p->timing_n += 1; }      // 100 MHz ticks -> ms
        const size_t half = (size_t)wt_plan::STAMP_SLOTS * sizeof(unsigned long long);
        WT_HIP_CHECK(hipMemset(p->stamps, 0xFF, half));
        WT_HIP_CHECK(hipMemset(p->stamps + wt_plan::STAMP_SLOTS, 0, half));
        WT_HIP_CHECK(hipDeviceSynchronize());
        p->stamp_next = 0;
    }
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

int wt_unit_run(const wt_plan* p, const float* x, float* y, void* workspace, void* stream) {
    if (!p || p->kind != WT_PLAN_UNIT_LSTM) { set_error("wt_unit_run: wrong plan kind"); return WT_ERR_INVALID; }
    if (!x || !y || !workspace) { set_error("wt_unit_run: null buffer"); return WT_ERR_INVALID; }
    RunCtx c{static_cast<char*>(workspace), static_cast<hipStream_t>(stream), x, y, nullptr, nullptr, 0};
    return run_plan(p, c);
}

int wt_codes_to_features(const wt_model* m, const int64_t* codes, int32_t K, int32_t B, int64_t L, float* features,
                         void* stream) {
    if (!m || !codes || !features) { set_error("wt_codes_to_features: null argument"); return WT_ERR_INVALID; }
    if (K < 1 || K > m->arch.num_quantizers) { set_error("wt_codes_to_features: K exceeds the number of codebooks"); return WT_ERR_INVALID; }
    DeviceGuard dg(m->device);
    if (!dg.ok) { set_error("hipSetDevice failed"); return WT_ERR_HIP; }
    return launch_codes_to_features(codes, m->embed, K, m->arch.vq_bins, B, L, 512, features, static_cast<hipStream_t>(stream),
                                    m->bad_codes_dev);
}

int wt_sconv1d(const float* x, const float* w, const float* bias, float* y, int32_t B, int64_t T, int32_t Cin,
               int32_t Cout, int32_t k, int32_t stride, int32_t dilation, int32_t elu_input, void* stream) {
    ConvW cw; cw.w = const_cast<float*>(w); cw.b = const_cast<float*>(bias); cw.cout = Cout; cw.cin = Cin; cw.k = k;
    GemmArgs a = sconv_args(cw, B, T, stride, dilation);
    a.A = x; a.C = y;
    return launch_gemm(a, elu_input ? PRO_ELU : PRO_NONE, EPI_BIAS, static_cast<hipStream_t>(stream));
}

// Both operands of a single-stage S32 call are split here (the plans' producers write S32 directly), each with a
// per-tensor power-of-two scale chosen on the device; `tail` = 256 spare bytes after the two S32 arrays
static int split_pair(const float* w, long nw, const float* x, long nx, char* ws_w, char* ws_x, char* tail, GemmArgs& a,
                      hipStream_t s) {
    unsigned* bits = reinterpret_cast<unsigned*>(tail);
    float* sc = reinterpret_cast<float*>(tail + 16);              // {scale_w, scale_x, 1 / (scale_w * scale_x)}
    if (int rc = launch_pow2_scales(w, nw, x, nx, bits, sc, s)) return rc;
    if (int rc = launch_split_s32(w, ws_w, nw, s, sc)) return rc;
    if (int rc = launch_split_s32(x, ws_x, nx, s, sc + 1)) return rc;
    a.W_hi = ws_w;
    a.A = reinterpret_cast<const float*>(ws_x);
    a.acc_scale_dev = sc + 2;
    return 0;
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
    if (f16x3 == 1) { set_error("wt_linear: mode 1 (the in-loop split kernel of round 1) was removed; use 2, 3 or 4"); return WT_ERR_INVALID; }
    char* xs = hi + (size_t)N * K * 4;
    if (int rc = split_pair(w, (long)N * K, x, (long)M * K, hi, xs, xs + (size_t)M * K * 4, a, s)) return rc;
    // timing-experiment builds (WT_GEMM16S_DBG: tools/gemm16s_bench.py) leave their clock stamps behind the scales
    a.dbg_stamps = reinterpret_cast<unsigned long long*>(xs + (size_t)M * K * 4 + 256);
    if (f16x3 == 4) return launch_gemm16s(a, EPI_BIAS_GELU, OUT_S32, s);      // ConvNeXt pwconv1: exact-erf GELU epilogue, S32 out
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
    if (int rc = split_pair(w, (long)Cout * k * Cin, x, (long)B * T * Cin, ws, xs, xs + (size_t)B * T * Cin * 4, a, s)) return rc;
    a.C = y;
    return launch_gemm16s(a, EPI_BIAS, 0, s);
}

static size_t al256(size_t b) { return (b + 255) / 256 * 256; }
size_t wt_vq_workspace_bytes(int64_t N, int32_t D, int32_t bins) {
    const size_t np = std::max(gemm_vq_parts(bins), gemm16s_vq_parts(bins));
    return al256((size_t)N * D * 4) + al256((size_t)bins * D * 4) + al256((size_t)N * sizeof(float)) +
           2 * al256((size_t)N * np * sizeof(float)) + al256((size_t)bins * sizeof(float)) + 512;
}

// the ee[] table here is rebuilt per call on the device by row_sumsq (same kernel as |x|^2)
static int vq_nearest(const float* x, const float* embed, int64_t N, int32_t D, int32_t bins, int64_t* codes_out,
                      void* workspace, void* stream, bool s32) {
    if (!x || !embed || !codes_out || !workspace) { set_error("wt_vq_nearest: null argument"); return WT_ERR_INVALID; }
    if (s32 && (D % 32)) { set_error("wt_vq_nearest: the split-f16 kernel needs D % 32 == 0"); return WT_ERR_INVALID; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int np = s32 ? gemm16s_vq_parts(bins) : gemm_vq_parts(bins);
    char* ws = static_cast<char*>(workspace);
    char* xs = ws; ws += al256((size_t)N * D * 4);
    char* es = ws; ws += al256((size_t)bins * D * 4);
    float* xx = reinterpret_cast<float*>(ws); ws += al256((size_t)N * sizeof(float));
    float* pv = reinterpret_cast<float*>(ws); ws += al256((size_t)N * np * sizeof(float));
    int* pi = reinterpret_cast<int*>(ws); ws += al256((size_t)N * np * sizeof(float));
    float* ee = reinterpret_cast<float*>(ws); ws += al256((size_t)bins * sizeof(float));
    if (int rc = launch_row_sumsq(x, xx, N, D, s)) return rc;
    if (int rc = launch_row_sumsq(embed, ee, bins, D, s)) return rc;
    GemmArgs a = linear_args(embed, nullptr, N, bins, D);
    a.A = x; a.vq_xx = xx; a.vq_ee = ee; a.vq_pval = pv; a.vq_pidx = pi; a.vq_nparts = np;
    if (s32) {
        // what the encoder plan launches: distances on gemm16s.hip, per-slab argmax in its epilogue
        if (int rc = split_pair(embed, (long)bins * D, x, (long)N * D, es, xs, ws, a, s)) return rc;
        if (int rc = launch_gemm16s(a, EPI_ARGMAX, OUT_F32, s)) return rc;
    } else {
        if (int rc = launch_gemm(a, PRO_NONE, EPI_ARGMAX, s)) return rc;
    }
    for (int64_t r0 = 0; r0 < N; r0 += 8192) {
        const int n = (int)std::min<int64_t>(8192, N - r0);
        if (int rc = launch_vq_finalize(pv + r0 * np, pi + r0 * np, np, embed, codes_out + r0, nullptr, 1, n, D, bins, s)) return rc;
    }
    return WT_OK;
}
int wt_vq_nearest(const float* x, const float* embed, int64_t N, int32_t D, int32_t bins, int64_t* codes_out,
                  void* workspace, void* stream) {
    return vq_nearest(x, embed, N, D, bins, codes_out, workspace, stream, true);
}
int wt_vq_nearest_f32(const float* x, const float* embed, int64_t N, int32_t D, int32_t bins, int64_t* codes_out,
                      void* workspace, void* stream) {
    return vq_nearest(x, embed, N, D, bins, codes_out, workspace, stream, false);
}

int wt_resblock(const float* x, const float* wav, const float* e0_w, const float* e0_b, const float* w3, const float* b3,
                const float* w1, const float* b1, const float* ws, const float* bs, float* y, int32_t B, int64_t T,
                int32_t C, int32_t elu_out, int32_t out_s32, int32_t fp32_chain, void* stream) {
    if ((!x && !wav) || !w3 || !b3 || !w1 || !b1 || !ws || !bs || !y) { set_error("wt_resblock: null argument"); return WT_ERR_INVALID; }
    if (wav && (!e0_w || !e0_b)) { set_error("wt_resblock: the folded first conv needs its weights"); return WT_ERR_INVALID; }
    if (B < 1 || T < 1 || (long)B * T >= (long)INT_MAX) { set_error("wt_resblock: bad shape"); return WT_ERR_INVALID; }
    if (fp32_chain && out_s32) { set_error("wt_resblock: the fp32 kernel writes fp32"); return WT_ERR_INVALID; }
    ResblockArgs a{};
    a.x = wav ? nullptr : x; a.wav = wav; a.e0_w = e0_w; a.e0_b = e0_b;
    a.W3 = w3; a.b3 = b3; a.W1 = w1; a.b1 = b1; a.Ws = ws; a.bs = bs;
    a.y = y; a.B = B; a.T = (int)T; a.C = C; a.elu_out = elu_out ? 1 : 0; a.out_s32 = out_s32 ? 1 : 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return fp32_chain ? launch_resblock(a, s) : launch_resblock16(a, s);
}

// The stage-1 kernel of the shipped encode plan: first conv + SEANetResnetBlock + ELU + the stage's down conv in one launch
// (resblock16.hip, DOWN); wd [64][2r][32], y_down [B][ceil(T / r)][64] fp32.
int wt_resblock_down(const float* wav, const float* e0_w, const float* e0_b, const float* w3, const float* b3, const float* w1,
                     const float* b1, const float* ws, const float* bs, const float* wd, const float* bd, float* y_down,
                     int32_t B, int64_t T, int32_t r, void* stream) {
    if (!wav || !e0_w || !e0_b || !w3 || !b3 || !w1 || !b1 || !ws || !bs || !wd || !bd || !y_down) {
        set_error("wt_resblock_down: null argument"); return WT_ERR_INVALID;
    }
    if (B < 1 || T < 1 || (long)B * T >= (long)INT_MAX) { set_error("wt_resblock_down: bad shape"); return WT_ERR_INVALID; }
    if (!resblock16_down_fusable(32, T, r, 2 * r)) {
        set_error("wt_resblock_down: needs stride 2 or 4 and T >= 1024"); return WT_ERR_INVALID;
    }
    ResblockArgs a{};
    a.wav = wav; a.e0_w = e0_w; a.e0_b = e0_b; a.W3 = w3; a.b3 = b3; a.W1 = w1; a.b1 = b1; a.Ws = ws; a.bs = bs;
    a.Wd = wd; a.bd = bd; a.y_down = y_down; a.R = r; a.B = B; a.T = (int)T; a.C = 32;
    return launch_resblock16_down(a, static_cast<hipStream_t>(stream));
}

}  // extern "C"
