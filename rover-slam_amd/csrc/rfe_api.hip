// rfe_api.hip -- the C ABI of librover_fe.so (see include/rover_fe.h for the reference interface
// each entry point replaces).  Host-side orchestration only: weight packing, workspaces, the two
// kernel pipelines (SuperPoint, LightGlue) and the batched stream mode.  No CPU compute path.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include "rfe_internal.h"

using namespace rfe;

static std::string g_init_error;
static std::mutex g_mu;

namespace rfe {

int fail(rfe_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    else { std::lock_guard<std::mutex> l(g_mu); g_init_error = msg; }
    return code;
}

int ensure_ws(rfe_ctx* c, void** p, size_t* cur, size_t need) {
    if (need <= *cur) return RFE_OK;
    if (*p) { RFE_HIP(c, hipStreamSynchronize(c->stream)); RFE_HIP(c, hipFree(*p)); *p = nullptr; *cur = 0; }
    need = (need + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
    hipError_t e = hipMalloc(p, need);
    if (e != hipSuccess) return fail(c, RFE_ERR_OOM, std::string("hipMalloc workspace: ") + hipGetErrorString(e));
    *cur = need;
    return RFE_OK;
}

// Pinned host staging (grow-only) for the host-pointer entries a tracking thread calls once per frame: the caller's arrays are pageable
// (std::vector, cv::Mat), and a hipMemcpyAsync on pageable memory is a synchronous, internally staged copy PER CALL -- four to six of them per
// entry.  Packing the inputs into one pinned block (one DMA in) and fetching the contiguous device results with one DMA out, then scattering
// on the host -- together with the runner writing straight into its output tensors -- measured through the drop-in classes (bench.py
// latency.dropin): one frame 0.914 -> 0.793 ms, one pair 3.28 -> 2.99 ms, one stereo frame 3.43 -> 3.25 ms (profiles/r04_ab_notes.md).
int ensure_pin(rfe_ctx* c, size_t need) {
    if (need <= c->h_pin_bytes) return RFE_OK;
    if (c->h_pin) { RFE_HIP(c, hipStreamSynchronize(c->stream)); RFE_HIP(c, hipHostFree(c->h_pin)); c->h_pin = nullptr; c->h_pin_bytes = 0; }
    need = (need + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
    hipError_t e = hipHostMalloc(&c->h_pin, need, hipHostMallocDefault);
    if (e != hipSuccess) return fail(c, RFE_ERR_OOM, std::string("hipHostMalloc staging: ") + hipGetErrorString(e));
    c->h_pin_bytes = need;
    return RFE_OK;
}

// Host blocks handed out by rfe_host_malloc (pinned, portable): a host entry whose descriptor output lies inside one of them lets the DMA engine write
// the K x 256 floats straight into the caller's memory instead of staging them through h_pin and copying 1 MB on the host afterwards.
static std::mutex g_pin_mu;
static std::vector<std::pair<char*, size_t>> g_pin_blocks;
bool is_lib_pinned(const void* p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (const auto& b : g_pin_blocks)
        if ((const char*)p >= b.first && (const char*)p + bytes <= b.first + b.second) return true;
    return false;
}

// RFE_OPT_HOST_GRAPH: run `enqueue` (the kernel launches of a host entry, on c->stream and -- forked and joined by events -- c->side_stream) as a replayed
// hipGraph.  A key is everything the kernel arguments bake in (shape, thresholds, workspace addresses, settings_gen).  The first HOST_GRAPH_REPEATS - 1
// calls of a key are ordinary launches (workspaces grow, function attributes are set -- neither is legal inside a capture -- and a shape that never
// comes back never pays a capture); the next one runs under hipStreamBeginCapture, is instantiated into one of four LRU slots, and launched; from then on
// one hipGraphLaunch per call.  Alternating shapes (left / right, stereo sizes) keep their graphs; a caller whose keypoint counts differ on every frame
// gets ordinary launches throughout -- the option only helps FIXED-CAPACITY callers (counts saturating Kmax, or padded to it).  Profiling and the test
// tap fall back to ordinary launches, and so does a failed capture or instantiation: the option never changes results, only how the work is submitted.
constexpr int HOST_GRAPH_REPEATS = 3;
template <typename F>
int run_host_graph(rfe_ctx* c, rfe_ctx::HostGraph& g, const std::string& key, F&& enqueue) {
    if (!c->opt_host_graph || c->prof || c->tap.armed) return enqueue();
    ++g.tick;
    for (auto& sl : g.slot)
        if (sl.exec && sl.key == key) { sl.used = g.tick; RFE_HIP(c, hipGraphLaunch(sl.exec, c->stream)); return RFE_OK; }
    rfe_ctx::HostGraph::Seen* sn = nullptr;
    for (auto& q : g.seen) if (q.count > 0 && q.key == key) sn = &q;
    if (!sn) {                                         // a new key takes the least recently used history entry
        sn = &g.seen[0];
        for (auto& q : g.seen) if (q.used < sn->used) sn = &q;
        sn->key = key; sn->count = 0;
    }
    sn->used = g.tick;
    if (++sn->count < HOST_GRAPH_REPEATS) return enqueue();
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); sn->count = 0; return enqueue(); }
    const int rc = enqueue();
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(c->stream, &graph);
    if (rc != RFE_OK || e != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        sn->count = 0;
        return rc != RFE_OK ? rc : enqueue();          // nothing ran during the capture: submit it the ordinary way
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess || !exec) { (void)hipGetLastError(); sn->count = 0; return enqueue(); }
    rfe_ctx::HostGraph::Slot* sl = &g.slot[0];
    for (auto& q : g.slot) { if (!q.exec) { sl = &q; break; } if (q.used < sl->used) sl = &q; }
    if (sl->exec) (void)hipGraphExecDestroy(sl->exec);
    sl->exec = exec; sl->key = key; sl->used = g.tick;
    sn->count = 0; sn->key.clear();                    // the history entry is free again: the slot remembers the key now
    RFE_HIP(c, hipGraphLaunch(sl->exec, c->stream));
    return RFE_OK;
}
static void host_graph_release(rfe_ctx::HostGraph& g) {
    for (auto& sl : g.slot) { if (sl.exec) (void)hipGraphExecDestroy(sl.exec); sl.exec = nullptr; sl.key.clear(); }
    for (auto& q : g.seen) { q.key.clear(); q.count = 0; }
}
static std::string host_graph_key(const rfe_ctx* c, const char* kind, std::initializer_list<long long> v) {
    std::string k = kind;
    for (long long x : v) { k += '|'; k += std::to_string(x); }
    k += "|g" + std::to_string(c->settings_gen) + "|" + std::to_string((unsigned long long)(uintptr_t)c->ws_sp) + "|" + std::to_string((unsigned long long)(uintptr_t)c->ws_lg) +
         "|" + std::to_string((unsigned long long)(uintptr_t)c->ws_io) + "|" + std::to_string((unsigned long long)(uintptr_t)c->sp_hold.get()) + "|" +
         std::to_string((unsigned long long)(uintptr_t)c->lg_hold.get());
    return k;
}

ProfScope::ProfScope(rfe_ctx* ctx, const char* name, hipStream_t on) : c(ctx), idx(-1), st(on ? on : ctx->stream) {
    if (!c->prof) return;
    if (!c->prof_filter.empty() && c->prof_filter != name) return;
    for (size_t i = 0; i < c->stages.size(); ++i) if (c->stages[i].name == name) idx = (int)i;
    if (idx < 0) { c->stages.push_back(Stage{name, 0, 0}); idx = (int)c->stages.size() - 1; }
    auto get = [&]() { hipEvent_t e; if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
    e0 = get(); e1 = get();
    (void)hipEventRecord(e0, st);
}
ProfScope::~ProfScope() {
    if (idx < 0) return;
    (void)hipEventRecord(e1, st);
    c->pending.push_back({idx, {e0, e1}});
}
void prof_collect(rfe_ctx* c) {
    for (auto& p : c->pending) {
        float ms = 0.f;
        (void)hipEventSynchronize(p.second.second);
        (void)hipEventElapsedTime(&ms, p.second.first, p.second.second);
        c->stages[p.first].ms += ms; c->stages[p.first].calls += 1;
        c->ev_pool.push_back(p.second.first); c->ev_pool.push_back(p.second.second);
    }
    c->pending.clear();
}

// bump allocator over a workspace
struct Bump {
    char* base; size_t off = 0;
    explicit Bump(void* b) : base((char*)b) {}
    template <typename T> T* take(size_t n) { T* p = (T*)(base + off); off += (n * sizeof(T) + 255) & ~(size_t)255; return p; }
};
static size_t al(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

}  // namespace rfe

// =====================================================================================
// lifetime
// =====================================================================================
extern "C" const char* rfe_version(void) { return "rover-fe 0.1 (gfx950)"; }

extern "C" const char* rfe_last_error(rfe_ctx* ctx) {
    if (ctx) return ctx->err.c_str();
    return g_init_error.c_str();
}

extern "C" int rfe_init(int device, rfe_ctx** out) {
    if (!out) return fail(nullptr, RFE_ERR_INVALID, "rfe_init: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, RFE_ERR_NO_DEVICE, "rfe_init: no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(nullptr, RFE_ERR_INVALID, "rfe_init: device index out of range");
    if ((e = hipSetDevice(device)) != hipSuccess)
        return fail(nullptr, RFE_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess)
        return fail(nullptr, RFE_ERR_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, RFE_ERR_NO_DEVICE, std::string("rfe_init: kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    rfe_ctx* c = new rfe_ctx();
    c->device = device;
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        delete c;
        return fail(nullptr, RFE_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    c->stream = c->own_stream;
    if ((e = hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming)) != hipSuccess) {
        rfe_destroy(c);
        return fail(nullptr, RFE_ERR_HIP, std::string("hipStreamCreate/hipEventCreate: ") + hipGetErrorString(e));
    }
    *out = c;
    return RFE_OK;
}

extern "C" void rfe_destroy(rfe_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    prof_collect(c);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    host_graph_release(c->g_extract);
    host_graph_release(c->g_match);
    auto fr = [](void* p) { if (p) (void)hipFree(p); };
    c->sp_hold.reset(); c->lg_hold.reset();   // the last ctx holding a device copy frees it
    fr(c->ws_sp); fr(c->ws_lg); fr(c->ws_io); fr(c->ws_tmp); fr(c->ws_st); fr(c->sp_cnt);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); (void)hipStreamDestroy(c->side_stream); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    (void)hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" int rfe_set_option(rfe_ctx* c, int option, int value) {
    if (!c) return RFE_ERR_INVALID;
    switch (option) {
        case RFE_OPT_LG_FOLD_WO: c->opt_lg_fold = value != 0; ++c->settings_gen; return RFE_OK;
        case RFE_OPT_LG_FP16X2: c->opt_lg_fp16x2 = value != 0; ++c->settings_gen; return RFE_OK;
        case RFE_OPT_HOST_GRAPH: c->opt_host_graph = value != 0; return RFE_OK;
        default: return fail(c, RFE_ERR_INVALID, "rfe_set_option: unknown option");
    }
}
extern "C" int rfe_get_option(rfe_ctx* c, int option, int* value) {
    if (!c || !value) return RFE_ERR_INVALID;
    switch (option) {
        case RFE_OPT_LG_FOLD_WO: *value = c->opt_lg_fold ? 1 : 0; return RFE_OK;
        case RFE_OPT_LG_FP16X2: *value = c->opt_lg_fp16x2 ? 1 : 0; return RFE_OK;
        case RFE_OPT_HOST_GRAPH: *value = c->opt_host_graph ? 1 : 0; return RFE_OK;
        default: return fail(c, RFE_ERR_INVALID, "rfe_get_option: unknown option");
    }
}

extern "C" int rfe_set_stream(rfe_ctx* c, void* s) {
    if (!c) return RFE_ERR_INVALID;
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return RFE_OK;
}
extern "C" int rfe_synchronize(rfe_ctx* c) {
    if (!c) return RFE_ERR_INVALID;
    RFE_HIP(c, hipSetDevice(c->device));
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    return RFE_OK;
}
extern "C" int rfe_malloc(rfe_ctx* c, size_t bytes, void** p) {
    if (!c || !p) return RFE_ERR_INVALID;
    RFE_HIP(c, hipSetDevice(c->device));
    hipError_t e = hipMalloc(p, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(c, RFE_ERR_OOM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return RFE_OK;
}
extern "C" int rfe_free(rfe_ctx* c, void* p) {
    if (!c) return RFE_ERR_INVALID;
    RFE_HIP(c, hipSetDevice(c->device));
    RFE_HIP(c, hipFree(p));
    return RFE_OK;
}
extern "C" int rfe_memcpy_h2d(rfe_ctx* c, void* d, const void* s, size_t n) {
    if (!c) return RFE_ERR_INVALID;
    RFE_HIP(c, hipSetDevice(c->device));
    RFE_HIP(c, hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, c->stream));
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    return RFE_OK;
}
extern "C" int rfe_memcpy_d2h(rfe_ctx* c, void* d, const void* s, size_t n) {
    if (!c) return RFE_ERR_INVALID;
    RFE_HIP(c, hipSetDevice(c->device));
    RFE_HIP(c, hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, c->stream));
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    return RFE_OK;
}

// =====================================================================================
// weights
// =====================================================================================
extern "C" int64_t rfe_weight_count(int kind) {
    return kind == RFE_KIND_SUPERPOINT ? SP_COUNT : kind == RFE_KIND_LIGHTGLUE ? LG_COUNT : -1;
}

// Read-only weights are shared: Rover-SLAM keeps 2-3 extractors and 3 matchers per process, each with a private runner
// (src/Tracking.cc:645-651, :70; LocalMapping.cc:45; LoopClosing.cc:46).  Every ctx that loads the same blob on the same
// device points at ONE device copy (looked up by device, kind and a 64-bit FNV-1a hash of the floats, and CONFIRMED by comparing
// the blob with the host copy the entry keeps: two different blobs with one hash get two entries); the copy is freed when the
// last ctx holding it is destroyed or loads something else.
namespace {
struct SpShared {
    rfe::SpWeightsDev w; int device = 0; std::vector<float> host;
    ~SpShared() {
        (void)hipSetDevice(device);
        if (w.conv1a_w) (void)hipFree(w.conv1a_w);
        for (int l = 0; l < 12; ++l) { if (w.packed[l]) (void)hipFree(w.packed[l]); if (w.bias[l]) (void)hipFree(w.bias[l]); }
    }
};
struct LgShared {
    rfe::LgWeightsDev w; int device = 0; std::vector<float> host;
    ~LgShared() { (void)hipSetDevice(device); if (w.blob) (void)hipFree(w.blob); if (w.extra) (void)hipFree(w.extra); if (w.h2) (void)hipFree(w.h2); }
};
std::mutex g_weights_mu;
std::map<std::tuple<int, int, uint64_t, int>, std::weak_ptr<void>> g_weights;   // (device, kind, hash, collision index) -> device copy

uint64_t fnv1a64(const float* p, size_t n) {
    const unsigned char* b = reinterpret_cast<const unsigned char*>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n * sizeof(float); ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
}  // namespace

static int upload(rfe_ctx* c, float** dst, const float* src, size_t n) {
    if (*dst) { RFE_HIP(c, hipFree(*dst)); *dst = nullptr; }
    RFE_HIP(c, hipMalloc((void**)dst, n * sizeof(float)));
    RFE_HIP(c, hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return RFE_OK;
}

static int set_sp_upload(rfe_ctx* c, const float* blob);
static int set_lg_upload(rfe_ctx* c, const float* blob);

static int set_sp(rfe_ctx* c, const float* blob) {
    const uint64_t hash = fnv1a64(blob, (size_t)SP_COUNT) ^ (uint64_t)conv_ck();
    std::lock_guard<std::mutex> lk(g_weights_mu);
    auto key = std::make_tuple(c->device, (int)RFE_KIND_SUPERPOINT, hash, 0);
    for (;; ++std::get<3>(key)) {   // same hash, different contents -> next collision index
        auto it = g_weights.find(key);
        if (it == g_weights.end()) break;
        auto sp = it->second.lock();
        if (!sp) break;             // expired entry: reuse its slot
        if (memcmp(static_cast<SpShared*>(sp.get())->host.data(), blob, (size_t)SP_COUNT * sizeof(float)) != 0) continue;
        c->sp_hold = sp; c->sp = static_cast<SpShared*>(sp.get())->w; c->has_sp = true;
        return RFE_OK;
    }
    c->has_sp = false; c->sp_hold.reset(); c->sp = SpWeightsDev();
    int rc = set_sp_upload(c, blob);
    auto sp = std::make_shared<SpShared>();
    sp->w = c->sp; sp->device = c->device;      // takes ownership of whatever was allocated, also after a partial failure
    if (rc) { c->sp = SpWeightsDev(); return rc; }
    sp->host.assign(blob, blob + SP_COUNT);
    c->sp_hold = sp; g_weights[key] = sp;
    return RFE_OK;
}

static int set_lg(rfe_ctx* c, const float* blob) {
    const uint64_t hash = fnv1a64(blob, (size_t)LG_COUNT);
    std::lock_guard<std::mutex> lk(g_weights_mu);
    auto key = std::make_tuple(c->device, (int)RFE_KIND_LIGHTGLUE, hash, 0);
    for (;; ++std::get<3>(key)) {
        auto it = g_weights.find(key);
        if (it == g_weights.end()) break;
        auto lg = it->second.lock();
        if (!lg) break;
        if (memcmp(static_cast<LgShared*>(lg.get())->host.data(), blob, (size_t)LG_COUNT * sizeof(float)) != 0) continue;
        c->lg_hold = lg; c->lg = static_cast<LgShared*>(lg.get())->w; c->has_lg = true;
        return RFE_OK;
    }
    c->has_lg = false; c->lg_hold.reset(); c->lg = LgWeightsDev();
    int rc = set_lg_upload(c, blob);
    auto lg = std::make_shared<LgShared>();
    lg->w = c->lg; lg->device = c->device;
    if (rc) { c->lg = LgWeightsDev(); return rc; }
    lg->host.assign(blob, blob + LG_COUNT);
    c->lg_hold = lg; g_weights[key] = lg;
    return RFE_OK;
}

static int set_sp_upload(rfe_ctx* c, const float* blob) {
    size_t off = 0;
    for (int l = 0; l < 12; ++l) {
        const SpLayer& L = kSpLayers[l];
        const float* w = blob + off;
        const size_t wn = (size_t)L.cout * L.cin * L.k * L.k;
        const float* b = w + wn;
        off += wn + L.cout;
        int rc;
        if (l == L_1A) {
            std::vector<float> t(9 * 64);
            for (int co = 0; co < 64; ++co) for (int k = 0; k < 9; ++k) t[k * 64 + co] = w[co * 9 + k];
            if ((rc = upload(c, &c->sp.conv1a_w, t.data(), t.size()))) return rc;
        } else if (L.k == 3) {
            std::vector<float> t;
            pack_conv3x3_weights(w, L.cin, L.cout, t);
            if ((rc = upload(c, &c->sp.packed[l], t.data(), t.size()))) return rc;
        } else {
            if ((rc = upload(c, &c->sp.packed[l], w, wn))) return rc;  // [N][K] as-is
        }
        if ((rc = upload(c, &c->sp.bias[l], b, L.cout))) return rc;
    }
    c->has_sp = true;
    return RFE_OK;
}

static int set_lg_upload(rfe_ctx* c, const float* blob) {
    int rc = upload(c, &c->lg.blob, blob, (size_t)LG_COUNT);
    if (rc) return rc;
    float* p = c->lg.blob;
    auto take = [&](size_t n) { float* r = p; p += n; return r; };
    LgWeightsDev& W = c->lg;
    W.wr = take(64);
    for (int l = 0; l < LG_LAYERS; ++l) {
        LgLayerDev& L = W.L[l];
        L.wqkv = take(768 * 256); L.bqkv = take(768); L.wo = take(256 * 256); L.bo = take(256);
        L.w1 = take(512 * 512); L.b1 = take(512); L.lng = take(512); L.lnb = take(512);
        L.w2 = take(256 * 512); L.b2 = take(256);
        L.cwqk = take(256 * 256); L.cbqk = take(256); L.cwv = take(256 * 256); L.cbv = take(256);
        L.cwo = take(256 * 256); L.cbo = take(256);
        L.cw1 = take(512 * 512); L.cb1 = take(512); L.clng = take(512); L.clnb = take(512);
        L.cw2 = take(256 * 512); L.cb2 = take(256);
    }
    W.wp = take(256 * 256); W.bp = take(256); W.wm = take(256); W.bm = take(1);
    if (p - c->lg.blob != LG_COUNT) return fail(c, RFE_ERR_INVALID, "internal: LightGlue blob layout mismatch");
    // derived weights, built once at load time:
    //  * the two cross-attention input projections of every layer packed into one [512][256] Linear;
    //  * the attention output projection (Wo, bo) folded into the message half of the first FFN Linear:
    //    the message m = ctx Wo^T + bo only ever feeds ffn.0, so  W1 [x | m] + b1 = [W1a | W1b Wo] [x | ctx] + (b1 + W1b bo).
    //    The product is formed in double precision on the host; it removes 18 of the 19 256x256 GEMM launches
    //    per forward (mathematically identical, rounding differs at the 1e-7 level; RFE_LG_NO_FOLD=1 keeps them).
    if (W.extra) { RFE_HIP(c, hipFree(W.extra)); W.extra = nullptr; }
    const size_t per = 512 * 256 + 512 + 2 * (512 * 512 + 512);
    RFE_HIP(c, hipMalloc((void**)&W.extra, per * LG_LAYERS * sizeof(float)));
    {
        std::vector<float> w1f(512 * 512), b1f(512);
        std::vector<double> acc(256);
        size_t off = 64;   // host blob walk, same order as above (Wr first)
        for (int l = 0; l < LG_LAYERS; ++l) {
            LgLayerDev& L = W.L[l];
            float* base = W.extra + per * l + 512 * 256 + 512;
            L.w1f = base; L.b1f = base + 512 * 512; L.cw1f = L.b1f + 512; L.cb1f = L.cw1f + 512 * 512;
            const float* h = blob + off;
            const float* s_wo = h + 768 * 256 + 768; const float* s_bo = s_wo + 256 * 256;
            const float* s_w1 = s_bo + 256; const float* s_b1 = s_w1 + 512 * 512;
            const float* cr = s_b1 + 512 + 512 + 512 + 256 * 512 + 256;           // start of the cross block
            const float* c_wo = cr + 2 * (256 * 256 + 256); const float* c_bo = c_wo + 256 * 256;
            const float* c_w1 = c_bo + 256; const float* c_b1 = c_w1 + 512 * 512;
            off += 1250560;   // floats per layer (self 658176 + cross 592384)
            for (int blk = 0; blk < 2; ++blk) {
                const float* wo = blk ? c_wo : s_wo; const float* bo = blk ? c_bo : s_bo;
                const float* w1 = blk ? c_w1 : s_w1; const float* b1 = blk ? c_b1 : s_b1;
                for (int i = 0; i < 512; ++i) {
                    const float* w1row = w1 + (size_t)i * 512;
                    for (int j = 0; j < 256; ++j) { w1f[(size_t)i * 512 + j] = w1row[j]; acc[j] = 0.0; }
                    double bacc = b1[i];
                    for (int k = 0; k < 256; ++k) {
                        const double wv = w1row[256 + k];
                        const float* worow = wo + (size_t)k * 256;
                        for (int j = 0; j < 256; ++j) acc[j] += wv * (double)worow[j];
                        bacc += wv * (double)bo[k];
                    }
                    for (int j = 0; j < 256; ++j) w1f[(size_t)i * 512 + 256 + j] = (float)acc[j];
                    b1f[i] = (float)bacc;
                }
                RFE_HIP(c, hipMemcpy(blk ? L.cw1f : L.w1f, w1f.data(), w1f.size() * 4, hipMemcpyHostToDevice));
                RFE_HIP(c, hipMemcpy(blk ? L.cb1f : L.b1f, b1f.data(), b1f.size() * 4, hipMemcpyHostToDevice));
            }
        }
    }
    for (int l = 0; l < LG_LAYERS; ++l) {
        LgLayerDev& L = W.L[l];
        L.cwqkv = W.extra + per * l; L.cbqkv = L.cwqkv + 512 * 256;
        RFE_HIP(c, hipMemcpy(L.cwqkv, L.cwqk, 256 * 256 * 4, hipMemcpyDeviceToDevice));
        RFE_HIP(c, hipMemcpy(L.cwqkv + 256 * 256, L.cwv, 256 * 256 * 4, hipMemcpyDeviceToDevice));
        RFE_HIP(c, hipMemcpy(L.cbqkv, L.cbqk, 256 * 4, hipMemcpyDeviceToDevice));
        RFE_HIP(c, hipMemcpy(L.cbqkv + 256, L.cbv, 256 * 4, hipMemcpyDeviceToDevice));
    }
    // fp16 (hi, lo) planes of both weight buffers for RFE_OPT_LG_FP16X2 (gemm_h2.hip): split once here, 2 x 2 bytes per float
    if (W.h2) { RFE_HIP(c, hipFree(W.h2)); W.h2 = nullptr; }
    W.n_blob = ((size_t)LG_COUNT + 63) & ~(size_t)63; W.n_extra = per * LG_LAYERS;     // plane starts stay 128-byte aligned (LG_COUNT is odd)
    RFE_HIP(c, hipMalloc((void**)&W.h2, 2 * (W.n_blob + W.n_extra) * sizeof(uint16_t)));
    launch_split_f16(c->stream, W.blob, W.h2, W.h2 + W.n_blob, (size_t)LG_COUNT);
    launch_split_f16(c->stream, W.extra, W.h2 + 2 * W.n_blob, W.h2 + 2 * W.n_blob + W.n_extra, W.n_extra);
    RFE_HIP(c, hipGetLastError());
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    c->has_lg = true;
    return RFE_OK;
}

extern "C" int rfe_set_weights(rfe_ctx* c, int kind, const float* blob, int64_t count) {
    if (!c || !blob) return fail(c, RFE_ERR_INVALID, "rfe_set_weights: null argument");
    RFE_HIP(c, hipSetDevice(c->device));
    if (count != rfe_weight_count(kind)) return fail(c, RFE_ERR_INVALID, "rfe_set_weights: wrong float count for this model kind");
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    const int rc = kind == RFE_KIND_SUPERPOINT ? set_sp(c, blob) : set_lg(c, blob);
    ++c->settings_gen;
    if (rc == RFE_OK) {
        // hyper-parameters belong to a weight set: a bare blob (and a version-1 file) carries none, so this kind's values go back to the published
        // defaults -- a v2 load followed by rfe_set_weights must not keep the earlier file's radius / border / top-k rule silently (rover_fe.h)
        const rfe_hparams d = rfe_default_hparams();
        if (kind == RFE_KIND_SUPERPOINT) {
            c->hp.sp_max_keypoints = d.sp_max_keypoints; c->hp.sp_detection_threshold = d.sp_detection_threshold; c->hp.sp_nms_radius = d.sp_nms_radius;
            c->hp.sp_remove_borders = d.sp_remove_borders; c->hp.sp_topk_always = d.sp_topk_always;
        } else {
            c->hp.lg_layers = d.lg_layers; c->hp.lg_heads = d.lg_heads; c->hp.lg_filter_threshold = d.lg_filter_threshold;
        }
    }
    return rc;
}

extern "C" int rfe_host_malloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return RFE_ERR_INVALID;
    *out = nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return RFE_ERR_OOM; }
    { std::lock_guard<std::mutex> lk(g_pin_mu); g_pin_blocks.push_back({(char*)p, bytes}); }
    *out = p;
    return RFE_OK;
}
extern "C" void rfe_host_free(void* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (size_t i = 0; i < g_pin_blocks.size(); ++i)
            if (g_pin_blocks[i].first == (char*)p) { g_pin_blocks.erase(g_pin_blocks.begin() + i); break; }
    }
    (void)hipHostFree(p);
}

extern "C" int64_t rfe_workspace_bytes(rfe_ctx* c) {
    return c ? (int64_t)(c->ws_sp_bytes + c->ws_lg_bytes + c->ws_io_bytes + c->ws_tmp_bytes + c->ws_st_bytes) : 0;
}

extern "C" uint64_t rfe_weights_id(rfe_ctx* c, int kind) {
    if (!c) return 0;
    const void* p = kind == RFE_KIND_SUPERPOINT ? c->sp_hold.get() : kind == RFE_KIND_LIGHTGLUE ? c->lg_hold.get() : nullptr;
    return (uint64_t)(uintptr_t)p;
}

// RFEW container (rover-slam_amd/weights.py).  Both versions: "RFEW" | u32 version | u32 kind | u64 float count.
//   version 1: the floats follow.
//   version 2: u32 hp_bytes | hp_bytes of graph hyper-parameters | the floats.  Hyper-parameter block (little endian):
//     kind 1 (SuperPoint): i32 max_keypoints, f32 detection_threshold, i32 nms_radius, i32 remove_borders, i32 topk_always   (20 bytes)
//     kind 2 (LightGlue):  i32 layers, i32 heads, f32 filter_threshold                                     (12 bytes)
//   A longer block (a later writer) is accepted, its known prefix used.
static int check_hparams(rfe_ctx* c, const rfe_hparams& h, const char* who) {
    if (h.sp_max_keypoints < 1 || h.sp_max_keypoints > 4096) return fail(c, RFE_ERR_INVALID, std::string(who) + ": sp_max_keypoints must be in 1..4096");
    if (!(h.sp_detection_threshold >= 0.f) || !(h.sp_detection_threshold < 1.f)) return fail(c, RFE_ERR_INVALID, std::string(who) + ": sp_detection_threshold must be in [0, 1)");
    if (h.sp_nms_radius < 1 || h.sp_nms_radius > NMS_MAX_RADIUS) return fail(c, RFE_ERR_INVALID, std::string(who) + ": sp_nms_radius must be in 1.." + std::to_string(NMS_MAX_RADIUS));
    if (h.sp_remove_borders < 0 || h.sp_remove_borders > 64) return fail(c, RFE_ERR_INVALID, std::string(who) + ": sp_remove_borders must be in 0..64");
    if (h.sp_topk_always != 0 && h.sp_topk_always != 1) return fail(c, RFE_ERR_INVALID, std::string(who) + ": sp_topk_always must be 0 or 1");
    if (h.lg_layers != LG_LAYERS || h.lg_heads != 4)
        return fail(c, RFE_ERR_INVALID, std::string(who) + ": the LightGlue kernels are built for 9 layers of 4 heads x 64, the file / caller says " +
                                        std::to_string(h.lg_layers) + " layers of " + std::to_string(h.lg_heads) + " heads");
    if (!(h.lg_filter_threshold >= 0.f) || !(h.lg_filter_threshold < 1.f)) return fail(c, RFE_ERR_INVALID, std::string(who) + ": lg_filter_threshold must be in [0, 1)");
    return RFE_OK;
}

static int load_rfew(rfe_ctx* c, const char* path, int want_kind) {
    FILE* f = fopen(path, "rb");
    if (!f) return fail(c, RFE_ERR_IO, std::string("cannot open weight file ") + path);
    unsigned char head[20];
    if (fread(head, 1, 20, f) != 20 || memcmp(head, "RFEW", 4) != 0) { fclose(f); return fail(c, RFE_ERR_IO, std::string("not an RFEW file: ") + path); }
    uint32_t ver, kind; uint64_t cnt;
    memcpy(&ver, head + 4, 4); memcpy(&kind, head + 8, 4); memcpy(&cnt, head + 12, 8);
    if ((ver != 1 && ver != 2) || (int)kind != want_kind || (int64_t)cnt != rfe_weight_count(want_kind)) { fclose(f); return fail(c, RFE_ERR_IO, std::string("RFEW header mismatch in ") + path); }
    rfe_hparams hp = c->hp;
    if (ver == 2) {
        uint32_t hb = 0;
        unsigned char blk[256];
        const uint32_t need = want_kind == RFE_KIND_SUPERPOINT ? 20u : 12u;
        if (fread(&hb, 4, 1, f) != 1 || hb < need || hb > sizeof(blk) || fread(blk, 1, hb, f) != hb) { fclose(f); return fail(c, RFE_ERR_IO, std::string("RFEW v2 hyper-parameter block damaged in ") + path); }
        if (want_kind == RFE_KIND_SUPERPOINT) {
            memcpy(&hp.sp_max_keypoints, blk, 4); memcpy(&hp.sp_detection_threshold, blk + 4, 4);
            memcpy(&hp.sp_nms_radius, blk + 8, 4); memcpy(&hp.sp_remove_borders, blk + 12, 4); memcpy(&hp.sp_topk_always, blk + 16, 4);
        } else {
            memcpy(&hp.lg_layers, blk, 4); memcpy(&hp.lg_heads, blk + 4, 4); memcpy(&hp.lg_filter_threshold, blk + 8, 4);
        }
        const int rc = check_hparams(c, hp, path);
        if (rc) { fclose(f); c->err = "RFEW v2 hyper-parameters refused: " + c->err; return RFE_ERR_IO; }
    }
    std::vector<float> blob(cnt);
    size_t got = fread(blob.data(), sizeof(float), cnt, f);
    fclose(f);
    if (got != cnt) return fail(c, RFE_ERR_IO, std::string("short read on ") + path);
    const int rc = rfe_set_weights(c, want_kind, blob.data(), (int64_t)cnt);      // resets this kind's hyper-parameters to the defaults
    if (rc == RFE_OK && ver == 2) {                                               // the file's hyper-parameters travel with its weights
        if (want_kind == RFE_KIND_SUPERPOINT) {
            c->hp.sp_max_keypoints = hp.sp_max_keypoints; c->hp.sp_detection_threshold = hp.sp_detection_threshold; c->hp.sp_nms_radius = hp.sp_nms_radius;
            c->hp.sp_remove_borders = hp.sp_remove_borders; c->hp.sp_topk_always = hp.sp_topk_always;
        } else {
            c->hp.lg_layers = hp.lg_layers; c->hp.lg_heads = hp.lg_heads; c->hp.lg_filter_threshold = hp.lg_filter_threshold;
        }
    }
    return rc;
}

extern "C" int rfe_get_hparams(rfe_ctx* c, rfe_hparams* out) {
    if (!c || !out) return RFE_ERR_INVALID;
    *out = c->hp;
    return RFE_OK;
}
extern "C" int rfe_set_hparams(rfe_ctx* c, const rfe_hparams* in) {
    if (!c || !in) return RFE_ERR_INVALID;
    const int rc = check_hparams(c, *in, "rfe_set_hparams");
    if (rc) return rc;
    c->hp = *in;
    ++c->settings_gen;
    return RFE_OK;
}

// An ONNX graph file (the reference's own onnxmodel/superpoint.onnx / lightglue_sim.onnx, src/Extractors/SPextractor.cc:92-94,
// src/Matchers/lightglue_onnx.cpp:38): initializers -> canonical blob, graph constants -> hyper-parameters (onnx_load.hip), then exactly what an RFEW
// v2 file does.  A graph whose hyper-parameters cannot be read is refused with the reason (RFE_ERR_IO).
static int load_onnx(rfe_ctx* c, const char* path, int want_kind) {
    std::vector<float> blob;
    rfe_hparams hp = rfe_default_hparams();
    std::string err;
    if (!rfe::onnx_convert(path, want_kind, blob, &hp, err)) return fail(c, RFE_ERR_IO, err);
    int rc = check_hparams(c, hp, path);
    if (rc) { c->err = "graph hyper-parameters refused: " + c->err; return RFE_ERR_IO; }
    if ((int64_t)blob.size() != rfe_weight_count(want_kind)) return fail(c, RFE_ERR_IO, std::string("converted weight count mismatch for ") + path);
    rc = rfe_set_weights(c, want_kind, blob.data(), (int64_t)blob.size());   // resets this kind's hyper-parameters to the defaults
    if (rc == RFE_OK) {
        if (want_kind == RFE_KIND_SUPERPOINT) {
            c->hp.sp_max_keypoints = hp.sp_max_keypoints; c->hp.sp_detection_threshold = hp.sp_detection_threshold; c->hp.sp_nms_radius = hp.sp_nms_radius;
            c->hp.sp_remove_borders = hp.sp_remove_borders; c->hp.sp_topk_always = hp.sp_topk_always;
        } else {
            c->hp.lg_layers = hp.lg_layers; c->hp.lg_heads = hp.lg_heads; c->hp.lg_filter_threshold = hp.lg_filter_threshold;
        }
    }
    return rc;
}

// RFEW container or ONNX graph, told apart by the file's first four bytes
static int load_any(rfe_ctx* c, const char* path, int want_kind) {
    FILE* f = fopen(path, "rb");
    if (!f) return fail(c, RFE_ERR_IO, std::string("cannot open weight file ") + path);
    unsigned char magic[4] = {0, 0, 0, 0};
    const size_t got = fread(magic, 1, 4, f);
    fclose(f);
    if (got == 4 && memcmp(magic, "RFEW", 4) == 0) return load_rfew(c, path, want_kind);
    return load_onnx(c, path, want_kind);
}

extern "C" int rfe_load_weights(rfe_ctx* c, const char* sp_path, const char* lg_path) {
    if (!c) return RFE_ERR_INVALID;
    int rc;
    if (sp_path && (rc = load_any(c, sp_path, RFE_KIND_SUPERPOINT))) return rc;
    if (lg_path && (rc = load_any(c, lg_path, RFE_KIND_LIGHTGLUE))) return rc;
    return RFE_OK;
}

extern "C" int rfe_load_onnx(rfe_ctx* c, const char* sp_path, const char* lg_path) {
    if (!c) return RFE_ERR_INVALID;
    int rc;
    if (sp_path && (rc = load_onnx(c, sp_path, RFE_KIND_SUPERPOINT))) return rc;
    if (lg_path && (rc = load_onnx(c, lg_path, RFE_KIND_LIGHTGLUE))) return rc;
    return RFE_OK;
}

// =====================================================================================
// SuperPoint pipeline
// =====================================================================================
namespace {

struct SpBuffers {
    float *a1, *p1, *a2, *p2, *a3, *p3, *a4, *f4, *pa, *logits, *da, *dmap, *smap, *nmap, *ss;
    uint8_t *mask, *supp;
    float* cand_score; int32_t* cand_idx;
    unsigned long long* sel_keys; int32_t* sel_n;     // selected (score, pixel) keys between select_kernel and select_rank_kernel
    bool tail_fused = false;                          // sp_tail_lat_kernel ran: candidates are 64-bit keys at cand_score (cand_score | cand_idx = 8 B per pixel)
};

// conv1a as its own launch (78.6 MB/frame activation in HBM) instead of recomputed inside conv1b: A/B and test switch
bool sp_unfused_conv1() { static const bool u = tune_env("RFE_UNFUSED_CONV1") != nullptr; return u; }

size_t sp_ws_bytes(int B, int H, int W) {
    const size_t hw = (size_t)B * H * W, cells = hw / 64;
    size_t t = 0;
    if (sp_unfused_conv1()) t += al(hw * 64 * 4);   // a1: never materialised on the default (fused) path
    t += al(hw / 4 * 64 * 4) * 2;  // p1, a2
    t += al(hw / 16 * 64 * 4);     // p2
    t += al(hw / 16 * 128 * 4);    // a3
    t += al(cells * 128 * 4) * 3;  // p3, a4, f4
    t += al(cells * 256 * 4) * 3;  // pa, da, dmap
    t += al(cells * 65 * 4);       // logits
    t += al(hw * 4) * 3;           // smap, nmap, ss
    t += al(hw) * 2;               // mask, supp
    t += al(hw * 4) * 2;           // cand
    t += al((size_t)B * 4096 * 8) + al((size_t)B * 4);   // sel_keys (Kmax <= 4096), sel_n
    return t + 4096;
}

void sp_carve(void* ws, int B, int H, int W, SpBuffers& b) {
    const size_t hw = (size_t)B * H * W, cells = hw / 64;
    Bump a(ws);
    b.a1 = sp_unfused_conv1() ? a.take<float>(hw * 64) : nullptr; b.p1 = a.take<float>(hw / 4 * 64); b.a2 = a.take<float>(hw / 4 * 64);
    b.p2 = a.take<float>(hw / 16 * 64); b.a3 = a.take<float>(hw / 16 * 128);
    b.p3 = a.take<float>(cells * 128); b.a4 = a.take<float>(cells * 128); b.f4 = a.take<float>(cells * 128);
    b.pa = a.take<float>(cells * 256); b.da = a.take<float>(cells * 256); b.dmap = a.take<float>(cells * 256);
    b.logits = a.take<float>(cells * 65);
    b.smap = a.take<float>(hw); b.nmap = a.take<float>(hw); b.ss = a.take<float>(hw);
    b.mask = a.take<uint8_t>(hw); b.supp = a.take<uint8_t>(hw);
    b.cand_score = a.take<float>(hw); b.cand_idx = a.take<int32_t>(hw);
    b.sel_keys = a.take<unsigned long long>((size_t)B * 4096); b.sel_n = a.take<int32_t>(B);
}

GemmArgs gemm_plain(const float* A, int lda, const float* Bw, int ldb, const float* bias, float* C, int ldc, int M, int N, int K) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.B = Bw; g.ldb = ldb; g.bias = bias; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.alpha = 1.0f; g.batch = 1;
    return g;
}

// LightGlue Linears: tolerance-checked, free to use the k-permuted GEMM path (GemmArgs::kperm)
GemmArgs gemm_lg(const float* A, int lda, const float* Bw, int ldb, const float* bias, float* C, int ldc, int M, int N, int K) {
    GemmArgs g = gemm_plain(A, lda, Bw, ldb, bias, C, ldc, M, N, K);
    g.kperm = 1;
    return g;
}
// A LightGlue Linear whose weight matrix Bw lives in the ctx's weight buffers: with RFE_OPT_LG_FP16X2 on, the fp16 (hi, lo) planes of
// the same matrix ride along and launch_gemm_nt takes the split GEMM for the throughput shapes (gemm_h2.hip)
GemmArgs gemm_lgw(const rfe_ctx* c, const float* A, int lda, const float* Bw, int ldb, const float* bias, float* C, int ldc, int M, int N, int K) {
    GemmArgs g = gemm_lg(A, lda, Bw, ldb, bias, C, ldc, M, N, K);
    const LgWeightsDev& W = c->lg;
    if (c->opt_lg_fp16x2 && W.h2) {
        if (Bw >= W.blob && Bw < W.blob + LG_COUNT) { g.Bh = W.h2 + (Bw - W.blob); g.Bl = g.Bh + W.n_blob; }
        else if (Bw >= W.extra && Bw < W.extra + W.n_extra) { g.Bh = W.h2 + 2 * W.n_blob + (Bw - W.extra); g.Bl = g.Bh + W.n_extra; }
    }
    return g;
}

int sp_check(rfe_ctx* c, int H, int W, int B, int Kmax) {
    if (!c) return RFE_ERR_INVALID;
    if (!c->has_sp) return fail(c, RFE_ERR_NO_WEIGHTS, "SuperPoint weights not loaded (rfe_load_weights / rfe_set_weights)");
    if (H < 8 || W < 8 || B <= 0) return fail(c, RFE_ERR_INVALID, "extract: H and W must be at least 8, B > 0");
    if (Kmax <= 0 || Kmax > 4096) return fail(c, RFE_ERR_INVALID, "extract: Kmax must be in 1..4096");
    return RFE_OK;
}

// backbone + heads up to the NMS'ed score map and the normalised descriptor map
// join = false: the caller still has detector-only work to enqueue and joins the descriptor stream itself
// (hipStreamWaitEvent(c->stream, c->ev_join)) when `forked` comes back true
int sp_forward_maps(rfe_ctx* c, const void* img, int H, int W, int stride, int B, SpBuffers& b, bool join, bool& forked, bool img_f32 = false,
                    long long frame_step = 0 /*pixels from frame b to b + 1; 0 = stride * H*/, float thr = 0.0005f /*candidate threshold of the fused tail*/,
                    bool want_maps = false /*test hook: the fused tail also writes the score map and the NMS'ed map*/) {
    forked = false;
    int rc = ensure_ws(c, &c->ws_sp, &c->ws_sp_bytes, sp_ws_bytes(B, H, W));
    if (rc) return rc;
    sp_carve(c->ws_sp, B, H, W, b);
    hipStream_t s = c->stream;
    const SpWeightsDev& w = c->sp;
    // Any H, W >= 8 (the ONNX graph has dynamic axes): every 2x2/2 max-pool floors, so the levels are H1 = H/2, H2 = H1/2,
    // Hc = H2/2 and the score map / NMS / selection live on the 8Hc x 8Wc frame (= the image when H, W are multiples of 8;
    // KITTI 1241 x 376 -> 155 x 47 cells, score map 1240 x 376).  The workspace carve (sized from B*H*W) is an upper bound.
    const int H1 = H / 2, W1 = W / 2, H2 = H1 / 2, W2 = W1 / 2, Hc = H2 / 2, Wc = W2 / 2, cells = B * Hc * Wc;
    if (sp_unfused_conv1()) {
        { ProfScope p(c, "conv1a"); launch_conv1a_u8(s, img, img_f32, stride, B, H, W, w.conv1a_w, w.bias[L_1A], b.a1); }
        { ProfScope p(c, "conv1b"); launch_conv3x3(s, b.a1, B, H, W, 64, w.packed[L_1B], w.bias[L_1B], 64, true, true, b.p1, L_1B); }
    } else {   // conv1a recomputed inside conv1b's LDS staging: the [B,H,W,64] activation never touches HBM
        ProfScope p(c, "conv1ab");
        launch_conv1ab_fused(s, img, img_f32, stride, B, H, W, w.conv1a_w, w.bias[L_1A], w.packed[L_1B], w.bias[L_1B], b.p1, frame_step);
    }
    { ProfScope p(c, "conv2a"); launch_conv3x3(s, b.p1, B, H1, W1, 64, w.packed[L_2A], w.bias[L_2A], 64, true, false, b.a2, L_2A); }
    { ProfScope p(c, "conv2b"); launch_conv3x3(s, b.a2, B, H1, W1, 64, w.packed[L_2B], w.bias[L_2B], 64, true, true, b.p2, L_2B); }
    { ProfScope p(c, "conv3a"); launch_conv3x3(s, b.p2, B, H2, W2, 64, w.packed[L_3A], w.bias[L_3A], 128, true, false, b.a3, L_3A); }
    { ProfScope p(c, "conv3b"); launch_conv3x3(s, b.a3, B, H2, W2, 128, w.packed[L_3B], w.bias[L_3B], 128, true, true, b.p3, L_3B); }
    { ProfScope p(c, "conv4a"); launch_conv3x3(s, b.p3, B, Hc, Wc, 128, w.packed[L_4A], w.bias[L_4A], 128, true, false, b.a4, L_4A); }
    { ProfScope p(c, "conv4b"); launch_conv3x3(s, b.a4, B, Hc, Wc, 128, w.packed[L_4B], w.bias[L_4B], 128, true, false, b.f4, L_4B); }
    // The two heads only share their input f4.  The descriptor head (convDa, convDb, L2 norm: MFMA work) runs on the
    // side stream while the detector head continues on the main one with its tail of small bandwidth / latency-bound
    // kernels (convPb, softmax, 5 NMS passes, selection), which would otherwise leave most of the chip idle.
    // With events around every stage (full profiling pass) the heads stay serial so that the stage times are clean.
    static const bool fork_env = tune_env("RFE_SP_NO_FORK") == nullptr;
    const bool fork = fork_env && !(c->prof && c->prof_filter.empty());
    hipStream_t sd = fork ? c->side_stream : s;
    if (fork) { RFE_HIP(c, hipEventRecord(c->ev_fork, s)); RFE_HIP(c, hipStreamWaitEvent(sd, c->ev_fork, 0)); }
    { ProfScope p(c, "convDa", sd); launch_conv3x3(sd, b.f4, B, Hc, Wc, 128, w.packed[L_DA], w.bias[L_DA], 256, true, false, b.da, L_DA); }
    { ProfScope p(c, "convDb", sd); launch_gemm_nt(sd, gemm_plain(b.da, 256, w.packed[L_DB], 256, w.bias[L_DB], b.dmap, 256, cells, 256, 256)); }
    { ProfScope p(c, "sp_post", sd); launch_descmap_norm(sd, b.dmap, cells); }
    if (fork) RFE_HIP(c, hipEventRecord(c->ev_join, sd));
    static const bool join_early = tune_env("RFE_SP_JOIN_EARLY") != nullptr;   // tuning build: two streams, but the heads one after the other (diagnostic)
    if (fork && join_early) RFE_HIP(c, hipStreamWaitEvent(s, c->ev_join, 0));
    { ProfScope p(c, "convPa"); launch_conv3x3(s, b.f4, B, Hc, Wc, 128, w.packed[L_PA], w.bias[L_PA], 256, true, false, b.pa, L_PA); }
    { ProfScope p(c, "convPb"); launch_gemm_nt(s, gemm_plain(b.pa, 256, w.packed[L_PB], 256, w.bias[L_PB], b.logits, 65, cells, 65, 256)); }
    { ProfScope p(c, "sp_post");
      // one to four frames, published radius: softmax + NMS + candidate compaction in ONE launch (sp_post.hip: sp_tail_lat_kernel); otherwise the separate launches
      if (B <= 4 && c->hp.sp_nms_radius == 4) {
          if (!c->sp_cnt) { RFE_HIP(c, hipMalloc((void**)&c->sp_cnt, 8 * sizeof(int32_t))); c->sp_cnt_dirty = true; }
          if (c->sp_cnt_dirty) { RFE_HIP(c, hipMemsetAsync(c->sp_cnt, 0, 8 * sizeof(int32_t), s)); c->sp_cnt_dirty = false; }
          b.tail_fused = launch_sp_tail_lat(s, b.logits, B, Hc, Wc, c->hp.sp_nms_radius, c->hp.sp_remove_borders, thr, (unsigned long long*)b.cand_score, c->sp_cnt,
                                            want_maps ? b.smap : nullptr, want_maps ? b.nmap : nullptr);
          if (b.tail_fused) c->sp_cnt_dirty = true;      // until the ranking kernel (which zeroes the counters) is enqueued behind it
      }
      if (!b.tail_fused) {
          launch_softmax65_d2s(s, b.logits, 65, B, Hc, Wc, b.smap);
          launch_nms(s, b.smap, B, 8 * Hc, 8 * Wc, c->hp.sp_nms_radius, c->hp.sp_remove_borders, b.ss, b.mask, b.supp, b.nmap);
      } }
    if (fork && join) RFE_HIP(c, hipStreamWaitEvent(s, c->ev_join, 0));
    forked = fork && !join;
    RFE_HIP(c, hipGetLastError());
    return RFE_OK;
}

int sp_forward(rfe_ctx* c, const void* img, int H, int W, int stride, int B, int Kmax, float thr,
               int32_t* n, int32_t* kxy, float* score, float* desc, uint8_t* desc_bin = nullptr, bool img_f32 = false, long long frame_step = 0) {
    SpBuffers b;
    bool forked;
    int rc = sp_forward_maps(c, img, H, W, stride, B, b, false, forked, img_f32, frame_step, thr);
    if (rc) return rc;
    const int Hc = H / 2 / 2 / 2, Wc = W / 2 / 2 / 2, Hs = 8 * Hc, Ws = 8 * Wc;   // score-map frame, see sp_forward_maps
    { ProfScope p(c, "sp_select");
      if (b.tail_fused) {
          launch_select_keys(c->stream, (const unsigned long long*)b.cand_score, c->sp_cnt, B, Hs, Ws, Kmax, c->hp.sp_topk_always != 0, n, kxy, score);
          c->sp_cnt_dirty = false;
      } else
      launch_select(c->stream, b.nmap, B, Hs, Ws, Kmax, thr, b.cand_score, b.cand_idx, n, kxy, score, (int32_t*)b.ss /*NMS scratch, free by now*/, c->hp.sp_topk_always != 0, b.sel_keys, b.sel_n);
      if (forked) RFE_HIP(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));   // descriptor map ready
      launch_desc_sample(c->stream, b.dmap, B, Hc, Wc, Hs, Ws, n, kxy, Kmax, desc, desc_bin); }
    RFE_HIP(c, hipGetLastError());
    return RFE_OK;
}

}  // namespace

extern "C" int rfe_extract_u8_bin_dev(rfe_ctx* c, const uint8_t* img, int H, int W, int stride, int B, int Kmax,
                                      float thr, int32_t* n, int32_t* kxy, float* score, float* desc, uint8_t* desc_bin) {
    int rc = sp_check(c, H, W, B, Kmax);
    if (rc) return rc;
    if (!img || !n || !kxy || !score || !desc || stride < W) return fail(c, RFE_ERR_INVALID, "extract: null pointer or stride < W");
    RFE_HIP(c, hipSetDevice(c->device));
    return sp_forward(c, img, H, W, stride, B, Kmax, thr, n, kxy, score, desc, desc_bin);
}

extern "C" int rfe_extract_u8_dev(rfe_ctx* c, const uint8_t* img, int H, int W, int stride, int B, int Kmax,
                                  float thr, int32_t* n, int32_t* kxy, float* score, float* desc) {
    return rfe_extract_u8_bin_dev(c, img, H, W, stride, B, Kmax, thr, n, kxy, score, desc, nullptr);
}

extern "C" int rfe_extract_u8(rfe_ctx* c, const uint8_t* img, int H, int W, int stride, int B, int Kmax, float thr,
                              int32_t* n, int32_t* kxy, float* score, float* desc) {
    return rfe_extract_u8_bin(c, img, H, W, stride, B, Kmax, thr, n, kxy, score, desc, nullptr);
}

extern "C" int rfe_extract_u8_bin(rfe_ctx* c, const uint8_t* img, int H, int W, int stride, int B, int Kmax, float thr,
                                  int32_t* n, int32_t* kxy, float* score, float* desc, uint8_t* desc_bin) {
    int rc = sp_check(c, H, W, B, Kmax);
    if (rc) return rc;
    if (!img || !n || !kxy || !score || !desc || stride < W) return fail(c, RFE_ERR_INVALID, "extract: null pointer or stride < W");
    RFE_HIP(c, hipSetDevice(c->device));
    // the device copy is tight (pitch W): a cv::Mat ROI has stride > W and its last row ends W bytes into the row, so
    // only W bytes of every row are read (B*H*stride bytes from the first pixel would run past a ROI at the bottom of
    // its parent buffer)
    const size_t ib = al((size_t)B * H * W), nb = al((size_t)B * 4), kb = al((size_t)B * Kmax * 8),
                 sb = al((size_t)B * Kmax * 4), db = al((size_t)B * Kmax * 1024), bb = desc_bin ? al((size_t)B * Kmax * 256) : 0;
    if ((rc = ensure_ws(c, &c->ws_io, &c->ws_io_bytes, ib + nb + kb + sb + db + bb))) return rc;
    if ((rc = ensure_pin(c, ib + nb + kb + sb + db + bb))) return rc;
    char* p = (char*)c->ws_io;
    char* hp = (char*)c->h_pin;
    uint8_t* d_img = (uint8_t*)p; int32_t* d_n = (int32_t*)(p + ib); int32_t* d_k = (int32_t*)(p + ib + nb);
    float* d_s = (float*)(p + ib + nb + kb); float* d_d = (float*)(p + ib + nb + kb + sb);
    uint8_t* d_b = desc_bin ? (uint8_t*)(p + ib + nb + kb + sb + db) : nullptr;
    for (size_t r = 0; r < (size_t)B * H; ++r) memcpy(hp + r * W, img + r * stride, (size_t)W);
    RFE_HIP(c, hipMemcpyAsync(d_img, hp, (size_t)B * H * W, hipMemcpyHostToDevice, c->stream));
    if ((rc = ensure_ws(c, &c->ws_sp, &c->ws_sp_bytes, sp_ws_bytes(B, H, W)))) return rc;   // before the key is formed: a capture must not allocate
    int thr_bits; memcpy(&thr_bits, &thr, 4);
    if ((rc = run_host_graph(c, c->g_extract, host_graph_key(c, "xu8", {H, W, B, Kmax, thr_bits, desc_bin != nullptr}),
                             [&] { return sp_forward(c, d_img, H, W, W, B, Kmax, thr, d_n, d_k, d_s, d_d, d_b); }))) return rc;
    // descriptors in rfe_host_malloc'ed memory (the class shims' tensors are): the DMA engine writes them where the caller wants them
    const bool direct = is_lib_pinned(desc, (size_t)B * Kmax * 1024);
    if (direct) {
        RFE_HIP(c, hipMemcpyAsync(hp + ib, p + ib, nb + kb + sb, hipMemcpyDeviceToHost, c->stream));           // [n | kxy | score]
        RFE_HIP(c, hipMemcpyAsync(desc, d_d, (size_t)B * Kmax * 1024, hipMemcpyDeviceToHost, c->stream));
        if (desc_bin) RFE_HIP(c, hipMemcpyAsync(hp + ib + nb + kb + sb + db, d_b, (size_t)B * Kmax * 256, hipMemcpyDeviceToHost, c->stream));
    } else {
        RFE_HIP(c, hipMemcpyAsync(hp + ib, p + ib, nb + kb + sb + db + bb, hipMemcpyDeviceToHost, c->stream));   // [n | kxy | score | desc | bin] in one DMA
    }
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(n, hp + ib, (size_t)B * 4);
    memcpy(kxy, hp + ib + nb, (size_t)B * Kmax * 8);
    memcpy(score, hp + ib + nb + kb, (size_t)B * Kmax * 4);
    if (!direct) memcpy(desc, hp + ib + nb + kb + sb, (size_t)B * Kmax * 1024);
    if (desc_bin) memcpy(desc_bin, hp + ib + nb + kb + sb + db, (size_t)B * Kmax * 256);
    prof_collect(c);
    return RFE_OK;
}

// The reference's float entry (Extractor_Inference on an already normalised CV_32F image, src/Extractors/superpoint_onnx.cc:88-118): the
// pixel values go into conv1a as they are, whatever their range -- no u8 round trip, no NormalizeImage.  stride in floats.
extern "C" int rfe_extract_f32_dev(rfe_ctx* c, const float* img, int H, int W, int stride, int B, int Kmax,
                                   float thr, int32_t* n, int32_t* kxy, float* score, float* desc) {
    int rc = sp_check(c, H, W, B, Kmax);
    if (rc) return rc;
    if (!img || !n || !kxy || !score || !desc || stride < W) return fail(c, RFE_ERR_INVALID, "extract: null pointer or stride < W");
    RFE_HIP(c, hipSetDevice(c->device));
    return sp_forward(c, img, H, W, stride, B, Kmax, thr, n, kxy, score, desc, nullptr, true);
}

extern "C" int rfe_extract_f32(rfe_ctx* c, const float* img, int H, int W, int stride, int B, int Kmax, float thr,
                               int32_t* n, int32_t* kxy, float* score, float* desc) {
    int rc = sp_check(c, H, W, B, Kmax);
    if (rc) return rc;
    if (!img || !n || !kxy || !score || !desc || stride < W) return fail(c, RFE_ERR_INVALID, "extract: null pointer or stride < W");
    RFE_HIP(c, hipSetDevice(c->device));
    const size_t ib = al((size_t)B * H * W * 4), nb = al((size_t)B * 4), kb = al((size_t)B * Kmax * 8),
                 sb = al((size_t)B * Kmax * 4), db = al((size_t)B * Kmax * 1024);
    if ((rc = ensure_ws(c, &c->ws_io, &c->ws_io_bytes, ib + nb + kb + sb + db))) return rc;
    if ((rc = ensure_pin(c, ib + nb + kb + sb + db))) return rc;
    char* p = (char*)c->ws_io;
    char* hp = (char*)c->h_pin;
    float* d_img = (float*)p; int32_t* d_n = (int32_t*)(p + ib); int32_t* d_k = (int32_t*)(p + ib + nb);
    float* d_s = (float*)(p + ib + nb + kb); float* d_d = (float*)(p + ib + nb + kb + sb);
    for (size_t r = 0; r < (size_t)B * H; ++r) memcpy(hp + r * W * 4, img + r * stride, (size_t)W * 4);
    RFE_HIP(c, hipMemcpyAsync(d_img, hp, (size_t)B * H * W * 4, hipMemcpyHostToDevice, c->stream));
    if ((rc = ensure_ws(c, &c->ws_sp, &c->ws_sp_bytes, sp_ws_bytes(B, H, W)))) return rc;
    int thr_bits; memcpy(&thr_bits, &thr, 4);
    if ((rc = run_host_graph(c, c->g_extract, host_graph_key(c, "xf32", {H, W, B, Kmax, thr_bits}),
                             [&] { return sp_forward(c, d_img, H, W, W, B, Kmax, thr, d_n, d_k, d_s, d_d, nullptr, true); }))) return rc;
    const bool direct = is_lib_pinned(desc, (size_t)B * Kmax * 1024);
    if (direct) {
        RFE_HIP(c, hipMemcpyAsync(hp + ib, p + ib, nb + kb + sb, hipMemcpyDeviceToHost, c->stream));
        RFE_HIP(c, hipMemcpyAsync(desc, d_d, (size_t)B * Kmax * 1024, hipMemcpyDeviceToHost, c->stream));
    } else {
        RFE_HIP(c, hipMemcpyAsync(hp + ib, p + ib, nb + kb + sb + db, hipMemcpyDeviceToHost, c->stream));
    }
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(n, hp + ib, (size_t)B * 4);
    memcpy(kxy, hp + ib + nb, (size_t)B * Kmax * 8);
    memcpy(score, hp + ib + nb + kb, (size_t)B * Kmax * 4);
    if (!direct) memcpy(desc, hp + ib + nb + kb + sb, (size_t)B * Kmax * 1024);
    prof_collect(c);
    return RFE_OK;
}

// =====================================================================================
// LightGlue pipeline.  Token layout: side-major sequences, seq = side*P + pair, each padded to L.
// =====================================================================================
namespace {

struct LgBuffers {
    float *x, *kn, *csn, *lnstat, *qkv, *ctx, *msg, *h, *md, *z, *sim, *rowlse, *collse, *mx0, *apart;
    int32_t *a0, *a1, *lens, *kvmap;
    char* extra;   // caller-sized scratch region carved after the fixed buffers
};

size_t lg_ws_bytes(int P, int L, size_t extra_bytes = 0) {
    const size_t rows = (size_t)2 * P * L;
    size_t t = 0;
    t += al(rows * 256 * 4) * 4;  // x ctx msg md
    t += al(rows * 2 * 4) + al(rows * 64 * 4) + al(rows * 32 * 4) + al(rows * 768 * 4) + al(rows * 512 * 4) + al(rows * 4);
    t += al((size_t)P * L * L * 4);
    t += al((size_t)P * L * 4) * 5;
    t += al((size_t)2 * P * 4) * 2;
    t += al(lg_attention_part_bytes(2 * P, L));
    return t + al(extra_bytes) + 4096;
}
void lg_carve(void* ws, int P, int L, LgBuffers& b, size_t extra_bytes = 0) {
    const size_t rows = (size_t)2 * P * L;
    Bump a(ws);
    b.x = a.take<float>(rows * 256); b.ctx = a.take<float>(rows * 256); b.msg = a.take<float>(rows * 256);
    b.md = a.take<float>(rows * 256);
    b.kn = a.take<float>(rows * 2); b.csn = a.take<float>(rows * 64); b.lnstat = a.take<float>(rows * 32);   // <= 16 partial pairs per row
    b.qkv = a.take<float>(rows * 768); b.h = a.take<float>(rows * 512); b.z = a.take<float>(rows);
    b.sim = a.take<float>((size_t)P * L * L);
    b.rowlse = a.take<float>((size_t)P * L); b.collse = a.take<float>((size_t)P * L); b.mx0 = a.take<float>((size_t)P * L);
    b.a0 = a.take<int32_t>((size_t)P * L); b.a1 = a.take<int32_t>((size_t)P * L);
    b.lens = a.take<int32_t>((size_t)2 * P); b.kvmap = a.take<int32_t>((size_t)2 * P);
    { const size_t pb = lg_attention_part_bytes(2 * P, L); b.apart = pb ? a.take<float>(pb / 4) : nullptr; }
    b.extra = a.take<char>(extra_bytes);
}

// scratch of the split-key attention: carved for 2P sequences; a call on fewer sequences (stream mode's per-frame self
// block) may use it whenever its own requirement fits
float* lg_part(const LgBuffers& b, int nseq, int L) { return (b.apart && lg_attention_part_bytes(nseq, L) > 0) ? b.apart : nullptr; }


// x + ffn([x | msg]) in place on x
void lg_ffn(rfe_ctx* c, LgBuffers& b, float* x, const float* second, int rows, const float* w1, const float* b1, const float* g,
            const float* be, const float* w2, const float* b2) {
    hipStream_t s = c->stream;
    // LayerNorm(512) + GELU between the two Linears is fused across them: ffn.0's epilogue leaves per-row partial sums next to the
    // raw h, ffn.3 normalises while it stages its A tiles -- h crosses HBM once in each direction instead of twice (268 MB per block
    // saved, one launch fewer).  Throughput tiles only: launch_gemm_nt returns 0 partials for small problems, which keep the
    // stand-alone lg_ln_gelu pass (as does RFE_LN_FUSE=0 in the tuning build).
    static const bool ln_fuse = tune_int("RFE_LN_FUSE", 1) != 0;
    int P = 0;
    { ProfScope p(c, "lg_ffn1");   // A = [x | second]: second is the message, or the attention context when Wo is folded into W1
      GemmArgs a = gemm_lgw(c, x, 256, w1, 512, b1, b.h, 512, rows, 512, 512);
      a.A2 = second; a.lda2 = 256; a.K1 = 256;
      if (ln_fuse && !gemm_latency_regime(a)) a.stats_out = b.lnstat;   // latency regime: the stand-alone pass below (see lg_kernels.hip)
      P = launch_gemm_nt(s, a); }
    if (P == 0 && !c->opt_lg_fp16x2) {
        // one / few pairs per call: LayerNorm + GELU inside ffn.3 (ffn2_lat.hip: the 16 x 512 panel normalised once per workgroup) -- no stand-alone
        // pass, no second round trip of h.  (The fp16x2 option keeps the split form of gemm_lat behind the stand-alone pass.)
        ProfScope p(c, "lg_ffn2");
        if (launch_ffn2_ln_lat(s, b.h, w2, b2, g, be, x, 256, x, 256, rows)) return;
    }
    if (P == 0) { ProfScope p(c, "lg_ln_gelu"); launch_lg_ln_gelu(s, b.h, g, be, rows); }   // small problems (and RFE_LN_FUSE=0): stand-alone pass
    { ProfScope p(c, "lg_ffn2");
      GemmArgs a = gemm_lgw(c, b.h, 512, w2, 512, b2, x, 256, rows, 256, 512);
      a.R = x; a.ldr = 256;
      if (P > 0) { a.stats_in = b.lnstat; a.stats_p = P; a.ln_g = g; a.ln_b = be; }
      launch_gemm_nt(s, a); }
}

// projection + attention of a self block: b.qkv <- [q | k | v] (q | k rotated, unless the fallback named below ran), b.ctx <- softmax(q k^T / 8) v per head.
// Returns whether b.qkv holds ROTATED q | k (false: the plain-epilogue fallback, rotary applied by the attention kernel on load).
bool lg_self_qkv_attention(rfe_ctx* c, LgBuffers& b, const LgLayerDev& Lw, const float* x, const float* csn, const int32_t* lens, int nseq, int L) {
    hipStream_t s = c->stream;
    const int rows = nseq * L;
    // q,k,v = Wqkv x + b, q and k rotated by the projection's epilogue -- gemm_lat.hip at one / few pairs, gemm.hip's ROPE tile at throughput shapes (table
    // rows staged into LDS by DMA under the K loop) -- so that every attention kernel runs without a table and takes its K tiles straight into LDS
    // (self blocks: lg_attention_dma_kernel 523 us against 565 us for the form that rotates every staged K tile; +9 us on the projection).
    // Rotating only K there and q as the attention loads it measured worse on both sides (RFE_QKV_ROPE=2, profiles/r05_ab_notes.md).
    // RFE_OPT_LG_FP16X2: gemm_h2.hip has no rotary epilogue, lg_attention_h2_kernel rotates both on load.
    bool roped = false, k_roped = false;
    { ProfScope p(c, "lg_qkv");
      GemmArgs a = gemm_lgw(c, x, 256, Lw.wqkv, 256, Lw.bqkv, b.qkv, 768, rows, 768, 256);
      static const int nt_rope = tune_int("RFE_QKV_ROPE", 1);   // tuning switch: 0 = plain epilogue, rotary on load in lg_attention_kernel<.., ROPE>; 2 = only k in the epilogue, q on load
      if (gemm_latency_regime(a) && launch_gemm_lat(s, a, csn, 512)) roped = true;
      else {
          if (nt_rope && csn) {
              a.rope_c0 = nt_rope == 2 ? 256 : 0; a.rope_c1 = 512;
              if (gemm_nt_rope_ok(a)) { a.rope_csn = csn; k_roped = true; roped = nt_rope != 2; }
          }
          launch_gemm_nt(s, a);
      } }
    { ProfScope p(c, "lg_attention");
      launch_lg_attention(s, b.qkv, b.qkv + 256, b.qkv + 512, 768, b.ctx, nseq, L, L, lens, lens, nullptr, lg_part(b, nseq, L), roped ? nullptr : csn, c->opt_lg_fp16x2, k_roped); }
    return roped;
}

// self block on `nseq` sequences of L tokens held in x (in place); scratch: b.qkv, b.ctx, b.msg, b.h
void lg_self_block(rfe_ctx* c, LgBuffers& b, const LgLayerDev& Lw, float* x, const float* csn, const int32_t* lens, int nseq, int L) {
    hipStream_t s = c->stream;
    const int rows = nseq * L;
    lg_self_qkv_attention(c, b, Lw, x, csn, lens, nseq, L);
    if (c->opt_lg_fold) {
        lg_ffn(c, b, x, b.ctx, rows, Lw.w1f, Lw.b1f, Lw.lng, Lw.lnb, Lw.w2, Lw.b2);
    } else {
        { ProfScope p(c, "lg_proj"); launch_gemm_nt(s, gemm_lgw(c, b.ctx, 256, Lw.wo, 256, Lw.bo, b.msg, 256, rows, 256, 256)); }
        lg_ffn(c, b, x, b.msg, rows, Lw.w1, Lw.b1, Lw.lng, Lw.lnb, Lw.w2, Lw.b2);
    }
}

// runs the 9 layers + assignment on already staged b.x / b.kn / b.lens / b.kvmap.
// first_self_done: b.x already holds the output of layer 0's self block and b.csn the rotary table
// (stream mode computes them once per FRAME instead of once per pair side).
int lg_forward(rfe_ctx* c, LgBuffers& b, int P, int L, float thr, int cap, int32_t* S, int32_t* pairs, float* ms,
               float* scores_opt, bool first_self_done = false, bool posenc_done = false) {
    hipStream_t s = c->stream;
    const LgWeightsDev& W = c->lg;
    const int rows = 2 * P * L, nseq = 2 * P;
    if (!first_self_done && !posenc_done) { ProfScope p(c, "lg_misc"); launch_lg_posenc(s, b.kn, W.wr, rows, b.csn); }
    for (int l = 0; l < LG_LAYERS; ++l) {
        const LgLayerDev& Lw = W.L[l];
        if (l > 0 || !first_self_done) lg_self_block(c, b, Lw, b.x, b.csn, b.lens, nseq, L);
        // ---- cross block
        { ProfScope p(c, "lg_cross_qkv"); launch_gemm_nt(s, gemm_lgw(c, b.x, 256, Lw.cwqkv, 256, Lw.cbqkv, b.qkv, 512, rows, 512, 256)); }
        { ProfScope p(c, "lg_attention"); launch_lg_attention(s, b.qkv, b.qkv, b.qkv + 256, 512, b.ctx, nseq, L, L, b.lens, b.lens, b.kvmap, lg_part(b, nseq, L), nullptr, c->opt_lg_fp16x2); }
        if (c->opt_lg_fold) {
            lg_ffn(c, b, b.x, b.ctx, rows, Lw.cw1f, Lw.cb1f, Lw.clng, Lw.clnb, Lw.cw2, Lw.cb2);
        } else {
            { ProfScope p(c, "lg_proj"); launch_gemm_nt(s, gemm_lgw(c, b.ctx, 256, Lw.cwo, 256, Lw.cbo, b.msg, 256, rows, 256, 256)); }
            lg_ffn(c, b, b.x, b.msg, rows, Lw.cw1, Lw.cb1, Lw.clng, Lw.clnb, Lw.cw2, Lw.cb2);
        }
    }
    // ---- assignment
    { ProfScope p(c, "lg_proj");
      GemmArgs a = gemm_lgw(c, b.x, 256, W.wp, 256, W.bp, b.md, 256, rows, 256, 256);
      a.alpha = 0.25f;  // / 256^(1/4)
      launch_gemm_nt(s, a); }
    { ProfScope p(c, "lg_sim");
      GemmArgs g = gemm_lg(b.md, 256, b.md + (size_t)P * L * 256, 256, nullptr, b.sim, L, L, L, 256);
      g.batch = P; g.sA = (long long)L * 256; g.sB = (long long)L * 256; g.sC = (long long)L * L;
      g.m_valid = b.lens;
      launch_gemm_nt(s, g); }
    // one-shot test tap (rfe_k_set_lightglue_tap): final token states [L,256] per side and the log-assignment matrix [L,L]
    // of ONE pair of this forward, whatever the entry point and tiling (batched, stream, stereo frame)
    const bool tap = c->tap.armed && c->tap.pair < P;
    c->tap.armed = false;
    int scores_pair = -1;
    if (tap && c->tap.scores && !scores_opt) { scores_opt = c->tap.scores; scores_pair = c->tap.pair; }
    { ProfScope p(c, "lg_assign");
      if (!lg_assign_few_pairs(P, L)) launch_lg_matchability(s, b.x, W.wm, W.bm, rows, b.z);   // few pairs: inside the row log-sum-exp launch
      launch_lg_assign(s, b.sim, b.z, b.z + (size_t)P * L, P, L, cap, b.lens, b.lens + P, thr, scores_opt, b.rowlse,
                       b.collse, b.a0, b.mx0, b.a1, S, pairs, ms, scores_pair, b.x, W.wm, W.bm, b.z); }
    if (tap) {
        if (c->tap.x0) RFE_HIP(c, hipMemcpyAsync(c->tap.x0, b.x + (size_t)c->tap.pair * L * 256, (size_t)L * 1024, hipMemcpyDeviceToDevice, s));
        if (c->tap.x1) RFE_HIP(c, hipMemcpyAsync(c->tap.x1, b.x + (size_t)(P + c->tap.pair) * L * 256, (size_t)L * 1024, hipMemcpyDeviceToDevice, s));
    }
    RFE_HIP(c, hipGetLastError());
    return RFE_OK;
}

// stage device inputs [P,Mmax,*]/[P,Nmax,*] into the padded side-major token layout: descriptors -> x, normalised keypoints -> kn,
// rows past the input capacity zeroed, clamped lengths + cross-attention map -- one launch (it replaced two memsets, four strided
// device-to-device copies and the set-up kernel: eight enqueues per call on the single-pair latency path)
__global__ __launch_bounds__(256) void lg_stage_kernel(const float* __restrict__ k0n, const float* __restrict__ k1n,
                                                       const float* __restrict__ d0, const float* __restrict__ d1,
                                                       const int32_t* __restrict__ m, const int32_t* __restrict__ n, int P, int Mmax,
                                                       int Nmax, int L, float* __restrict__ x, float* __restrict__ kn,
                                                       int32_t* __restrict__ lens, int32_t* __restrict__ kvmap,
                                                       const float* __restrict__ wr, float2* __restrict__ csn) {
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < 2 * P; i += 256) {
            int v = i < P ? m[i] : n[i - P];
            const int cap = i < P ? Mmax : Nmax;
            v = v < 0 ? 0 : (v > cap ? cap : v);
            lens[i] = v;
            kvmap[i] = i < P ? i + P : i - P;
        }
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);     // side-major: row = (side * P + pair) * L + i
    if (row >= (int64_t)2 * P * L) return;
    const int lane = threadIdx.x & 63;
    const int i = (int)(row % L), sp = (int)(row / L), side = sp >= P, pair = side ? sp - P : sp;
    const int cap = side ? Nmax : Mmax;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 kv = make_float2(0.f, 0.f);
    if (i < cap) {
        const size_t src = (size_t)pair * cap + i;
        v = reinterpret_cast<const float4*>((side ? d1 : d0) + src * 256)[lane];
        if (lane == 0) kv = reinterpret_cast<const float2*>(side ? k1n : k0n)[src];
    }
    reinterpret_cast<float4*>(x + row * 256)[lane] = v;
    if (lane == 0) reinterpret_cast<float2*>(kn)[row] = kv;
    if (csn) {   // the rotary table row of this token (lg_posenc_kernel's arithmetic): saves the stand-alone launch in front of every forward
        const float kx = __shfl(kv.x, 0), ky = __shfl(kv.y, 0);
        if (lane < 32) {
            const float th = fmaf(wr[2 * lane + 1], ky, wr[2 * lane] * kx);
            csn[row * 32 + lane] = make_float2(cosf(th), sinf(th));
        }
    }
}

int lg_stage(rfe_ctx* c, LgBuffers& b, const float* k0n, const float* k1n, const float* d0, const float* d1,
             const int32_t* m, const int32_t* n, int P, int Mmax, int Nmax, int L) {
    const int64_t rows = (int64_t)2 * P * L;
    hipLaunchKernelGGL(lg_stage_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, c->stream, k0n, k1n, d0, d1, m, n, P, Mmax, Nmax, L,
                       b.x, b.kn, b.lens, b.kvmap, c->lg.wr, reinterpret_cast<float2*>(b.csn));   // the rotary table too: lg_forward(posenc_done = true)
    return RFE_OK;
}

int lg_check(rfe_ctx* c, int P, int Mmax, int Nmax) {
    if (!c) return RFE_ERR_INVALID;
    if (!c->has_lg) return fail(c, RFE_ERR_NO_WEIGHTS, "LightGlue weights not loaded (rfe_load_weights / rfe_set_weights)");
    if (P <= 0 || Mmax <= 0 || Nmax <= 0 || Mmax > 4096 || Nmax > 4096) return fail(c, RFE_ERR_INVALID, "match: P > 0 and 1 <= Mmax,Nmax <= 4096 required");
    return RFE_OK;
}

}  // namespace

extern "C" int rfe_match_dev(rfe_ctx* c, const float* k0n, const float* k1n, const float* d0, const float* d1,
                             const int32_t* m, const int32_t* n, int P, int Mmax, int Nmax, float thr, int32_t* S,
                             int32_t* pairs, float* ms) {
    int rc = lg_check(c, P, Mmax, Nmax);
    if (rc) return rc;
    if (!k0n || !k1n || !d0 || !d1 || !m || !n || !S || !pairs || !ms) return fail(c, RFE_ERR_INVALID, "match: null pointer");
    RFE_HIP(c, hipSetDevice(c->device));
    const int L = ((std::max(Mmax, Nmax) + 3) / 4) * 4;
    if ((rc = ensure_ws(c, &c->ws_lg, &c->ws_lg_bytes, lg_ws_bytes(P, L)))) return rc;
    LgBuffers b;
    lg_carve(c->ws_lg, P, L, b);
    if ((rc = lg_stage(c, b, k0n, k1n, d0, d1, m, n, P, Mmax, Nmax, L))) return rc;
    return lg_forward(c, b, P, L, thr, std::min(Mmax, Nmax), S, pairs, ms, nullptr, false, true);
}

extern "C" int rfe_match(rfe_ctx* c, const float* k0n, const float* k1n, const float* d0, const float* d1,
                         const int32_t* m, const int32_t* n, int P, int Mmax, int Nmax, float thr, int32_t* S,
                         int32_t* pairs, float* ms) {
    int rc = lg_check(c, P, Mmax, Nmax);
    if (rc) return rc;
    if (!k0n || !k1n || !d0 || !d1 || !m || !n || !S || !pairs || !ms) return fail(c, RFE_ERR_INVALID, "match: null pointer");
    RFE_HIP(c, hipSetDevice(c->device));
    const int cap = std::min(Mmax, Nmax);
    const size_t bk0 = al((size_t)P * Mmax * 8), bk1 = al((size_t)P * Nmax * 8), bd0 = al((size_t)P * Mmax * 1024),
                 bd1 = al((size_t)P * Nmax * 1024), bi = al((size_t)P * 4), bp = al((size_t)P * cap * 8), bs = al((size_t)P * cap * 4);
    if ((rc = ensure_ws(c, &c->ws_io, &c->ws_io_bytes, bk0 + bk1 + bd0 + bd1 + 3 * bi + bp + bs))) return rc;
    char* p = (char*)c->ws_io;
    float* dk0 = (float*)p; p += bk0; float* dk1 = (float*)p; p += bk1; float* dd0 = (float*)p; p += bd0; float* dd1 = (float*)p; p += bd1;
    int32_t* dm = (int32_t*)p; p += bi; int32_t* dn = (int32_t*)p; p += bi; int32_t* dS = (int32_t*)p; p += bi;
    int32_t* dp = (int32_t*)p; p += bp; float* dms = (float*)p;
    hipStream_t s = c->stream;
    // inputs packed into the pinned mirror of the device block (one DMA in), results fetched with one DMA out: see ensure_pin
    const size_t in_bytes = bk0 + bk1 + bd0 + bd1 + 2 * bi, out_bytes = bi + bp + bs;
    if ((rc = ensure_pin(c, in_bytes + out_bytes))) return rc;
    char* hp = (char*)c->h_pin;
    // two DMAs, so that the host copy of the second descriptor block (1 MB at K = 1024) runs while the first one is on the bus
    memcpy(hp, k0n, (size_t)P * Mmax * 8);
    memcpy(hp + bk0, k1n, (size_t)P * Nmax * 8);
    memcpy(hp + bk0 + bk1, d0, (size_t)P * Mmax * 1024);
    const size_t first = bk0 + bk1 + bd0;
    RFE_HIP(c, hipMemcpyAsync(c->ws_io, hp, first, hipMemcpyHostToDevice, s));
    memcpy(hp + first, d1, (size_t)P * Nmax * 1024);
    memcpy(hp + first + bd1, m, (size_t)P * 4);
    memcpy(hp + first + bd1 + bi, n, (size_t)P * 4);
    RFE_HIP(c, hipMemcpyAsync((char*)c->ws_io + first, hp + first, in_bytes - first, hipMemcpyHostToDevice, s));
    {
        const int L = ((std::max(Mmax, Nmax) + 3) / 4) * 4;
        if ((rc = ensure_ws(c, &c->ws_lg, &c->ws_lg_bytes, lg_ws_bytes(P, L)))) return rc;
        int thr_bits; memcpy(&thr_bits, &thr, 4);
        if ((rc = run_host_graph(c, c->g_match, host_graph_key(c, "m", {P, Mmax, Nmax, thr_bits}),
                                 [&] { return rfe_match_dev(c, dk0, dk1, dd0, dd1, dm, dn, P, Mmax, Nmax, thr, dS, dp, dms); }))) return rc;
    }
    RFE_HIP(c, hipMemcpyAsync(hp + in_bytes, dS, out_bytes, hipMemcpyDeviceToHost, s));
    RFE_HIP(c, hipStreamSynchronize(s));
    memcpy(S, hp + in_bytes, (size_t)P * 4);
    memcpy(pairs, hp + in_bytes + bi, (size_t)P * cap * 8);
    memcpy(ms, hp + in_bytes + bi + bp, (size_t)P * cap * 4);
    prof_collect(c);
    return RFE_OK;
}

extern "C" int rfe_match_fused(rfe_ctx* c, const float* kp0, int M, const float* kp1, int N, const float* desc0,
                               const float* desc1, int rows, int cols, float filter_thr, float match_thresh,
                               int32_t* vnMatches12) {
    if (!c) return RFE_ERR_INVALID;
    if (M < 0 || N < 0 || !vnMatches12) return fail(c, RFE_ERR_INVALID, "match_fused: bad argument");
    for (int i = 0; i < M; ++i) vnMatches12[i] = -1;   // vnMatches12.resize(M, -1): SPmatcher.cc:375,413,460
    if (M == 0 || N == 0) return 0;
    // NormalizeKeypoints, reference src/Matchers/transform.cpp:19-32
    std::vector<float> k0((size_t)M * 2), k1((size_t)N * 2);
    const float sx = (float)cols / 2, sy = (float)rows / 2, scale = (float)std::max(cols, rows) / 2;
    for (int i = 0; i < M; ++i) { k0[2 * i] = (kp0[2 * i] - sx) / scale; k0[2 * i + 1] = (kp0[2 * i + 1] - sy) / scale; }
    for (int i = 0; i < N; ++i) { k1[2 * i] = (kp1[2 * i] - sx) / scale; k1[2 * i + 1] = (kp1[2 * i + 1] - sy) / scale; }
    const int cap = std::min(M, N);
    std::vector<int32_t> pairs((size_t)cap * 2);
    std::vector<float> ms(cap);
    int32_t S = 0, m = M, n = N;
    int rc = rfe_match(c, k0.data(), k1.data(), desc0, desc1, &m, &n, 1, M, N, filter_thr, &S, pairs.data(), ms.data());
    if (rc) return rc;
    // Matcher_PostProcess_fused, reference src/Matchers/lightglue_onnx.cpp:437-453
    int size = 0;
    for (int i = 0; i < S; ++i)
        if (ms[i] > match_thresh) { ++size; vnMatches12[pairs[2 * i]] = pairs[2 * i + 1]; }
    return size;
}

// =====================================================================================
// batched stream: extract B frames, match (i, i+1)
// =====================================================================================
extern "C" int rfe_extract_match_stream_dev(rfe_ctx* c, const uint8_t* img, int H, int W, int stride, int B, int Kmax,
                                            float thr, float filter_thr, int32_t* n, int32_t* kxy, float* score,
                                            float* desc, int32_t* S, int32_t* pairs, float* ms) {
    int rc = rfe_extract_u8_dev(c, img, H, W, stride, B, Kmax, thr, n, kxy, score, desc);
    if (rc || B < 2) return rc;
    if ((rc = lg_check(c, B - 1, Kmax, Kmax))) return rc;
    if (!S || !pairs || !ms) return fail(c, RFE_ERR_INVALID, "stream: null match output");
    const int P = B - 1, L = ((Kmax + 3) / 4) * 4;
    const size_t kn_bytes = al((size_t)B * Kmax * 8);        // normalised keypoints of all B frames
    const size_t rot_bytes = al((size_t)B * L * 64 * 4);      // per-FRAME rotary table (cos, sin pairs)
    const size_t extra_bytes = kn_bytes + rot_bytes;
    if ((rc = ensure_ws(c, &c->ws_lg, &c->ws_lg_bytes, lg_ws_bytes(P, L, extra_bytes)))) return rc;
    LgBuffers b;
    lg_carve(c->ws_lg, P, L, b, extra_bytes);
    float* kn_all = (float*)b.extra;
    hipStream_t s = c->stream;
    static const bool dedup = tune_env("RFE_NO_SELF_DEDUP") == nullptr;   // tuning / test switch
    if (!dedup || L != Kmax) {
        { ProfScope p(c, "lg_misc");
          launch_normalize_kpts(s, kxy, (int64_t)B * Kmax, H, W, kn_all);
          if ((rc = lg_stage(c, b, kn_all, kn_all + (size_t)Kmax * 2, desc, desc + (size_t)Kmax * 256, n, n + 1, P, Kmax, Kmax, L))) return rc; }
        return lg_forward(c, b, P, L, filter_thr, Kmax, S, pairs, ms, nullptr, false, true);
    }
    // Every interior frame is side 1 of pair i-1 and side 0 of pair i, and layer 0's self block depends on
    // the frame alone: run it (and the positional encoding) once per FRAME, then scatter into the pair layout.
    // B = 2 (one pair): the per-frame layout IS the pair layout, the block runs in place and nothing is scattered
    float* xf = P == 1 ? b.x : b.md;         // [B, L, 256]: md ([2P, L, 256], B <= 2P) is only used by the assignment at the end
    float* csnf = P == 1 ? b.csn : (float*)(b.extra + kn_bytes);   // [B*L, 32, 2], own scratch (the similarity buffer [P, L, L] is too small
                                                                   //  for it when L < 64 (P+1)/P)
    { ProfScope p(c, "lg_misc");   // one launch: NormalizeKeypoints + rotary table + descriptors -> token rows + lengths / cross map
      launch_lg_frame_prologue(s, kxy, desc, c->lg.wr, n, B, L, H, W, kn_all, csnf, xf, b.lens, b.kvmap); }
    lg_self_block(c, b, c->lg.L[0], xf, csnf, n, B, L);
    if (P > 1) {
      ProfScope p(c, "lg_misc");
      const size_t half = (size_t)P * L;
      launch_copy_f32(s, xf, b.x, (int64_t)half * 256);
      launch_copy_f32(s, xf + (size_t)L * 256, b.x + half * 256, (int64_t)half * 256);
      launch_copy_f32(s, csnf, b.csn, (int64_t)half * 64);
      launch_copy_f32(s, csnf + (size_t)L * 64, b.csn + half * 64, (int64_t)half * 64); }
    RFE_HIP(c, hipGetLastError());
    return lg_forward(c, b, P, L, filter_thr, Kmax, S, pairs, ms, nullptr, true);
}

// =====================================================================================
// sparse stereo matching (Frame::ComputeStereoMatches, src/Frame.cc:1159-1446)
// =====================================================================================
extern "C" int rfe_stereo_match_dev(rfe_ctx* c, const uint8_t* imgL, const uint8_t* imgR, int H, int W, int stride,
                                    const float* kL, int N, const float* kR, int Nr, const float* dL, const float* dR,
                                    float mb, float mbf, float* uRight, float* depth) {
    if (!c) return RFE_ERR_INVALID;
    if (N < 0 || Nr < 0 || N > 4096 || H <= 0 || W <= 0 || stride < W || !(mb > 0.f)) return fail(c, RFE_ERR_INVALID, "stereo_match: bad argument (0 <= N <= 4096, mb > 0, stride >= W)");
    if (N == 0) return RFE_OK;
    if (!imgL || !imgR || !kL || !dL || !uRight || !depth || (Nr > 0 && (!kR || !dR))) return fail(c, RFE_ERR_INVALID, "stereo_match: null pointer");
    RFE_HIP(c, hipSetDevice(c->device));
    int rc = ensure_ws(c, &c->ws_tmp, &c->ws_tmp_bytes, al((size_t)N * 4));
    if (rc) return rc;
    ProfScope p(c, "stereo_match");
    launch_stereo_match(c->stream, imgL, imgR, H, W, stride, kL, N, kR, Nr, dL, dR, mb, mbf, uRight, depth, (int32_t*)c->ws_tmp);
    RFE_HIP(c, hipGetLastError());
    return RFE_OK;
}

extern "C" int rfe_stereo_match(rfe_ctx* c, const uint8_t* imgL, const uint8_t* imgR, int H, int W, int stride,
                                const float* kL, int N, const float* kR, int Nr, const float* dL, const float* dR,
                                float mb, float mbf, float* uRight, float* depth) {
    if (!c) return RFE_ERR_INVALID;
    if (N < 0 || Nr < 0 || N > 4096 || H <= 0 || W <= 0 || stride < W) return fail(c, RFE_ERR_INVALID, "stereo_match: bad argument");
    if (N == 0) return RFE_OK;
    RFE_HIP(c, hipSetDevice(c->device));
    const size_t bi = al((size_t)H * W), bkl = al((size_t)N * 8), bkr = al((size_t)std::max(Nr, 1) * 8),
                 bdl = al((size_t)N * 1024), bdr = al((size_t)std::max(Nr, 1) * 1024), bo = al((size_t)N * 4);
    int rc = ensure_ws(c, &c->ws_io, &c->ws_io_bytes, 2 * bi + bkl + bkr + bdl + bdr + 2 * bo);
    if (rc) return rc;
    char* p = (char*)c->ws_io;
    uint8_t* dIL = (uint8_t*)p; p += bi; uint8_t* dIR = (uint8_t*)p; p += bi;
    float* dkl = (float*)p; p += bkl; float* dkr = (float*)p; p += bkr; float* ddl = (float*)p; p += bdl; float* ddr = (float*)p; p += bdr;
    float* du = (float*)p; p += bo; float* dz = (float*)p;
    hipStream_t s = c->stream;
    RFE_HIP(c, hipMemcpy2DAsync(dIL, (size_t)W, imgL, (size_t)stride, (size_t)W, (size_t)H, hipMemcpyHostToDevice, s));   // tight device copy (ROI-safe)
    RFE_HIP(c, hipMemcpy2DAsync(dIR, (size_t)W, imgR, (size_t)stride, (size_t)W, (size_t)H, hipMemcpyHostToDevice, s));
    RFE_HIP(c, hipMemcpyAsync(dkl, kL, (size_t)N * 8, hipMemcpyHostToDevice, s));
    RFE_HIP(c, hipMemcpyAsync(ddl, dL, (size_t)N * 1024, hipMemcpyHostToDevice, s));
    if (Nr > 0) {
        RFE_HIP(c, hipMemcpyAsync(dkr, kR, (size_t)Nr * 8, hipMemcpyHostToDevice, s));
        RFE_HIP(c, hipMemcpyAsync(ddr, dR, (size_t)Nr * 1024, hipMemcpyHostToDevice, s));
    }
    if ((rc = rfe_stereo_match_dev(c, dIL, dIR, H, W, W, dkl, N, dkr, Nr, ddl, ddr, mb, mbf, du, dz))) return rc;
    RFE_HIP(c, hipMemcpyAsync(uRight, du, (size_t)N * 4, hipMemcpyDeviceToHost, s));
    RFE_HIP(c, hipMemcpyAsync(depth, dz, (size_t)N * 4, hipMemcpyDeviceToHost, s));
    RFE_HIP(c, hipStreamSynchronize(s));
    prof_collect(c);
    return RFE_OK;
}

// =====================================================================================
// stereo stream (BASELINE configs[4]): one device-resident entry point per stereo frame
// =====================================================================================
__global__ void st_zero_count_kernel(int32_t* S) { S[0] = 0; }

// One launch in front of the temporal match of a stereo frame (round 5: it replaced normalize_kpts + lg_stage + the save kernel behind the match).  One wave per
// token row of the pair layout [side 0 = this left view | side 1 = the previous left view], L rows each:
//   side 0: NormalizeKeypoints (reference src/Matchers/transform.cpp:19-32) of the integer keypoint, descriptor row -> x, rotary table row -> csn, AND both into
//           the state slot that becomes "previous" for the next frame (two slots, flipped per frame: nothing is copied after the match);
//   side 1: the stored normalised keypoint / descriptor of the previous view -> x, csn;
// workgroup 0: clamped lengths, cross-attention map, the next slot's keypoint count.  have_prev = 0 (first frame of a stream): side 1 is zero-filled, length 0.
__global__ __launch_bounds__(256) void st_stage_kernel(const int32_t* __restrict__ kxy, const float* __restrict__ desc, const int32_t* __restrict__ n, int Kmax, int L,
                                                       float sx, float sy, float scale, const float* __restrict__ kn_prev, const float* __restrict__ desc_prev,
                                                       const int32_t* __restrict__ n_prev, int have_prev, const float* __restrict__ wr, float* __restrict__ x,
                                                       float* __restrict__ kn, float2* __restrict__ csn, int32_t* __restrict__ lens, int32_t* __restrict__ kvmap,
                                                       float* __restrict__ kn_next, float* __restrict__ desc_next, int32_t* __restrict__ n_next) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int v0 = n[0]; v0 = v0 < 0 ? 0 : (v0 > Kmax ? Kmax : v0);
        int v1 = have_prev ? n_prev[0] : 0; v1 = v1 < 0 ? 0 : (v1 > Kmax ? Kmax : v1);
        lens[0] = v0; lens[1] = v1; kvmap[0] = 1; kvmap[1] = 0;
        n_next[0] = n[0];
    }
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= 2 * L) return;
    const int side = row >= L, i = side ? row - L : row;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    float kx = 0.f, ky = 0.f;
    if (i < Kmax) {
        if (!side) {
            v = reinterpret_cast<const float4*>(desc + (size_t)i * 256)[lane];
            kx = ((float)kxy[2 * i] - sx) / scale; ky = ((float)kxy[2 * i + 1] - sy) / scale;
            reinterpret_cast<float4*>(desc_next + (size_t)i * 256)[lane] = v;
            if (lane == 0) reinterpret_cast<float2*>(kn_next)[i] = make_float2(kx, ky);
        } else if (have_prev) {
            v = reinterpret_cast<const float4*>(desc_prev + (size_t)i * 256)[lane];
            const float2 kp = reinterpret_cast<const float2*>(kn_prev)[i];
            kx = kp.x; ky = kp.y;
        }
    }
    reinterpret_cast<float4*>(x + (size_t)row * 256)[lane] = v;
    if (lane == 0) reinterpret_cast<float2*>(kn)[row] = make_float2(kx, ky);
    if (lane < 32) {
        const float th = fmaf(wr[2 * lane + 1], ky, wr[2 * lane] * kx);
        csn[(size_t)row * 32 + lane] = make_float2(cosf(th), sinf(th));
    }
}

extern "C" int rfe_stereo_frame_dev(rfe_ctx* c, const uint8_t* imgL, const uint8_t* imgR, int H, int W, int stride, int Kmax,
                                    float thr, float filter_thr, float mb, float mbf, int reset, int32_t* n, int32_t* kxy,
                                    float* score, float* desc, float* uRight, float* depth, int32_t* S, int32_t* pairs,
                                    float* ms) {
    int rc = sp_check(c, H, W, 2, Kmax);
    if (rc) return rc;
    if ((rc = lg_check(c, 1, Kmax, Kmax))) return rc;
    if (!imgL || !imgR || !n || !kxy || !score || !desc || !uRight || !depth || !S || !pairs || !ms || stride < W || !(mb > 0.f))
        return fail(c, RFE_ERR_INVALID, "stereo_frame: null pointer, stride < W or mb <= 0");
    RFE_HIP(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    // state: [2,H,W] staged views (unfused-conv1 tuning path only) | sadv [Kmax] | two slots of { kn [Kmax,2], desc [Kmax,256], n [1] }: the previous left
    // view lives in slot st_flip, this frame's staging kernel fills the other one, then the slots swap -- nothing is copied behind the match
    const size_t b_img = al((size_t)2 * H * W), b_sad = al((size_t)Kmax * 4), b_kn = al((size_t)Kmax * 8), b_desc = al((size_t)Kmax * 1024), b_n = al(4);
    const bool fresh = c->st_H != H || c->st_W != W || c->st_K != Kmax;
    if ((rc = ensure_ws(c, &c->ws_st, &c->ws_st_bytes, b_img + b_sad + 2 * (b_kn + b_desc + b_n) + 256))) return rc;
    if (fresh || reset) { c->st_have_prev = false; c->st_H = H; c->st_W = W; c->st_K = Kmax; }
    char* p = (char*)c->ws_st;
    uint8_t* d_img = (uint8_t*)p; p += b_img;
    int32_t* sadv = (int32_t*)p; p += b_sad;
    float* kn_slot[2]; float* desc_slot[2]; int32_t* n_slot[2];
    for (int q = 0; q < 2; ++q) { kn_slot[q] = (float*)p; p += b_kn; desc_slot[q] = (float*)p; p += b_desc; n_slot[q] = (int32_t*)p; p += b_n; }
    const int prev = c->st_flip & 1, next = prev ^ 1;
    // both views as ONE batch of 2 (the reference runs them on two threads, src/Frame.cc:142-147).  Measured alternative: the
    // right view + stereo match on a second lane (own streams / workspace) next to left view + LightGlue -- 3.52 ms per stereo
    // frame against 3.26 ms for this form: two batch-1 extractions are no faster than one batch of 2, and the co-running
    // kernels slow the latency-bound LightGlue chain (profiles/r02_ab_notes.md).  The two views are read where the caller has them (round 5: conv1's
    // tile loader takes the distance between frame 0 and frame 1 -- any distance, here imgR - imgL; the two staging copies are gone).  The caller keeps
    // them valid until the ctx stream has passed this call, like every input of a *_dev entry.
    const uint8_t *vL = imgL, *vR = imgR;
    int vstride = stride;
    if (sp_unfused_conv1()) {     // tuning path (stand-alone conv1a has no frame step): stage as before
        RFE_HIP(c, hipMemcpy2DAsync(d_img, (size_t)W, imgL, (size_t)stride, (size_t)W, (size_t)H, hipMemcpyDeviceToDevice, s));
        RFE_HIP(c, hipMemcpy2DAsync(d_img + (size_t)H * W, (size_t)W, imgR, (size_t)stride, (size_t)W, (size_t)H, hipMemcpyDeviceToDevice, s));
        vL = d_img; vR = d_img + (size_t)H * W; vstride = W;
    }
    if ((rc = sp_forward(c, vL, H, W, vstride, 2, Kmax, thr, n, kxy, score, desc, nullptr, false, (long long)(vR - vL)))) return rc;
    // Frame::ComputeStereoMatches (src/Frame.cc:1159-1446) on the device-resident features; counts stay on the device
    // ... on the SIDE stream: the stereo kernels (45 us of small launches) and the temporal LightGlue match below only share their inputs, and
    // the one-pair LightGlue is a chain of latency-bound kernels that leaves room next to it (with an event pair around every stage, full
    // profiling pass, everything stays serial so that the stage times are clean)
    const bool st_fork = c->st_have_prev && !(c->prof && c->prof_filter.empty()) && tune_env("RFE_ST_NO_FORK") == nullptr;
    hipStream_t ss = st_fork ? c->side_stream : s;
    if (st_fork) { RFE_HIP(c, hipEventRecord(c->ev_fork, s)); RFE_HIP(c, hipStreamWaitEvent(ss, c->ev_fork, 0)); }
    { ProfScope ps(c, "stereo_match", ss);
      launch_stereo_match_counts(ss, vL, vR, H, W, vstride, kxy, kxy + (size_t)Kmax * 2, Kmax, n, desc,
                                 desc + (size_t)Kmax * 256, mb, mbf, uRight, depth, sadv); }
    if (st_fork) RFE_HIP(c, hipEventRecord(c->ev_join, ss));
    // every exit below -- the error returns of ensure_ws / lg_forward included -- joins the side stream first: the caller's NEXT call
    // rewrites uRight / depth (and its image buffers) on the ctx stream, which must not overtake stereo kernels still reading or writing them
    struct JoinGuard { rfe_ctx* c; hipStream_t s; bool on; ~JoinGuard() { if (on) (void)hipStreamWaitEvent(s, c->ev_join, 0); } } join_guard{c, s, st_fork};
    // temporal match exactly as Tracking issues it: SearchBySP(mCurrentFrame, mLastFrame) (src/Tracking.cc:3465) ->
    // MatchingPoints_onnx(CurrentFrame, LastFrame, vnMatches1) (src/Matchers/SPmatcher.cc:1050-1054): THIS left view is set 0,
    // the previous left view set 1, so pairs are (current index, previous index) like vnMatches1[IdxCF] = IdxLF; true image
    // size like the Frame overload (:457-542, :463-464)
    {
        const int L = ((Kmax + 3) / 4) * 4;
        if ((rc = ensure_ws(c, &c->ws_lg, &c->ws_lg_bytes, lg_ws_bytes(1, L)))) return rc;
        LgBuffers b;
        lg_carve(c->ws_lg, 1, L, b);
        const float sx = (float)W / 2, sy = (float)H / 2, scale = (float)(H > W ? H : W) / 2;     // launch_normalize_kpts' constants
        { ProfScope ps(c, "lg_misc");   // ONE staging launch: normalise, rotary table, token rows of both sides, lengths -- and this view into the next slot
          hipLaunchKernelGGL(st_stage_kernel, dim3((unsigned)((2 * L + 3) / 4)), dim3(256), 0, s, kxy, desc, n, Kmax, L, sx, sy, scale, kn_slot[prev], desc_slot[prev],
                             n_slot[prev], c->st_have_prev ? 1 : 0, c->lg.wr, b.x, b.kn, reinterpret_cast<float2*>(b.csn), b.lens, b.kvmap, kn_slot[next], desc_slot[next],
                             n_slot[next]); }
        if (c->st_have_prev) {
            if ((rc = lg_forward(c, b, 1, L, filter_thr, Kmax, S, pairs, ms, nullptr, false, true))) return rc;
        } else {
            hipLaunchKernelGGL(st_zero_count_kernel, dim3(1), dim3(1), 0, s, S);
        }
    }
    c->st_flip = next;
    if (st_fork) { join_guard.on = false; RFE_HIP(c, hipStreamWaitEvent(s, c->ev_join, 0)); }   // uRight / depth are complete when the ctx stream is
    c->st_have_prev = true;
    RFE_HIP(c, hipGetLastError());
    return RFE_OK;
}

// =====================================================================================
// descriptor helpers of the callers' classic searches.  The "_dev" forms take device pointers and are
// asynchronous on the ctx stream (descriptors usually ARE device resident: they come out of rfe_extract_u8_dev /
// rfe_stereo_frame_dev); the host-pointer forms validate, stage through ws_io and call them.
// =====================================================================================
extern "C" int rfe_l2_distance_matrix_dev(rfe_ctx* c, const float* a, int M, const float* b, int N, float* out) {
    if (!c) return RFE_ERR_INVALID;
    if (M < 0 || N < 0 || (M > 0 && N > 0 && (!a || !b || !out))) return fail(c, RFE_ERR_INVALID, "l2_distance_matrix: bad argument");
    if (M == 0 || N == 0) return RFE_OK;
    RFE_HIP(c, hipSetDevice(c->device));
    { ProfScope ps(c, "l2_matrix"); launch_l2_matrix(c->stream, a, M, b, N, out); }
    RFE_HIP(c, hipGetLastError());
    return RFE_OK;
}

extern "C" int rfe_l2_distance_matrix(rfe_ctx* c, const float* a, int M, const float* b, int N, float* out) {
    if (!c) return RFE_ERR_INVALID;
    if (M < 0 || N < 0 || (M > 0 && N > 0 && (!a || !b || !out))) return fail(c, RFE_ERR_INVALID, "l2_distance_matrix: bad argument");
    if (M == 0 || N == 0) return RFE_OK;
    RFE_HIP(c, hipSetDevice(c->device));
    const size_t ba = al((size_t)M * 1024), bb = al((size_t)N * 1024), bo = al((size_t)M * N * 4);
    int rc = ensure_ws(c, &c->ws_io, &c->ws_io_bytes, ba + bb + bo);
    if (rc) return rc;
    float* da = (float*)c->ws_io; float* db = (float*)((char*)c->ws_io + ba); float* dout = (float*)((char*)c->ws_io + ba + bb);
    RFE_HIP(c, hipMemcpyAsync(da, a, (size_t)M * 1024, hipMemcpyHostToDevice, c->stream));
    RFE_HIP(c, hipMemcpyAsync(db, b, (size_t)N * 1024, hipMemcpyHostToDevice, c->stream));
    if ((rc = rfe_l2_distance_matrix_dev(c, da, M, db, N, dout))) return rc;
    RFE_HIP(c, hipMemcpyAsync(out, dout, (size_t)M * N * 4, hipMemcpyDeviceToHost, c->stream));
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    return RFE_OK;
}

extern "C" int rfe_binarize_descriptors_dev(rfe_ctx* c, const float* desc, int rows, uint8_t* out) {
    if (!c) return RFE_ERR_INVALID;
    if (rows < 0 || (rows > 0 && (!desc || !out))) return fail(c, RFE_ERR_INVALID, "binarize_descriptors: bad argument");
    if (rows == 0) return RFE_OK;
    RFE_HIP(c, hipSetDevice(c->device));
    launch_binarize(c->stream, desc, rows, out);
    RFE_HIP(c, hipGetLastError());
    return RFE_OK;
}

extern "C" int rfe_binarize_descriptors(rfe_ctx* c, const float* desc, int rows, uint8_t* out) {
    if (!c) return RFE_ERR_INVALID;
    if (rows < 0 || (rows > 0 && (!desc || !out))) return fail(c, RFE_ERR_INVALID, "binarize_descriptors: bad argument");
    if (rows == 0) return RFE_OK;
    RFE_HIP(c, hipSetDevice(c->device));
    const size_t bd = al((size_t)rows * 1024), bo = al((size_t)rows * 256);
    int rc = ensure_ws(c, &c->ws_io, &c->ws_io_bytes, bd + bo);
    if (rc) return rc;
    float* dd = (float*)c->ws_io; uint8_t* dout = (uint8_t*)c->ws_io + bd;
    RFE_HIP(c, hipMemcpyAsync(dd, desc, (size_t)rows * 1024, hipMemcpyHostToDevice, c->stream));
    if ((rc = rfe_binarize_descriptors_dev(c, dd, rows, dout))) return rc;
    RFE_HIP(c, hipMemcpyAsync(out, dout, (size_t)rows * 256, hipMemcpyDeviceToHost, c->stream));
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    return RFE_OK;
}

// best / second-best scan of SPmatcher::SearchByProjection1 (src/Matchers/SPmatcher.cc:1218-1248) over device-resident CSR
// candidate lists.  The lists cannot be validated from the host without a synchronisation: the kernel ignores candidate
// indices outside [0, Nf); offsets must be non-decreasing with offsets[0] = 0 (the caller's contract, as for the host form).
extern "C" int rfe_search_candidates_dev(rfe_ctx* c, const float* q, int Nq, const float* f, int Nf, const int32_t* offsets,
                                         const int32_t* cand, const uint8_t* skip, int32_t* best_idx, float* best_dist,
                                         float* second_dist) {
    if (!c) return RFE_ERR_INVALID;
    if (Nq < 0 || Nf < 0) return fail(c, RFE_ERR_INVALID, "search_candidates: negative count");
    if (Nq == 0) return RFE_OK;
    if (!q || !offsets || !best_idx || !best_dist || !second_dist || (Nf > 0 && (!f || !cand)))
        return fail(c, RFE_ERR_INVALID, "search_candidates: null pointer");
    RFE_HIP(c, hipSetDevice(c->device));
    { ProfScope ps(c, "search_candidates");
      launch_search_candidates(c->stream, q, Nq, f, Nf, offsets, cand, skip, best_idx, best_dist, second_dist); }
    RFE_HIP(c, hipGetLastError());
    return RFE_OK;
}

extern "C" int rfe_search_candidates(rfe_ctx* c, const float* q, int Nq, const float* f, int Nf, const int32_t* offsets,
                                     const int32_t* cand, const uint8_t* skip, int32_t* best_idx, float* best_dist,
                                     float* second_dist) {
    if (!c) return RFE_ERR_INVALID;
    if (Nq < 0 || Nf < 0) return fail(c, RFE_ERR_INVALID, "search_candidates: negative count");
    if (Nq == 0) return RFE_OK;
    if (!q || !offsets || !best_idx || !best_dist || !second_dist) return fail(c, RFE_ERR_INVALID, "search_candidates: null pointer");
    if (offsets[0] != 0) return fail(c, RFE_ERR_INVALID, "search_candidates: offsets[0] must be 0");
    for (int i = 0; i < Nq; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(c, RFE_ERR_INVALID, "search_candidates: offsets must be non-decreasing");
    const int nnz = offsets[Nq];
    if (nnz > 0 && (!cand || !f)) return fail(c, RFE_ERR_INVALID, "search_candidates: null candidate list");
    for (int k = 0; k < nnz; ++k)
        if (cand[k] < 0 || cand[k] >= Nf) return fail(c, RFE_ERR_INVALID, "search_candidates: candidate index out of range");
    RFE_HIP(c, hipSetDevice(c->device));
    const size_t bq = al((size_t)Nq * 1024), bf = al((size_t)std::max(Nf, 1) * 1024), bo = al((size_t)(Nq + 1) * 4),
                 bc = al((size_t)std::max(nnz, 1) * 4), bs = al((size_t)std::max(Nf, 1)), br = al((size_t)Nq * 4);
    int rc = ensure_ws(c, &c->ws_io, &c->ws_io_bytes, bq + bf + bo + bc + bs + 3 * br);
    if (rc) return rc;
    char* p = (char*)c->ws_io;
    float* dq = (float*)p; p += bq; float* df = (float*)p; p += bf; int32_t* doff = (int32_t*)p; p += bo;
    int32_t* dc = (int32_t*)p; p += bc; uint8_t* dsk = (uint8_t*)p; p += bs;
    int32_t* dbi = (int32_t*)p; p += br; float* dbd = (float*)p; p += br; float* dsd = (float*)p;
    hipStream_t s = c->stream;
    RFE_HIP(c, hipMemcpyAsync(dq, q, (size_t)Nq * 1024, hipMemcpyHostToDevice, s));
    if (Nf > 0 && f) RFE_HIP(c, hipMemcpyAsync(df, f, (size_t)Nf * 1024, hipMemcpyHostToDevice, s));
    RFE_HIP(c, hipMemcpyAsync(doff, offsets, (size_t)(Nq + 1) * 4, hipMemcpyHostToDevice, s));
    if (nnz > 0) RFE_HIP(c, hipMemcpyAsync(dc, cand, (size_t)nnz * 4, hipMemcpyHostToDevice, s));
    if (skip && Nf > 0) RFE_HIP(c, hipMemcpyAsync(dsk, skip, (size_t)Nf, hipMemcpyHostToDevice, s));
    if ((rc = rfe_search_candidates_dev(c, dq, Nq, df, Nf, doff, dc, skip ? dsk : nullptr, dbi, dbd, dsd))) return rc;
    RFE_HIP(c, hipMemcpyAsync(best_idx, dbi, (size_t)Nq * 4, hipMemcpyDeviceToHost, s));
    RFE_HIP(c, hipMemcpyAsync(best_dist, dbd, (size_t)Nq * 4, hipMemcpyDeviceToHost, s));
    RFE_HIP(c, hipMemcpyAsync(second_dist, dsd, (size_t)Nq * 4, hipMemcpyDeviceToHost, s));
    RFE_HIP(c, hipStreamSynchronize(s));
    prof_collect(c);
    return RFE_OK;
}

// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:438-530) for Np map points whose observed descriptors and CSR
// offsets live on the device.  `total` >= offsets[Np] (the number of descriptors, grid size) and `maxn` >= the largest
// observation count (LDS row, <= 8192) come from the caller, who built the lists; a point with more observations than maxn
// rounded up to a power of two is reported as best = -2 instead of computed.
extern "C" int rfe_distinctive_descriptors_dev(rfe_ctx* c, const float* desc, const int32_t* offsets, int Np, int total, int maxn,
                                               int32_t* best, float* median) {
    if (!c) return RFE_ERR_INVALID;
    if (Np < 0 || total < 0 || maxn < 0) return fail(c, RFE_ERR_INVALID, "distinctive_descriptors: negative count");
    if (Np == 0) return RFE_OK;
    if (!offsets || !best || !median || (total > 0 && !desc)) return fail(c, RFE_ERR_INVALID, "distinctive_descriptors: null pointer");
    if (maxn > 8192) return fail(c, RFE_ERR_INVALID, "distinctive_descriptors: more than 8192 observations of one map point");
    RFE_HIP(c, hipSetDevice(c->device));
    int rc = ensure_ws(c, &c->ws_tmp, &c->ws_tmp_bytes, al((size_t)std::max(total, 1) * 4));   // per-descriptor medians
    if (rc) return rc;
    { ProfScope ps(c, "distinctive");
      launch_distinctive(c->stream, desc, offsets, total, Np, std::max(maxn, 1), (float*)c->ws_tmp, best, median); }
    RFE_HIP(c, hipGetLastError());
    return RFE_OK;
}

extern "C" int rfe_distinctive_descriptors(rfe_ctx* c, const float* desc, const int32_t* offsets, int Np, int32_t* best,
                                           float* median) {
    if (!c) return RFE_ERR_INVALID;
    if (Np < 0) return fail(c, RFE_ERR_INVALID, "distinctive_descriptors: negative count");
    if (Np == 0) return RFE_OK;
    if (!offsets || !best || !median) return fail(c, RFE_ERR_INVALID, "distinctive_descriptors: null pointer");
    if (offsets[0] != 0) return fail(c, RFE_ERR_INVALID, "distinctive_descriptors: offsets[0] must be 0");
    int maxn = 0;
    for (int p = 0; p < Np; ++p) {
        const int n = offsets[p + 1] - offsets[p];
        if (n < 0) return fail(c, RFE_ERR_INVALID, "distinctive_descriptors: offsets must be non-decreasing");
        maxn = std::max(maxn, n);
    }
    if (maxn > 8192) return fail(c, RFE_ERR_INVALID, "distinctive_descriptors: more than 8192 observations of one map point");
    const int total = offsets[Np];
    if (total > 0 && !desc) return fail(c, RFE_ERR_INVALID, "distinctive_descriptors: null descriptors");
    RFE_HIP(c, hipSetDevice(c->device));
    const size_t bd = al((size_t)std::max(total, 1) * 1024), bo = al((size_t)(Np + 1) * 4), br = al((size_t)Np * 4);
    int rc = ensure_ws(c, &c->ws_io, &c->ws_io_bytes, bd + bo + 2 * br);
    if (rc) return rc;
    char* w = (char*)c->ws_io;
    float* dd = (float*)w; w += bd; int32_t* doff = (int32_t*)w; w += bo; int32_t* dbest = (int32_t*)w; w += br; float* dmedian = (float*)w;
    hipStream_t s = c->stream;
    if (total > 0) RFE_HIP(c, hipMemcpyAsync(dd, desc, (size_t)total * 1024, hipMemcpyHostToDevice, s));
    RFE_HIP(c, hipMemcpyAsync(doff, offsets, (size_t)(Np + 1) * 4, hipMemcpyHostToDevice, s));
    if ((rc = rfe_distinctive_descriptors_dev(c, dd, doff, Np, total, maxn, dbest, dmedian))) return rc;
    RFE_HIP(c, hipMemcpyAsync(best, dbest, (size_t)Np * 4, hipMemcpyDeviceToHost, s));
    RFE_HIP(c, hipMemcpyAsync(median, dmedian, (size_t)Np * 4, hipMemcpyDeviceToHost, s));
    RFE_HIP(c, hipStreamSynchronize(s));
    prof_collect(c);
    return RFE_OK;
}

// =====================================================================================
// profiling
// =====================================================================================
extern "C" int rfe_profile_enable(rfe_ctx* c, int on) { if (!c) return RFE_ERR_INVALID; c->prof = on != 0; return RFE_OK; }
extern "C" int rfe_profile_filter(rfe_ctx* c, const char* stage) {
    if (!c) return RFE_ERR_INVALID;
    c->prof_filter = stage ? stage : "";
    return RFE_OK;
}
extern "C" int rfe_profile_reset(rfe_ctx* c) {
    if (!c) return RFE_ERR_INVALID;
    (void)hipStreamSynchronize(c->stream);
    prof_collect(c);
    c->stages.clear();
    return RFE_OK;
}
extern "C" int rfe_profile_read(rfe_ctx* c, char* names, size_t names_cap, double* ms, int64_t* calls, int cap) {
    if (!c) return RFE_ERR_INVALID;
    (void)hipStreamSynchronize(c->stream);
    prof_collect(c);
    std::string all;
    int k = 0;
    for (auto& st : c->stages) {
        if (k >= cap) break;
        if (k) all += ";";
        all += st.name; ms[k] = st.ms; calls[k] = st.calls; ++k;
    }
    if (names && names_cap) { strncpy(names, all.c_str(), names_cap - 1); names[names_cap - 1] = 0; }
    return k;
}

// =====================================================================================
// kernel-level test hooks
// =====================================================================================
extern "C" int rfe_k_conv3x3(rfe_ctx* c, const float* in, int B, int H, int W, int Cin, const float* w, const float* bias,
                             int Cout, int relu, int pool, float* out) {
    if (!c) return RFE_ERR_INVALID;
    if ((Cin != 16 && Cin != 32 && Cin != 64 && Cin != 128) || (Cout % 64)) return fail(c, RFE_ERR_INVALID, "k_conv3x3: Cin in {16,32,64,128}, Cout % 64 == 0");
    RFE_HIP(c, hipSetDevice(c->device));
    std::vector<float> packed;
    pack_conv3x3_weights(w, Cin, Cout, packed);
    int rc = ensure_ws(c, &c->ws_tmp, &c->ws_tmp_bytes, al(packed.size() * 4) + al((size_t)Cout * 4));
    if (rc) return rc;
    float* dw = (float*)c->ws_tmp; float* db = (float*)((char*)c->ws_tmp + al(packed.size() * 4));
    RFE_HIP(c, hipMemcpyAsync(dw, packed.data(), packed.size() * 4, hipMemcpyHostToDevice, c->stream));
    RFE_HIP(c, hipMemcpyAsync(db, bias, (size_t)Cout * 4, hipMemcpyHostToDevice, c->stream));
    launch_conv3x3(c->stream, in, B, H, W, Cin, dw, db, Cout, relu != 0, pool != 0, out, 0);
    RFE_HIP(c, hipGetLastError());
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    return RFE_OK;
}

extern "C" int rfe_k_linear(rfe_ctx* c, const float* a, int M, int K, const float* w, const float* bias, int N, int relu,
                            float* out) {
    if (!c) return RFE_ERR_INVALID;
    if (K % 32) return fail(c, RFE_ERR_INVALID, "k_linear: K % 32 == 0 required");
    RFE_HIP(c, hipSetDevice(c->device));
    int rc = ensure_ws(c, &c->ws_tmp, &c->ws_tmp_bytes, al((size_t)N * K * 4) + al((size_t)N * 4));
    if (rc) return rc;
    float* dw = (float*)c->ws_tmp; float* db = (float*)((char*)c->ws_tmp + al((size_t)N * K * 4));
    RFE_HIP(c, hipMemcpyAsync(dw, w, (size_t)N * K * 4, hipMemcpyHostToDevice, c->stream));
    if (bias) RFE_HIP(c, hipMemcpyAsync(db, bias, (size_t)N * 4, hipMemcpyHostToDevice, c->stream));
    GemmArgs g = gemm_plain(a, K, dw, K, bias ? db : nullptr, out, N, M, N, K);
    g.relu = relu;
    launch_gemm_nt(c->stream, g);
    RFE_HIP(c, hipGetLastError());
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    return RFE_OK;
}

extern "C" int rfe_k_scoremap(rfe_ctx* c, const uint8_t* img, int H, int W, int stride, int B, float* scoremap,
                              float* nms, float* descmap) {
    int rc = sp_check(c, H, W, B, 1);
    if (rc) return rc;
    RFE_HIP(c, hipSetDevice(c->device));
    SpBuffers b;
    bool forked;
    if ((rc = sp_forward_maps(c, img, H, W, stride, B, b, true, forked, false, 0, 0.0005f, true))) return rc;   // sp_cnt stays dirty: zeroed before the next forward
    const size_t hw = (size_t)B * (H / 8 * 8) * (W / 8 * 8);   // maps are on the score-map frame: [B, 8*(H/8), 8*(W/8)]
    if (scoremap) RFE_HIP(c, hipMemcpyAsync(scoremap, b.smap, hw * 4, hipMemcpyDeviceToDevice, c->stream));
    if (nms) RFE_HIP(c, hipMemcpyAsync(nms, b.nmap, hw * 4, hipMemcpyDeviceToDevice, c->stream));
    if (descmap) RFE_HIP(c, hipMemcpyAsync(descmap, b.dmap, hw / 64 * 256 * 4, hipMemcpyDeviceToDevice, c->stream));
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    return RFE_OK;
}

// keypoint selection alone on a caller-provided post-NMS map [B,H,W] (device): candidates > thr, the top Kmax by (score descending, pixel
// index ascending) or row-major order, through the forward's own launch_select -- so B selects the form (rank-all up to 4 frames, radix
// select + rank sort above).  Lets the tests drive candidate counts and tie patterns no network output produces.
extern "C" int rfe_k_select(rfe_ctx* c, const float* nms, int B, int H, int W, int Kmax, float thr, int topk_always, int32_t* n, int32_t* kxy,
                            float* score) {
    int rc = sp_check(c, H, W, B, Kmax);
    if (rc) return rc;
    if (!nms || !n || !kxy || !score) return fail(c, RFE_ERR_INVALID, "k_select: null pointer");
    RFE_HIP(c, hipSetDevice(c->device));
    if ((rc = ensure_ws(c, &c->ws_sp, &c->ws_sp_bytes, sp_ws_bytes(B, H, W)))) return rc;
    SpBuffers b;
    sp_carve(c->ws_sp, B, H, W, b);
    launch_select(c->stream, nms, B, H, W, Kmax, thr, b.cand_score, b.cand_idx, n, kxy, score, (int32_t*)b.ss, topk_always != 0, b.sel_keys, b.sel_n);
    RFE_HIP(c, hipGetLastError());
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    return RFE_OK;
}

// the same selection through the latency regime's key form (B <= 4): the map's candidates are appended as 64-bit keys in a scrambled order by a helper
// kernel (one atomic per candidate -- the order sp_tail_lat_kernel's workgroups leave is just as arbitrary), then select_rankall_keys_kernel ranks them
extern "C" int rfe_k_select_keys(rfe_ctx* c, const float* nms, int B, int H, int W, int Kmax, float thr, int topk_always, int32_t* n, int32_t* kxy,
                                 float* score) {
    int rc = sp_check(c, H, W, B, Kmax);
    if (rc) return rc;
    if (!nms || !n || !kxy || !score || B > 4) return fail(c, RFE_ERR_INVALID, "k_select_keys: null pointer or more than four frames");
    RFE_HIP(c, hipSetDevice(c->device));
    if ((rc = ensure_ws(c, &c->ws_sp, &c->ws_sp_bytes, sp_ws_bytes(B, H, W)))) return rc;
    SpBuffers b;
    sp_carve(c->ws_sp, B, H, W, b);
    if (!c->sp_cnt) RFE_HIP(c, hipMalloc((void**)&c->sp_cnt, 8 * sizeof(int32_t)));
    RFE_HIP(c, hipMemsetAsync(c->sp_cnt, 0, 8 * sizeof(int32_t), c->stream));
    launch_keys_from_map(c->stream, nms, B, H * W, thr, (unsigned long long*)b.cand_score, c->sp_cnt);
    launch_select_keys(c->stream, (const unsigned long long*)b.cand_score, c->sp_cnt, B, H, W, Kmax, topk_always != 0, n, kxy, score);
    c->sp_cnt_dirty = false;
    RFE_HIP(c, hipGetLastError());
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    int32_t left[8];
    RFE_HIP(c, hipMemcpy(left, c->sp_cnt, sizeof(left), hipMemcpyDeviceToHost));
    for (int q = 0; q < 8; ++q) if (left[q] != 0) return fail(c, RFE_ERR_HIP, "k_select_keys: the ranking kernel did not leave the candidate counters at zero");
    return RFE_OK;
}

// x + ffn([x | second]) of one LightGlue block with the loaded weights (unfolded W1: `second` is the attention message), through
// the same lg_ffn the forward uses -- so `rows` selects the path: >= 32768 rows take the 128x256 tiles with the LayerNorm + GELU
// fused across ffn.0 / ffn.3, a few thousand rows the 64-row tiles with the stand-alone lg_ln_gelu pass.
extern "C" int rfe_k_lightglue_ffn(rfe_ctx* c, int layer, int cross, const float* x, const float* second, int rows, float* out) {
    int rc = lg_check(c, 1, 4, 4);
    if (rc) return rc;
    if (layer < 0 || layer >= LG_LAYERS || rows <= 0 || !x || !second || !out) return fail(c, RFE_ERR_INVALID, "k_lightglue_ffn: bad argument");
    RFE_HIP(c, hipSetDevice(c->device));
    const int L = 1024, P = (rows + 2 * L - 1) / (2 * L);
    if ((rc = ensure_ws(c, &c->ws_lg, &c->ws_lg_bytes, lg_ws_bytes(P, L)))) return rc;
    LgBuffers b;
    lg_carve(c->ws_lg, P, L, b);
    const LgLayerDev& Lw = c->lg.L[layer];
    RFE_HIP(c, hipMemcpyAsync(out, x, (size_t)rows * 1024, hipMemcpyDeviceToDevice, c->stream));
    if (cross) lg_ffn(c, b, out, second, rows, Lw.cw1, Lw.cb1, Lw.clng, Lw.clnb, Lw.cw2, Lw.cb2);
    else lg_ffn(c, b, out, second, rows, Lw.w1, Lw.b1, Lw.lng, Lw.lnb, Lw.w2, Lw.b2);
    RFE_HIP(c, hipGetLastError());
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    return RFE_OK;
}

extern "C" int rfe_k_attention(rfe_ctx* c, const float* q, const float* k, const float* v, int ld, float* out, int nseq, int Lq, int Lk,
                               const int32_t* qlen, const int32_t* klen, const int32_t* kv_map, const float* rope) {
    if (!c) return RFE_ERR_INVALID;
    if (!q || !k || !v || !out || nseq <= 0 || Lq <= 0 || Lk <= 0 || ld < 256) return fail(c, RFE_ERR_INVALID, "k_attention: bad argument");
    RFE_HIP(c, hipSetDevice(c->device));
    float* part = nullptr;
    const size_t pb = lg_attention_part_bytes(nseq, Lq);
    if (pb) RFE_HIP(c, hipMalloc((void**)&part, pb));
    launch_lg_attention(c->stream, q, k, v, ld, out, nseq, Lq, Lk, qlen, klen, kv_map, part, rope, c->opt_lg_fp16x2);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (part) (void)hipFree(part);
    RFE_HIP(c, e);
    return RFE_OK;
}

extern "C" int rfe_k_set_lightglue_tap(rfe_ctx* c, int pair, float* x0, float* x1, float* scores) {
    if (!c) return RFE_ERR_INVALID;
    if (pair < 0) { c->tap.armed = false; return RFE_OK; }
    c->tap.armed = true; c->tap.pair = pair; c->tap.x0 = x0; c->tap.x1 = x1; c->tap.scores = scores;
    return RFE_OK;
}

// projection + attention of layer `layer`'s self block on caller-provided token rows x [nseq * L, 256] with the rotary table csn [nseq * L, 32] (cos, sin),
// through the forward's own lg_self_qkv_attention -- so nseq * L selects the path (throughput: rotary in gemm.hip's epilogue + lg_attention_dma_kernel;
// one / few pairs: gemm_lat.hip + lg_attention_lat_kernel; shapes neither takes: plain epilogue + rotary on load).  qkv_out [nseq * L, 768], ctx_out [nseq * L, 256];
// *qk_rotated = 1 when qkv_out's q | k columns are rotated.
extern "C" int rfe_k_lightglue_self_attention(rfe_ctx* c, int layer, const float* x, const float* csn, const int32_t* lens, int nseq, int L,
                                              float* qkv_out, float* ctx_out, int32_t* qk_rotated) {
    int rc = lg_check(c, 1, 4, 4);
    if (rc) return rc;
    if (layer < 0 || layer >= LG_LAYERS || nseq <= 0 || L <= 0 || (L % 4) || !x || !csn || !lens) return fail(c, RFE_ERR_INVALID, "k_lightglue_self_attention: bad argument");
    RFE_HIP(c, hipSetDevice(c->device));
    const int P = (nseq + 1) / 2;
    if ((rc = ensure_ws(c, &c->ws_lg, &c->ws_lg_bytes, lg_ws_bytes(P, L)))) return rc;
    LgBuffers b;
    lg_carve(c->ws_lg, P, L, b);
    const bool rot = lg_self_qkv_attention(c, b, c->lg.L[layer], x, csn, lens, nseq, L);
    RFE_HIP(c, hipGetLastError());
    if (qkv_out) RFE_HIP(c, hipMemcpyAsync(qkv_out, b.qkv, (size_t)nseq * L * 768 * 4, hipMemcpyDeviceToDevice, c->stream));
    if (ctx_out) RFE_HIP(c, hipMemcpyAsync(ctx_out, b.ctx, (size_t)nseq * L * 256 * 4, hipMemcpyDeviceToDevice, c->stream));
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    if (qk_rotated) *qk_rotated = rot ? 1 : 0;
    return RFE_OK;
}

extern "C" int rfe_k_lightglue_taps(rfe_ctx* c, const float* k0n, const float* k1n, const float* d0, const float* d1,
                                    int M, int N, float* x0, float* x1, float* scores) {
    int rc = lg_check(c, 1, M, N);
    if (rc) return rc;
    RFE_HIP(c, hipSetDevice(c->device));
    const int L = ((std::max(M, N) + 3) / 4) * 4, cap = std::min(M, N);
    const size_t extra = al((size_t)L * L * 4) + al(64) * 3 + al((size_t)cap * 8) + al((size_t)cap * 4);
    if ((rc = ensure_ws(c, &c->ws_lg, &c->ws_lg_bytes, lg_ws_bytes(1, L, extra)))) return rc;
    LgBuffers b;
    lg_carve(c->ws_lg, 1, L, b, extra);
    char* p = b.extra;
    float* sc = (float*)p; p += al((size_t)L * L * 4);
    int32_t* dm = (int32_t*)p; p += al(64); int32_t* dn = (int32_t*)p; p += al(64); int32_t* dS = (int32_t*)p; p += al(64);
    int32_t* dp = (int32_t*)p; p += al((size_t)cap * 8); float* dms = (float*)p;
    RFE_HIP(c, hipMemcpyAsync(dm, &M, 4, hipMemcpyHostToDevice, c->stream));
    RFE_HIP(c, hipMemcpyAsync(dn, &N, 4, hipMemcpyHostToDevice, c->stream));
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    if ((rc = lg_stage(c, b, k0n, k1n, d0, d1, dm, dn, 1, M, N, L))) return rc;
    if ((rc = lg_forward(c, b, 1, L, 0.1f, cap, dS, dp, dms, scores ? sc : nullptr, false, true))) return rc;
    if (x0) RFE_HIP(c, hipMemcpyAsync(x0, b.x, (size_t)M * 1024, hipMemcpyDeviceToDevice, c->stream));
    if (x1) RFE_HIP(c, hipMemcpyAsync(x1, b.x + (size_t)L * 256, (size_t)N * 1024, hipMemcpyDeviceToDevice, c->stream));
    if (scores) RFE_HIP(c, hipMemcpy2DAsync(scores, (size_t)N * 4, sc, (size_t)L * 4, (size_t)N * 4, M, hipMemcpyDeviceToDevice, c->stream));
    RFE_HIP(c, hipStreamSynchronize(c->stream));
    return RFE_OK;
}
