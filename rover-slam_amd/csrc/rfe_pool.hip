// rfe_pool.hip -- multi-device pool behind the C ABI: configs[3] (a stream of F frames, consecutive pairs matched) for a C++ host
// without Python.  The reference is single device (src/Extractors/superpoint_onnx.cc:19, src/Matchers/lightglue_onnx.cpp:24) and a
// C++ program (src/Tracking.cc:645-651 constructs its extractors), so the sharding of rover-slam_amd/sharding.py is restated here
// in C++: one rfe_ctx + one host worker thread per member; member r extracts frames [first_r, first_r + pairs_r] (ONE overlap frame)
// and matches its pairs_r consecutive pairs -- no inter-device dependency on the data path.  The only exchange is the gather of
// every member's results into ONE root buffer (global frame / pair order) on member 0's device:
//   RCCL transport: grouped ncclSend / ncclRecv over the root's direct xGMI links (librccl is dlopen'ed when a pool is created, so
//                   librover_fe.so itself carries no RCCL dependency); one communicator per member, ncclCommInitAll in this process
//   COPY transport: hipMemcpyPeerAsync / device-to-device copies issued by the root (members that share a device, or no RCCL)
// then one set of device-to-host copies into the caller's arrays.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <shared_mutex>
#include "rfe_internal.h"

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;     // optional
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& why) {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { why = std::string("librccl not found (") + dlerror() + ")"; return false; }
#define RFE_SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(lib, sym)); if (!field) { why = std::string("librccl lacks ") + sym; return false; }
        RFE_SYM(CommInitAll, "ncclCommInitAll") RFE_SYM(CommDestroy, "ncclCommDestroy") RFE_SYM(GroupStart, "ncclGroupStart")
        RFE_SYM(GroupEnd, "ncclGroupEnd") RFE_SYM(Send, "ncclSend") RFE_SYM(Recv, "ncclRecv") RFE_SYM(GetErrorString, "ncclGetErrorString")
#undef RFE_SYM
        CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(lib, "ncclCommAbort"));
        return true;
    }
};

// the seven result arrays of a stream, per frame or per pair; `row` = bytes per frame / pair
struct Field { int per_pair; size_t row; };
enum { F_N, F_S, F_KXY, F_PAIRS, F_MS, F_SCORE, F_DESC, F_COUNT };
inline void field_table(int Kmax, Field* f) {
    f[F_N] = {0, 4}; f[F_S] = {1, 4}; f[F_KXY] = {0, (size_t)Kmax * 8}; f[F_PAIRS] = {1, (size_t)Kmax * 8};
    f[F_MS] = {1, (size_t)Kmax * 4}; f[F_SCORE] = {0, (size_t)Kmax * 4}; f[F_DESC] = {0, (size_t)Kmax * 1024};
}
inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
// byte offsets of the seven arrays in a pack that holds `frames` frames (frames - 1 pairs, at least one pair slot)
inline size_t pack_layout(int frames, int Kmax, size_t* off) {
    Field f[F_COUNT]; field_table(Kmax, f);
    size_t o = 0;
    for (int i = 0; i < F_COUNT; ++i) { off[i] = o; o += al256((size_t)(f[i].per_pair ? std::max(frames - 1, 1) : frames) * f[i].row); }
    return o;
}

struct Job {
    const uint8_t* img; int H, W, stride, F, Kmax; float thr, fthr; bool rccl, with_score, with_desc;
    int phase;   // 0 = upload + extract + match (ends with a stream synchronisation), 1 = RCCL gather.  Two phases so that a member
                 // which fails in phase 0 (out of memory, bad weights) never leaves the others waiting inside a collective
};

struct Member {
    rfe_ctx* ctx = nullptr;
    int device = 0;
    void* frames = nullptr; size_t frames_bytes = 0;     // this member's shard of the input stream (device)
    void* pack = nullptr; size_t pack_bytes = 0;         // this member's results (device), pack_layout
    ncclComm_t comm = nullptr;
    std::thread th;
    int rc = 0;
    std::string err;
};

}  // namespace

struct rfe_pool {
    std::vector<Member> m;
    std::string err;
    RcclApi api;
    bool rccl_up = false;
    std::atomic<bool> rccl_broken{false};                 // a member's gather failed: communicators are torn down, the pool continues on COPY
    std::mutex abort_mu;                                  // taking a member's handle out of Member::comm for its abort (own abort and a peer's abort may meet)
    std::shared_mutex comm_mu;                            // shared: a member is posting its group (uses its comm handle); exclusive: every communicator is being aborted
    std::atomic<int> inject_fail{-1};                     // test hook (rfe_k_pool_inject_gather_failure): this member's next gather fails inside its group
    std::string rccl_why;                                 // why the RCCL transport is unavailable
    void* root = nullptr; size_t root_bytes = 0;          // gathered results in global order on member 0's device
    // job hand-off to the persistent workers
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    uint64_t gen = 0;
    int pending = 0;
    bool quit = false;
    Job job{};
};

static std::string g_pool_error;

static int pfail(rfe_pool* p, int code, const std::string& msg) {
    if (p) p->err = msg; else g_pool_error = msg;
    return code;
}

extern "C" int rfe_pool_shard(int F, int n, int member, int* first_frame, int* frames, int* pairs) {
    // F - 1 consecutive pairs split as evenly as possible, the first (F - 1) % n members take one more; a member extracts its pairs'
    // frames = pairs + 1 (one overlap frame).  For (F - 1) % n == 0 this is rover-slam_amd/sharding.py:shard_frames.  A member without
    // pairs is idle (frames = 0), except member 0 of a one-frame stream (extraction only).
    if (F < 1 || n < 1 || member < 0 || member >= n) return RFE_ERR_INVALID;
    const int P = F - 1, base = P / n, rem = P % n;
    const int own = base + (member < rem ? 1 : 0);
    const int first = member * base + std::min(member, rem);
    if (first_frame) *first_frame = first;
    if (pairs) *pairs = own;
    if (frames) *frames = own > 0 ? own + 1 : (F == 1 && member == 0 ? 1 : 0);
    return RFE_OK;
}

static int grow(Member& mb, void** p, size_t* cur, size_t need) {
    if (*cur >= need) return RFE_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *cur = 0; }
    if (hipMalloc(p, need) != hipSuccess) { mb.err = "hipMalloc of " + std::to_string(need) + " bytes failed"; return RFE_ERR_OOM; }
    *cur = need;
    return RFE_OK;
}

#define POOL_HIP(mb, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (mb).err = std::string(#call) + ": " + hipGetErrorString(e_); return RFE_ERR_HIP; } } while (0)
#define POOL_NCCL(p, mb, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { (mb).err = std::string(#call) + ": " + (p)->api.GetErrorString(r_); return RFE_ERR_HIP; } } while (0)

// what one member does for one call (on its own thread): upload its shard, run the stream entry point, take part in the gather
static int run_member(rfe_pool* p, int r) {
    Member& mb = p->m[r];
    const Job& j = p->job;
    const int n = (int)p->m.size();
    int first, frames, own;
    rfe_pool_shard(j.F, n, r, &first, &frames, &own);
    POOL_HIP(mb, hipSetDevice(mb.device));
    hipStream_t s = mb.ctx->stream;
    size_t off[F_COUNT];
    pack_layout(std::max(frames, 1), j.Kmax, off);
    if (j.phase == 0 && frames > 0) {
        int rc;
        if ((rc = grow(mb, &mb.frames, &mb.frames_bytes, (size_t)frames * j.H * j.W))) return rc;
        if ((rc = grow(mb, &mb.pack, &mb.pack_bytes, pack_layout(frames, j.Kmax, off)))) return rc;
        // frames of the stream are j.stride * j.H bytes apart: one pitched copy of frames * H rows into a tight device image
        POOL_HIP(mb, hipMemcpy2DAsync(mb.frames, (size_t)j.W, j.img + (size_t)first * j.stride * j.H, (size_t)j.stride, (size_t)j.W,
                                      (size_t)frames * j.H, hipMemcpyHostToDevice, s));
        char* pk = (char*)mb.pack;
        rc = rfe_extract_match_stream_dev(mb.ctx, (const uint8_t*)mb.frames, j.H, j.W, j.W, frames, j.Kmax, j.thr, j.fthr, (int32_t*)(pk + off[F_N]),
                                          (int32_t*)(pk + off[F_KXY]), (float*)(pk + off[F_SCORE]), (float*)(pk + off[F_DESC]),
                                          (int32_t*)(pk + off[F_S]), (int32_t*)(pk + off[F_PAIRS]), (float*)(pk + off[F_MS]));
        if (rc) { mb.err = rfe_last_error(mb.ctx); return rc; }
    }
    if (j.phase == 1 && j.rccl) {
        // every member sends the rows it owns to member 0, array by array, in one group; member 0 posts the matching receives at
        // the rows' global positions (its own rows travel through a self send / receive, so that a pool of one exercises RCCL too).
        // Sends and receives between one pair of ranks match in issue order: both sides walk the arrays and the members in the same order.
        Field f[F_COUNT]; field_table(j.Kmax, f);
        size_t goff[F_COUNT];
        pack_layout(j.F, j.Kmax, goff);
        // A failure between GroupStart and GroupEnd must not leave the group open: the peers have posted (or will post) their
        // matching operations and would wait for ever.  The first failure is recorded, the remaining operations of the group are
        // skipped, GroupEnd is ALWAYS called, and on any failure this member aborts its communicator (ncclCommAbort: the peers'
        // pending operations with it fail instead of hanging; the pool then falls back to the COPY transport, see rfe_pool_extract_match_stream).
        // A PARTIAL failure must not strand the healthy members either: they have closed their groups and sit in hipStreamSynchronize on
        // transfer kernels that wait for the failed peer, and the main thread waits for them before it looks at rccl_broken.  So the failing
        // member aborts EVERY member's communicator (exclusive comm_mu: no peer is between GroupStart and GroupEnd, i.e. nobody is using
        // a handle); the peers' kernels then end, their synchronisations return, and the call falls back to COPY.
        ncclResult_t first = ncclSuccess;
        const char* where = "ncclGroupStart";
        {
            std::shared_lock<std::shared_mutex> posting(p->comm_mu);
            if (!mb.comm) { mb.err = "communicator aborted by a failing peer"; return RFE_ERR_HIP; }
            first = p->api.GroupStart();
            const bool opened = first == ncclSuccess;
            auto rows_of = [&](int q, int i, int* first_row) {   // rows of array i that member q contributes, and where they start globally
                int fq, frq, oq;
                rfe_pool_shard(j.F, n, q, &fq, &frq, &oq);
                *first_row = fq;
                if (f[i].per_pair) return oq;
                int last = -1;                                   // the last member with frames also contributes its overlap frame
                for (int t = 0; t < n; ++t) { int a, b, c; rfe_pool_shard(j.F, n, t, &a, &b, &c); if (b > 0) last = t; }
                return frq == 0 ? 0 : (q == last ? frq : oq);
            };
            if (opened && p->inject_fail.load() == r) { p->inject_fail.store(-1); first = ncclInternalError; where = "ncclSend (injected by rfe_k_pool_inject_gather_failure)"; }
            for (int i = 0; i < F_COUNT && first == ncclSuccess; ++i) {
                if ((i == F_SCORE && !j.with_score) || (i == F_DESC && !j.with_desc)) continue;
                int fr;
                const int rows = rows_of(r, i, &fr);
                if (rows > 0 && (first = p->api.Send((char*)mb.pack + off[i], (size_t)rows * f[i].row, ncclUint8, 0, mb.comm, s)) != ncclSuccess) { where = "ncclSend"; break; }
                if (r == 0)
                    for (int q = 0; q < n && first == ncclSuccess; ++q) {
                        int fq;
                        const int rq = rows_of(q, i, &fq);
                        if (rq > 0 && (first = p->api.Recv((char*)p->root + goff[i] + (size_t)fq * f[i].row, (size_t)rq * f[i].row, ncclUint8, q, mb.comm, s)) != ncclSuccess) where = "ncclRecv";
                    }
            }
            if (opened) {
                const ncclResult_t e = p->api.GroupEnd();
                if (first == ncclSuccess && e != ncclSuccess) { first = e; where = "ncclGroupEnd"; }
            }
        }
        if (first != ncclSuccess) {
            mb.err = std::string(where) + ": " + p->api.GetErrorString(first);
            p->rccl_broken.store(true);
            auto abort_member = [&](Member& q) {
                ncclComm_t c;
                { std::lock_guard<std::mutex> g(p->abort_mu); c = q.comm; q.comm = nullptr; }
                if (c) { (void)hipSetDevice(q.device); (void)(p->api.CommAbort ? p->api.CommAbort(c) : p->api.CommDestroy(c)); }
            };
            // OWN communicator first and at once (only this thread posts on that handle, and it has left its group): a peer whose ncclGroupEnd blocks on
            // this member -- first-use connection set-up does -- holds the SHARED lock until this abort releases it; waiting for the exclusive lock
            // before any abort would be a deadlock.  Then, exclusively (no peer between GroupStart and GroupEnd), everybody else's.
            abort_member(mb);
            {
                std::unique_lock<std::shared_mutex> all(p->comm_mu);
                for (auto& q : p->m) abort_member(q);
            }
            (void)hipSetDevice(mb.device);
            return RFE_ERR_HIP;
        }
        if (hipStreamSynchronize(s) != hipSuccess || p->rccl_broken.load()) {   // a peer failed and aborted the communicators under this member's transfers
            (void)hipGetLastError();
            mb.err = "gather aborted by a failing peer";
            return RFE_ERR_HIP;
        }
        return RFE_OK;
    }
    POOL_HIP(mb, hipStreamSynchronize(s));
    return RFE_OK;
}

static void worker(rfe_pool* p, int r) {
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_go.wait(lk, [&] { return p->quit || p->gen != seen; });
            if (p->quit) return;
            seen = p->gen;
        }
        Member& mb = p->m[r];
        mb.err.clear();
        mb.rc = run_member(p, r);
        {
            std::lock_guard<std::mutex> lk(p->mu);
            if (--p->pending == 0) p->cv_done.notify_all();
        }
    }
}

extern "C" const char* rfe_pool_last_error(rfe_pool* p) { return p ? p->err.c_str() : g_pool_error.c_str(); }
extern "C" int rfe_pool_size(rfe_pool* p) { return p ? (int)p->m.size() : 0; }
extern "C" rfe_ctx* rfe_pool_ctx(rfe_pool* p, int member) { return (p && member >= 0 && member < (int)p->m.size()) ? p->m[member].ctx : nullptr; }
extern "C" int rfe_pool_has_rccl(rfe_pool* p) { return p && p->rccl_up ? 1 : 0; }

extern "C" void rfe_pool_destroy(rfe_pool* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->quit = true;
    }
    p->cv_go.notify_all();
    for (auto& mb : p->m) if (mb.th.joinable()) mb.th.join();
    for (auto& mb : p->m) {
        (void)hipSetDevice(mb.device);
        if (mb.comm) (void)p->api.CommDestroy(mb.comm);
        if (mb.frames) (void)hipFree(mb.frames);
        if (mb.pack) (void)hipFree(mb.pack);
        if (mb.ctx) rfe_destroy(mb.ctx);
    }
    if (p->root) { (void)hipSetDevice(p->m.empty() ? 0 : p->m[0].device); (void)hipFree(p->root); }
    if (p->api.lib) dlclose(p->api.lib);
    delete p;
}

extern "C" int rfe_pool_create(const int* devices, int n, rfe_pool** out) {
    if (!out) return pfail(nullptr, RFE_ERR_INVALID, "rfe_pool_create: out is NULL");
    *out = nullptr;
    if (!devices || n < 1 || n > 64) return pfail(nullptr, RFE_ERR_INVALID, "rfe_pool_create: need 1..64 devices");
    rfe_pool* p = new rfe_pool();
    p->m.resize(n);
    for (int r = 0; r < n; ++r) {
        p->m[r].device = devices[r];
        const int rc = rfe_init(devices[r], &p->m[r].ctx);
        if (rc) {
            const std::string why = std::string("rfe_pool_create: member ") + std::to_string(r) + " (device " + std::to_string(devices[r]) + "): " + rfe_last_error(nullptr);
            rfe_pool_destroy(p);
            return pfail(nullptr, rc, why);
        }
    }
    // RCCL communicators: one rank per member, all in this process; needs pairwise distinct devices
    bool distinct = true;
    for (int a = 0; a < n; ++a) for (int b = a + 1; b < n; ++b) distinct &= devices[a] != devices[b];
    if (!distinct) p->rccl_why = "members share a device (RCCL needs one device per rank)";
    else if (p->api.load(p->rccl_why)) {
        std::vector<ncclComm_t> comms(n);
        const ncclResult_t r = p->api.CommInitAll(comms.data(), n, devices);
        if (r == ncclSuccess) { for (int q = 0; q < n; ++q) p->m[q].comm = comms[q]; p->rccl_up = true; }
        else p->rccl_why = std::string("ncclCommInitAll: ") + p->api.GetErrorString(r);
    }
    for (int r = 0; r < n; ++r) p->m[r].th = std::thread(worker, p, r);
    *out = p;
    return RFE_OK;
}

extern "C" int rfe_k_pool_inject_gather_failure(rfe_pool* p, int member) {
    if (!p || member < 0 || member >= (int)p->m.size()) return RFE_ERR_INVALID;
    p->inject_fail.store(member);
    return RFE_OK;
}

extern "C" int rfe_pool_set_weights(rfe_pool* p, int kind, const float* blob, int64_t count) {
    if (!p) return RFE_ERR_INVALID;
    for (size_t r = 0; r < p->m.size(); ++r) {   // members on one device share one device copy (rfe_set_weights' cache)
        const int rc = rfe_set_weights(p->m[r].ctx, kind, blob, count);
        if (rc) return pfail(p, rc, "member " + std::to_string(r) + ": " + rfe_last_error(p->m[r].ctx));
    }
    return RFE_OK;
}

extern "C" int rfe_pool_load_weights(rfe_pool* p, const char* sp_path, const char* lg_path) {
    if (!p) return RFE_ERR_INVALID;
    for (size_t r = 0; r < p->m.size(); ++r) {
        const int rc = rfe_load_weights(p->m[r].ctx, sp_path, lg_path);
        if (rc) return pfail(p, rc, "member " + std::to_string(r) + ": " + rfe_last_error(p->m[r].ctx));
    }
    return RFE_OK;
}

extern "C" int rfe_pool_set_option(rfe_pool* p, int option, int value) {
    if (!p) return RFE_ERR_INVALID;
    for (size_t r = 0; r < p->m.size(); ++r) {
        const int rc = rfe_set_option(p->m[r].ctx, option, value);
        if (rc) return pfail(p, rc, "member " + std::to_string(r) + ": " + rfe_last_error(p->m[r].ctx));
    }
    return RFE_OK;
}

extern "C" int rfe_pool_set_hparams(rfe_pool* p, const rfe_hparams* in) {
    if (!p) return RFE_ERR_INVALID;
    for (size_t r = 0; r < p->m.size(); ++r) {
        const int rc = rfe_set_hparams(p->m[r].ctx, in);
        if (rc) return pfail(p, rc, "member " + std::to_string(r) + ": " + rfe_last_error(p->m[r].ctx));
    }
    return RFE_OK;
}

extern "C" int rfe_pool_extract_match_stream(rfe_pool* p, const uint8_t* img, int H, int W, int stride, int F, int Kmax, float thr,
                                             float filter_thr, int transport, int32_t* n, int32_t* kxy, float* score, float* desc,
                                             int32_t* S, int32_t* pairs, float* ms) {
    if (!p) return RFE_ERR_INVALID;
    if (!img || F < 1 || H < 8 || W < 8 || stride < W || Kmax < 1 || !n || !kxy) return pfail(p, RFE_ERR_INVALID, "pool stream: bad argument");
    if (F > 1 && (!S || !pairs || !ms)) return pfail(p, RFE_ERR_INVALID, "pool stream: null match output");
    if (transport != RFE_POOL_AUTO && transport != RFE_POOL_RCCL && transport != RFE_POOL_COPY) return pfail(p, RFE_ERR_INVALID, "pool stream: unknown transport");
    if (transport == RFE_POOL_RCCL && !p->rccl_up) return pfail(p, RFE_ERR_INVALID, "pool stream: RCCL transport unavailable: " + p->rccl_why);
    bool rccl = transport == RFE_POOL_RCCL || (transport == RFE_POOL_AUTO && p->rccl_up && p->m.size() > 1);
    const int nm = (int)p->m.size();
    // the call works on the members' devices; the caller's current device is restored on every exit path
    struct DeviceGuard { int dev = -1; DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; } ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); } } device_guard;
    Member& root = p->m[0];
    size_t goff[F_COUNT];
    const size_t gbytes = pack_layout(F, Kmax, goff);
    if (hipSetDevice(root.device) != hipSuccess) return pfail(p, RFE_ERR_HIP, "pool stream: hipSetDevice(root)");
    if (p->root_bytes < gbytes) {
        if (p->root) (void)hipFree(p->root);
        p->root = nullptr; p->root_bytes = 0;
        if (hipMalloc(&p->root, gbytes) != hipSuccess) return pfail(p, RFE_ERR_OOM, "pool stream: root buffer of " + std::to_string(gbytes) + " bytes");
        p->root_bytes = gbytes;
    }
    for (int phase = 0; phase < (rccl ? 2 : 1); ++phase) {
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->job = Job{img, H, W, stride, F, Kmax, thr, filter_thr, rccl, score != nullptr, desc != nullptr, phase};
            p->pending = nm;
            ++p->gen;
        }
        p->cv_go.notify_all();
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_done.wait(lk, [&] { return p->pending == 0; });
        }
        if (phase == 1 && p->rccl_broken.load()) {
            // a member could not post its part of the gather: it has closed its group and aborted its communicator, so nobody is left
            // inside a collective.  Tear the communicators down for good and deliver THIS call's results through the COPY transport
            // (phase 0 left every member's rows complete in its own buffer); an explicit RFE_POOL_RCCL request is answered with the error.
            std::string why;
            for (int r = 0; r < nm; ++r) {
                Member& mb = p->m[r];
                if (mb.rc && why.empty()) why = "member " + std::to_string(r) + ": " + mb.err;
                (void)hipSetDevice(mb.device);
                if (mb.comm) { (void)(p->api.CommAbort ? p->api.CommAbort(mb.comm) : p->api.CommDestroy(mb.comm)); mb.comm = nullptr; }
                (void)hipStreamSynchronize(mb.ctx->stream);
                (void)hipGetLastError();
                mb.rc = 0;
            }
            p->rccl_up = false;
            p->rccl_why = "a gather failed (" + why + "); the pool continues on the COPY transport";
            if (transport == RFE_POOL_RCCL) return pfail(p, RFE_ERR_HIP, "pool stream: RCCL gather failed: " + why);
            rccl = false;
            if (hipSetDevice(root.device) != hipSuccess) return pfail(p, RFE_ERR_HIP, "pool stream: hipSetDevice(root)");
            break;
        }
        for (int r = 0; r < nm; ++r)
            if (p->m[r].rc) return pfail(p, p->m[r].rc, "pool stream: member " + std::to_string(r) + " (device " + std::to_string(p->m[r].device) + "): " + p->m[r].err);
    }

    Field f[F_COUNT]; field_table(Kmax, f);
    hipStream_t s = root.ctx->stream;
    auto hipf = [&](hipError_t e, const char* what) { return e == hipSuccess ? RFE_OK : pfail(p, RFE_ERR_HIP, std::string("pool stream: ") + what + ": " + hipGetErrorString(e)); };
    if (!rccl) {
        // COPY transport: the root pulls every member's rows to their global positions (members have synchronised their streams)
        int last = -1;
        for (int t = 0; t < nm; ++t) { int a, b, c; rfe_pool_shard(F, nm, t, &a, &b, &c); if (b > 0) last = t; }
        for (int r = 0; r < nm; ++r) {
            int first, frames, own;
            rfe_pool_shard(F, nm, r, &first, &frames, &own);
            if (frames == 0) continue;
            size_t off[F_COUNT];
            pack_layout(frames, Kmax, off);
            for (int i = 0; i < F_COUNT; ++i) {
                if ((i == F_SCORE && !score) || (i == F_DESC && !desc)) continue;
                const int rows = f[i].per_pair ? own : (r == last ? frames : own);
                if (rows == 0) continue;
                char* dst = (char*)p->root + goff[i] + (size_t)first * f[i].row;
                const char* src = (const char*)p->m[r].pack + off[i];
                const size_t bytes = (size_t)rows * f[i].row;
                int rc = p->m[r].device == root.device ? hipf(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s), "device-to-device copy")
                                                       : hipf(hipMemcpyPeerAsync(dst, root.device, src, p->m[r].device, bytes, s), "peer copy");
                if (rc) return rc;
            }
        }
    }
    void* outs[F_COUNT] = {n, S, kxy, pairs, ms, score, desc};
    for (int i = 0; i < F_COUNT; ++i) {
        const int rows = f[i].per_pair ? F - 1 : F;
        if (!outs[i] || rows == 0) continue;
        int rc = hipf(hipMemcpyAsync(outs[i], (char*)p->root + goff[i], (size_t)rows * f[i].row, hipMemcpyDeviceToHost, s), "device-to-host copy");
        if (rc) return rc;
    }
    return hipf(hipStreamSynchronize(s), "hipStreamSynchronize");
}
