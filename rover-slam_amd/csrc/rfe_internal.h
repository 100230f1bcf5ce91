// rfe_internal.h -- shared declarations of librover_fe.so (HIP, gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <memory>
#include <string>
#include <vector>
#include "../../include/rover_fe.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace rfe {

// Tuning / ablation switches read from the environment exist only in a -DRFE_TUNING build (`make tuning` ->
// librover_fe_tuning.so, used by tools/ for A/B measurements).  The shipped library never consults the environment:
// its numerics and kernel choices depend on its arguments and on rfe_set_option alone.
#ifdef RFE_TUNING
inline const char* tune_env(const char* name) { return getenv(name); }
#else
inline const char* tune_env(const char*) { return nullptr; }
#endif
inline int tune_int(const char* name, int dflt) { const char* e = tune_env(name); return e ? atoi(e) : dflt; }

// ---------------------------------------------------------------- SuperPoint layer table
// names follow the reference's dead libtorch header include/SuperPoint.h:24-41
struct SpLayer { int cin, cout, k; };
static const SpLayer kSpLayers[12] = {
    {1, 64, 3}, {64, 64, 3}, {64, 64, 3}, {64, 64, 3}, {64, 128, 3}, {128, 128, 3},
    {128, 128, 3}, {128, 128, 3}, {128, 256, 3}, {256, 65, 1}, {128, 256, 3}, {256, 256, 1}};
enum { L_1A, L_1B, L_2A, L_2B, L_3A, L_3B, L_4A, L_4B, L_PA, L_PB, L_DA, L_DB };

constexpr int LG_LAYERS = 9;
constexpr int64_t SP_COUNT = 1300865;
constexpr int64_t LG_COUNT = 11321153;

// conv3x3 implicit-GEMM tiling (see sp_conv.hip)
constexpr int CONV_CK = 8;    // input channels per LDS chunk (RFE_CONV_CK=16 selects the larger chunk)
constexpr int CONV_NT = 64;   // output channels per workgroup

// device-side packed SuperPoint weights
struct SpWeightsDev {
    float* conv1a_w = nullptr;           // [9][64]   (kappa-major)
    float* packed[12] = {nullptr};       // 3x3 layers: [Cout/64][Cin/16][144][64]; 1x1: [N][K] as-is
    float* bias[12] = {nullptr};
};

struct LgLayerDev {
    float *wqkv, *bqkv, *wo, *bo, *w1, *b1, *lng, *lnb, *w2, *b2;
    float *cwqk, *cbqk, *cwv, *cbv, *cwo, *cbo, *cw1, *cb1, *clng, *clnb, *cw2, *cb2;
    float *cwqkv, *cbqkv;   // packed [512][256] = [Wqk ; Wv] and [512] bias (device copy made at load time)
    // attention output projection folded into the first FFN Linear (made at load time, see set_lg):
    // ffn1([x | ctx Wo^T + bo]) == [x | ctx] [W1a | W1b Wo]^T + (b1 + W1b bo)
    float *w1f, *b1f, *cw1f, *cb1f;
};
struct LgWeightsDev {
    float* blob = nullptr;  // whole canonical blob on device; pointers below index into it
    float* extra = nullptr; // packed cross-attention projection weights (cwqkv / cbqkv of every layer)
    float* wr;
    LgLayerDev L[LG_LAYERS];
    float *wp, *bp, *wm, *bm;
    // fp16 (hi, lo) planes of every float of `blob` and `extra`, made at load time for RFE_OPT_LG_FP16X2 (gemm_h2.hip):
    // h2 = [hi blob | lo blob | hi extra | lo extra]; the planes of a weight matrix sit at the matrix' own offset inside its buffer
    uint16_t* h2 = nullptr;
    size_t n_blob = 0, n_extra = 0;
};

// ---------------------------------------------------------------- profiling
struct Stage { std::string name; double ms = 0; int64_t calls = 0; };

struct GemmArgs {
    const float* A; int lda;          // rows of A, K-contiguous
    const float* A2; int lda2; int K1; // optional second A source for k >= K1 (concat [x | msg])
    const float* B; int ldb;          // B[n][k], K-contiguous (PyTorch Linear weight layout)
    const float* bias;                // [N] or null
    const float* R; int ldr;          // optional residual added after alpha/relu
    float* C; int ldc;
    int M, N, K;
    float alpha; int relu;
    long long sA, sA2, sB, sC, sR;    // batch strides (grid.z)
    const int* m_valid;               // optional per-batch valid row count (rows >= m_valid skipped)
    int batch;
    // LayerNorm(N) + GELU fused across two GEMMs (LightGlue ffn.0 -> LN -> GELU -> ffn.3): the producer's epilogue writes per-row
    // partial (mean, sum of squared deviations from it) of what it stores -- one pair per (column tile, wave column) = per 128
    // columns: stats_out [M][P][2], P returned by
    // launch_gemm_nt --, the consumer normalises its A operand while staging it: gelu(((a - mean) * rstd) * ln_g[k] + ln_b[k])
    float* stats_out;
    const float* stats_in; int stats_p; const float* ln_g; const float* ln_b;
    // optional fp16 (hi, lo) planes of B, same [N][K] layout and ldb (RFE_OPT_LG_FP16X2): the shapes that would take the 128 x 256
    // throughput tile run gemm_h2.hip's split GEMM on the f16 matrix pipe instead; everything else ignores them
    const uint16_t* Bh; const uint16_t* Bl;
    // kperm != 0: the caller does not need the k-ascending reduction order (LightGlue: tolerance-checked, not bit-exact).  The 128-row
    // tiles then use the 16-byte-swizzled LDS layout with 128-bit fragment reads, which consume k in the order (s, 16 + s) per K tile.
    int kperm;
    // rope_csn != null: LightGlue's rotary encoding applied by the epilogue to output columns rope_c0 <= n < rope_c1 (the k columns of the qkv projection;
    // multiples of 256): (t0, t1) -> (t0 c - t1 s, t1 c + t0 s) on adjacent column pairs, (c, s) = rope_csn[row][(column & 63) / 2].  Plain projections only.
    const float* rope_csn; int rope_c0, rope_c1;
    // xcd != 0 (set by launch_gemm_nt when the grid allows it): workgroups are dealt round-robin over the 8 XCDs in launch order, so the column tiles of ONE row
    // panel -- which read the same A rows -- are decoded onto one XCD (one L2 fetch of the panel instead of gridDim.x of them); the tiles are the same, only which
    // workgroup computes which tile changes
    int xcd;
#ifdef RFE_TUNING
    int abl;   // timing ablations (wrong results): 1 = global loads of the first K tile only, 2 = LDS stores / barriers of the first K tile only, 4 = no epilogue stores
#endif
};

}  // namespace rfe

// the published LightGlue-style export settings (include/rover_fe.h: rfe_hparams defaults)
inline rfe_hparams rfe_default_hparams() { return rfe_hparams{1024, 0.0005f, 4, 4, 0, rfe::LG_LAYERS, 4, 0.1f}; }

struct rfe_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;   // descriptor head of SuperPoint runs here, concurrently with the detector head
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::string err;
    bool has_sp = false, has_lg = false;
    bool opt_lg_fold = true;             // RFE_OPT_LG_FOLD_WO
    bool opt_lg_fp16x2 = false;          // RFE_OPT_LG_FP16X2
    bool opt_host_graph = false;         // RFE_OPT_HOST_GRAPH: the synchronous host entries replay a captured hipGraph of their kernel sequence
    // one instantiated graph per host entry kind (extract / match): `key` names the call shape + every pointer and setting baked into the kernel arguments,
    // `seen` the shape of the last ordinary call (a shape is captured on its SECOND call: the first one allocates workspaces and sets function attributes)
    // RFE_OPT_HOST_GRAPH: a few instantiated graphs per entry (LRU), and a short history of the keys seen lately with their counts -- a shape is captured
    // only once it has come back HOST_GRAPH_REPEATS times, so a caller whose shapes never repeat (per-frame keypoint counts) pays no capture at all
    struct HostGraph {
        struct Slot { std::string key; hipGraphExec_t exec = nullptr; unsigned long long used = 0; };
        struct Seen { std::string key; int count = 0; unsigned long long used = 0; };
        Slot slot[4]; Seen seen[8]; unsigned long long tick = 0;
    };
    HostGraph g_extract, g_match;
    unsigned long long settings_gen = 0; // bumped by everything a captured graph bakes in: weights, hyper-parameters, options
    rfe_hparams hp = rfe_default_hparams();   // graph hyper-parameters (RFEW v2 header / rfe_set_hparams)
    rfe::SpWeightsDev sp;                // views into *sp_hold / *lg_hold
    rfe::LgWeightsDev lg;
    std::shared_ptr<void> sp_hold, lg_hold;   // device copies, shared by every ctx of the process that loaded the same blob on the same device
    // grow-only workspaces
    void* ws_sp = nullptr; size_t ws_sp_bytes = 0;
    void* ws_lg = nullptr; size_t ws_lg_bytes = 0;
    void* ws_io = nullptr; size_t ws_io_bytes = 0;   // staging for host-pointer entry points
    void* h_pin = nullptr; size_t h_pin_bytes = 0;   // pinned host mirror of ws_io for the per-frame host entries (extract / match): one DMA each way
    void* ws_tmp = nullptr; size_t ws_tmp_bytes = 0; // test hooks
    int32_t* sp_cnt = nullptr;           // [4][2] (count, tickets) of the fused detector tail (sp_post.hip: sp_tail_lat_kernel); zero between calls
    bool sp_cnt_dirty = false;           // the tail ran and its ranking kernel was not enqueued behind it (an error in between): zeroed before the next use
    void* ws_st = nullptr; size_t ws_st_bytes = 0;   // stereo stream state: staged views, previous left view's features
    int st_H = 0, st_W = 0, st_K = 0; bool st_have_prev = false; int st_flip = 0;   // st_flip: which of the two state slots holds the previous left view
    // one-shot test tap (rfe_k_set_lightglue_tap): the next LightGlue forward of this ctx, whatever entry point runs it,
    // copies the final token states / log-assignment matrix of one pair to these device buffers
    struct { bool armed = false; int pair = 0; float *x0 = nullptr, *x1 = nullptr, *scores = nullptr; } tap;
    // profiling
    bool prof = false;
    std::string prof_filter;          // non-empty: only this stage records events
    std::vector<rfe::Stage> stages;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending;
    std::vector<hipEvent_t> ev_pool;
};

namespace rfe {

int fail(rfe_ctx* c, int code, const std::string& msg);
#define RFE_HIP(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return rfe::fail((ctx), RFE_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// profiling scope: records two events around a stage when ctx->prof is on
struct ProfScope {
    rfe_ctx* c; int idx; hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st;
    ProfScope(rfe_ctx* ctx, const char* name, hipStream_t on = nullptr);
    ~ProfScope();
};
void prof_collect(rfe_ctx* c);

int ensure_ws(rfe_ctx* c, void** p, size_t* cur, size_t need);

// ---------------------------------------------------------------- kernel launchers
// sp_conv.hip
void pack_conv3x3_weights(const float* w_oihw, int cin, int cout, std::vector<float>& out);
size_t packed_conv3x3_count(int cin, int cout);
int conv_ck();   // input channels per LDS chunk the 3x3 weights are packed for (8; RFE_CONV_CK=16 for A/B)
// img: u8 pixels (NormalizeImage fused) or, img_f32, already normalised float pixels; stride in pixels
void launch_conv1a_u8(hipStream_t s, const void* img, bool img_f32, int stride, int B, int H, int W,
                      const float* w9x64, const float* bias, float* out);
void launch_conv3x3(hipStream_t s, const float* in, int B, int H, int W, int cin,
                    const float* wpacked, const float* bias, int cout, bool relu, bool pool, float* out,
                    int tag = 0);
void launch_conv1ab_fused(hipStream_t s, const void* img, bool img_f32, int stride, int B, int H, int W, const float* w1a,
                          const float* b1a, const float* wp, const float* bias, float* out, long long frame_step = 0 /*pixels between frames; 0 = stride * H*/);
// gemm.hip
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the CURRENT device's copy of a kernel: set it once per (kernel, device) -- pools run
// one ctx per device.  `done` is the caller's per-kernel flag array (static bool [64]); races only repeat the idempotent call.
inline void ensure_dynamic_lds(const void* kernel, int bytes, bool* done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !done[dev]) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (dev >= 0 && dev < 64) done[dev] = true;
    }
}
bool launch_gemm_ln_ws(hipStream_t s, const GemmArgs& g);   // gemm_ws.hip: the LayerNorm-fused consumer with producer / math waves; false = shape not served
int launch_gemm_h2(hipStream_t s, const GemmArgs& g);   // gemm_h2.hip: split GEMM on the f16 matrix pipe (GemmArgs::Bh / Bl), called by launch_gemm_nt
void launch_split_f16(hipStream_t s, const float* x, uint16_t* hi, uint16_t* lo, size_t n);
bool gemm_nt_rope_ok(const GemmArgs& g);                  // may launch_gemm_nt carry GemmArgs::rope_csn for this shape? (set rope_c0 / rope_c1 first)
int launch_gemm_nt(hipStream_t s, const GemmArgs& g);   // returns the number of partial-statistics pairs per row it wrote (0 without stats_out)
// The throughput tiles (128-row) take a problem when they give every CU a workgroup; everything smaller is the LATENCY regime (one or a
// few pairs per call -- what the reference itself runs): gemm_lat.hip / lg_attention_lat.hip serve it, gemm.hip's 64-row tiles what they refuse.
inline bool gemm_latency_regime(const GemmArgs& g) {
    const long long b = g.batch > 0 ? g.batch : 1;
    auto tiles = [&](int bm, int bn) { return (long long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * b; };
    const bool big = (g.N % 256 == 0 && tiles(128, 256) >= 256) || tiles(128, 128) >= 256 || g.M > 8192;
    static const bool on = tune_int("RFE_LAT", 1) != 0;   // tuning build: RFE_LAT=0 = the round-3 latency path
    return on && !big;
}
// gemm_lat.hip: false = shape not served (nothing launched).  rope_csn != null: rotary epilogue on output columns < rope_cols (qkv).
bool launch_gemm_lat(hipStream_t s, const GemmArgs& g, const float* rope_csn, int rope_cols);
// onnx_load.hip (host only): an ONNX graph file -> canonical weight blob of `kind` + that kind's fields of *hp; false + reason when the file cannot be
// read, a tensor cannot be placed, or a hyper-parameter cannot be read from the graph (nothing is guessed)
bool onnx_convert(const std::string& path, int kind, std::vector<float>& blob, rfe_hparams* hp, std::string& err);
// ffn2_lat.hip: ffn.3 of a one- / few-pair forward with LayerNorm(512) + GELU of its activation fused in (h = ffn.0's raw output); false = not served
bool launch_ffn2_ln_lat(hipStream_t s, const float* h, const float* w2, const float* b2, const float* ln_g, const float* ln_b, const float* R, int ldr,
                        float* C, int ldc, int M);
// lg_attention_lat.hip: one-/few-pair attention without rotary (q, k rotated by the projection); false = shape not served.
bool launch_lg_attention_lat(hipStream_t s, const float* q, const float* k, const float* v, int ld, float* out, int nseq, int Lq, int Lk,
                             const int* qlen, const int* klen, const int* kv_map, bool h2 = false /*RFE_OPT_LG_FP16X2: split products*/);
// sp_post.hip
void launch_softmax65_d2s(hipStream_t s, const float* logits, int ld, int B, int Hc, int Wc, float* score);
constexpr int NMS_MAX_RADIUS = 8;
void launch_nms(hipStream_t s, const float* score, int B, int H, int W, int radius /*1..NMS_MAX_RADIUS*/, int border, float* tmp_ss,
                uint8_t* tmp_mask, uint8_t* tmp_supp, float* out);
void launch_select(hipStream_t s, const float* nms, int B, int H, int W, int Kmax, float thr,
                   float* cand_score, int32_t* cand_idx, int32_t* n_out, int32_t* kxy, float* score,
                   int32_t* chunk_cnt /*B * ceil(H*W/4096) ints of scratch*/, bool topk_always /*rfe_hparams::sp_topk_always*/,
                   unsigned long long* sel_keys /*[B,Kmax] scratch*/, int32_t* sel_n /*[B] scratch*/);
// the detector tail of one to four frames in one launch (softmax65 + depth-to-space + simple_nms(4) + border + threshold -> unordered 64-bit candidate keys)
// and the ranking that consumes it; cand_cnt = two ints per frame, zero between calls (the ranking kernel resets them).  false = not served.
bool launch_sp_tail_lat(hipStream_t s, const float* logits, int B, int Hc, int Wc, int radius, int border, float thr, unsigned long long* cand_keys /*[B, 64 Hc Wc]*/,
                        int32_t* cand_cnt /*[B, 2]*/, float* smap_out /*optional [B, 8 Hc, 8 Wc]*/, float* nmap_out /*optional*/);
void launch_select_keys(hipStream_t s, const unsigned long long* cand_keys, int32_t* cand_cnt, int B, int H, int W, int Kmax, bool topk_always,
                        int32_t* n_out, int32_t* kxy, float* score);
void launch_keys_from_map(hipStream_t s, const float* nms, int B, int HW, float thr, unsigned long long* cand_keys, int32_t* cand_cnt);   // test hook
void launch_descmap_norm(hipStream_t s, float* dmap, int64_t cells);
void launch_desc_sample(hipStream_t s, const float* dmap, int B, int Hc, int Wc, int H, int W,
                        const int32_t* n, const int32_t* kxy, int Kmax, float* desc, uint8_t* desc_bin /*optional u8 [B,Kmax,256] = desc > 0*/);
// lg_kernels.hip
void launch_lg_posenc(hipStream_t s, const float* kn, const float* wr, int rows, float* csn /*[rows,32] (cos, sin) pairs*/);
void launch_lg_attention(hipStream_t s, const float* q, const float* k, const float* v, int ld /*row stride of q,k,v*/,
                         float* out, int nseq, int Lq, int Lk, const int* qlen, const int* klen,
                         const int* kv_map /*seq -> kv seq index, or null = identity*/,
                         float* part /*lg_attention_part_bytes(nseq, Lq) of scratch for the split-key variant, or null*/,
                         const float* rope_csn = nullptr /*self blocks: rotary table [nseq*Lq, 32] of (cos, sin) pairs, applied to q and k on load*/,
                         bool fp16x2 = false /*RFE_OPT_LG_FP16X2: problems of >= 32 768 query rows take lg_attention_h2.hip's split products on the f16 matrix pipe*/,
                         bool k_roped = false /*with rope_csn: k is already rotated (gemm.hip's rotary epilogue), only q is rotated on load*/);
void launch_lg_attention_h2(hipStream_t s, const float* q, const float* k, const float* v, int ld, float* out, int nseq, int Lq, int Lk,
                            const int* qlen, const int* klen, const int* kv_map, const float* rope_csn);   // lg_attention_h2.hip
size_t lg_attention_part_bytes(int nseq, int Lq);
void launch_lg_ln_gelu(hipStream_t s, float* h, const float* g, const float* b, int64_t rows);
void launch_lg_assign(hipStream_t s, const float* sim, const float* z0, const float* z1, int P, int L,
                      int cap, const int* m, const int* n, float thr, float* scores_opt, float* rowlse,
                      float* collse, int32_t* a0, float* mx0, int32_t* a1, int32_t* S, int32_t* pairs,
                      float* ms, int scores_pair /*>= 0: scores_opt is [L,L] and receives that pair only; < 0: every pair into [P,L,L]*/,
                      const float* x, const float* wm, const float* bm, float* z /*few-pair shapes: the matchability head rides in the row log-sum-exp launch*/);
bool lg_assign_few_pairs(int P, int L);   // true: launch_lg_assign computes the matchability itself (launch_lg_matchability must not be called)
void launch_lg_frame_prologue(hipStream_t s, const int32_t* kxy, const float* desc, const float* wr, const int32_t* nkp, int B, int L, int rows, int cols,
                              float* kn, float* csn, float* x, int32_t* lens, int32_t* kvmap);
void launch_lg_matchability(hipStream_t s, const float* x, const float* w, const float* b, int64_t rows, float* z);
void launch_copy_f32(hipStream_t s, const float* src, float* dst, int64_t n);
void launch_normalize_kpts(hipStream_t s, const int32_t* kxy, int64_t n, int rows, int cols, float* out);

// stereo.hip
void launch_stereo_match(hipStream_t s, const uint8_t* imgL, const uint8_t* imgR, int H, int W, int stride,
                         const float* kL, int N, const float* kR, int Nr, const float* dL, const float* dR, float mb,
                         float mbf, float* uRight, float* depth, int32_t* sadv);

void launch_stereo_match_counts(hipStream_t s, const uint8_t* imgL, const uint8_t* imgR, int H, int W, int stride,
                                const int32_t* kL, const int32_t* kR, int Kmax, const int32_t* counts, const float* dL,
                                const float* dR, float mb, float mbf, float* uRight, float* depth, int32_t* sadv);

void launch_l2_matrix(hipStream_t s, const float* a, int M, const float* b, int N, float* out);
void launch_binarize(hipStream_t s, const float* d, int64_t rows, uint8_t* out);
void launch_search_candidates(hipStream_t s, const float* q, int Nq, const float* f, int Nf, const int32_t* offsets, const int32_t* cand,
                              const uint8_t* skip, int32_t* best_idx, float* best_dist, float* second_dist);
void launch_distinctive(hipStream_t s, const float* desc, const int32_t* offsets, int total /*>= offsets[Np]*/, int Np, int maxn,
                        float* med /*[total] scratch*/, int32_t* best, float* median);

}  // namespace rfe
