/*
 * rfe_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the arithmetic the reference delegates to ONNXRuntime:
 *   superpoint.onnx      (reference call site src/Extractors/superpoint_onnx.cc:133-136)
 *   lightglue_sim.onnx   (reference call site src/Matchers/lightglue_onnx.cpp:210-214)
 * plus the host-side pre/post-processing around those calls
 *   NormalizeImage       src/Matchers/transform.cpp:3-17
 *   NormalizeKeypoints   src/Matchers/transform.cpp:19-32
 *   Matcher_PostProcess_fused  src/Matchers/lightglue_onnx.cpp:396-482
 *
 * PARITY UNPINNED: both .onnx blobs are absent from the reference checkout
 * (.MISSING_LARGE_BLOBS:4-5), onnxruntime 1.16.3 is not installed, and the reference has no
 * tests or golden vectors for this path.  The network arithmetic below restates the published
 * SuperPoint / LightGlue architectures (layer list corroborated by include/SuperPoint.h:24-41)
 * and is cross-checked in-container against the independent `transformers` 5.15 modelling code
 * (tools/gen_golden.py -> tests/golden/).  See DESIGN.md "Oracle".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 *
 * Canonical arithmetic (the HIP kernels reproduce these orders bit-for-bit for SuperPoint):
 *   conv / linear : acc = bias; for kappa ascending (kappa = ci*9 + ky*3 + kx, i.e. the memory
 *                   order of a PyTorch OIHW weight row): acc = fmaf(in, w, acc); zero padding is
 *                   multiplied, not skipped.
 *   softmax65     : m = max; e_c = rfe_expf(l_c - m); s = e_0 + e_1 + ... (index order); p = e/s
 *   l2norm256     : lane partial p[l] = fma-chain of x[4l..4l+3]^2, then xor-butterfly
 *                   (offsets 32,16,8,4,2,1): p[l] += p[l^off]
 */
#ifndef RFE_ORACLE_H
#define RFE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- weight blob geometry (floats) ---- */
int64_t rfo_sp_weight_count(void);           /* 1 300 865 */
int64_t rfo_lg_weight_count(void);
int64_t rfo_sp_layer_offset(int layer, int want_bias); /* offset of W (OIHW) or bias of layer 0..11 */

/* ---- primitive restatements (kernel-level KATs) ---- */
float rfo_expf(float x);
/* NHWC conv3x3 pad 1 (+bias, optional ReLU, optional fused 2x2/2 max-pool). w is OIHW. */
void rfo_conv3x3(const float* in, int H, int W, int Cin, const float* w, const float* bias,
                 int Cout, int relu, int pool, float* out);
/* out[M,N] = a[M,K] . w[N,K]^T + bias (bias may be NULL -> 0) */
void rfo_linear(const float* a, int M, int K, const float* w, const float* bias, int N, float* out);
void rfo_softmax65_d2s(const float* logits /*[Hc*Wc,65]*/, int Hc, int Wc, float* score /*[8Hc,8Wc]*/);
void rfo_nms(const float* score, int H, int W, int radius, float* out);
float rfo_sumsq256(const float* x);
void rfo_l2norm256(const float* x, float* y);

/* ---- SuperPoint end to end (one frame) ----
 * img: u8 [H,W] (H,W multiples of 8).  Outputs padded to Kmax.  Optional debug taps may be NULL.
 * Returns number of keypoints n (<= Kmax).  kxy = (x,y) pairs. */
int rfo_superpoint(const float* weights, const uint8_t* img, int H, int W, int Kmax, float thr,
                   int nms_radius, int border, int32_t* kxy, float* score, float* desc,
                   float* dbg_scoremap /*[H,W] pre-NMS*/, float* dbg_nms /*[H,W] post-NMS,border*/,
                   float* dbg_descmap /*[Hc,Wc,256] normalised*/, float* dbg_feat /*[Hc,Wc,128] conv4b*/);

/* The same with the selection rule of exports that apply top-k unconditionally (torch.topk(scores, min(k, n)): the TopK node sits
 * behind a Min in the graph): topk_always != 0 orders the keypoints by (score descending, pixel index ascending) also when no more
 * than Kmax candidates pass the threshold; 0 = the published top_k_keypoints (row-major order in that case) = rfo_superpoint. */
int rfo_superpoint_ex(const float* weights, const uint8_t* img, int H, int W, int Kmax, float thr,
                      int nms_radius, int border, int topk_always, int32_t* kxy, float* score, float* desc,
                      float* dbg_scoremap, float* dbg_nms, float* dbg_descmap, float* dbg_feat);

/* The same from an already normalised float image [H,W] (the reference's Extractor_Inference takes a CV_32F cv::Mat,
 * superpoint_onnx.cc:88-118); rfo_superpoint_ex is NormalizeImage + this. */
int rfo_superpoint_f32(const float* weights, const float* img, int H, int W, int Kmax, float thr,
                       int nms_radius, int border, int topk_always, int32_t* kxy, float* score, float* desc,
                       float* dbg_scoremap, float* dbg_nms, float* dbg_descmap, float* dbg_feat);

/* ---- LightGlue end to end (one pair) ----
 * k0n/k1n: normalised keypoints [M,2]/[N,2]; d0/d1: [M,256]/[N,256].
 * pairs: [min(M,N),2] (i,j) ascending i; ms: scores.  Returns S.
 * Optional taps: dbg_x0/x1 = final token states [M,256]/[N,256]; dbg_scores = [M,N] log-assignment. */
int rfo_lightglue(const float* weights, const float* k0n, const float* k1n, const float* d0,
                  const float* d1, int M, int N, float filter_thr, int32_t* pairs, float* ms,
                  float* dbg_x0, float* dbg_x1, float* dbg_scores);

/* host-side glue restatements */
void rfo_normalize_keypoints(const float* kxy, int n, int h, int w, float* out); /* transform.cpp:19-32 */
int  rfo_postprocess_fused(const int32_t* pairs, const float* ms, int S, float match_thresh,
                           int32_t* vnMatches12, int M);                       /* lightglue_onnx.cpp:437-453 */

/* sparse stereo matching, Frame::ComputeStereoMatches (src/Frame.cc:1159-1446), nLevels == 1 */
void rfo_stereo_match(const uint8_t* imgL, const uint8_t* imgR, int H, int W, const float* kL, int N,
                      const float* kR, int Nr, const float* dL, const float* dR, float mb, float mbf,
                      float* uRight, float* depth);

/* classic-search descriptor arithmetic: SPmatcher.cc:1218-1248, MapPoint.cc:438-530 */
void rfo_search_candidates(const float* q, int Nq, const float* f, const int32_t* offsets, const int32_t* cand,
                           const uint8_t* skip, int32_t* best_idx, float* best_dist, float* second_dist);
void rfo_distinctive_descriptors(const float* desc, const int32_t* offsets, int Np, int32_t* best, float* median);

/* OpenMP thread count of the loops above (oracle.py sets it to the CPUs the process may actually use) */
void rfo_set_num_threads(int n);
int rfo_get_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
