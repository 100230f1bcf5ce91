/*
 * rover_fe.h -- C ABI of librover_fe.so: MI355X-native SuperPoint extractor + LightGlue matcher.
 *
 * This is the drop-in boundary for Rover-SLAM's learned front end.  Each entry point names the
 * reference interface it replaces (paths relative to the reference checkout):
 *
 *   rfe_init / rfe_destroy        SuperPointOnnxRunner::InitOrtEnv   src/Extractors/superpoint_onnx.cc:4-66
 *                                 LightGlueDecoupleOnnxRunner::InitOrtEnv  src/Matchers/lightglue_onnx.cpp:4-98
 *   rfe_load_weights              the two model paths: src/Extractors/SPextractor.cc:93 (superpoint.onnx),
 *                                 src/Matchers/lightglue_onnx.cpp:38 (lightglue_sim.onnx)
 *   rfe_extract_u8                NormalizeImage (src/Matchers/transform.cpp:3-17) +
 *                                 SuperPointOnnxRunner::Extractor_Inference (superpoint_onnx.cc:88-162,
 *                                 Session::Run at :133-136) + the tensor unpacking half of
 *                                 Extractor_PostProcess (superpoint_onnx.cc:165-255)
 *   rfe_match                     LightGlueDecoupleOnnxRunner::Matcher_Inference
 *                                 (lightglue_onnx.cpp:162-240, Session::Run at :210-214); inputs are the
 *                                 already normalised keypoints of Matcher_PreProcess (:140-159)
 *   rfe_match_fused               rfe_match + Matcher_PostProcess_fused (lightglue_onnx.cpp:396-482):
 *                                 fills vnMatches12 exactly like SPmatcher::MatchingPoints_onnx
 *                                 (src/Matchers/SPmatcher.cc:359-542)
 *   rfe_extract_match_stream      batched-frames mode (BASELINE config 4): extract B frames and match
 *                                 frame i with frame i+1, everything device resident
 *
 * Conventions: plain pointers and sizes only.  Functions return RFE_OK (0) or a negative
 * rfe_status; nothing throws across the boundary; rfe_last_error() gives the message
 * (the reference prints to std::cerr and returns EXIT_FAILURE, superpoint_onnx.cc:59-65).
 * A ctx is single-caller (not re-entrant); several ctxs may be used concurrently from different
 * threads (the reference gives each SPextractor / SPmatcher its own ORT session,
 * src/Tracking.cc:645-651, :70).  "_dev" variants take DEVICE pointers and are asynchronous on
 * the ctx stream; the plain variants take HOST pointers and return after the results are copied.
 *
 * There is no CPU fallback: every entry point fails with RFE_ERR_NO_DEVICE if no gfx950 GPU
 * can be opened.
 */
#ifndef ROVER_FE_H
#define ROVER_FE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rfe_ctx rfe_ctx;

typedef enum {
    RFE_OK = 0,
    RFE_ERR_INVALID = -1,    /* bad argument */
    RFE_ERR_NO_DEVICE = -2,  /* no HIP device / wrong arch */
    RFE_ERR_HIP = -3,        /* a HIP runtime call failed */
    RFE_ERR_IO = -4,         /* weight file problem */
    RFE_ERR_NO_WEIGHTS = -5, /* model used before weights were set */
    RFE_ERR_OOM = -6
} rfe_status;

#define RFE_DESC_DIM 256
#define RFE_KIND_SUPERPOINT 1
#define RFE_KIND_LIGHTGLUE 2

/* ---- lifetime ---- */
int rfe_init(int device, rfe_ctx** out);
void rfe_destroy(rfe_ctx* ctx);
const char* rfe_last_error(rfe_ctx* ctx); /* ctx may be NULL: last error of a failed rfe_init */
const char* rfe_version(void);

/* ---- weights: ONNX graph files, RFEW container files, or host blobs in the canonical layout ----
 * rfe_load_weights takes, per path, EITHER the reference's own model file -- onnxmodel/superpoint.onnx (src/Extractors/SPextractor.cc:92-94) /
 * onnxmodel/lightglue_sim.onnx (src/Matchers/lightglue_onnx.cpp:38): the initializers are re-packed into the canonical layout and the graph's
 * baked-in hyper-parameters (see rfe_hparams below) are read from its nodes, in C++ (rover-slam_amd/csrc/onnx_load.hip; no Python step, no
 * protobuf / onnx library); a graph whose constants cannot be read, or whose tensors cannot be placed, is refused with the reason
 * (RFE_ERR_IO, rfe_last_error) -- OR an RFEW container written by rover-slam_amd/onnx_weights.py / weights.py; the file's first four bytes
 * decide.  rfe_load_onnx insists on ONNX. */
int rfe_load_weights(rfe_ctx* ctx, const char* sp_path, const char* lg_path); /* either may be NULL */
int rfe_load_onnx(rfe_ctx* ctx, const char* sp_path, const char* lg_path);    /* either may be NULL */
int rfe_set_weights(rfe_ctx* ctx, int kind, const float* blob, int64_t count);
int64_t rfe_weight_count(int kind);
/* Read-only weights are shared inside a process: every ctx that loads the same blob on the same device uses one device
 * copy (the reference keeps 2-3 extractors and 3 matchers per process, each with a private session: src/Tracking.cc:645-651,
 * :70, src/LocalMapping.cc:45, src/LoopClosing.cc:46).  rfe_weights_id: opaque id of the copy a ctx uses (0 = none);
 * equal ids = shared copy. */
uint64_t rfe_weights_id(rfe_ctx* ctx, int kind);

/* ---- graph hyper-parameters (RFEW version 2; per ctx) ----
 * The reference's two model files carry constants that never appear in its C++: the extractor takes K from the SHAPE of the
 * `keypoints` output (src/Extractors/superpoint_onnx.cc:169-181) because max_num_keypoints, the detection threshold, the NMS
 * radius and the border width are baked into onnxmodel/superpoint.onnx at export time (src/Extractors/SPextractor.cc:92-94),
 * and the matcher reads whatever matches0 / mscores0 hold (src/Matchers/lightglue_onnx.cpp:404-409) because depth, heads and the
 * in-graph filter threshold are baked into onnxmodel/lightglue_sim.onnx (lightglue_onnx.cpp:38).  rover-slam_amd/onnx_weights.py
 * reads them from the graphs and writes them into the RFEW v2 header; rfe_load_weights applies them to the ctx.  Hyper-parameters belong
 * to a weight set: version-1 files and rfe_set_weights (bare blobs) carry none and RESET the loaded kind's values to the defaults below
 * (so a v2 load followed by rfe_set_weights does not keep the earlier file's radius / border / top-k rule); call rfe_set_hparams AFTER
 * the weights to state other values.
 *   sp_max_keypoints / sp_detection_threshold / lg_filter_threshold are what a drop-in caller passes as Kmax / thr / filter_thr
 *     (the C++ shims do exactly that); the entry points themselves keep taking them as arguments.
 *   sp_nms_radius (1..8) and sp_remove_borders (0..64) act inside rfe_extract_* (simple_nms window 2r+1, border set to -1).
 *   sp_topk_always: 0 = the published top_k_keypoints (score-descending order only when more than Kmax candidates pass, row-major
 *     otherwise); 1 = exports that run torch.topk(scores, min(k, n)) unconditionally (TopK behind a Min in the graph): keypoints
 *     always ordered by (score descending, pixel index ascending).
 *   lg_layers / lg_heads describe the file; the kernels are built for 9 layers of 4 heads x 64 and rfe_set_hparams /
 *     rfe_load_weights refuse anything else (RFE_ERR_INVALID / RFE_ERR_IO). */
typedef struct rfe_hparams {
    int32_t sp_max_keypoints;       /* default 1024 */
    float sp_detection_threshold;   /* default 0.0005 */
    int32_t sp_nms_radius;          /* default 4 */
    int32_t sp_remove_borders;      /* default 4 */
    int32_t sp_topk_always;         /* default 0 */
    int32_t lg_layers;              /* 9 */
    int32_t lg_heads;               /* 4 */
    float lg_filter_threshold;      /* default 0.1 */
} rfe_hparams;
int rfe_get_hparams(rfe_ctx* ctx, rfe_hparams* out);
int rfe_set_hparams(rfe_ctx* ctx, const rfe_hparams* in);

/* ---- options (per ctx; the library never reads the environment) ----
 * RFE_OPT_LG_FOLD_WO (default 1): every LightGlue attention output projection (Wo, bo) is multiplied into the message
 *   half of the following ffn.0 Linear at load time (W1 [x | ctx Wo^T + bo] + b1 = [W1a | W1b Wo] [x | ctx] + (b1 + W1b bo),
 *   formed in double precision): 18 fewer GEMM launches per forward (+4 % throughput).  Mathematically identical; the
 *   message is simply never rounded to fp32.  0 keeps the graph of lightglue_sim.onnx node for node.  Measured over 40
 *   cases at K = 1024 (profiles/r02_lg_tolerance.md): match lists identical either way; match scores move by <= 3.2e-4
 *   (folded) / <= 2.0e-4 (unfolded) against the fp32 CPU oracle, which itself sits up to 2.6e-4 from a float64 evaluation
 *   of the same graph (HIP vs float64: 2.4e-4 / 1.8e-4) -- the stated tolerance is 5e-4 for both.
 * RFE_OPT_LG_FP16X2 (default 0 = every LightGlue matrix product on the fp32 matrix instructions): 1 = the Linears (every call: the throughput
 *   tiles of batched calls, gemm_h2.hip, and the one- / few-pair latency tiles, gemm_lat.hip) AND the fused attention (batched calls of
 *   >= 32 768 token rows: lg_attention_h2.hip; one / few pairs: the split form of lg_attention_lat.hip) run as SPLIT products on the f16 matrix pipe --
 *   every fp32 operand as fp16 hi + fp16 lo (22 of 24 significand bits), three of the four cross products, fp32 accumulation, softmax
 *   and LayerNorm / GELU in fp32 as before (rover-slam_amd/csrc/gemm_h2.hip, lg_attention_h2.hip).  Stand-alone the Linears are
 *   2.2-2.8x and the attention 2.4-2.6x faster than the fp32 kernels; error against float64: Linears rms 3.2e-8 of sum|a||b| (fp32 fmaf
 *   chain: 2.8e-8), attention context 3.7e-6 (fp32 kernels: 4.8e-6) (profiles/r03_ab_notes.md, tests/test_gpu_attention.py); end to end
 *   over the 40-case study the match lists are identical and the scores sit 1.6e-4 (Wo folded) / 2.6e-4 (unfolded) from float64, inside
 *   the spread of the fp32 evaluations of the same graph (1.8e-4 .. 2.6e-4, profiles/r03_lg_tolerance.md).  Valid while activations
 *   stay below fp16's 65504 in magnitude (LightGlue's are O(1..100)); past it the attention's operands saturate (finite results), the
 *   Linears' inputs too (MODE.FP16_OVFL).  One pair per call: 2.43 -> 2.06 ms incl. both extractions, a stereo frame 2.59 -> 2.25 ms
 *   (bench.py `latency.resident_fp16x2`).  The assignment and all of SuperPoint stay fp32; bench.py reports the throughput step as
 *   `variants.fp16x2`, never as the headline. */
#define RFE_OPT_LG_FOLD_WO 1
#define RFE_OPT_LG_FP16X2 2
/* RFE_OPT_HOST_GRAPH (default 0): the synchronous host entries (rfe_extract_u8 / _bin / _f32, rfe_match / rfe_match_fused) submit their kernel sequence as ONE
 *   replayed hipGraph per call shape instead of 17 .. 100 stream launches (captured on the THIRD call of a shape, four shapes kept per entry, least recently
 *   used replaced; weights, hyper-parameters, options, workspaces and shape are part of the key; profiling falls back).  For rfe_match the shape holds Mmax
 *   and Nmax, so the option only helps FIXED-CAPACITY callers (keypoint counts that saturate, or are padded to, a capacity): a caller whose counts differ on
 *   every frame never repeats a shape, is never captured and gets ordinary launches.  Results are identical.  The device is busy for 98 % of a one-pair call either way, so the median
 *   does not move; the option is there for hosts whose threads get preempted inside the enqueue burst (tail latency).  Measured: profiles/r05_ab_notes.md. */
#define RFE_OPT_HOST_GRAPH 3
int rfe_set_option(rfe_ctx* ctx, int option, int value);
int rfe_get_option(rfe_ctx* ctx, int option, int* value);

/* ---- stream / sync / device memory helpers (so a pure-C caller needs no HIP headers) ---- */
int rfe_set_stream(rfe_ctx* ctx, void* hip_stream); /* NULL -> ctx's own stream */
int rfe_synchronize(rfe_ctx* ctx);
int rfe_malloc(rfe_ctx* ctx, size_t bytes, void** dev_ptr);
int rfe_free(rfe_ctx* ctx, void* dev_ptr);
/* Pinned (page-locked, portable) host memory.  Optional: every host entry point takes ANY host pointer.  When the descriptor output of
 * rfe_extract_u8 / rfe_extract_f32 / rfe_extract_u8_bin lies inside a block from rfe_host_malloc, the K x 256 floats are DMA'd straight into it
 * (otherwise they are staged through the ctx's pinned block and copied on the host: ~1 MB per 1024-keypoint frame).  The C++ runner shim
 * keeps its output tensors in such blocks.  Not tied to a ctx; rfe_host_free(NULL) is a no-op. */
int rfe_host_malloc(size_t bytes, void** host_ptr);
void rfe_host_free(void* host_ptr);
/* bytes of device workspace the ctx holds right now (grow-only: the high-water mark of every call so far; weights not included) */
int64_t rfe_workspace_bytes(rfe_ctx* ctx);
int rfe_memcpy_h2d(rfe_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int rfe_memcpy_d2h(rfe_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);

/* ---- SuperPoint ----
 * img:   u8 grayscale, B frames of H x W, row pitch `stride` bytes, frame pitch stride*H
 *        (any H, W >= 8, as the reference graph's dynamic axes allow: the three 2x2 max-pools floor, keypoints come from
 *        the 8*(H/8) x 8*(W/8) top-left frame -- KITTI's 1241 x 376 gives a 1240 x 376 score map; the reference asserts
 *        CV_8UC1, SPextractor.cc:525)
 * Kmax:  capacity per frame (the export-time max_num_keypoints of the reference graph)
 * thr:   detection threshold (0.0005 in the LightGlue-ONNX SuperPoint export)
 * n:     [B]            number of keypoints per frame
 * kxy:   [B,Kmax,2] i32 (x,y) integer pixel coordinates; rows >= n[b] are zero
 * score: [B,Kmax]
 * desc:  [B,Kmax,256]   L2-normalised descriptors
 * Order: score-descending (ties: row-major pixel index ascending) when more than Kmax candidates
 * pass the threshold, row-major otherwise. */
int rfe_extract_u8(rfe_ctx* ctx, const uint8_t* img, int H, int W, int stride, int B, int Kmax,
                   float thr, int32_t* n, int32_t* kxy, float* score, float* desc);
int rfe_extract_u8_dev(rfe_ctx* ctx, const uint8_t* img_dev, int H, int W, int stride, int B,
                       int Kmax, float thr, int32_t* n_dev, int32_t* kxy_dev, float* score_dev,
                       float* desc_dev);

/* The float entry: img is an ALREADY normalised single-channel float image (what NormalizeImage produced, or any CV_32F the
 * caller hands to Extractor_Inference, src/Extractors/superpoint_onnx.cc:88-118); the values enter conv1a unchanged, whatever
 * their range.  stride in floats.  For img[i] = u8[i] * (1/255.f) the outputs are bit-identical to rfe_extract_u8's. */
int rfe_extract_f32(rfe_ctx* ctx, const float* img, int H, int W, int stride, int B, int Kmax,
                    float thr, int32_t* n, int32_t* kxy, float* score, float* desc);
int rfe_extract_f32_dev(rfe_ctx* ctx, const float* img_dev, int H, int W, int stride, int B,
                        int Kmax, float thr, int32_t* n_dev, int32_t* kxy_dev, float* score_dev,
                        float* desc_dev);

/* The same with a second descriptor output for the loop-closure side: desc_bin u8 [B,Kmax,256] = desc > 0,
 * i.e. Frame::binarize_descriptors (src/Frame.cc:1034-1043) written by the sampling kernel itself, ready for
 * Converter::toDescriptorVector + the DBoW3 vocabulary of Frame::ComputeBoW3 (:1044-1054), which stay on the CPU.
 * desc_bin may be NULL (= the plain call). */
int rfe_extract_u8_bin(rfe_ctx* ctx, const uint8_t* img, int H, int W, int stride, int B, int Kmax, float thr, int32_t* n,
                       int32_t* kxy, float* score, float* desc, uint8_t* desc_bin);
int rfe_extract_u8_bin_dev(rfe_ctx* ctx, const uint8_t* img_dev, int H, int W, int stride, int B, int Kmax, float thr,
                           int32_t* n_dev, int32_t* kxy_dev, float* score_dev, float* desc_dev, uint8_t* desc_bin_dev);

/* ---- LightGlue ----
 * P pairs.  k0n/k1n: normalised keypoints [P,Mmax,2] / [P,Nmax,2]; d0/d1: [P,Mmax,256] /
 * [P,Nmax,256]; m/n: [P] valid counts.  filter_thr: in-graph match filter (0.1).
 * S: [P] number of matches; pairs: [P,min(Mmax,Nmax),2] (i,j) with i ascending; ms: scores. */
int rfe_match(rfe_ctx* ctx, const float* k0n, const float* k1n, const float* d0, const float* d1,
              const int32_t* m, const int32_t* n, int P, int Mmax, int Nmax, float filter_thr,
              int32_t* S, int32_t* pairs, float* ms);
int rfe_match_dev(rfe_ctx* ctx, const float* k0n, const float* k1n, const float* d0,
                  const float* d1, const int32_t* m, const int32_t* n, int P, int Mmax, int Nmax,
                  float filter_thr, int32_t* S, int32_t* pairs, float* ms);

/* One pair, pixel keypoints in, vnMatches12 out (length M, -1 = unmatched); returns the number
 * of accepted matches (>= 0) or a negative rfe_status.  (rows, cols) is the image size used by
 * NormalizeKeypoints; pass 300,400 to reproduce the hard-coded quirk of three of the reference's
 * four overloads (SPmatcher.cc:360-361,376-377,414-415). */
int rfe_match_fused(rfe_ctx* ctx, const float* kpts0_xy, int M, const float* kpts1_xy, int N,
                    const float* desc0, const float* desc1, int rows, int cols, float filter_thr,
                    float match_thresh, int32_t* vnMatches12);

/* ---- batched stream (device resident): extract B frames, match (i, i+1) for i < B-1 ----
 * Outputs as in rfe_extract_u8_dev, plus S:[B-1], pairs:[B-1,Kmax,2], ms:[B-1,Kmax]
 * (any match output pointer may be NULL to skip the copy-out of that array). */
int rfe_extract_match_stream_dev(rfe_ctx* ctx, const uint8_t* img_dev, int H, int W, int stride,
                                 int B, int Kmax, float thr, float filter_thr, int32_t* n_dev,
                                 int32_t* kxy_dev, float* score_dev, float* desc_dev,
                                 int32_t* S_dev, int32_t* pairs_dev, float* ms_dev);

/* ---- sparse stereo matching ----
 * Frame::ComputeStereoMatches (src/Frame.cc:1159-1446) for nLevels == 1: for every left keypoint, best right
 * keypoint within +-2 rows and the disparity range [0, mbf/mb) by 256-d L2 distance
 * (SPmatcher::DescriptorDistance_sp, src/Matchers/SPmatcher.cc:2184-2189; accepted below (TH_HIGH+TH_LOW)/2 = 1.3),
 * 11x11 SAD refinement over +-5 px on the raw images, parabola sub-pixel fit, median outlier cut.
 * kL/kR: pixel keypoints [N,2]/[Nr,2]; dL/dR: descriptors [N,256]/[Nr,256]; mb, mbf: baseline (m) and
 * baseline*fx of the Frame.  Outputs mvuRight / mvDepth: [N], -1 = no match.  Deviation: left keypoints whose
 * 11x11 patch leaves the image are skipped (the reference's cv::Mat::rowRange would throw). */
int rfe_stereo_match(rfe_ctx* ctx, const uint8_t* imgL, const uint8_t* imgR, int H, int W, int stride,
                     const float* kL, int N, const float* kR, int Nr, const float* dL, const float* dR,
                     float mb, float mbf, float* uRight, float* depth);
int rfe_stereo_match_dev(rfe_ctx* ctx, const uint8_t* imgL, const uint8_t* imgR, int H, int W, int stride,
                         const float* kL, int N, const float* kR, int Nr, const float* dL, const float* dR,
                         float mb, float mbf, float* uRight, float* depth);

/* ---- stereo stream (BASELINE configs[4]): ONE device-resident call per stereo frame ----
 * What the reference does per stereo frame, fused: the stereo Frame constructor extracts the left and right views on two
 * threads (src/Frame.cc:106-171, :142-147) and calls ComputeStereoMatches (:165, :1159-1446); Tracking then matches the
 * frame against the previous one with LightGlue (SPmatcher::MatchingPoints_onnx, Frame overload, src/Matchers/SPmatcher.cc:457-542,
 * called from :1050-1080).  Here: both views through SuperPoint as one batch of 2, the sparse stereo match on the device-
 * resident features (keypoint counts never visit the host), and one LightGlue match of THIS left view (set 0) against the
 * PREVIOUS left view (set 1, kept inside the ctx) with the true image size -- the argument order of
 * SearchBySP(mCurrentFrame, mLastFrame) (src/Tracking.cc:3465; SPmatcher.cc:1050-1054), so pairs[k] = (index in the current
 * frame, index in the previous frame) like vnMatches1[IdxCF] = IdxLF.  Asynchronous on the ctx stream.
 * imgL/imgR: device u8 [H,W], row pitch `stride`.  reset != 0 (or a change of H, W, Kmax) starts a new sequence: S = 0.
 * Outputs (device): n [2] (left, right), kxy [2,Kmax,2], score [2,Kmax], desc [2,Kmax,256] as rfe_extract_u8_dev;
 * uRight / depth [Kmax] as rfe_stereo_match (entries >= n[0] are -1); S [1], pairs [Kmax,2], ms [Kmax] as rfe_match_dev. */
int rfe_stereo_frame_dev(rfe_ctx* ctx, const uint8_t* imgL_dev, const uint8_t* imgR_dev, int H, int W, int stride, int Kmax,
                         float thr, float filter_thr, float mb, float mbf, int reset, int32_t* n_dev, int32_t* kxy_dev,
                         float* score_dev, float* desc_dev, float* uRight_dev, float* depth_dev, int32_t* S_dev,
                         int32_t* pairs_dev, float* ms_dev);

/* ---- descriptor helpers for the callers' classic searches, host pointers (device forms below) ----
 * rfe_l2_distance_matrix: out[i*N + j] = SPmatcher::DescriptorDistance_sp(a_i, b_j)
 *   (src/Matchers/SPmatcher.cc:2184-2189) for all pairs of a [M,256] x b [N,256]; the candidate lists of
 *   SearchByProjection / Fuse (SPmatcher.cc:1170-1354, 49-357) stay with the caller.
 * rfe_binarize_descriptors: Frame::binarize_descriptors (src/Frame.cc:1034-1043): out u8 [rows,256] = desc > 0. */
int rfe_l2_distance_matrix(rfe_ctx* ctx, const float* a, int M, const float* b, int N, float* out);
int rfe_binarize_descriptors(rfe_ctx* ctx, const float* desc, int rows, uint8_t* out);

/* rfe_search_candidates: the best / second-best descriptor scan of SPmatcher::SearchByProjection1
 *   (src/Matchers/SPmatcher.cc:1218-1248; the same loop in SearchByProjection :755-800 and Fuse :150-200) for Nq
 *   map-point descriptors q [Nq,256] against the frame descriptors f [Nf,256].  Candidate lists are the caller's
 *   (Frame::GetFeaturesInArea) in CSR form: cand[offsets[i] .. offsets[i+1]) are the frame features near map point i,
 *   scanned in that order.  skip [Nf] (may be NULL): non-zero = feature already owns a MapPoint with observations
 *   (:1226-1228).  Per map point: best_idx (-1 = none), best_dist and second_dist (both start at 256, strict '<').
 *   The TH_HIGH cut and the greedy assignment (:1253-1262) stay with the caller.
 * rfe_distinctive_descriptors: MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:438-530) for Np map points at
 *   once: desc holds the observed descriptors of all points back to back, offsets [Np+1] delimits them (<= 8192 per
 *   point).  best[p] = index within the point's own list of the descriptor with the least median distance to the
 *   others (first strict minimum, -1 for an empty point), median[p] = that median. */
int rfe_search_candidates(rfe_ctx* ctx, const float* q, int Nq, const float* f, int Nf, const int32_t* offsets,
                          const int32_t* cand, const uint8_t* skip, int32_t* best_idx, float* best_dist, float* second_dist);
int rfe_distinctive_descriptors(rfe_ctx* ctx, const float* desc, const int32_t* offsets, int Np, int32_t* best, float* median);

/* Device-resident forms of the four helpers above (descriptors normally never leave HBM: they come out of rfe_extract_u8_dev /
 * rfe_stereo_frame_dev, and the host forms re-upload 1 KB per descriptor per call): every pointer is a DEVICE pointer, the call is
 * asynchronous on the ctx stream, the host forms are thin staging wrappers around these.  What the host forms validate and these
 * cannot without a synchronisation is the caller's contract instead: offsets[0] = 0, offsets non-decreasing;
 * rfe_search_candidates_dev ignores candidate indices outside [0, Nf) (SPmatcher.cc:1218-1262 semantics otherwise unchanged);
 * rfe_distinctive_descriptors_dev (MapPoint.cc:438-530) takes `total` >= offsets[Np] (descriptor count) and `maxn` >= the largest
 * observation count of any point (<= 8192) from the caller, and reports a point with more observations than maxn rounded up to
 * a power of two as best = -2. */
int rfe_l2_distance_matrix_dev(rfe_ctx* ctx, const float* a_dev, int M, const float* b_dev, int N, float* out_dev);
int rfe_binarize_descriptors_dev(rfe_ctx* ctx, const float* desc_dev, int rows, uint8_t* out_dev);
int rfe_search_candidates_dev(rfe_ctx* ctx, const float* q_dev, int Nq, const float* f_dev, int Nf, const int32_t* offsets_dev,
                              const int32_t* cand_dev, const uint8_t* skip_dev, int32_t* best_idx_dev, float* best_dist_dev,
                              float* second_dist_dev);
int rfe_distinctive_descriptors_dev(rfe_ctx* ctx, const float* desc_dev, const int32_t* offsets_dev, int Np, int total, int maxn,
                                    int32_t* best_dev, float* median_dev);

/* ---- multi-device pool (BASELINE configs[3] for a C / C++ host) ----
 * The reference runs on ONE device (device_id = 0, src/Extractors/superpoint_onnx.cc:19, src/Matchers/lightglue_onnx.cpp:24) and
 * is a C++ program (src/Tracking.cc:645-651); a pool gives such a host the frame sharding of rover-slam_amd/sharding.py without
 * Python: one ctx + one host worker thread per member device.  A stream of F frames has F - 1 consecutive pairs; they are split
 * as evenly as possible (rfe_pool_shard), member r extracts its pairs' frames -- ONE overlap frame with its neighbour, no
 * inter-device dependency -- and matches them; every member's results are then gathered into one root buffer on member 0's device
 * in global frame / pair order and copied to the caller's HOST arrays (layouts as rfe_extract_match_stream_dev with B = F;
 * score / desc may be NULL = not gathered).
 * transport: RFE_POOL_RCCL = grouped ncclSend / ncclRecv into the root (librccl is dlopen'ed at pool creation; needs pairwise
 * distinct devices), RFE_POOL_COPY = peer / device-to-device copies issued by the root (members may share a device),
 * RFE_POOL_AUTO = RCCL when the pool has more than one member and the communicators are up, else COPY.
 * `devices` may name one device several times (functional tests on a one-GPU box; weights are then shared, see rfe_weights_id).
 * A pool is single-caller like a ctx.  rfe_pool_ctx exposes a member's ctx (options, profiling); do not run it while a pool call
 * is in flight. */
typedef struct rfe_pool rfe_pool;
#define RFE_POOL_AUTO 0
#define RFE_POOL_RCCL 1
#define RFE_POOL_COPY 2
int rfe_pool_create(const int* devices, int n, rfe_pool** out);
void rfe_pool_destroy(rfe_pool* pool);
const char* rfe_pool_last_error(rfe_pool* pool); /* pool may be NULL: last error of a failed rfe_pool_create */
int rfe_pool_size(rfe_pool* pool);
rfe_ctx* rfe_pool_ctx(rfe_pool* pool, int member);
int rfe_pool_has_rccl(rfe_pool* pool);           /* 1 = RCCL communicators are up */
int rfe_pool_set_weights(rfe_pool* pool, int kind, const float* blob, int64_t count);
int rfe_pool_load_weights(rfe_pool* pool, const char* sp_path, const char* lg_path);
int rfe_pool_set_option(rfe_pool* pool, int option, int value);
int rfe_pool_set_hparams(rfe_pool* pool, const rfe_hparams* in);   /* every member; rfe_pool_load_weights applies a v2 file's values by itself */
/* the sharding rule itself (pure function, no device): member `member` of n extracts frames [first_frame, first_frame + frames)
 * of an F-frame stream and owns `pairs` consecutive pairs starting at first_frame (frames = pairs + 1; 0 / 0 = idle member) */
int rfe_pool_shard(int F, int n, int member, int* first_frame, int* frames, int* pairs);
int rfe_pool_extract_match_stream(rfe_pool* pool, const uint8_t* img_host, int H, int W, int stride, int F, int Kmax, float thr,
                                  float filter_thr, int transport, int32_t* n, int32_t* kxy, float* score, float* desc,
                                  int32_t* S, int32_t* pairs, float* ms);

/* ---- per-stage timing (hipEvent on the ctx stream), for bench.py's roofline object ----
 * Enable, run, then read back: names is a ';'-separated list of stage names, ms / calls the
 * accumulated time and launch count per stage since the last reset.
 * rfe_profile_filter restricts the events to one stage (NULL or "" = all stages): each pair of events keeps
 * consecutive kernels from overlapping their launch and drain (about 2 % of a batched step when every stage is
 * instrumented), so a throughput measurement instruments only the kernel it reports. */
int rfe_profile_enable(rfe_ctx* ctx, int on);
int rfe_profile_filter(rfe_ctx* ctx, const char* stage);
int rfe_profile_reset(rfe_ctx* ctx);
int rfe_profile_read(rfe_ctx* ctx, char* names, size_t names_cap, double* ms, int64_t* calls, int cap);

/* ---- kernel-level test hooks (device pointers, synchronous): used only by tests/ ----
 * NHWC activations, canonical weight layouts as in the oracle. */
int rfe_k_conv3x3(rfe_ctx* ctx, const float* in_dev, int B, int H, int W, int Cin,
                  const float* w_oihw_host, const float* bias_host, int Cout, int relu, int pool,
                  float* out_dev);
int rfe_k_linear(rfe_ctx* ctx, const float* a_dev, int M, int K, const float* w_nk_host,
                 const float* bias_host, int N, int relu, float* out_dev);
int rfe_k_scoremap(rfe_ctx* ctx, const uint8_t* img_dev, int H, int W, int stride, int B,
                   float* scoremap_dev /*[B,Hs,Ws] pre-NMS, Hs = 8*(H/8), Ws = 8*(W/8)*/,
                   float* nms_dev /*[B,Hs,Ws] post-NMS+border*/, float* descmap_dev /*[B,H/8,W/8,256]*/);
/* keypoint selection alone on a post-NMS map (H, W multiples of 8 as the score-map frame always is): n [B], kxy [B,Kmax,2], score [B,Kmax] */
int rfe_k_select(rfe_ctx* ctx, const float* nms_dev /*[B,H,W]*/, int B, int H, int W, int Kmax, float thr, int topk_always,
                 int32_t* n_dev, int32_t* kxy_dev, float* score_dev);
/* the same through the form one to four frames take in the forward: an UNORDERED list of 64-bit candidate keys (as the fused detector tail leaves it), ranked */
int rfe_k_select_keys(rfe_ctx* ctx, const float* nms_dev /*[B,H,W], B <= 4*/, int B, int H, int W, int Kmax, float thr, int topk_always,
                      int32_t* n_dev, int32_t* kxy_dev, float* score_dev);
int rfe_k_lightglue_taps(rfe_ctx* ctx, const float* k0n, const float* k1n, const float* d0,
                         const float* d1, int M, int N, float* x0_dev, float* x1_dev,
                         float* scores_dev /*[M,N]*/);

/* One FFN block of LightGlue with the loaded weights: out = x + ffn.3(gelu(LayerNorm(ffn.0([x | second])))) for `rows` token rows
 * (x, second, out: device [rows,256]; out may not alias x), self (cross = 0) or cross block of `layer`.  Goes through the forward's
 * own code, so the row count selects the tiling: >= 32768 rows = throughput tiles with the fused LayerNorm + GELU. */
int rfe_k_lightglue_ffn(rfe_ctx* ctx, int layer, int cross, const float* x_dev, const float* second_dev, int rows, float* out_dev);
/* The fused attention alone: out[seq][q][head*64 + d] = softmax(Q K^T / 8) V per (sequence, head), 4 heads of 64 (device pointers, row
 * stride ld floats for q / k / v, out [nseq*Lq, 256]); qlen / klen: valid rows per sequence or NULL, kv_map: sequence -> sequence whose
 * k / v it attends to or NULL, rope: LightGlue's rotary table [nseq*Lq, 32] of (cos, sin) pairs applied to q and k, or NULL.  Goes through
 * the forward's launcher, so nseq * Lq and RFE_OPT_LG_FP16X2 select the kernel exactly as inside a match call. */
int rfe_k_attention(rfe_ctx* ctx, const float* q_dev, const float* k_dev, const float* v_dev, int ld, float* out_dev, int nseq, int Lq, int Lk,
                    const int32_t* qlen_dev, const int32_t* klen_dev, const int32_t* kv_map_dev, const float* rope_dev);
/* Projection + attention of one self block with the loaded weights, through the forward's own code: x_dev [nseq*L, 256] token rows, csn_dev [nseq*L, 32]
 * (cos, sin) rotary pairs, lens_dev [nseq] valid rows; qkv_out_dev [nseq*L, 768] = [q | k | v] as the attention reads them, ctx_out_dev [nseq*L, 256]
 * (either may be NULL).  nseq * L selects the path as inside a match call: throughput shapes rotate q | k in the projection's epilogue (gemm.hip) and run
 * the LDS-DMA attention kernel, one / few pairs take gemm_lat.hip + lg_attention_lat.hip; *qk_rotated = 1 when qkv_out's q | k are rotated (0: the
 * fallback that rotates on load).  L % 4 == 0. */
int rfe_k_lightglue_self_attention(rfe_ctx* ctx, int layer, const float* x_dev, const float* csn_dev, const int32_t* lens_dev, int nseq, int L,
                                   float* qkv_out_dev, float* ctx_out_dev, int32_t* qk_rotated);
/* One-shot tap for the NEXT LightGlue forward of this ctx, whichever entry point runs it (rfe_match[_dev] with P pairs,
 * rfe_extract_match_stream_dev, rfe_stereo_frame_dev) and therefore whichever tiling it selects: after the last layer the final
 * token states of pair `pair` are copied to x0_dev / x1_dev ([L,256] each, L = max(Mmax,Nmax) rounded up to 4; rows past the
 * pair's keypoint counts are padding) and its log-assignment matrix to scores_dev ([L,L], row stride L; only the m x n block is
 * written).  Any pointer may be NULL; pair < 0 disarms.  A pair index >= P of the next forward is ignored. */
int rfe_k_set_lightglue_tap(rfe_ctx* ctx, int pair, float* x0_dev, float* x1_dev, float* scores_dev);
/* Fault injection for the pool's RCCL gather: the NEXT gather of `member` fails inside its ncclGroup (as a refused ncclSend would).  The
 * failing member aborts every member's communicator, nobody is left waiting in a collective, and the call delivers its results through
 * the COPY transport (RFE_POOL_AUTO) or reports the failure (RFE_POOL_RCCL); later calls use COPY.  tests/test_pool.py. */
int rfe_k_pool_inject_gather_failure(rfe_pool* pool, int member);
/* The ONNX reader without a ctx (and without a GPU): `path` of `kind` -> blob [rfe_weight_count(kind)] and that kind's fields of *hp (weights_only != 0:
 * the initializers alone, no hyper-parameter readers).  RFE_OK, or RFE_ERR_IO with the reason in err.  tests/test_onnx_cpp.py holds it bit for bit
 * against rover-slam_amd/onnx_weights.py. */
int rfe_k_onnx_convert(const char* path, int kind, int weights_only, float* blob, rfe_hparams* hp, char* err, int errlen);

#ifdef __cplusplus
}
#endif
#endif /* ROVER_FE_H */
