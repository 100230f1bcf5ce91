// sp_post.hip -- the non-GEMM tail of SuperPoint: softmax(65)+depth-to-space, simple_nms(r),
// border/threshold/top-k selection, descriptor-map L2 normalisation and bilinear descriptor sampling.
// In the reference all of this is inside superpoint.onnx (Ort::Session::Run,
// src/Extractors/superpoint_onnx.cc:133-136); outputs follow the tensor contract read by
// Extractor_PostProcess (superpoint_onnx.cc:169-181): keypoints (x,y), scores, descriptors[.,256].
// All kernels here are HBM/latency bound (a few MB per frame); arithmetic orders are the canonical
// ones of oracle/rfe_oracle.c so that scores / keypoints / descriptors are bit-exact.
#include "rfe_internal.h"

namespace rfe {

// ---------------------------------------------------------------- canonical expf (== rfo_expf)
__device__ __forceinline__ float rfe_expf(float x) {
    x = fmaxf(x, -87.0f);
    x = fminf(x, 88.0f);
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    const float r2 = r * r;
    const float y = fmaf(p, r2, r) + 1.0f;
    return y * __uint_as_float((uint32_t)((int)n + 127) << 23);
}

// one thread per 8x8 cell: 65-way softmax in index order, drop dustbin, scatter to the H x W map
__global__ __launch_bounds__(256) void softmax65_d2s_kernel(const float* __restrict__ logits, int ld,
                                                            int cells_total, int Hc, int Wc,
                                                            float* __restrict__ score) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= cells_total) return;
    const int b = gid / (Hc * Wc), cell = gid % (Hc * Wc);
    const float* l = logits + (size_t)gid * ld;
    float e[65];
    float m = l[0];
#pragma unroll
    for (int c = 0; c < 65; ++c) { e[c] = l[c]; m = fmaxf(m, e[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 65; ++c) { e[c] = rfe_expf(e[c] - m); s = s + e[c]; }
    const int W = Wc * 8;
    const int cy = cell / Wc, cx = cell % Wc;
    float* o = score + (size_t)b * Hc * 8 * W + (size_t)(cy * 8) * W + cx * 8;
#pragma unroll
    for (int dy = 0; dy < 8; ++dy) {
        float4 v0 = make_float4(e[dy * 8] / s, e[dy * 8 + 1] / s, e[dy * 8 + 2] / s, e[dy * 8 + 3] / s);
        float4 v1 = make_float4(e[dy * 8 + 4] / s, e[dy * 8 + 5] / s, e[dy * 8 + 6] / s, e[dy * 8 + 7] / s);
        *reinterpret_cast<float4*>(o + (size_t)dy * W) = v0;
        *reinterpret_cast<float4*>(o + (size_t)dy * W + 4) = v1;
    }
}

void launch_softmax65_d2s(hipStream_t s, const float* logits, int ld, int B, int Hc, int Wc, float* score) {
    const int total = B * Hc * Wc;
    hipLaunchKernelGGL(softmax65_d2s_kernel, dim3((total + 255) / 256), dim3(256), 0, s, logits, ld, total, Hc, Wc, score);
}

// ---------------------------------------------------------------- simple_nms
// published recurrence (SuperPoint / LightGlue):  max_mask = s == mp(s);
//   2x { supp = mp(max_mask) > 0; ss = supp ? 0 : s; new = ss == mp(ss); max_mask |= new & ~supp }
//   out = max_mask ? s : 0   then the `border`-px frame is set to -1.
// mp = max_pool2d(2 r + 1, stride 1, padding r).  The radius and the border are hyper-parameters of the reference's graph
// (rfe_hparams: baked into superpoint.onnx at export time): CR = 4 is the compile-time instance of the published default,
// CR = 0 takes the radius at run time (1..NMS_MAX_RADIUS, LDS tiles sized for the maximum).
constexpr int NTH = 32, NTW = 64;

template <int CR, typename LoadFn>
__device__ __forceinline__ void tile_maxpool(LoadFn load, int y0, int x0, int H, int W, int r, float* t0, float* t1) {
    const int R = CR ? CR : r, NIH = NTH + 2 * R, NIW = NTW + 2 * R;
    for (int idx = threadIdx.x; idx < NIH * NIW; idx += 256) {
        const int py = idx / NIW, px = idx % NIW;
        const int gy = y0 - R + py, gx = x0 - R + px;
        t0[idx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? load(gy, gx) : -INFINITY;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NIH * NTW; idx += 256) {
        const int py = idx / NTW, px = idx % NTW;
        const float* rp = t0 + py * NIW + px;
        float m = rp[0];
        if (CR) {
#pragma unroll
            for (int d = 1; d <= 2 * CR; ++d) m = fmaxf(m, rp[d]);
        } else {
            for (int d = 1; d <= 2 * R; ++d) m = fmaxf(m, rp[d]);
        }
        t1[idx] = m;
    }
    __syncthreads();
}
template <int CR>
__device__ __forceinline__ float tile_colmax(const float* t1, int py, int px, int r) {
    const float* c = t1 + py * NTW + px;
    float m = c[0];
    if (CR) {
#pragma unroll
        for (int d = 1; d <= 2 * CR; ++d) m = fmaxf(m, c[d * NTW]);
    } else {
        for (int d = 1; d <= 2 * r; ++d) m = fmaxf(m, c[d * NTW]);
    }
    return m;
}

// MODE 0: mask = (s == mp(s))
// MODE 1: supp = mp(mask) > 0 ; ss = supp ? 0 : s
// MODE 2: mask |= (ss == mp(ss)) & ~supp ; FINAL: out = mask ? s : 0, border -> -1
template <int MODE, bool FINAL, int CR>
__global__ __launch_bounds__(256) void nms_pass_kernel(const float* __restrict__ s, float* __restrict__ ss,
                                                       uint8_t* __restrict__ mask, uint8_t* __restrict__ supp,
                                                       float* __restrict__ out, int H, int W, int r, int border) {
    constexpr int RM = CR ? CR : NMS_MAX_RADIUS;
    __shared__ float t0[(NTH + 2 * RM) * (NTW + 2 * RM)];
    __shared__ float t1[(NTH + 2 * RM) * NTW];
    const int R = CR ? CR : r, NIW = NTW + 2 * R;
    const size_t fo = (size_t)blockIdx.z * H * W;
    const int y0 = blockIdx.y * NTH, x0 = blockIdx.x * NTW;
    if (MODE == 0) tile_maxpool<CR>([&](int y, int x) { return s[fo + (size_t)y * W + x]; }, y0, x0, H, W, r, t0, t1);
    if (MODE == 1) tile_maxpool<CR>([&](int y, int x) { return mask[fo + (size_t)y * W + x] ? 1.f : 0.f; }, y0, x0, H, W, r, t0, t1);
    if (MODE == 2) tile_maxpool<CR>([&](int y, int x) { return ss[fo + (size_t)y * W + x]; }, y0, x0, H, W, r, t0, t1);
    for (int idx = threadIdx.x; idx < NTH * NTW; idx += 256) {
        const int py = idx / NTW, px = idx % NTW;
        const int y = y0 + py, x = x0 + px;
        if (y >= H || x >= W) continue;
        const float m = tile_colmax<CR>(t1, py, px, r);
        const size_t o = fo + (size_t)y * W + x;
        if (MODE == 0) {
            mask[o] = (s[o] == m) ? 1 : 0;
        } else if (MODE == 1) {
            const bool sp = m > 0.f;
            supp[o] = sp ? 1 : 0;
            ss[o] = sp ? 0.f : s[o];
        } else {
            const float c = t0[(py + R) * NIW + px + R];  // ss at this pixel
            bool mk = mask[o] != 0;
            if (c == m && !supp[o]) mk = true;
            if (FINAL) {
                float v = mk ? s[o] : 0.f;
                if (y < border || y >= H - border || x < border || x >= W - border) v = -1.f;
                out[o] = v;
            } else {
                mask[o] = mk ? 1 : 0;
            }
        }
    }
}

template <int CR>
static void launch_nms_r(hipStream_t st, const float* score, int B, int H, int W, int r, int border, float* tmp_ss,
                         uint8_t* tmp_mask, uint8_t* tmp_supp, float* out) {
    dim3 grid((W + NTW - 1) / NTW, (H + NTH - 1) / NTH, B), blk(256);
    hipLaunchKernelGGL((nms_pass_kernel<0, false, CR>), grid, blk, 0, st, score, tmp_ss, tmp_mask, tmp_supp, out, H, W, r, border);
    hipLaunchKernelGGL((nms_pass_kernel<1, false, CR>), grid, blk, 0, st, score, tmp_ss, tmp_mask, tmp_supp, out, H, W, r, border);
    hipLaunchKernelGGL((nms_pass_kernel<2, false, CR>), grid, blk, 0, st, score, tmp_ss, tmp_mask, tmp_supp, out, H, W, r, border);
    hipLaunchKernelGGL((nms_pass_kernel<1, false, CR>), grid, blk, 0, st, score, tmp_ss, tmp_mask, tmp_supp, out, H, W, r, border);
    hipLaunchKernelGGL((nms_pass_kernel<2, true, CR>), grid, blk, 0, st, score, tmp_ss, tmp_mask, tmp_supp, out, H, W, r, border);
}

// ---- the whole recurrence in ONE launch (radius 4, the published default): a workgroup takes a 32 x 64 output tile with a halo of
// 5 r = 20 pixels (five nested 9-windows), keeps the score tile in LDS and runs the five max-pools there -- row pass into a scratch plane,
// column pass fused with the stage's elementwise step.  Each stage is valid on a frame 4 px smaller than the previous one; after the fifth
// exactly the output tile is left.  Same comparisons on the same floats as the five-launch form (bit-identical output); the mask /
// suppression planes and four of the five kernel boundaries (10 - 16 us each at batch 1, where the whole detector tail is latency) are gone.
// Pixels outside the image read -inf in every plane, like max_pool2d's padding.  Work items are QUADS (four consecutive outputs of a row /
// of a column): 12 loaded values give four 9-wide maxima in 17 max operations (shared middle, left and right running partials), and all
// index arithmetic divides by compile-time constants (a first version with one output per item and run-time frame sizes took 78 us per
// frame -- slower than the five launches it replaced).
// The output tile is TH x 64 with TH = 32 / 40 / 48 (LDS 127 / 141 / 156 KB, always one workgroup per CU): the launcher takes the smallest
// one that puts all frames of the call into ONE round of 256 workgroups (two VGA frames: 2 x 150 tiles of 32 rows are two rounds, 2 x 120 of
// 40 rows one: 52.6 -> see profiles/r04_ab_notes.md).
constexpr int NF_R = 4, NF_HALO = 5 * NF_R, NF_IW = NTW + 2 * NF_HALO;   // 104 columns
static_assert(NF_IW % 4 == 0, "quad items");

// four 9-wide maxima from 12 consecutive values v[0..11]: out[i] = max(v[i .. i + 8])
__device__ __forceinline__ void max9x4(const float (&v)[12], float (&o)[4]) {
    const float mid = fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7]));
    const float l3 = v[3], l2 = fmaxf(v[2], l3), l1 = fmaxf(v[1], l2), l0 = fmaxf(v[0], l1);
    const float r8 = v[8], r9 = fmaxf(r8, v[9]), r10 = fmaxf(r9, v[10]), r11 = fmaxf(r10, v[11]);
    o[0] = fmaxf(fmaxf(l0, mid), r8); o[1] = fmaxf(fmaxf(l1, mid), r9);
    o[2] = fmaxf(fmaxf(l2, mid), r10); o[3] = fmaxf(fmaxf(l3, mid), r11);
}
// row pass: src valid on the frame of margin M -> T[y][x] = max(src[y][x - 4 .. x + 4]) for rows of that frame, columns of margin M + 4
template <int M, int NF_IH, int NT = 256>
__device__ __forceinline__ void nf_rowpass(const float* __restrict__ src, float* __restrict__ T, int tid) {
    constexpr int ROWS = NF_IH - 2 * M, QUADS = (NF_IW - 2 * (M + NF_R)) / 4;
    for (int idx = tid; idx < ROWS * QUADS; idx += NT) {
        const int py = M + idx / QUADS, px = M + NF_R + 4 * (idx % QUADS);
        const f32x4* p = reinterpret_cast<const f32x4*>(src + py * NF_IW + px);
        const f32x4 a = p[-1], b = p[0], c = p[1];
        const float v[12] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3]};
        float o[4];
        max9x4(v, o);
        *reinterpret_cast<f32x4*>(T + py * NF_IW + px) = f32x4{o[0], o[1], o[2], o[3]};
    }
}
// column pass over T on the frame of margin M (a multiple of 4): calls f(py, px, pooled value) for every pixel of that frame; an item is
// four consecutive rows of one column, lanes run along x (conflict-free reads)
template <int M, int NF_IH, int NT = 256, typename F>
__device__ __forceinline__ void nf_colpass(const float* __restrict__ T, int tid, F f) {
    constexpr int COLS = NF_IW - 2 * M, QROWS = (NF_IH - 2 * M) / 4;
    for (int idx = tid; idx < QROWS * COLS; idx += NT) {
        const int py = M + 4 * (idx / COLS), px = M + idx % COLS;
        const float* p = T + (py - NF_R) * NF_IW + px;
        float v[12];
#pragma unroll
        for (int d = 0; d < 12; ++d) v[d] = p[d * NF_IW];
        float o[4];
        max9x4(v, o);
#pragma unroll
        for (int e = 0; e < 4; ++e) f(py + e, px, o[e]);
    }
}

template <int TH_>
__global__ __launch_bounds__(256) void nms_fused_kernel(const float* __restrict__ s, float* __restrict__ out, int H, int W, int border) {
    constexpr int NF_IH = TH_ + 2 * NF_HALO;
    static_assert(NF_IH % 4 == 0, "quad items");
    extern __shared__ __attribute__((aligned(16))) float nf_lds[];
    float* const S = nf_lds;                       // scores
    float* const T = S + NF_IH * NF_IW;            // row-pass scratch
    float* const Mk = T + NF_IH * NF_IW;           // max_mask as 1 / 0 (-inf outside the image)
    float* const SS = Mk + NF_IH * NF_IW;          // supp_scores
    uint8_t* const SUP = reinterpret_cast<uint8_t*>(SS + NF_IH * NF_IW);   // supp_mask
    const size_t fo = (size_t)blockIdx.z * H * W;
    const int y0 = blockIdx.y * TH_ - NF_HALO, x0 = blockIdx.x * NTW - NF_HALO;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < NF_IH * NF_IW; idx += 256) {
        const int py = idx / NF_IW, px = idx % NF_IW, gy = y0 + py, gx = x0 + px;
        S[idx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? s[fo + (size_t)gy * W + gx] : -INFINITY;
    }
    __syncthreads();
    auto inside = [&](int py, int px) { const int gy = y0 + py, gx = x0 + px; return gy >= 0 && gy < H && gx >= 0 && gx < W; };
    // stage 1: max_mask = s == mp(s)                                                    (score frame 0 -> mask frame 4)
    nf_rowpass<0, NF_IH>(S, T, tid);
    __syncthreads();
    nf_colpass<NF_R, NF_IH>(T, tid, [&](int py, int px, float m) {
        const int o = py * NF_IW + px;
        Mk[o] = inside(py, px) ? (S[o] == m ? 1.f : 0.f) : -INFINITY;
    });
    __syncthreads();
    // round 1: supp_mask = mp(max_mask) > 0 ; supp_scores = supp ? 0 : s                (mask frame 4 -> frame 8)
    nf_rowpass<NF_R, NF_IH>(Mk, T, tid);
    __syncthreads();
    nf_colpass<2 * NF_R, NF_IH>(T, tid, [&](int py, int px, float m) {
        const int o = py * NF_IW + px;
        const bool sp = m > 0.f;
        SUP[o] = sp ? 1 : 0;
        SS[o] = inside(py, px) ? (sp ? 0.f : S[o]) : -INFINITY;
    });
    __syncthreads();
    //          new_max_mask = supp_scores == mp(supp_scores) ; max_mask |= new & ~supp    (frame 8 -> frame 12)
    nf_rowpass<2 * NF_R, NF_IH>(SS, T, tid);
    __syncthreads();
    nf_colpass<3 * NF_R, NF_IH>(T, tid, [&](int py, int px, float m) {
        const int o = py * NF_IW + px;
        if (inside(py, px)) Mk[o] = (Mk[o] != 0.f || (SS[o] == m && !SUP[o])) ? 1.f : 0.f;      // stays -inf outside the image
    });
    __syncthreads();
    // round 2                                                                            (mask frame 12 -> frame 16 -> frame 20 = the output tile)
    nf_rowpass<3 * NF_R, NF_IH>(Mk, T, tid);
    __syncthreads();
    nf_colpass<4 * NF_R, NF_IH>(T, tid, [&](int py, int px, float m) {
        const int o = py * NF_IW + px;
        const bool sp = m > 0.f;
        SUP[o] = sp ? 1 : 0;
        SS[o] = inside(py, px) ? (sp ? 0.f : S[o]) : -INFINITY;
    });
    __syncthreads();
    nf_rowpass<4 * NF_R, NF_IH>(SS, T, tid);
    __syncthreads();
    nf_colpass<5 * NF_R, NF_IH>(T, tid, [&](int py, int px, float m) {
        const int o = py * NF_IW + px;
        if (!inside(py, px)) return;
        const bool mk = Mk[o] != 0.f || (SS[o] == m && !SUP[o]);
        const int gy = y0 + py, gx = x0 + px;
        float v = mk ? S[o] : 0.f;
        if (gy < border || gy >= H - border || gx < border || gx >= W - border) v = -1.f;
        out[fo + (size_t)gy * W + gx] = v;
    });
}

void launch_nms(hipStream_t st, const float* score, int B, int H, int W, int radius, int border, float* tmp_ss,
                uint8_t* tmp_mask, uint8_t* tmp_supp, float* out) {
    // One or two frames per call only: the fused kernel needs 127 KB of LDS (one workgroup per CU, 150 tiles per VGA frame -- fine when
    // there are one or two frames), at 33 frames its 4950 workgroups run in 20 rounds and the five light launches win (0.57 vs 0.25 ms per step)
    static const int fused_frames = tune_int("RFE_NMS_FUSED", 4);   // tuning build: 0 = the five-launch form at every batch size
    if (radius == NF_R && B <= fused_frames) {
        const int gx = (W + NTW - 1) / NTW;
        auto wgs = [&](int th) { return (long long)gx * ((H + th - 1) / th) * B; };
        static const int th_env = tune_int("RFE_NMS_TH", 0);   // tuning build: force the tile height
        const int th = th_env ? th_env : (wgs(32) <= 256 ? 32 : wgs(40) <= 256 ? 40 : wgs(48) <= 256 ? 48 : 32);
#define RFE_NMS_FUSED_GO(TH_)                                                                                                        \
        do {                                                                                                                         \
            constexpr int bytes = (TH_ + 2 * NF_HALO) * NF_IW * 17;   /* four float planes + one byte plane */                         \
            static bool ls_[64];                                                                                                     \
            ensure_dynamic_lds((const void*)nms_fused_kernel<TH_>, bytes, ls_);                                                      \
            hipLaunchKernelGGL(nms_fused_kernel<TH_>, dim3(gx, (H + TH_ - 1) / TH_, B), dim3(256), bytes, st, score, out, H, W, border); \
        } while (0)
        if (th == 48) RFE_NMS_FUSED_GO(48);
        else if (th == 40) RFE_NMS_FUSED_GO(40);
        else RFE_NMS_FUSED_GO(32);
#undef RFE_NMS_FUSED_GO
        return;
    }
    if (radius == 4) launch_nms_r<4>(st, score, B, H, W, radius, border, tmp_ss, tmp_mask, tmp_supp, out);
    else launch_nms_r<0>(st, score, B, H, W, radius, border, tmp_ss, tmp_mask, tmp_supp, out);
}

// ---------------------------------------------------------------- threshold + top-k selection
// (1) ordered (row-major) compaction of pixels with nms > thr (count + compact kernels over 4096-pixel chunks);
// (2) one 1024-thread workgroup per frame: if more than Kmax candidates: 4-pass radix select on the score bits for the
// Kmax-th largest score, ties resolved by ascending pixel index (candidate order), then a bitonic
// sort of the selected 64-bit keys (score bits << 32 | ~index) in LDS -> score-descending output.
constexpr int SEL_T = 1024;

__device__ __forceinline__ int block_excl_scan_flag(bool flag, int* wave_tot /*[16]*/, int& total) {
    const unsigned long long bal = __ballot(flag);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int lane_prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wv] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SEL_T / 64; ++w) { const int c = wave_tot[w]; if (w < wv) off += c; tot += c; }
    __syncthreads();
    total = tot;
    return off + lane_prefix;
}

// exclusive block scan of one int per thread (wave scan by shuffles + 16 wave totals in LDS)
__device__ __forceinline__ int block_excl_scan_int(int v, int* wave_tot /*[16]*/, int& total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off); if (lane >= off) x += y; }
    if (lane == 63) wave_tot[wv] = x;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SEL_T / 64; ++w) { const int c = wave_tot[w]; if (w < wv) off += c; tot += c; }
    __syncthreads();
    total = tot;
    return off + x - v;
}

// Ordered compaction spread over the chip (one 1024-thread workgroup scanning a 640 x 480 map alone took 95 us of the
// 105 us of this stage at batch 1): workgroup (chunk, frame) handles SEL_CHUNK consecutive pixels, 16 per thread.
// Pass 1 counts the candidates of every chunk, pass 2 writes them at (sum of the earlier chunks' counts) + block scan.
constexpr int SEL_CHUNK = 4096;

__device__ __forceinline__ int sel_load16(const float* __restrict__ s, int HW, int p0, float thr, float (&v)[16]) {
    int cnt = 0;
    if (p0 + 15 < HW) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(s + p0 + 4 * q);
            v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = p0 + e < HW ? s[p0 + e] : -1.f;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) cnt += v[e] > thr ? 1 : 0;
    return cnt;
}

__global__ __launch_bounds__(256) void select_count_kernel(const float* __restrict__ nms, int HW, int nch, float thr,
                                                           int32_t* __restrict__ chunk_cnt) {
    __shared__ int wtot[4];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    float v[16];
    int cnt = sel_load16(nms + (size_t)b * HW, HW, chunk * SEL_CHUNK + tid * 16, thr, v);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((tid & 63) == 0) wtot[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) chunk_cnt[b * nch + chunk] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
}

__global__ __launch_bounds__(256) void select_compact_kernel(const float* __restrict__ nms, int HW, int nch, float thr,
                                                             const int32_t* __restrict__ chunk_cnt,
                                                             float* __restrict__ cand_score, int32_t* __restrict__ cand_idx) {
    __shared__ int wtot[4];
    __shared__ int sbase;
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // base = candidates in the earlier chunks of this frame
    int part = 0;
    for (int c = tid; c < chunk; c += 256) part += chunk_cnt[b * nch + c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) wtot[wv] = part;
    __syncthreads();
    if (tid == 0) sbase = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    __syncthreads();
    const int base = sbase;
    const int p0 = chunk * SEL_CHUNK + tid * 16;
    float v[16];
    const int cnt = sel_load16(nms + (size_t)b * HW, HW, p0, thr, v);
    int x = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off); if (lane >= off) x += y; }
    __syncthreads();
    if (lane == 63) wtot[wv] = x;
    __syncthreads();
    int pos = base + x - cnt;
    for (int w = 0; w < wv; ++w) pos += wtot[w];
    float* cs = cand_score + (size_t)b * HW;
    int32_t* ci = cand_idx + (size_t)b * HW;
#pragma unroll
    for (int e = 0; e < 16; ++e)
        if (v[e] > thr) { cs[pos] = v[e]; ci[pos] = p0 + e; ++pos; }
}

__global__ __launch_bounds__(SEL_T) void select_kernel(const float* __restrict__ nms, int H, int W, int Kmax,
                                                       int P2, float thr, float* __restrict__ cand_score,
                                                       int32_t* __restrict__ cand_idx, int32_t* __restrict__ n_out,
                                                       int32_t* __restrict__ kxy, float* __restrict__ score,
                                                       const int32_t* __restrict__ chunk_cnt, int nch, int topk_always,
                                                       unsigned long long* __restrict__ sel_keys, int32_t* __restrict__ sel_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);  // [P2]
    __shared__ int wave_tot[SEL_T / 64];
    __shared__ int hist[256];
    __shared__ unsigned int sh_prefix;
    __shared__ int sh_remaining;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int HW = H * W;
    (void)nms;
    float* cs = cand_score + (size_t)b * HW;
    int32_t* ci = cand_idx + (size_t)b * HW;

    // candidates were compacted in row-major order by select_count_kernel / select_compact_kernel
    int count = 0;
    for (int c = 0; c < nch; ++c) count += chunk_cnt[b * nch + c];

    int32_t* okxy = kxy + (size_t)b * Kmax * 2;
    float* osc = score + (size_t)b * Kmax;
    if (count <= Kmax && !topk_always) {     // published top_k_keypoints: nothing to cut -> row-major order
        if (tid == 0) { n_out[b] = count; sel_n[b] = -1; }      // -1: select_rank_kernel has nothing to do for this frame
        for (int k = tid; k < Kmax; k += SEL_T) {
            if (k < count) {
                const int idx = ci[k];
                okxy[2 * k] = idx % W; okxy[2 * k + 1] = idx / W; osc[k] = cs[k];
            } else {
                okxy[2 * k] = 0; okxy[2 * k + 1] = 0; osc[k] = 0.f;
            }
        }
        return;
    }
    const int nsel = count < Kmax ? count : Kmax;
    if (count <= Kmax) {
        // topk_always (TopK behind Min(k, n) in the graph): every candidate is kept, ordered like the cut set below
        for (int k = tid; k < P2; k += SEL_T)
            keys[k] = k < count ? (((unsigned long long)__float_as_uint(cs[k]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)ci[k])) : 0ull;
        __syncthreads();
    } else {
    // ---- radix select: Kmax-th largest score (scores are positive floats: bit pattern is monotonic)
    if (tid == 0) { sh_prefix = 0u; sh_remaining = Kmax; }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned int prefix = sh_prefix;
        const unsigned int pmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int k = tid; k < count; k += SEL_T) {
            const unsigned int bits = __float_as_uint(cs[k]);
            if ((bits & pmask) == prefix) atomicAdd(&hist[(bits >> shift) & 255], 1);
        }
        __syncthreads();
        {   // bin holding the rem-th largest: scan the histogram from the top bin down, in parallel
            const int rem = sh_remaining;
            const int v = tid < 256 ? hist[255 - tid] : 0;
            int tot;
            const int above = block_excl_scan_int(v, wave_tot, tot);   // candidates in higher bins
            if (tid < 256 && above < rem && above + v >= rem) {
                sh_prefix = prefix | ((unsigned int)(255 - tid) << shift);
                sh_remaining = rem - above;
            }
        }
        __syncthreads();
    }
    const unsigned int T = sh_prefix;
    const int r_eq = sh_remaining;  // how many candidates with score == T are taken (lowest index first)
    for (int k = tid; k < P2; k += SEL_T) keys[k] = 0ull;
    __syncthreads();
    int eq_seen = 0, sel_seen = 0;
    for (int base = 0; base < count; base += SEL_T) {
        const int k = base + tid;
        const unsigned int bits = k < count ? __float_as_uint(cs[k]) : 0u;
        const bool gt = k < count && bits > T;
        const bool eq = k < count && bits == T;
        int tot_eq;
        const int eq_rank = eq_seen + block_excl_scan_flag(eq, wave_tot, tot_eq);
        const bool sel = gt || (eq && eq_rank < r_eq);
        int tot_sel;
        const int pos = sel_seen + block_excl_scan_flag(sel, wave_tot, tot_sel);
        if (sel) keys[pos] = ((unsigned long long)bits << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)ci[k]);
        eq_seen += tot_eq; sel_seen += tot_sel;
    }
    __syncthreads();
    }
    // the selected keys (distinct: the pixel index is part of the key) go to memory; select_rank_kernel orders them on many CUs
    if (tid == 0) { n_out[b] = nsel; sel_n[b] = nsel; }
    for (int t = tid; t < nsel; t += SEL_T) sel_keys[(size_t)b * Kmax + t] = keys[t];
}

// Rank sort, descending, of the nsel selected keys of every frame: the number of larger keys IS the output position.  It used to be the tail
// of select_kernel -- one workgroup per frame pulling Kmax^2 = 1M broadcast LDS reads through a single CU's LDS port, 30 of that kernel's
// 41 us at batch 1 -- and is now spread over Kmax / 64 workgroups per frame: each loads the frame's keys into LDS (8 KB at Kmax = 1024), four
// threads share one key and scan a quarter of the list each.  Rows [nsel, Kmax) are zeroed.
__global__ __launch_bounds__(256) void select_rank_kernel(const unsigned long long* __restrict__ sel_keys, const int32_t* __restrict__ sel_n,
                                                          int W, int Kmax, int32_t* __restrict__ kxy, float* __restrict__ score) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    const int b = blockIdx.y, tid = threadIdx.x;
    const int nsel = sel_n[b];
    if (nsel < 0) return;                 // row-major frame: select_kernel wrote the outputs itself
    const unsigned long long* src = sel_keys + (size_t)b * Kmax;
    for (int k = tid; k < nsel; k += 256) keys[k] = src[k];
    __syncthreads();
    const int t = blockIdx.x * 64 + (tid >> 2), part = tid & 3;
    int32_t* okxy = kxy + (size_t)b * Kmax * 2;
    float* osc = score + (size_t)b * Kmax;
    if (t >= Kmax) return;
    if (t >= nsel) { if (part == 0) { okxy[2 * t] = 0; okxy[2 * t + 1] = 0; osc[t] = 0.f; } return; }
    const unsigned long long key = keys[t];
    int rank = 0;
    // part p scans the key pairs p, p + 4, ... (16-byte LDS reads, eight in flight: one dependent 8-byte read per iteration was 17 us)
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const u64x2* kp = reinterpret_cast<const u64x2*>(keys);
    const int npair = nsel >> 1;
#pragma unroll 8
    for (int j = part; j < npair; j += 4) { const u64x2 kk = kp[j]; rank += (kk[0] > key ? 1 : 0) + (kk[1] > key ? 1 : 0); }
    if ((nsel & 1) && part == 0) rank += keys[nsel - 1] > key ? 1 : 0;
    rank += __shfl_xor(rank, 1);
    rank += __shfl_xor(rank, 2);
    if (part == 0) {
        const int idx = (int)(0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull));
        okxy[2 * rank] = idx % W; okxy[2 * rank + 1] = idx / W;
        osc[rank] = __uint_as_float((unsigned int)(key >> 32));
    }
}

// Latency regime (up to four frames per call): selection AND ordering as one rank computation spread over the chip.  Every candidate's
// rank among ALL candidates of its frame (key = score bits << 32 | ~pixel index, distinct) is the number of larger keys; a candidate with
// rank < Kmax is a selected keypoint and its rank IS its output row -- no radix select, no single-workgroup tail (select_kernel: 21 us of
// barriers on one CU, + 5 us of select_rank_kernel).  A workgroup loads the frame's keys into LDS and ranks 32 candidates, eight threads
// per candidate scanning an eighth of the list each with 16-byte reads.  count <= Kmax without the unconditional top-k: row-major copy
// (the published top_k_keypoints).  More than RA_MAX candidates (a radius-4 NMS leaves at most one survivor per 5 x 5 block, 12 288 on a
// VGA frame -- but equal scores survive together, and larger frames / smaller radii exist): the key list is walked in windows of RA_MAX and the
// workgroups stride over the candidates, so the kernel is complete for any count and nothing is launched behind it.
constexpr int RA_MAX = 8192, RA_PER = 32;   // keys per LDS window (64 KB: two workgroups per CU; a VGA frame has 5 000 - 6 500 candidates); candidates ranked per workgroup and pass
__global__ __launch_bounds__(256) void select_rankall_kernel(const float* __restrict__ cand_score, const int32_t* __restrict__ cand_idx,
                                                             const int32_t* __restrict__ chunk_cnt, int nch, int HW, int W, int Kmax,
                                                             int topk_always, int32_t* __restrict__ n_out, int32_t* __restrict__ kxy,
                                                             float* __restrict__ score) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    const int b = blockIdx.y, tid = threadIdx.x;
    int count = 0;
    for (int c = 0; c < nch; ++c) count += chunk_cnt[b * nch + c];
    const float* cs = cand_score + (size_t)b * HW;
    const int32_t* ci = cand_idx + (size_t)b * HW;
    int32_t* okxy = kxy + (size_t)b * Kmax * 2;
    float* osc = score + (size_t)b * Kmax;
    const int nsel = count < Kmax ? count : Kmax;
    if (blockIdx.x == 0 && tid == 0) n_out[b] = nsel;
    const int span = count > Kmax ? count : Kmax;
    const int nwin = (count + RA_MAX - 1) / RA_MAX;
    const bool rowmajor = count <= Kmax && !topk_always;      // nothing to cut: row-major order
    auto make_key = [](float sc, int ix) { return ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)ix); };
    // keys [w0, w0 + wn) of the frame into LDS, four independent (score, index) loads in flight per thread
    auto load_window = [&](int w0, int wn) {
        for (int k0 = tid; k0 < wn; k0 += 4 * 256) {
            float sc[4]; int ix[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int k = k0 + u * 256; sc[u] = k < wn ? cs[w0 + k] : 0.f; ix[u] = k < wn ? ci[w0 + k] : 0; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int k = k0 + u * 256; if (k < wn) keys[k] = make_key(sc[u], ix[u]); }
        }
    };
    bool loaded = false;
    for (int t0 = blockIdx.x * RA_PER; t0 < span; t0 += gridDim.x * RA_PER) {
        const int t = t0 + (tid >> 3), part = tid & 7;      // eight threads per candidate, each scans an eighth of the key pairs
        if (part == 0 && t >= nsel && t < Kmax) { okxy[2 * t] = 0; okxy[2 * t + 1] = 0; osc[t] = 0.f; }   // rows [nsel, Kmax) of the padded outputs
        if (t0 >= count) continue;
        if (rowmajor) {
            if (part == 0 && t < count) { const int idx = ci[t]; okxy[2 * t] = idx % W; okxy[2 * t + 1] = idx / W; osc[t] = cs[t]; }
            continue;
        }
        const bool live = t < count;
        const unsigned long long key = live ? make_key(cs[t], ci[t]) : ~0ull;
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        const u64x2* kp = reinterpret_cast<const u64x2*>(keys);
        int rank = 0;
        for (int w = 0; w < nwin; ++w) {
            const int w0 = w * RA_MAX, wn = count - w0 < RA_MAX ? count - w0 : RA_MAX;
            if (nwin > 1 || !loaded) {
                if (loaded) __syncthreads();                 // everybody has left the previous window
                load_window(w0, wn);
                loaded = true;
                __syncthreads();
            }
            const int npair = wn >> 1;
#pragma unroll 8
            for (int j = part; j < npair; j += 8) { const u64x2 kk = kp[j]; rank += (kk[0] > key ? 1 : 0) + (kk[1] > key ? 1 : 0); }
            if ((wn & 1) && part == 0) rank += keys[wn - 1] > key ? 1 : 0;
        }
        rank += __shfl_xor(rank, 1);
        rank += __shfl_xor(rank, 2);
        rank += __shfl_xor(rank, 4);
        if (live && part == 0 && rank < Kmax) {
            const int idx = (int)(0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull));
            okxy[2 * rank] = idx % W; okxy[2 * rank + 1] = idx / W;
            osc[rank] = __uint_as_float((unsigned int)(key >> 32));
        }
    }
}

void launch_select(hipStream_t s, const float* nms, int B, int H, int W, int Kmax, float thr, float* cand_score,
                   int32_t* cand_idx, int32_t* n_out, int32_t* kxy, float* score, int32_t* chunk_cnt, bool topk_always,
                   unsigned long long* sel_keys, int32_t* sel_n) {
    int P2 = 1;
    while (P2 < Kmax) P2 <<= 1;
    const int HW = H * W, nch = (HW + SEL_CHUNK - 1) / SEL_CHUNK;   // chunk_cnt: B * nch ints of scratch
    hipLaunchKernelGGL(select_count_kernel, dim3(nch, B), dim3(256), 0, s, nms, HW, nch, thr, chunk_cnt);
    hipLaunchKernelGGL(select_compact_kernel, dim3(nch, B), dim3(256), 0, s, nms, HW, nch, thr, chunk_cnt, cand_score, cand_idx);
    // one or two frames: rank-all over the chip; more: one select workgroup per frame already runs the frames in parallel
    static const int ra_frames = tune_int("RFE_SELECT_RANKALL", 4);   // tuning build: 0 = the radix-select form at every batch size
    if (B <= ra_frames) {
        const int cap = HW < RA_MAX ? HW : RA_MAX, span = cap > Kmax ? cap : Kmax;
        static bool ls_[64];
        ensure_dynamic_lds((const void*)select_rankall_kernel, RA_MAX * 8, ls_);
        hipLaunchKernelGGL(select_rankall_kernel, dim3((span + RA_PER - 1) / RA_PER, B), dim3(256), (size_t)cap * 8, s, cand_score, cand_idx, chunk_cnt, nch, HW, W,
                           Kmax, topk_always ? 1 : 0, n_out, kxy, score);
        return;
    }
    hipLaunchKernelGGL(select_kernel, dim3(B), dim3(SEL_T), (size_t)P2 * 8, s, nms, H, W, Kmax, P2, thr,
                       cand_score, cand_idx, n_out, kxy, score, chunk_cnt, nch, topk_always ? 1 : 0, sel_keys, sel_n);
    hipLaunchKernelGGL(select_rank_kernel, dim3((Kmax + 63) / 64, B), dim3(256), (size_t)Kmax * 8, s, sel_keys, sel_n, W, Kmax, kxy, score);
}


// ---------------------------------------------------------------- the detector tail of the latency regime in ONE launch
// One to four frames per call -- the reference's own call pattern (batch 1, src/Extractors/superpoint_onnx.cc:100).  What used to be five launches
// between convPb and the ranking (softmax65_d2s 10 us, nms_fused 27.6, select_count 4.7, select_compact 5.1 at one VGA frame: 47 us in which no
// matrix instruction runs, profiles/r05_latency_kernel_stats.md) is one kernel of 1024-thread workgroups:
//   head: the 65-way softmax of the cells under the haloed tile straight from convPb's logits -- the logits of the tile's cell rows are copied into LDS
//         (coalesced, 260 B per cell), one thread per cell takes the maximum and, after the all-thread exponentials, the sum IN INDEX ORDER (the oracle's
//         order: rfo softmax), every pixel is e / sum with the same division -- bit-identical to softmax65_d2s_kernel, and the score map never visits HBM;
//   body: nms_fused_kernel's five max-pool rounds, 1024 threads instead of 256 on the same LDS planes (each pass is a handful of items per thread);
//   tail: threshold + border on the output tile, block scan, ONE atomicAdd per workgroup on the frame's candidate counter, candidates appended as
//         64-bit keys (score bits << 32 | ~pixel index) in ARBITRARY order -- select_rankall_keys_kernel places every candidate by its rank, so the order
//         of the list never reaches a result (the row-major case ranks by pixel index).
// cand_cnt: two ints per frame (count, tickets), zero between calls: the ranking kernel's last workgroup resets them (see there).
constexpr int ST_NT = 1024;
template <int TH_>
__global__ __launch_bounds__(ST_NT, 1) void sp_tail_lat_kernel(const float* __restrict__ logits, int Hc, int Wc, int border, float thr,
                                                               unsigned long long* __restrict__ cand_keys, int32_t* __restrict__ cand_cnt,
                                                               float* __restrict__ smap_out, float* __restrict__ nmap_out) {
    constexpr int NF_IH = TH_ + 2 * NF_HALO;
    constexpr int NCY = (NF_IH + 7) / 8 + 1, NCX = (NF_IW + 7) / 8 + 1;     // cell rows / columns a haloed tile can touch
    static_assert(NF_IH % 4 == 0, "quad items");
    static_assert(NCY * NCX * 65 <= 3 * NF_IH * NF_IW, "the exponentials live in the T | Mk | SS planes");
    static_assert(ST_NT == SEL_T, "block scan");
    extern __shared__ __attribute__((aligned(16))) float nf_lds[];
    float* const S = nf_lds;                       // scores
    float* const T = S + NF_IH * NF_IW;            // row-pass scratch
    float* const Mk = T + NF_IH * NF_IW;           // max_mask as 1 / 0 (-inf outside the image); at the end: the NMS'ed scores of the output tile
    float* const SS = Mk + NF_IH * NF_IW;          // supp_scores
    uint8_t* const SUP = reinterpret_cast<uint8_t*>(SS + NF_IH * NF_IW);   // supp_mask
    float* const E = T;                            // head: logits, then exponentials, of the tile's cells [cell][65]
    __shared__ float cmax[NCY * NCX], csum[NCY * NCX];
    __shared__ int wtot[ST_NT / 64];
    __shared__ int sbase;
    const int H = 8 * Hc, W = 8 * Wc, HW = H * W;
    const int b = blockIdx.z;
    const size_t fo = (size_t)b * HW;
    const int y0 = blockIdx.y * TH_ - NF_HALO, x0 = blockIdx.x * NTW - NF_HALO;
    const int tid = threadIdx.x;
    // ---- head
    const int cy_lo = y0 > 0 ? y0 >> 3 : 0, cx_lo = x0 > 0 ? x0 >> 3 : 0;
    int cy_hi = (y0 + NF_IH - 1) >> 3; cy_hi = cy_hi < Hc - 1 ? cy_hi : Hc - 1;
    int cx_hi = (x0 + NF_IW - 1) >> 3; cx_hi = cx_hi < Wc - 1 ? cx_hi : Wc - 1;
    const int ncy = cy_hi - cy_lo + 1, ncx = cx_hi - cx_lo + 1, ncell = ncy * ncx, rowlen = ncx * 65;
    {
        const float* lb = logits + ((size_t)b * Hc * Wc) * 65;
        for (int r = 0; r < ncy; ++r) {
            const float* src = lb + ((size_t)(cy_lo + r) * Wc + cx_lo) * 65;
            for (int j = tid; j < rowlen; j += ST_NT) E[r * rowlen + j] = src[j];
        }
    }
    __syncthreads();
    for (int cell = tid; cell < ncell; cell += ST_NT) {
        const float* e = E + cell * 65;
        float m = e[0];
#pragma unroll
        for (int c = 0; c < 65; ++c) m = fmaxf(m, e[c]);
        cmax[cell] = m;
    }
    __syncthreads();
    for (int i = tid; i < ncell * 65; i += ST_NT) E[i] = rfe_expf(E[i] - cmax[i / 65]);
    __syncthreads();
    for (int cell = tid; cell < ncell; cell += ST_NT) {
        const float* e = E + cell * 65;
        float sm = 0.f;
#pragma unroll
        for (int c = 0; c < 65; ++c) sm = sm + e[c];
        csum[cell] = sm;
    }
    __syncthreads();
    for (int idx = tid; idx < NF_IH * NF_IW; idx += ST_NT) {
        const int py = idx / NF_IW, px = idx % NF_IW, gy = y0 + py, gx = x0 + px;
        float v = -INFINITY;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const int cell = ((gy >> 3) - cy_lo) * ncx + ((gx >> 3) - cx_lo);
            v = E[cell * 65 + (gy & 7) * 8 + (gx & 7)] / csum[cell];
            if (smap_out && py >= NF_HALO && py < NF_HALO + TH_ && px >= NF_HALO && px < NF_HALO + NTW) smap_out[fo + (size_t)gy * W + gx] = v;
        }
        S[idx] = v;
    }
    __syncthreads();     // (also: everybody is done with E before the first row pass overwrites T)
    // ---- body: nms_fused_kernel's recurrence
    auto inside = [&](int py, int px) { const int gy = y0 + py, gx = x0 + px; return gy >= 0 && gy < H && gx >= 0 && gx < W; };
    nf_rowpass<0, NF_IH, ST_NT>(S, T, tid);
    __syncthreads();
    nf_colpass<NF_R, NF_IH, ST_NT>(T, tid, [&](int py, int px, float m) {
        const int o = py * NF_IW + px;
        Mk[o] = inside(py, px) ? (S[o] == m ? 1.f : 0.f) : -INFINITY;
    });
    __syncthreads();
    nf_rowpass<NF_R, NF_IH, ST_NT>(Mk, T, tid);
    __syncthreads();
    nf_colpass<2 * NF_R, NF_IH, ST_NT>(T, tid, [&](int py, int px, float m) {
        const int o = py * NF_IW + px;
        const bool sp = m > 0.f;
        SUP[o] = sp ? 1 : 0;
        SS[o] = inside(py, px) ? (sp ? 0.f : S[o]) : -INFINITY;
    });
    __syncthreads();
    nf_rowpass<2 * NF_R, NF_IH, ST_NT>(SS, T, tid);
    __syncthreads();
    nf_colpass<3 * NF_R, NF_IH, ST_NT>(T, tid, [&](int py, int px, float m) {
        const int o = py * NF_IW + px;
        if (inside(py, px)) Mk[o] = (Mk[o] != 0.f || (SS[o] == m && !SUP[o])) ? 1.f : 0.f;
    });
    __syncthreads();
    nf_rowpass<3 * NF_R, NF_IH, ST_NT>(Mk, T, tid);
    __syncthreads();
    nf_colpass<4 * NF_R, NF_IH, ST_NT>(T, tid, [&](int py, int px, float m) {
        const int o = py * NF_IW + px;
        const bool sp = m > 0.f;
        SUP[o] = sp ? 1 : 0;
        SS[o] = inside(py, px) ? (sp ? 0.f : S[o]) : -INFINITY;
    });
    __syncthreads();
    nf_rowpass<4 * NF_R, NF_IH, ST_NT>(SS, T, tid);
    __syncthreads();
    nf_colpass<5 * NF_R, NF_IH, ST_NT>(T, tid, [&](int py, int px, float m) {
        const int o = py * NF_IW + px;
        if (!inside(py, px)) return;                 // Mk stays -inf there: never a candidate
        const bool mk = Mk[o] != 0.f || (SS[o] == m && !SUP[o]);
        const int gy = y0 + py, gx = x0 + px;
        float v = mk ? S[o] : 0.f;
        if (gy < border || gy >= H - border || gx < border || gx >= W - border) v = -1.f;
        Mk[o] = v;                                    // this thread owns o in this pass
        if (nmap_out) nmap_out[fo + (size_t)gy * W + gx] = v;
    });
    __syncthreads();
    // ---- tail: the candidates of the TH_ x 64 output tile
    constexpr int PPT = (TH_ * NTW + ST_NT - 1) / ST_NT;
    float cv[PPT]; int ci[PPT];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = tid + k * ST_NT;
        cv[k] = -INFINITY; ci[k] = 0;
        if (p < TH_ * NTW) {
            const int py = NF_HALO + p / NTW, px = NF_HALO + p % NTW;
            const float v = Mk[py * NF_IW + px];
            cv[k] = v; ci[k] = (y0 + py) * W + (x0 + px);
        }
        cnt += cv[k] > thr ? 1 : 0;
    }
    int total;
    int pos = block_excl_scan_int(cnt, wtot, total);
    if (tid == 0) sbase = total > 0 ? atomicAdd(cand_cnt + 2 * b, total) : 0;
    __syncthreads();
    pos += sbase;
    unsigned long long* ck = cand_keys + fo;
#pragma unroll
    for (int k = 0; k < PPT; ++k)
        if (cv[k] > thr) { ck[pos] = ((unsigned long long)__float_as_uint(cv[k]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)ci[k]); ++pos; }
}

// Ranking of the unordered key list sp_tail_lat_kernel leaves: select_rankall_kernel's rank computation (see there) with
//   * the keys read as they are (one 16-byte load per two candidates, every load of a thread in flight at once: the two 4-byte gathers per key behind a
//     seven-iteration loop were 14 of that kernel's 19.6 us);
//   * the row-major case (count <= Kmax, no unconditional top-k) ranked too -- by ascending pixel index (key halves swapped), since the list has no order;
//   * the frame's (count, tickets) pair reset by the LAST workgroup of the frame to leave: every workgroup reads the count first and takes a ticket when it
//     is done, so whoever draws the last ticket knows nobody will read the count again, and the next call finds zeros without a memset on the stream.
__global__ __launch_bounds__(256) void select_rankall_keys_kernel(const unsigned long long* __restrict__ cand_keys, int32_t* __restrict__ cand_cnt,
                                                                  int HW, int W, int Kmax, int topk_always, int32_t* __restrict__ n_out,
                                                                  int32_t* __restrict__ kxy, float* __restrict__ score) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const int b = blockIdx.y, tid = threadIdx.x;
    const int count = cand_cnt[2 * b];
    const unsigned long long* ck = cand_keys + (size_t)b * HW;
    int32_t* okxy = kxy + (size_t)b * Kmax * 2;
    float* osc = score + (size_t)b * Kmax;
    const int nsel = count < Kmax ? count : Kmax;
    if (blockIdx.x == 0 && tid == 0) n_out[b] = nsel;
    const int span = count > Kmax ? count : Kmax;
    const int nwin = (count + RA_MAX - 1) / RA_MAX;
    const bool rowmajor = count <= Kmax && !topk_always;      // nothing to cut: row-major order = rank by ascending pixel index
    auto xf = [&](unsigned long long k) { return rowmajor ? (k << 32) | (k >> 32) : k; };
    auto load_window = [&](int w0, int wn) {                   // w0 is a multiple of RA_MAX: 16-byte aligned
        const u64x2* src = reinterpret_cast<const u64x2*>(ck + w0);
        u64x2* dst = reinterpret_cast<u64x2*>(keys);
        const int npair = wn >> 1;
        for (int j0 = tid; j0 < npair; j0 += 8 * 256) {
            u64x2 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = j0 + u * 256; t[u] = j < npair ? src[j] : u64x2{0ull, 0ull}; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = j0 + u * 256; if (j < npair) dst[j] = u64x2{xf(t[u][0]), xf(t[u][1])}; }
        }
        if ((wn & 1) && tid == 0) keys[wn - 1] = xf(ck[w0 + wn - 1]);
    };
    bool loaded = false;
    for (int t0 = blockIdx.x * RA_PER; t0 < span; t0 += gridDim.x * RA_PER) {
        const int t = t0 + (tid >> 3), part = tid & 7;      // eight threads per candidate, each scans an eighth of the key pairs
        if (part == 0 && t >= nsel && t < Kmax) { okxy[2 * t] = 0; okxy[2 * t + 1] = 0; osc[t] = 0.f; }   // rows [nsel, Kmax) of the padded outputs
        if (t0 >= count) continue;
        const bool live = t < count;
        const unsigned long long key = live ? xf(ck[t]) : ~0ull;
        const u64x2* kp = reinterpret_cast<const u64x2*>(keys);
        int rank = 0;
        for (int w = 0; w < nwin; ++w) {
            const int w0 = w * RA_MAX, wn = count - w0 < RA_MAX ? count - w0 : RA_MAX;
            if (nwin > 1 || !loaded) {
                if (loaded) __syncthreads();                 // everybody has left the previous window
                load_window(w0, wn);
                loaded = true;
                __syncthreads();
            }
            const int npair = wn >> 1;
#pragma unroll 8
            for (int j = part; j < npair; j += 8) { const u64x2 kk = kp[j]; rank += (kk[0] > key ? 1 : 0) + (kk[1] > key ? 1 : 0); }
            if ((wn & 1) && part == 0) rank += keys[wn - 1] > key ? 1 : 0;
        }
        rank += __shfl_xor(rank, 1);
        rank += __shfl_xor(rank, 2);
        rank += __shfl_xor(rank, 4);
        if (live && part == 0 && rank < Kmax) {
            const unsigned long long k0 = rowmajor ? (key << 32) | (key >> 32) : key;      // back to (score bits, ~index)
            const int idx = (int)(0xFFFFFFFFu - (unsigned int)(k0 & 0xFFFFFFFFull));
            okxy[2 * rank] = idx % W; okxy[2 * rank + 1] = idx / W;
            osc[rank] = __uint_as_float((unsigned int)(k0 >> 32));
        }
    }
    // ---- ticket: the last workgroup of the frame to get here zeroes (count, tickets) for the next call
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        const int tk = atomicAdd(cand_cnt + 2 * b + 1, 1);
        if (tk == (int)gridDim.x - 1) { cand_cnt[2 * b] = 0; cand_cnt[2 * b + 1] = 0; __threadfence(); }
    }
}

// false = not served (more than four frames, another NMS radius): the caller runs the separate launches
bool launch_sp_tail_lat(hipStream_t st, const float* logits, int B, int Hc, int Wc, int radius, int border, float thr, unsigned long long* cand_keys,
                        int32_t* cand_cnt, float* smap_out, float* nmap_out) {
    static const int frames = tune_int("RFE_SP_TAIL_FUSED", 4);   // tuning build: 0 = the separate launches at every batch size
    if (radius != NF_R || B > frames || B < 1) return false;
    const int H = 8 * Hc, W = 8 * Wc;
    const int gx = (W + NTW - 1) / NTW;
    auto wgs = [&](int th) { return (long long)gx * ((H + th - 1) / th) * B; };
    static const int th_env = tune_int("RFE_NMS_TH", 0);
    const int th = th_env ? th_env : (wgs(32) <= 256 ? 32 : wgs(40) <= 256 ? 40 : wgs(48) <= 256 ? 48 : 32);
#define RFE_SP_TAIL_GO(TH_)                                                                                                                  \
    do {                                                                                                                                     \
        constexpr int bytes = (TH_ + 2 * NF_HALO) * NF_IW * 17;                                                                               \
        static bool ls_[64];                                                                                                                 \
        ensure_dynamic_lds((const void*)sp_tail_lat_kernel<TH_>, bytes, ls_);                                                                \
        hipLaunchKernelGGL(sp_tail_lat_kernel<TH_>, dim3(gx, (H + TH_ - 1) / TH_, B), dim3(ST_NT), bytes, st, logits, Hc, Wc, border, thr,   \
                           cand_keys, cand_cnt, smap_out, nmap_out);                                                                         \
    } while (0)
    if (th == 48) RFE_SP_TAIL_GO(48);
    else if (th == 40) RFE_SP_TAIL_GO(40);
    else RFE_SP_TAIL_GO(32);
#undef RFE_SP_TAIL_GO
    return true;
}

// test hook (rfe_k_select_keys): the candidates of a caller's post-NMS map as an unordered key list -- pixels visited in a scrambled order, one atomic each
__global__ __launch_bounds__(256) void keys_from_map_kernel(const float* __restrict__ nms, int HW, float thr, unsigned long long* __restrict__ cand_keys,
                                                            int32_t* __restrict__ cand_cnt) {
    const int b = blockIdx.y;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= HW) return;
    const int p = (int)((t * 7919 + 13) % HW);            // 7919 is prime: a permutation of [0, HW) whenever 7919 does not divide HW
    const float v = nms[(size_t)b * HW + p];
    if (v > thr) {
        const int pos = atomicAdd(cand_cnt + 2 * b, 1);
        cand_keys[(size_t)b * HW + pos] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)p);
    }
}
void launch_keys_from_map(hipStream_t s, const float* nms, int B, int HW, float thr, unsigned long long* cand_keys, int32_t* cand_cnt) {
    hipLaunchKernelGGL(keys_from_map_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, s, nms, HW, thr, cand_keys, cand_cnt);
}

void launch_select_keys(hipStream_t s, const unsigned long long* cand_keys, int32_t* cand_cnt, int B, int H, int W, int Kmax, bool topk_always,
                        int32_t* n_out, int32_t* kxy, float* score) {
    const int HW = H * W;
    const int cap = HW < RA_MAX ? HW : RA_MAX, span = cap > Kmax ? cap : Kmax;
    static bool ls_[64];
    ensure_dynamic_lds((const void*)select_rankall_keys_kernel, RA_MAX * 8, ls_);
    hipLaunchKernelGGL(select_rankall_keys_kernel, dim3((span + RA_PER - 1) / RA_PER, B), dim3(256), (size_t)cap * 8, s, cand_keys, cand_cnt, HW, W, Kmax,
                       topk_always ? 1 : 0, n_out, kxy, score);
}

// ---------------------------------------------------------------- 256-d L2 normalisation
// canonical tree (== rfo_sumsq256): lane l owns channels 4l..4l+3, then xor butterfly 32..1
__device__ __forceinline__ float wave_sumsq256(float4 v) {
    float p = v.x * v.x;
    p = fmaf(v.y, v.y, p);
    p = fmaf(v.z, v.z, p);
    p = fmaf(v.w, v.w, p);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off);
    return p;
}

__global__ __launch_bounds__(256) void descmap_norm_kernel(float* __restrict__ dmap, int64_t cells) {
    const int64_t cell = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= cells) return;
    const int lane = threadIdx.x & 63;
    float4* p = reinterpret_cast<float4*>(dmap + cell * 256) + lane;
    float4 v = *p;
    const float d = fmaxf(sqrtf(wave_sumsq256(v)), 1e-12f);
    v.x = v.x / d; v.y = v.y / d; v.z = v.z / d; v.w = v.w / d;
    *p = v;
}
void launch_descmap_norm(hipStream_t s, float* dmap, int64_t cells) {
    hipLaunchKernelGGL(descmap_norm_kernel, dim3((unsigned)((cells + 3) / 4)), dim3(256), 0, s, dmap, cells);
}

// one wave per keypoint: bilinear grid_sample(align_corners=True, zero padding) of the normalised
// descriptor map at ((x-3.5)/(W-4.5), (y-3.5)/(H-4.5)), then L2 normalise.
__global__ __launch_bounds__(256) void desc_sample_kernel(const float* __restrict__ dmap, int Hc, int Wc, int H, int W,
                                                          const int32_t* __restrict__ n, const int32_t* __restrict__ kxy,
                                                          int Kmax, float* __restrict__ desc, uint8_t* __restrict__ desc_bin) {
    const int b = blockIdx.y;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= Kmax) return;
    const int lane = threadIdx.x & 63;
    float4* o = reinterpret_cast<float4*>(desc + ((size_t)b * Kmax + k) * 256) + lane;
    // optional second output: Frame::binarize_descriptors (src/Frame.cc:1034-1043, cv::threshold(row, 0, 1, THRESH_BINARY)):
    // u8 [K,256] = desc > 0, what ComputeBoW3 hands to the DBoW3 vocabulary -- written here so that no second pass reads desc
    uchar4* ob = desc_bin ? reinterpret_cast<uchar4*>(desc_bin + ((size_t)b * Kmax + k) * 256) + lane : nullptr;
    if (k >= n[b]) { *o = make_float4(0.f, 0.f, 0.f, 0.f); if (ob) *ob = make_uchar4(0, 0, 0, 0); return; }
    const int x = kxy[((size_t)b * Kmax + k) * 2], y = kxy[((size_t)b * Kmax + k) * 2 + 1];
    const float gx = (((float)x - 3.5f) / ((float)W - 4.5f)) * 2.0f - 1.0f;
    const float gy = (((float)y - 3.5f) / ((float)H - 4.5f)) * 2.0f - 1.0f;
    const float ix = ((gx + 1.0f) * 0.5f) * (float)(Wc - 1);
    const float iy = ((gy + 1.0f) * 0.5f) * (float)(Hc - 1);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = (fx0 + 1.0f) - ix, wy0 = (fy0 + 1.0f) - iy;
    const float wnw = wx0 * wy0, wne = wx1 * wy0, wsw = wx0 * wy1, wse = wx1 * wy1;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* mb = dmap + (size_t)b * Hc * Wc * 256;
    const bool vx0 = x0 >= 0 && x0 < Wc, vx1 = x1 >= 0 && x1 < Wc, vy0 = y0 >= 0 && y0 < Hc, vy1 = y1 >= 0 && y1 < Hc;
    const float4 nw = (vx0 && vy0) ? reinterpret_cast<const float4*>(mb + ((size_t)y0 * Wc + x0) * 256)[lane] : z;
    const float4 ne = (vx1 && vy0) ? reinterpret_cast<const float4*>(mb + ((size_t)y0 * Wc + x1) * 256)[lane] : z;
    const float4 sw = (vx0 && vy1) ? reinterpret_cast<const float4*>(mb + ((size_t)y1 * Wc + x0) * 256)[lane] : z;
    const float4 se = (vx1 && vy1) ? reinterpret_cast<const float4*>(mb + ((size_t)y1 * Wc + x1) * 256)[lane] : z;
    float4 v;
    v.x = fmaf(se.x, wse, fmaf(sw.x, wsw, fmaf(ne.x, wne, nw.x * wnw)));
    v.y = fmaf(se.y, wse, fmaf(sw.y, wsw, fmaf(ne.y, wne, nw.y * wnw)));
    v.z = fmaf(se.z, wse, fmaf(sw.z, wsw, fmaf(ne.z, wne, nw.z * wnw)));
    v.w = fmaf(se.w, wse, fmaf(sw.w, wsw, fmaf(ne.w, wne, nw.w * wnw)));
    const float d = fmaxf(sqrtf(wave_sumsq256(v)), 1e-12f);
    v.x = v.x / d; v.y = v.y / d; v.z = v.z / d; v.w = v.w / d;
    *o = v;
    if (ob) *ob = make_uchar4(v.x > 0.f, v.y > 0.f, v.z > 0.f, v.w > 0.f);
}
void launch_desc_sample(hipStream_t s, const float* dmap, int B, int Hc, int Wc, int H, int W, const int32_t* n,
                        const int32_t* kxy, int Kmax, float* desc, uint8_t* desc_bin) {
    hipLaunchKernelGGL(desc_sample_kernel, dim3((Kmax + 3) / 4, B), dim3(256), 0, s, dmap, Hc, Wc, H, W, n, kxy, Kmax, desc, desc_bin);
}

}  // namespace rfe
