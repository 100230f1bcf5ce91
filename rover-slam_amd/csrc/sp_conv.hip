// sp_conv.hip -- SuperPoint 3x3 convolutions on CDNA4 (gfx950).
//
// Replaces the conv nodes of the reference's superpoint.onnx (executed by ONNXRuntime at
// src/Extractors/superpoint_onnx.cc:133-136).  Layout: NHWC fp32 activations.
//
// conv3x3_mfma_kernel: implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact f32:
// the accumulate is bit-for-bit an fmaf chain in k order).  Per workgroup (256 threads, 4 waves):
//   output tile 8 rows x 32 cols x 64 output channels; wave w owns rows 2w,2w+1 (two 32-pixel
//   M-blocks) x two 32-channel N-blocks -> 4 accumulators of 16 VGPRs.
//   The K loop runs over chunks of 16 input channels: the haloed input tile is staged into LDS
//   channel-planar ([ci][10][34]) so that the 32 lanes of an A-fragment read 32 consecutive words
//   (bank-conflict free), the weight chunk ([144][64]) is a linear copy of a pre-packed blob.
//   Canonical reduction order kappa = ci*9 + ky*3 + kx ascending, accumulator initialised with the
//   bias -- identical to oracle/rfe_oracle.c:rfo_conv3x3, so results are bit-exact.
//   Epilogue: ReLU, optional fused 2x2/2 max-pool (all four taps of a window live in one lane:
//   D rows (r, r+1) of the two M-blocks), coalesced 128-B stores.
// Roofline: compute bound on the fp32 MFMA peak (157.3 TFLOP/s); algorithmic 2*9*Cin*Cout FLOP/px.
#include <stdlib.h>
#include "rfe_internal.h"

namespace rfe {

constexpr int TH = 8, TW = 32, IH = TH + 2, IW = TW + 2;
constexpr int TWS = IW;            // LDS row stride (words)
constexpr int PLANE = IH * TWS;    // 340 words per input-channel plane
constexpr int NT = CONV_NT;

// input channels per LDS chunk: 16 (58.6 KB LDS, 2 workgroups/CU) or 8 (29.3 KB, 3-4 workgroups/CU)
int conv_ck() {
    static const int ck = [] { const char* e = tune_env("RFE_CONV_CK"); const int v = e ? atoi(e) : CONV_CK; return (v == 8 || v == 16) ? v : CONV_CK; }();
    return ck;
}

// two images of the same weights: the implicit-GEMM tiles' [Cout/64][Cin/CK][CK*9][64] and, behind it (Cin a multiple of 16, Cout of 16), the 16 x 16 x 4 tiles'
// [Cout/16][Cin/16][9][4][16][4] (see conv3x3_t16d_kernel)
size_t packed_conv3x3_count(int cin, int cout) { return (size_t)cout * cin * 9 * ((cin % 16 == 0 && cout % 16 == 0) ? 2 : 1); }

// [Cout/64][Cin/CK][CK*9][64] : element (ct, ch, kl, j) = w[ct*64+j][ch*CK + kl/9][(kl%9)/3][kl%3]
void pack_conv3x3_weights(const float* w, int cin, int cout, std::vector<float>& out) {
    out.assign(packed_conv3x3_count(cin, cout), 0.f);
    const int CK = conv_ck(), KCH = CK * 9;
    const int nch = cin / CK, nct = cout / NT;
    for (int ct = 0; ct < nct; ++ct)
        for (int ch = 0; ch < nch; ++ch)
            for (int kl = 0; kl < KCH; ++kl)
                for (int j = 0; j < NT; ++j) {
                    int co = ct * NT + j, ci = ch * CK + kl / 9, tap = kl % 9;
                    out[(((size_t)ct * nch + ch) * KCH + kl) * NT + j] = w[((size_t)co * cin + ci) * 9 + tap];
                }
    if (cin % 16 || cout % 16) return;
    // the 16 x 16 x 4 tiles (round 6): element (block of 16 output channels b, 16-channel stage st, group g of 16 kappa, lane group q, channel co, step s) =
    // w[16 b + co][16 st + kap / 9][tap = kap % 9] with kap = 16 g + 4 s + q -- the four weights ONE lane (co, q) feeds to the four consecutive k-steps 4 g .. 4 g + 3
    // are 16 contiguous bytes, and the 64 lanes of a group read 1 KB in lane order (one conflict-free ds_read_b128 instead of four ds_read_b32).  The order in which
    // the products are ACCUMULATED does not change: k-steps ascending, kappa = 4 s36 + q inside a step.
    float* t16 = out.data() + (size_t)cout * cin * 9;
    const int nst = cin / 16;
    for (int b = 0; b < cout / 16; ++b)
        for (int st = 0; st < nst; ++st)
            for (int g = 0; g < 9; ++g)
                for (int q = 0; q < 4; ++q)
                    for (int co = 0; co < 16; ++co)
                        for (int sq = 0; sq < 4; ++sq) {
                            const int kap = 16 * g + 4 * sq + q, ci = st * 16 + kap / 9, tap = kap % 9;
                            t16[((((((size_t)b * nst + st) * 9 + g) * 4 + q) * 16 + co) * 4) + sq] = w[((size_t)(b * 16 + co) * cin + ci) * 9 + tap];
                        }
}

// per-lane A offsets for the 9 k-steps of an 18-kappa period (two input channels)
__device__ __forceinline__ void conv_a_offsets(int (&aoff)[9], int h, int wave, int col) {
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const int k0 = 2 * s, k1 = 2 * s + 1;
        const int o0 = (k0 / 9) * PLANE + ((k0 % 9) / 3) * TWS + (k0 % 9) % 3;
        const int o1 = (k1 / 9) * PLANE + ((k1 % 9) / 3) * TWS + (k1 % 9) % 3;
        aoff[s] = (h ? o1 : o0) + (2 * wave) * TWS + col;
    }
}

// one staged chunk of CK input channels: CK*9/2 k-steps x 4 MFMA per wave
template <int CK, int PLANE_ = PLANE, int BLK1_ = TWS>
__device__ __forceinline__ void conv_chunk_mma(const float* lds_in, const float* lds_w, const int (&aoff)[9], int boff,
                                               f32x16 (&acc)[2][2]) {
#pragma unroll 1
    for (int cp = 0; cp < CK / 2; ++cp) {
        const float* ap = lds_in + cp * 2 * PLANE_;
        const float* bp = lds_w + cp * 18 * NT + boff;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const float a0 = ap[aoff[s]], a1 = ap[aoff[s] + BLK1_];
            const float b0 = bp[2 * s * NT], b1 = bp[2 * s * NT + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
}

template <int CK, int NTHR = 256>
__device__ __forceinline__ void conv_stage_weights(float* lds_w, const float* src_chunk, int tid) {
    constexpr int KCH = CK * 9;
    const float4* src = reinterpret_cast<const float4*>(src_chunk);
    float4* dst = reinterpret_cast<float4*>(lds_w);
#pragma unroll
    for (int it = 0; it < (KCH * NT / 4 + NTHR - 1) / NTHR; ++it)
        if (it * NTHR + tid < KCH * NT / 4) dst[it * NTHR + tid] = src[it * NTHR + tid];
}

// epilogue.  D layout: lane holds channel (lane&31), pixel column (r&3)+8*(r>>2)+4*h of rows 2*wave+{0,1}
template <bool POOL, bool RELU>
__device__ __forceinline__ void conv_store(const f32x16 (&acc)[2][2], float* __restrict__ out, int b, int H, int W, int COUT,
                                           int x0, int y0, int co0, int wave, int col, int h) {
    if (!POOL) {
        float* out_b = out + (size_t)b * H * W * COUT;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int y = y0 + 2 * wave + mb;
            if (y >= H) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (x >= W) continue;
                float v0 = acc[mb][0][r], v1 = acc[mb][1][r];
                if (RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                float* o = out_b + ((size_t)y * W + x) * COUT + co0 + col;
                o[0] = v0; o[32] = v1;
            }
        }
    } else {
        const int Ho = H >> 1, Wo = W >> 1;
        float* out_b = out + (size_t)b * Ho * Wo * COUT;
        const int yo = (y0 >> 1) + wave;
        if (yo < Ho) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int xo = (x0 >> 1) + ((r & 3) >> 1) + 4 * (r >> 2) + 2 * h;
                if (xo >= Wo) continue;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    float v = fmaxf(fmaxf(acc[0][nb][r], acc[0][nb][r + 1]), fmaxf(acc[1][nb][r], acc[1][nb][r + 1]));
                    if (RELU) v = fmaxf(v, 0.f);
                    out_b[((size_t)yo * Wo + xo) * COUT + co0 + nb * 32 + col] = v;
                }
            }
        }
    }
}

// XCD-aware block decode (1-D grid): blocks are dealt round-robin to the 8 XCDs, each with its own L2.  The nct output-channel
// tiles of one pixel tile read the same input tile: they are made consecutive on ONE XCD (linear = ((u / 8) * nct + ct) * 8 + u % 8
// for pixel tile u), so the haloed input tile is fetched into one L2 instead of nct of them (conv3b: 2, the 256-channel heads: 4).
struct ConvBlock { int b, bx, by, ct; bool valid; };
__device__ __forceinline__ ConvBlock conv_decode(int nct, int gx, int gy, int ntiles) {
    const int L = blockIdx.x, xcd = L & 7, t = L >> 3;
    ConvBlock k;
    k.ct = t % nct;
    const int u = (t / nct) * 8 + xcd;
    k.valid = u < ntiles;
    k.bx = u % gx; k.by = (u / gx) % gy; k.b = u / (gx * gy);
    return k;
}
static inline unsigned conv_grid(int gx, int gy, int B, int nct) { return (unsigned)(((gx * gy * B + 7) / 8) * 8 * nct); }

// TAG only gives each SuperPoint layer its own kernel symbol (per-layer rows in rocprofv3 --stats)
template <int CIN, bool POOL, bool RELU, int TAG, int CK>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(
    const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ out, int H, int W, int COUT, int gx, int gy, int ntiles) {
    constexpr int KCH = CK * 9;
    __shared__ __attribute__((aligned(16))) float lds[CK * PLANE + KCH * NT];
    float* lds_in = lds;
    float* lds_w = lds + CK * PLANE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, h = lane >> 5;
    const ConvBlock blk = conv_decode(COUT / NT, gx, gy, ntiles);
    if (!blk.valid) return;
    const int b = blk.b, ct = blk.ct;
    const int x0 = blk.bx * TW, y0 = blk.by * TH;
    const int co0 = ct * NT;

    f32x16 acc[2][2];
    {
        const float b0 = bias[co0 + col], b1 = bias[co0 + 32 + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][0][r] = b0; acc[1][0][r] = b0; acc[0][1][r] = b1; acc[1][1][r] = b1; }
    }
    int aoff[9];
    conv_a_offsets(aoff, h, wave, col);
    const int boff = h * NT + col;

    const float* in_b = in + (size_t)b * H * W * CIN;
    const float* wp_ct = wp + (size_t)ct * (CIN / CK) * KCH * NT;

    // per-thread staging slots (pixel, 4-channel group) are the same for every chunk: hoist the index arithmetic
    constexpr int S_IT = (IH * IW * (CK / 4) + 255) / 256;
    int s_goff[S_IT], s_loff[S_IT];   // global element offset (-1: zero padding / unused slot), LDS word offset
#pragma unroll
    for (int it = 0; it < S_IT; ++it) {
        const int idx = tid + it * 256;
        const int cq = idx % (CK / 4), pix = idx / (CK / 4);
        const int py = pix / IW, px = pix % IW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool slot = idx < IH * IW * (CK / 4);
        s_loff[it] = slot ? (cq * 4) * PLANE + py * TWS + px : -1;
        s_goff[it] = (slot && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (gy * W + gx) * CIN + cq * 4 : -1;
    }

    // The input of TWO consecutive chunks (2 x 32 B of every pixel = half a 128-B line of an NHWC row with 64 channels) is
    // fetched together on the even pass and the second half waits in registers for the odd pass: every line of the tile is
    // requested from L2 in half as many passes (the resident tiles of an XCD, 128 x 44 KB per pass, do not fit its 4 MB L2, so
    // each pass over a line used to be a fresh fetch), and the odd passes have no global input load on their staging path.
    constexpr bool PAIR = (CIN / CK) % 2 == 0;   // (a single 16-channel chunk in the generic test instantiation: no pairing)
    float4 v_odd[S_IT];
    for (int ch = 0; ch < CIN / CK; ++ch) {
        __syncthreads();
        // ---- stage input tile: NHWC global -> channel-planar LDS (zero padding materialised)
#pragma unroll
        for (int it = 0; it < S_IT; ++it) {
            if (s_loff[it] < 0) continue;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!PAIR) {
                if (s_goff[it] >= 0) v = *reinterpret_cast<const float4*>(in_b + s_goff[it] + ch * CK);
            } else if ((ch & 1) == 0) {
                v_odd[it] = v;
                if (s_goff[it] >= 0) {
                    v = *reinterpret_cast<const float4*>(in_b + s_goff[it] + ch * CK);
                    v_odd[it] = *reinterpret_cast<const float4*>(in_b + s_goff[it] + (ch + 1) * CK);
                }
            } else {
                v = v_odd[it];
            }
            float* d = lds_in + s_loff[it];
            d[0] = v.x; d[PLANE] = v.y; d[2 * PLANE] = v.z; d[3 * PLANE] = v.w;
        }
        conv_stage_weights<CK>(lds_w, wp_ct + (size_t)ch * KCH * NT, tid);
        __syncthreads();
        conv_chunk_mma<CK>(lds_in, lds_w, aoff, boff, acc);
    }
    conv_store<POOL, RELU>(acc, out, b, H, W, COUT, x0, y0, co0, wave, col, h);
}

// ------------------------------------------------------------------------------------------
// Small-tile variant for latency-bound launches (one or two frames: the 60 x 80 layers give the tiles above only
// 30-60 workgroups for 256 CUs, conv4a at batch 1 ran at 8 TFLOP/s).  Workgroup = 4 waves covering 4 rows x 32 columns
// x 32 output channels: wave w owns row w (one M-block) and ONE accumulator, so the grid is 4x larger (180-360
// workgroups on 60 x 80) and each wave's serial MFMA chain is 4x shorter.  Reads the same packed weights (half of each
// 64-wide row).  Same reduction order (bit-exact).  POOL: the 2x2 windows span two waves (rows w, w+1), so the ReLU'd tile is
// exchanged through LDS (4 x 32 px x 32 channels, the staging buffers are free by then) and pooled from there.
constexpr int STH = 4, SNT = 32;

// ROWS (round 6): rows per tile = waves per workgroup, 4 or 5 (no pool).  240 x 320 x 64 channels of ONE frame are 1200 four-row workgroups for 1024 resident slots -- four
// rounds' worth run together, 176 stragglers follow alone -- but 960 five-row workgroups, all resident at once (conv2a of one frame).
template <int CIN, bool RELU, int CK, bool POOL = false, int ROWS = STH>
__global__ __launch_bounds__(64 * ROWS, 4) void conv3x3_small_kernel(
    const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ out, int H, int W, int COUT, int gx, int gy, int ntiles) {
    constexpr int KCH = CK * 9;
    static_assert(ROWS == STH || !POOL, "the pooling exchange is written for four rows");
    constexpr int NTH = 64 * ROWS, RIH = ROWS + 2, RPLANE = RIH * TWS;     // threads, haloed rows, words per input-channel plane
    constexpr int LDS_WORDS = (CK * RPLANE + KCH * SNT) > (POOL ? STH * TW * SNT : 0) ? (CK * RPLANE + KCH * SNT) : STH * TW * SNT;
    __shared__ __attribute__((aligned(16))) float lds[LDS_WORDS];
    float* lds_in = lds;
    float* lds_w = lds + CK * RPLANE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, h = lane >> 5;
    const ConvBlock blk = conv_decode(COUT / SNT, gx, gy, ntiles);
    if (!blk.valid) return;
    const int b = blk.b, ct = blk.ct;
    const int x0 = blk.bx * TW, y0 = blk.by * ROWS;
    const int co0 = ct * SNT;

    f32x16 acc;
    {
        const float b0 = bias[co0 + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b0;
    }
    int aoff[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const int k0 = 2 * s, k1 = 2 * s + 1;
        const int o0 = (k0 / 9) * RPLANE + ((k0 % 9) / 3) * TWS + (k0 % 9) % 3;
        const int o1 = (k1 / 9) * RPLANE + ((k1 % 9) / 3) * TWS + (k1 % 9) % 3;
        aoff[s] = (h ? o1 : o0) + wave * TWS + col;
    }
    const int boff = h * SNT + col;
    const float* in_b = in + (size_t)b * H * W * CIN;
    // packed layout [Cout/64][Cin/CK][KCH][64]: this workgroup's 32 channels are half (ct & 1) of 64-tile ct >> 1
    const float* wp_ct = wp + (size_t)(ct >> 1) * (CIN / CK) * KCH * NT + (ct & 1) * SNT;

    for (int ch = 0; ch < CIN / CK; ++ch) {
        __syncthreads();
        for (int idx = tid; idx < RIH * IW * (CK / 4); idx += NTH) {
            const int cq = idx % (CK / 4), pix = idx / (CK / 4);
            const int py = pix / IW, px = pix % IW;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const float4*>(in_b + ((size_t)gy * W + gx) * CIN + ch * CK + cq * 4);
            float* d = lds_in + (cq * 4) * RPLANE + py * TWS + px;
            d[0] = v.x; d[RPLANE] = v.y; d[2 * RPLANE] = v.z; d[3 * RPLANE] = v.w;
        }
        {
            const float* src = wp_ct + (size_t)ch * KCH * NT;
            for (int idx = tid; idx < KCH * (SNT / 4); idx += NTH) {
                const int kl = idx / (SNT / 4), q = idx % (SNT / 4);
                reinterpret_cast<float4*>(lds_w)[idx] = *reinterpret_cast<const float4*>(src + kl * NT + q * 4);
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int cp = 0; cp < CK / 2; ++cp) {
            const float* ap = lds_in + cp * 2 * RPLANE;
            const float* bp = lds_w + cp * 18 * SNT + boff;
#pragma unroll
            for (int s = 0; s < 9; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[aoff[s]], bp[2 * s * SNT], acc, 0, 0, 0);
        }
    }
    if (POOL) {
        __syncthreads();                                   // every wave is done with the staging buffers
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int xl = (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = acc[r];
            if (RELU) v = fmaxf(v, 0.f);
            lds[(wave * TW + xl) * SNT + col] = v;         // [row][column][channel]
        }
        __syncthreads();
        const int Ho = H >> 1, Wo = W >> 1;
        float* out_b = out + (size_t)b * Ho * Wo * COUT;
        const int chn = tid & 31;
#pragma unroll
        for (int it = 0; it < (STH / 2) * (TW / 2) / 8; ++it) {
            const int pp = (tid >> 5) + 8 * it, pr = pp / (TW / 2), pc = pp % (TW / 2);
            const int yo = (y0 >> 1) + pr, xo = (x0 >> 1) + pc;
            if (yo >= Ho || xo >= Wo) continue;
            const float* e = lds + ((2 * pr) * TW + 2 * pc) * SNT + chn;
            out_b[((size_t)yo * Wo + xo) * COUT + co0 + chn] = fmaxf(fmaxf(e[0], e[SNT]), fmaxf(e[TW * SNT], e[TW * SNT + SNT]));
        }
        return;
    }
    const int y = y0 + wave;
    if (y < H) {
        float* out_b = out + (size_t)b * H * W * COUT;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (x >= W) continue;
            float v = acc[r];
            if (RELU) v = fmaxf(v, 0.f);
            out_b[((size_t)y * W + x) * COUT + co0 + col] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// 16 x 16 x 4 variant for the latency regime (one or two frames -- the reference's own call pattern, batch 1:
// src/Extractors/superpoint_onnx.cc:100): the 60 x 80 layers (and the 120 x 160 layer without a pool at one frame).  conv3x3_small_kernel
// gives every wave ONE 32 x 32 x 2 accumulator, i.e. one chain of Cin * 9 / 2 = 576 dependent 64-cycle matrix instructions (22 us before
// any staging), and its 32-pixel M-blocks waste a sixth of an 80-column row.  Here a wave owns 16 pixels x 32 output channels as two
// v_mfma_f32_16x16x4_f32 accumulators (32-cycle issue, two independent chains of Cin * 9 / 4), the workgroup 4 rows x 16 columns x 32
// channels: 60 x 80 x 128 channels = 1200 waves for 1024 SIMDs instead of 720, 80 = 5 x 16 columns exactly.  The product is formed
// transposed (matrix-A operand = weights, B = pixels): a lane ends up with four consecutive output channels of one pixel (16-byte bias load
// and store).  Reduction order: k-step s covers kappa = 4 s .. 4 s + 3 (lane group q = kappa & 3), ascending inside the instruction and
// across instructions, accumulator initialised with the bias -- the oracle's order, bit-exact like every other variant.
constexpr int T16_H = 4, T16_W = 16, T16_IH = T16_H + 2, T16_IW = T16_W + 2, T16_PLANE = T16_IH * T16_IW, T16_NC = 32, T16_CK = 16;

// ------------------------------------------------------------------------------------------
// Staging: BOTH operands are copied by global_load_lds_dwordx4 into three-stage LDS rings (two 16-channel stages in flight under the matrix
// instructions of a third): no staging registers, no LDS stores, one counted s_waitcnt + one bare barrier per stage.  A first form staged
// through registers (double-buffered, next stage's loads issued before this stage's matrix instructions): 36 - 38 us per 60 x 80 layer
// against 11 us of chain time -- a stage's 72 matrix instructions are shorter than the load latency they were supposed to hide, and the
// compiler drains the vector-memory counter wherever staged registers are live -- this form 25 us (profiles/r04_ab_notes.md).
//   * weights: the packed rows [kappa][64 co] as they are; one instruction moves 8 rows x 128 B (this tile's half of the 64 channels);
//   * input: the haloed 6 x 18 pixel tile stays PIXEL-major (a pixel's 16 channels of the stage = 64 B = 4 granules; one instruction moves
//     16 pixels); granule g of pixel p is stored at g ^ ((p >> 2) & 3) -- swizzle on the source address -- so that the B-operand read of the
//     16 x-consecutive pixels of a lane group (word p * 16 + ...) is conflict-free; pixels outside the image read a zero line;
//   * every wave issues the same number of copies per stage (5 + 2; surplus slots re-copy a valid line into padding).
typedef __attribute__((address_space(3))) void* conv_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* conv_gptr_t;
constexpr int T16D_WROWS = 144, T16D_PIX = 128;    // ring-stage capacities: weight rows (18 copy slots of 8), pixels (108 used: 8 slots of 16); 3 x 26 KB = 78 KB
__device__ __attribute__((aligned(64))) float t16d_zero_line[16];   // zero-initialised: the source of every out-of-image pixel

// NC: output channels per workgroup, 32 (two accumulator chains per wave) or 16 (ONE chain per wave, twice the workgroups: round 5, for the launches whose
// 32-channel grid leaves SIMDs idle or badly balanced -- a 60 x 80 layer of one frame is 300 workgroups for 256 CUs; the arithmetic and its order are the same).
// POOL (round 6): the fused 2 x 2 / 2 max-pool of conv2b / conv3b.  In the D layout a lane holds four output channels of ONE pixel (row = wave, column = lane & 15):
// the horizontal neighbour is lane ^ 1 (one DPP move), the vertical one sits in wave ^ 1 -- the even-column lanes park their horizontal maxima in the (by then
// idle) first ring stage, one barrier, and 2 x 8 x NC / 4 threads write the pooled 2 x 8 pixel block with 16-byte stores.  max and ReLU commute: ReLU first, as
// the epilogue already does.
template <int CIN, bool RELU, int NC = T16_NC, bool POOL = false>
__global__ __launch_bounds__(256, 2) void conv3x3_t16d_kernel(
    const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ out, int H, int W, int COUT, int gx, int gy, int ntiles) {
    static_assert(NC == 32 || NC == 16, "32 or 16 output channels per workgroup");
    constexpr int NB = NC / 16;                                                 // accumulator chains per wave
    constexpr int NST = CIN / T16_CK;                                           // 16 channels = 144 kappa per stage
    constexpr int IST_W = T16D_PIX * T16_CK, WST_W = T16D_WROWS * NC;            // 2048 + 5120 (2560) words per stage
    // dynamic LDS on purpose (a static array makes the compiler order every ds_read behind ALL outstanding copies): 3 x (8 + 20) KB
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const lin = lds;
    float* const lwr = lds + 3 * IST_W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, q = lane >> 4;
    const ConvBlock blk = conv_decode(COUT / NC, gx, gy, ntiles);
    if (!blk.valid) return;
    const int b = blk.b, ct = blk.ct;
    const int x0 = blk.bx * T16_W, y0 = blk.by * T16_H;
    const int co0 = ct * NC;

    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = *reinterpret_cast<const f32x4*>(bias + co0 + nb * 16 + 4 * q);
    // pixel-operand addressing of the 9 k-steps of a 36-kappa period (4 input channels = granule c4 of the stage): kappa = 4 s + q ->
    // (channel e = kappa / 9 inside the granule, tap) -> haloed pixel p; word = p * 16 + ((c4 ^ swizzle(p)) << 2) + e
    // (round 6) all 36 word offsets are formed ONCE: on gfx950 a wave's fp32 matrix instruction keeps its SIMD to itself (tools/kbench/mfma_valu_share.hip: VALU, integer
    // and LDS instructions of any wave on that SIMD ADD to the matrix time), and the xor / shift / add per pixel read were two vector instructions per 32-cycle matrix instruction
    int poff[4][9];
#pragma unroll
    for (int s9 = 0; s9 < 9; ++s9) {
        const int kap = 4 * s9 + q, e = kap / 9, tap = kap % 9;
        const int p = (wave + tap / 3) * T16_IW + px + tap % 3;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) poff[c4][s9] = p * T16_CK + e + ((c4 ^ ((p >> 2) & 3)) << 2);
    }
    const float* in_b = in + (size_t)b * H * W * CIN;
    // weights: the 16 x 16 x 4 image of the blob (behind the implicit-GEMM image, pack_conv3x3_weights): a stage of a 16-channel block is 9 KB contiguous; one wave
    // instruction copies 1 KB = one group of 16 kappa; 9 NB slots per stage, wave w issues slots w, w + 4, .. -- WCP each, so that the vector-memory counter advances
    // alike in every wave: a surplus copy repeats the wave's previous one (same data to the same place)
    const float* wp_t16 = wp + (size_t)COUT * CIN * 9;
    constexpr int NSLOT = 9 * NB, WCP = (NSLOT + 3) / 4, BLK_W = 9 * 256;                 // slots, copies per wave, words of one 16-channel block's stage
    int w_slot[WCP];
    const float* w_src[WCP];
#pragma unroll
    for (int u = 0; u < WCP; ++u) {
        w_slot[u] = wave + 4 * u < NSLOT ? wave + 4 * u : wave + 4 * (u - 1);
        const int nbk = w_slot[u] / 9, gs = w_slot[u] % 9;
        w_src[u] = wp_t16 + ((size_t)(ct * NB + nbk) * NST) * BLK_W + gs * 256 + lane * 4;      // + st * BLK_W per stage
    }
    // input copies: lane -> pixel slot * 16 + (lane >> 2), physical granule lane & 3; wave w issues slots w, w + 4 (pixels >= 108: padding)
    const float* i_src[2];
    bool i_img[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int p = (wave + 4 * u) * 16 + (lane >> 2);
        const int py = p / T16_IW, pxx = p % T16_IW;
        const int gyy = y0 - 1 + py, gxx = x0 - 1 + pxx;
        const int g = (lane & 3) ^ ((p >> 2) & 3);                             // global granule this LDS slot holds
        i_img[u] = p < T16_PLANE && gyy >= 0 && gyy < H && gxx >= 0 && gxx < W;
        i_src[u] = i_img[u] ? in_b + ((size_t)gyy * W + gxx) * CIN + g * 4 : t16d_zero_line + g * 4;
    }
    auto issue = [&](int st) {
        float* wdst = lwr + (st % 3) * WST_W;
        float* idst = lin + (st % 3) * IST_W;
#pragma unroll
        for (int u = 0; u < WCP; ++u)
            __builtin_amdgcn_global_load_lds((conv_gptr_t)(w_src[u] + (size_t)st * BLK_W), (conv_lds_ptr_t)(wdst + w_slot[u] * 256), 16, 0, 0);
#pragma unroll
        for (int u = 0; u < 2; ++u)
            __builtin_amdgcn_global_load_lds((conv_gptr_t)(i_src[u] + (i_img[u] ? st * T16_CK : 0)), (conv_lds_ptr_t)(idst + (wave + 4 * u) * 256), 16, 0, 0);
    };
    issue(0);
    if (NST > 1) issue(1);
    for (int st = 0; st < NST; ++st) {
        // stage st has landed (this wave's copies: the WCP + 2 of stage st + 1 may still be in flight; everybody's: barrier), and every wave has left
        // stage st - 1, whose buffers the next request reuses
        // The sched_barriers pin the wait + barrier against the machine scheduler: with the stage loop fully unrolled it hoisted the first LDS read of stage
        // st ABOVE this barrier (found in round 5: the 16-channel form gave wrong descriptor rows whenever the two heads ran concurrently -- the read
        // raced the copy it was waiting for; the "memory" clobber of the asm statement does not stop the post-RA scheduler)
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < NST) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WCP + 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (st + 2 < NST) issue(st + 2);
        const float* li = lin + (st % 3) * IST_W;
        const float* lw = lwr + (st % 3) * WST_W + (q * 16 + px) * 4;        // this lane's 16 bytes of group g, block nb: + nb * BLK_W + g * 256
        // 9 groups of 16 kappa = 4 k-steps each; k-step s36 = 4 g + sq covers kappa 4 s36 + q (pixel operand: input channel 4 (s36 / 9) + .., poff[s36 / 9][s36 % 9])
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            f32x4 wf[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wf[nb] = *reinterpret_cast<const f32x4*>(lw + nb * BLK_W + g * 256);
#pragma unroll
            for (int sq = 0; sq < 4; ++sq) {
                constexpr int dummy = 0; (void)dummy;
                const int s36 = 4 * g + sq;
                const float pv = li[poff[s36 / 9][s36 % 9]];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nb][sq], pv, acc[nb], 0, 0, 0);
            }
        }
    }
    if (RELU) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { acc[nb][0] = fmaxf(acc[nb][0], 0.f); acc[nb][1] = fmaxf(acc[nb][1], 0.f); acc[nb][2] = fmaxf(acc[nb][2], 0.f); acc[nb][3] = fmaxf(acc[nb][3], 0.f); }
    }
    if (POOL) {
        // every wave has left the last stage's buffers before stage 0's are overwritten: the last stage is (NST - 1) % 3, stage 0's buffer is only reused when that is
        // not 0 -- a barrier makes it safe either way
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        float* ex = lds;                                    // [4 rows][8 pooled columns][NC]
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 hm;
#pragma unroll
            for (int e = 0; e < 4; ++e) hm[e] = fmaxf(acc[nb][e], __shfl_xor(acc[nb][e], 1));
            if (!(px & 1)) *reinterpret_cast<f32x4*>(ex + (wave * 8 + (px >> 1)) * NC + nb * 16 + 4 * q) = hm;
        }
        __syncthreads();
        constexpr int CG = NC / 4;                          // 16-byte channel groups per pixel
        if (tid < 2 * 8 * CG) {
            const int cg = tid % CG, ppx = (tid / CG) & 7, prow = tid / (8 * CG);
            const f32x4 a = *reinterpret_cast<const f32x4*>(ex + ((2 * prow) * 8 + ppx) * NC + cg * 4);
            const f32x4 c = *reinterpret_cast<const f32x4*>(ex + ((2 * prow + 1) * 8 + ppx) * NC + cg * 4);
            const int Ho = H >> 1, Wo = W >> 1, yo = (y0 >> 1) + prow, xo = (x0 >> 1) + ppx;
            if (yo < Ho && xo < Wo)
                *reinterpret_cast<f32x4*>(out + (((size_t)b * Ho + yo) * Wo + xo) * COUT + co0 + cg * 4) =
                    f32x4{fmaxf(a[0], c[0]), fmaxf(a[1], c[1]), fmaxf(a[2], c[2]), fmaxf(a[3], c[3])};
        }
        return;
    }
    const int y = y0 + wave, x = x0 + px;
    if (y < H && x < W) {
        float* o = out + (((size_t)b * H + y) * W + x) * COUT + co0 + 4 * q;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) *reinterpret_cast<f32x4*>(o + nb * 16) = acc[nb];
    }
}

// ------------------------------------------------------------------------------------------
// Composite-tile variant of conv3x3_mfma_kernel for batches of feature maps that the 8 x 32 tile does not divide (the 60 x 80 grid
// of 640 x 480 frames: 7.5 x 2.5 tiles, 22 % of the MFMAs wasted).  The frames are laid out on a VIRTUAL canvas -- two frames side by
// side (2W columns), the pairs stacked (ceil(B/2) * H rows) -- and the canvas is cut into 8 x 32 tiles, so a tile may consist of up to
// 2 x 2 pieces that come from different frames (rows 56-59 of one frame pair + rows 0-3 of the next, columns 64-79 of the left frame +
// columns 0-15 of the right one).  Every piece is staged with its own zero-padded halo: the LDS tile is 12 x 36 instead of 10 x 34
// and a lane's A offsets just start two rows / columns further in when its pixel lies in the second piece.  160 x 1020 canvas for 33
// frames: 640 tiles, 96.7 % of the MFMA rows are real pixels.  Same per-pixel reduction order as every other variant (bit-exact).
constexpr int CIH = TH + 4, CIW = TW + 4, CPLANE = CIH * CIW;
template <int CIN, bool RELU, int TAG, int CK>
__global__ __launch_bounds__(256, 2) void conv3x3_comp_kernel(
    const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ out, int B, int H, int W, int COUT, int gx, int gy, int ntiles) {
    constexpr int KCH = CK * 9;
    __shared__ __attribute__((aligned(16))) float lds[CK * CPLANE + KCH * NT];
    float* lds_in = lds;
    float* lds_w = lds + CK * CPLANE;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, h = lane >> 5;
    const ConvBlock blk = conv_decode(COUT / NT, gx, gy, ntiles);
    if (!blk.valid) return;
    const int ct = blk.ct, co0 = ct * NT;
    const int x0 = blk.bx * TW, y0 = blk.by * TH;              // canvas coordinates
    const int pair0 = y0 / H, yin = y0 % H, half0 = x0 / W, xin = x0 % W;
    const int rb = (H - yin) < TH ? (H - yin) : TH;            // output rows [0, rb) = piece 0, [rb, 8) = piece 1 (next frame pair)
    const int cb = (W - xin) < TW ? (W - xin) : TW;            // output columns [0, cb) = piece 0, [cb, 32) = piece 1 (right frame)
    auto frame_of = [&](int pr, int pc) { const int hf = half0 + pc; const int f = 2 * (pair0 + pr) + hf; return (hf <= 1 && f < B) ? f : -1; };
    if (frame_of(0, 0) < 0 && frame_of(0, 1) < 0 && frame_of(1, 0) < 0 && frame_of(1, 1) < 0) return;   // tile holds no real pixel

    f32x16 acc[2][2];
    {
        const float b0 = bias[co0 + col], b1 = bias[co0 + 32 + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][0][r] = b0; acc[1][0][r] = b0; acc[0][1][r] = b1; acc[1][1][r] = b1; }
    }
    int aoff[9];
    {
        const int r0 = 2 * wave;                                // rb is even (H even): rows 2w and 2w+1 lie in the same piece
        const int base = (r0 + (r0 >= rb ? 2 : 0)) * CIW + col + (col >= cb ? 2 : 0);
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int k0 = 2 * s, k1 = 2 * s + 1;
            const int o0 = (k0 / 9) * CPLANE + ((k0 % 9) / 3) * CIW + (k0 % 9) % 3;
            const int o1 = (k1 / 9) * CPLANE + ((k1 % 9) / 3) * CIW + (k1 % 9) % 3;
            aoff[s] = (h ? o1 : o0) + base;
        }
    }
    const int boff = h * NT + col;
    const float* wp_ct = wp + (size_t)ct * (CIN / CK) * KCH * NT;

    constexpr int S_IT = (CIH * CIW * (CK / 4) + 255) / 256;
    int s_goff[S_IT], s_loff[S_IT];   // global element offset (-1: zero padding), LDS word offset (-1: unused slot)
#pragma unroll
    for (int it = 0; it < S_IT; ++it) {
        const int idx = tid + it * 256;
        const int cq = idx % (CK / 4), pix = idx / (CK / 4);
        const int lr = pix / CIW, lc = pix % CIW;
        const int pr = lr >= rb + 2, hr = lr - (pr ? rb + 2 : 0), nr = pr ? TH - rb : rb;   // haloed row hr of a piece with nr rows
        const int pc = lc >= cb + 2, hc = lc - (pc ? cb + 2 : 0), nc = pc ? TW - cb : cb;
        const bool slot = idx < CIH * CIW * (CK / 4) && nr > 0 && nc > 0 && hr < nr + 2 && hc < nc + 2;
        const int y = (pr ? 0 : yin) + hr - 1, x = (pc ? 0 : xin) + hc - 1;
        const int f = frame_of(pr, pc);
        s_loff[it] = slot ? (cq * 4) * CPLANE + lr * CIW + lc : -1;
        s_goff[it] = (slot && f >= 0 && y >= 0 && y < H && x >= 0 && x < W) ? ((f * H + y) * W + x) * CIN + cq * 4 : -1;
    }
    static_assert((CIN / CK) % 2 == 0, "paired chunk fetch");
    float4 v_odd[S_IT];
    for (int ch = 0; ch < CIN / CK; ++ch) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < S_IT; ++it) {
            if (s_loff[it] < 0) continue;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((ch & 1) == 0) {
                v_odd[it] = v;
                if (s_goff[it] >= 0) {
                    v = *reinterpret_cast<const float4*>(in + s_goff[it] + ch * CK);
                    v_odd[it] = *reinterpret_cast<const float4*>(in + s_goff[it] + (ch + 1) * CK);
                }
            } else {
                v = v_odd[it];
            }
            float* d = lds_in + s_loff[it];
            d[0] = v.x; d[CPLANE] = v.y; d[2 * CPLANE] = v.z; d[3 * CPLANE] = v.w;
        }
        conv_stage_weights<CK>(lds_w, wp_ct + (size_t)ch * KCH * NT, tid);
        __syncthreads();
        conv_chunk_mma<CK, CPLANE, CIW>(lds_in, lds_w, aoff, boff, acc);
    }
    // epilogue: D layout as in conv_store, every pixel mapped back to (frame, y, x)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int r = 2 * wave + mb, pr = r >= rb;
        const int y = pr ? r - rb : yin + r;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = (q & 3) + 8 * (q >> 2) + 4 * h, pc = c >= cb;
            const int x = pc ? c - cb : xin + c;
            const int f = frame_of(pr, pc);
            if (f < 0) continue;
            float v0 = acc[mb][0][q], v1 = acc[mb][1][q];
            if (RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
            float* o = out + (((size_t)f * H + y) * W + x) * COUT + co0 + col;
            o[0] = v0; o[32] = v1;
        }
    }
}

// ------------------------------------------------------------------------------------------
// conv1a + conv1b fused: the 64-channel input tile of conv1b is never read from HBM; every staged
// 8-channel chunk is recomputed in LDS from the u8 image tile (NormalizeImage * 1/255, conv1a 1->64,
// bias, ReLU -- same fmaf chain as conv1a_u8_kernel / the oracle, so still bit-exact), then conv1b +
// ReLU + 2x2 max-pool as above.  Removes the 78.6 MB/frame round trip of the largest activation.
// PX = uint8_t: the reference's NormalizeImage (x * 1/255) is applied here; PX = float: an already normalised CV_32F image as the reference's
// Extractor_Inference takes it (src/Extractors/superpoint_onnx.cc:88-118) is used as it is.  stride in pixels.
template <typename PX>
__device__ __forceinline__ float px_value(PX v) {
    if constexpr (sizeof(PX) == 1) return (float)v * 0.003921568859368563f;
    else return v;
}

template <int CK, typename PX>
__global__ __launch_bounds__(256, 2) void conv1ab_fused_kernel(
    const PX* __restrict__ img, int stride, const float* __restrict__ w1a /*[9][64]*/, const float* __restrict__ b1a,
    const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ out, int H, int W, long long frame_step /*pixels from frame b to b + 1*/) {
    constexpr int CIN = 64, COUT = 64, KCH = CK * 9;
    constexpr int MH = TH + 4, MW = TW + 4;   // image tile with a 2-pixel halo
    __shared__ __attribute__((aligned(16))) float lds[CK * PLANE + KCH * NT + MH * MW];
    float* lds_in = lds;
    float* lds_w = lds + CK * PLANE;
    float* lds_img = lds_w + KCH * NT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, h = lane >> 5;
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;

    f32x16 acc[2][2];
    {
        const float b0 = bias[col], b1 = bias[32 + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][0][r] = b0; acc[1][0][r] = b0; acc[0][1][r] = b1; acc[1][1][r] = b1; }
    }
    int aoff[9];
    conv_a_offsets(aoff, h, wave, col);
    const int boff = h * NT + col;

    const PX* im = img + (long long)b * frame_step;     // usually stride * H; a stereo pair hands over two separate views (rfe_stereo_frame_dev)
    for (int idx = tid; idx < MH * MW; idx += 256) {
        const int py = idx / MW, px = idx % MW;
        const int gy = y0 - 2 + py, gx = x0 - 2 + px;
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = px_value(im[(size_t)gy * stride + gx]);
        lds_img[idx] = v;
    }

    // per-thread pixel slots of the haloed tile are the same for every chunk: hoist index arithmetic and the
    // nine image taps (they only depend on the pixel), leaving 9 fmaf per channel in the chunk loop
    constexpr int P_IT = (IH * IW + 255) / 256;
    int p_loff[P_IT]; bool p_inb[P_IT]; float p_iv[P_IT][9];
    __syncthreads();   // image tile complete
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
        const int p = tid + it * 256;
        const int py = p / IW, px = p % IW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        p_loff[it] = p < IH * IW ? py * TWS + px : -1;
        p_inb[it] = gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
        for (int k = 0; k < 9; ++k) p_iv[it][k] = p < IH * IW ? lds_img[(py + k / 3) * MW + px + k % 3] : 0.f;
    }

    for (int ch = 0; ch < CIN / CK; ++ch) {
        __syncthreads();
        // ---- conv1a for CK channels on the haloed tile; pixels outside the image are conv1b's zero padding
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            if (p_loff[it] < 0) continue;
#pragma unroll
            for (int e = 0; e < CK; ++e) {
                const int c = ch * CK + e;
                float a = b1a[c];
#pragma unroll
                for (int k = 0; k < 9; ++k) a = fmaf(p_iv[it][k], w1a[k * 64 + c], a);
                lds_in[e * PLANE + p_loff[it]] = p_inb[it] ? fmaxf(a, 0.f) : 0.f;
            }
        }
        conv_stage_weights<CK>(lds_w, wp + (size_t)ch * KCH * NT, tid);
        __syncthreads();
        conv_chunk_mma<CK>(lds_in, lds_w, aoff, boff, acc);
    }
    conv_store<true, true>(acc, out, b, H, W, COUT, x0, y0, 0, wave, col, h);
}

void launch_conv1ab_fused(hipStream_t s, const void* img, bool img_f32, int stride, int B, int H, int W, const float* w1a,
                          const float* b1a, const float* wp, const float* bias, float* out, long long frame_step) {
    dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, B);
    if (frame_step == 0) frame_step = (long long)stride * H;
    if (img_f32) {
        if (conv_ck() == 8)
            hipLaunchKernelGGL((conv1ab_fused_kernel<8, float>), grid, dim3(256), 0, s, (const float*)img, stride, w1a, b1a, wp, bias, out, H, W, frame_step);
        else
            hipLaunchKernelGGL((conv1ab_fused_kernel<16, float>), grid, dim3(256), 0, s, (const float*)img, stride, w1a, b1a, wp, bias, out, H, W, frame_step);
        return;
    }
    if (conv_ck() == 8)
        hipLaunchKernelGGL((conv1ab_fused_kernel<8, uint8_t>), grid, dim3(256), 0, s, (const uint8_t*)img, stride, w1a, b1a, wp, bias, out, H, W, frame_step);
    else
        hipLaunchKernelGGL((conv1ab_fused_kernel<16, uint8_t>), grid, dim3(256), 0, s, (const uint8_t*)img, stride, w1a, b1a, wp, bias, out, H, W, frame_step);
}

#define RFE_CONV_LAUNCH(CIN, POOL, RELU, TAG)                                                                            \
    do {                                                                                                                \
        if (ck8) hipLaunchKernelGGL((conv3x3_mfma_kernel<CIN, POOL, RELU, TAG, 8>), dim3(conv_grid(gx, gy, B, cout / NT)), dim3(256), 0, s, in, wp, bias, out, H, W, cout, gx, gy, gx * gy * B); \
        else hipLaunchKernelGGL((conv3x3_mfma_kernel<CIN, POOL, RELU, TAG, 16>), dim3(conv_grid(gx, gy, B, cout / NT)), dim3(256), 0, s, in, wp, bias, out, H, W, cout, gx, gy, gx * gy * B);   \
    } while (0)

// tag: SuperPoint layer id (L_1B .. L_DA) for the production path, 0 for the generic test hook
void launch_conv3x3(hipStream_t s, const float* in, int B, int H, int W, int cin, const float* wp,
                    const float* bias, int cout, bool relu, bool pool, float* out, int tag) {
    const int gx = (W + TW - 1) / TW, gy = (H + TH - 1) / TH;
    const bool ck8 = conv_ck() == 8;
    // latency regime: too few 8 x 32 x 64 tiles to occupy the chip -> 4 x 32 x 32 tiles (4x the workgroups)
    static const int small_thr = tune_int("RFE_CONV_SMALL", 700);   // 0 disables; 700: conv2a / conv2b of two VGA frames (600 tiles) take the small tile (117 -> 111 us each), two 752 x 480 frames (720 tiles) do not (measured slower: 2.613 against 2.595 ms per stereo frame)
    // ... and, among those, the layers WITHOUT a pool of up to 12 000 pixels (Cin = 128: conv4a / conv4b / convPa / convDa of one or two frames, 60 x 80 or
    // 60 x 94 each) or 20 000 pixels (Cin = 64: conv3a of one frame): 16 x 16 x 4 tiles with LDS-DMA rings.  Measured at 640 x 480, event-timed stages
    // (profiles/r04_ab_notes.md): one frame conv4a / 4b 39 -> 29 us, convPa / Da 49 -> 45, conv3a 41 -> 36; two frames conv4a / 4b 54 -> 45, convPa / Da
    // 74 -> 68; beyond those budgets the 32-channel tiles re-read the weights too often (conv3a at two frames 60 -> 66 us) and conv3x3_small_kernel stays.
    static const int t16_px = tune_int("RFE_CONV_T16", 12000), t16_px64 = tune_int("RFE_CONV_T16_64", 20000);   // pixel budgets; 0 disables (tuning build)
    if (ck8 && relu && !pool && (long long)gx * gy * B * (cout / NT) < small_thr && (cin == 64 || cin == 128) &&
        (long long)B * H * W <= (cin == 64 ? t16_px64 : t16_px) && cout % T16_NC == 0) {
        const int sx = (W + T16_W - 1) / T16_W, sy = (H + T16_H - 1) / T16_H;
        // 16 output channels per workgroup (one accumulator chain per wave, twice the workgroups, 48 KB of LDS: three workgroups per CU) while the
        // 32-channel grid is below nc16_thr workgroups: a 60 x 80 layer of ONE frame is 300 (conv4a / 4b) or 600 (convPa / Da) workgroups of 32 channels
        // for 256 CUs x 2 -- 88 CUs run three, 168 run two, and a wave alone on its SIMD issues its chain at 40 instead of 32 cycles
        // Measured on one box (tools/tune_sweep.py, 200 steps, off / 700 / 1300 / 2500): one 640 x 480 frame 0.630 -> 0.605 ms (conv4a / 4b 28.7 -> 22.3 us,
        // convPa / Da 45 -> 33.7), one pair 2.314 -> 2.276, one 752 x 480 stereo frame 2.517 -> 2.487 (2500: convPa / Da of two 60 x 94 maps too); the
        // 64-channel-input layer (conv3a of one frame, 1200 workgroups) does not gain (36.6 -> 37.2 us) and keeps 32.
        static const int nc16_thr = tune_int("RFE_CONV_T16_NC16", 1000);   // round 6 (b128 weight fragments): convPa / Da of TWO frames (1200 - 1440 workgroups of 32 channels) are better off with 32 (60.2 -> 58.4, 71 -> 69 us), everything smaller with 16   // 0 disables (tuning build A/B)
        if (cin == 128 && (long long)sx * sy * B * (cout / T16_NC) < nc16_thr && cout % 16 == 0) {
            const dim3 g16(conv_grid(sx, sy, B, cout / 16));
            constexpr int b16 = 3 * (T16D_PIX * T16_CK + T16D_WROWS * 16) * 4;        // 51 KB
            static bool l16_[2][64];
            ensure_dynamic_lds((const void*)conv3x3_t16d_kernel<128, true, 16>, b16, l16_[0]);
            hipLaunchKernelGGL((conv3x3_t16d_kernel<128, true, 16>), g16, dim3(256), b16, s, in, wp, bias, out, H, W, cout, sx, sy, sx * sy * B);
            return;
        }
        const dim3 gs(conv_grid(sx, sy, B, cout / T16_NC));
        constexpr int bytes = 3 * (T16D_PIX * T16_CK + T16D_WROWS * T16_NC) * 4;   // 78 KB: two workgroups per CU
        static bool ls_[2][64];
        if (cin == 128) {
            ensure_dynamic_lds((const void*)conv3x3_t16d_kernel<128, true>, bytes, ls_[0]);
            hipLaunchKernelGGL((conv3x3_t16d_kernel<128, true>), gs, dim3(256), bytes, s, in, wp, bias, out, H, W, cout, sx, sy, sx * sy * B);
        } else {
            ensure_dynamic_lds((const void*)conv3x3_t16d_kernel<64, true>, bytes, ls_[1]);
            hipLaunchKernelGGL((conv3x3_t16d_kernel<64, true>), gs, dim3(256), bytes, s, in, wp, bias, out, H, W, cout, sx, sy, sx * sy * B);
        }
        return;
    }
    // the pooling layer with 128 input channels (conv3b) of ONE frame on the 16 x 16 x 4 tiles with 16 output channels per workgroup (round 6,
    // conv3x3_t16d_kernel<128, true, 16, POOL>): 120 x 160 x 128 channels = 2400 workgroups of one accumulator chain per wave instead of 600 workgroups of
    // 32 x 32 x 2 chains that leave 88 CUs with three and 168 with two.  Measured (tools/tune_sweep.py, c2, 200 steps, two alternating runs, profiles/r06_ab_notes.md):
    // conv3b 73.4 -> 64.0 us (32 channels per workgroup: 69.3); two frames with 32 channels 110 -> 128 us, the 64-input-channel layer (conv2b, 240 x 320)
    // 59.4 -> 68.3 us -- neither is taken.  Pixel budgets per Cin, 0 = off.
    static const int t16p_px64 = tune_int("RFE_CONV_T16P_64", 0), t16p_px128 = tune_int("RFE_CONV_T16P_128", 20000), t16p_nc16 = tune_int("RFE_CONV_T16P_NC16", 1);
    if (ck8 && relu && pool && (cin == 64 || cin == 128) && (long long)B * H * W <= (cin == 64 ? t16p_px64 : t16p_px128) && cout % T16_NC == 0 && H % 2 == 0 && W % 2 == 0) {
        const int sx = (W + T16_W - 1) / T16_W, sy = (H + T16_H - 1) / T16_H;
        static bool lp_[4][64];
        if (t16p_nc16) {
            constexpr int b16 = 3 * (T16D_PIX * T16_CK + T16D_WROWS * 16) * 4;
            const dim3 g16(conv_grid(sx, sy, B, cout / 16));
            if (cin == 128) { ensure_dynamic_lds((const void*)conv3x3_t16d_kernel<128, true, 16, true>, b16, lp_[0]);
                              hipLaunchKernelGGL((conv3x3_t16d_kernel<128, true, 16, true>), g16, dim3(256), b16, s, in, wp, bias, out, H, W, cout, sx, sy, sx * sy * B); }
            else { ensure_dynamic_lds((const void*)conv3x3_t16d_kernel<64, true, 16, true>, b16, lp_[1]);
                   hipLaunchKernelGGL((conv3x3_t16d_kernel<64, true, 16, true>), g16, dim3(256), b16, s, in, wp, bias, out, H, W, cout, sx, sy, sx * sy * B); }
            return;
        }
        constexpr int bytes = 3 * (T16D_PIX * T16_CK + T16D_WROWS * T16_NC) * 4;
        const dim3 gs(conv_grid(sx, sy, B, cout / T16_NC));
        if (cin == 128) { ensure_dynamic_lds((const void*)conv3x3_t16d_kernel<128, true, T16_NC, true>, bytes, lp_[2]);
                          hipLaunchKernelGGL((conv3x3_t16d_kernel<128, true, T16_NC, true>), gs, dim3(256), bytes, s, in, wp, bias, out, H, W, cout, sx, sy, sx * sy * B); }
        else { ensure_dynamic_lds((const void*)conv3x3_t16d_kernel<64, true, T16_NC, true>, bytes, lp_[3]);
               hipLaunchKernelGGL((conv3x3_t16d_kernel<64, true, T16_NC, true>), gs, dim3(256), bytes, s, in, wp, bias, out, H, W, cout, sx, sy, sx * sy * B); }
        return;
    }
    if (ck8 && (!pool || relu) && (long long)gx * gy * B * (cout / NT) < small_thr && (cin == 64 || cin == 128)) {
        const int sx = (W + TW - 1) / TW, sy = (H + STH - 1) / STH;
        const dim3 gs(conv_grid(sx, sy, B, cout / SNT));
        if (pool) {   // (every pooling layer of SuperPoint has a ReLU)
            if (cin == 128) hipLaunchKernelGGL((conv3x3_small_kernel<128, true, 8, true>), gs, dim3(256), 0, s, in, wp, bias, out, H, W, cout, sx, sy, sx * sy * B);
            else hipLaunchKernelGGL((conv3x3_small_kernel<64, true, 8, true>), gs, dim3(256), 0, s, in, wp, bias, out, H, W, cout, sx, sy, sx * sy * B);
            return;
        }
        // five-row tiles (round 6, tuning switch): the 64-channel pool-less layer of ONE 240 x 320 map (conv2a of one frame) as 960 workgroups of five waves
        static const int r5_px = tune_int("RFE_CONV_SMALL_R5", 0);   // pixel budget, 0 = off
        if (cin == 64 && relu && (long long)B * H * W <= r5_px) {
            const int sy5 = (H + 4) / 5;
            hipLaunchKernelGGL((conv3x3_small_kernel<64, true, 8, false, 5>), dim3(conv_grid(sx, sy5, B, cout / SNT)), dim3(320), 0, s, in, wp, bias, out, H, W, cout, sx, sy5, sx * sy5 * B);
            return;
        }
        if (cin == 128 && relu) hipLaunchKernelGGL((conv3x3_small_kernel<128, true, 8>), gs, dim3(256), 0, s, in, wp, bias, out, H, W, cout, sx, sy, sx * sy * B);
        else if (cin == 128) hipLaunchKernelGGL((conv3x3_small_kernel<128, false, 8>), gs, dim3(256), 0, s, in, wp, bias, out, H, W, cout, sx, sy, sx * sy * B);
        else if (relu) hipLaunchKernelGGL((conv3x3_small_kernel<64, true, 8>), gs, dim3(256), 0, s, in, wp, bias, out, H, W, cout, sx, sy, sx * sy * B);
        else hipLaunchKernelGGL((conv3x3_small_kernel<64, false, 8>), gs, dim3(256), 0, s, in, wp, bias, out, H, W, cout, sx, sy, sx * sy * B);
        return;
    }
    // 60 x 80 grid (conv4a/4b, convPa/Da), TFLOP/s at batch 33: plain 8x32 tiles 94 / 103 (22 % of the MFMA rows are padding); round-2a
    // tilings that divide the grid exactly with M-blocks of 2 rows x 16 columns (4x80 tile of 5 waves, 12x16 tile of 3 waves) 99 / 110;
    // composite 8x32 tiles over the two-frames-wide canvas 125 / 127 -> adopted, the 2x16-block kernels are gone (profiles/r02_ab_notes.md)
    static const int comp = tune_int("RFE_CONV_COMP", 1);   // composite tiles over the two-frames-wide canvas (0: round-2a tilings)
    if (comp && ck8 && !pool && relu && cin == 128 && (tag == L_4A || tag == L_4B || tag == L_PA || tag == L_DA) && B >= 2 &&
        H % 2 == 0 && W >= TW && H >= TH && (H % TH != 0 || W % TW != 0) &&
        (long long)B * H * W * cin < (1ll << 31)) {   // 32-bit element offsets over the whole batch inside the kernel
        const int cx = (2 * W + TW - 1) / TW, cy = (((B + 1) / 2) * H + TH - 1) / TH;
        const dim3 gc(conv_grid(cx, cy, 1, cout / NT));
        switch (tag) {
            case L_4A: hipLaunchKernelGGL((conv3x3_comp_kernel<128, true, L_4A, 8>), gc, dim3(256), 0, s, in, wp, bias, out, B, H, W, cout, cx, cy, cx * cy); return;
            case L_4B: hipLaunchKernelGGL((conv3x3_comp_kernel<128, true, L_4B, 8>), gc, dim3(256), 0, s, in, wp, bias, out, B, H, W, cout, cx, cy, cx * cy); return;
            case L_PA: hipLaunchKernelGGL((conv3x3_comp_kernel<128, true, L_PA, 8>), gc, dim3(256), 0, s, in, wp, bias, out, B, H, W, cout, cx, cy, cx * cy); return;
            default: hipLaunchKernelGGL((conv3x3_comp_kernel<128, true, L_DA, 8>), gc, dim3(256), 0, s, in, wp, bias, out, B, H, W, cout, cx, cy, cx * cy); return;
        }
    }
    switch (tag) {
        case L_1B: RFE_CONV_LAUNCH(64, true, true, L_1B); return;
        case L_2A: RFE_CONV_LAUNCH(64, false, true, L_2A); return;
        case L_2B: RFE_CONV_LAUNCH(64, true, true, L_2B); return;
        case L_3A: RFE_CONV_LAUNCH(64, false, true, L_3A); return;
        case L_3B: RFE_CONV_LAUNCH(128, true, true, L_3B); return;
        case L_4A: RFE_CONV_LAUNCH(128, false, true, L_4A); return;
        case L_4B: RFE_CONV_LAUNCH(128, false, true, L_4B); return;
        case L_PA: RFE_CONV_LAUNCH(128, false, true, L_PA); return;
        case L_DA: RFE_CONV_LAUNCH(128, false, true, L_DA); return;
        default: break;
    }
    // generic (kernel-level test hook): any Cin in {16,32,64,128}, any relu/pool combination
#define RFE_CONV_GEN(CIN)                                                    \
    do {                                                                     \
        if (pool && relu) RFE_CONV_LAUNCH(CIN, true, true, 0);               \
        else if (relu) RFE_CONV_LAUNCH(CIN, false, true, 0);                 \
        else if (pool) RFE_CONV_LAUNCH(CIN, true, false, 0);                 \
        else RFE_CONV_LAUNCH(CIN, false, false, 0);                          \
    } while (0)
    if (cin == 64) RFE_CONV_GEN(64);
    else if (cin == 128) RFE_CONV_GEN(128);
    else if (cin == 16) RFE_CONV_GEN(16);
    else if (cin == 32) RFE_CONV_GEN(32);
}

// ------------------------------------------------------------------------------------------
// conv1a: u8 image -> (x * 1/255) -> conv3x3 1->64 + bias + ReLU, NHWC out.  HBM bound (writes
// 256 B per pixel).  16 lanes per pixel, 4 output channels per lane: a wave stores 1 KB contiguous.
// NormalizeImage (reference src/Matchers/transform.cpp:11) is fused here.
template <typename PX>
__global__ __launch_bounds__(256) void conv1a_u8_kernel(const PX* __restrict__ img, int stride,
                                                        int H, int W, const float* __restrict__ w9x64,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ out) {
    const int b = blockIdx.y;
    const int cg = threadIdx.x & 15;  // channel group: channels 4cg..4cg+3
    float wr[9][4], br[4];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) wr[k][e] = w9x64[k * 64 + cg * 4 + e];
#pragma unroll
    for (int e = 0; e < 4; ++e) br[e] = bias[cg * 4 + e];
    const PX* im = img + (size_t)b * stride * H;
    float* ob = out + (size_t)b * H * W * 64;
    const int npix = H * W;
    for (int p = blockIdx.x * 16 + (threadIdx.x >> 4); p < npix; p += gridDim.x * 16) {
        const int y = p / W, x = p % W;
        float a[4] = {br[0], br[1], br[2], br[3]};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int gy = y + ky - 1, gx = x + kx - 1;
                float v = 0.f;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = px_value(im[(size_t)gy * stride + gx]);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = fmaf(v, wr[ky * 3 + kx][e], a[e]);
            }
        float4 o = make_float4(fmaxf(a[0], 0.f), fmaxf(a[1], 0.f), fmaxf(a[2], 0.f), fmaxf(a[3], 0.f));
        *reinterpret_cast<float4*>(ob + (size_t)p * 64 + cg * 4) = o;
    }
}

void launch_conv1a_u8(hipStream_t s, const void* img, bool img_f32, int stride, int B, int H, int W,
                      const float* w9x64, const float* bias, float* out) {
    int blocks = (H * W + 15) / 16;
    if (blocks > 4096) blocks = 4096;
    if (img_f32) hipLaunchKernelGGL(conv1a_u8_kernel<float>, dim3(blocks, B), dim3(256), 0, s, (const float*)img, stride, H, W, w9x64, bias, out);
    else hipLaunchKernelGGL(conv1a_u8_kernel<uint8_t>, dim3(blocks, B), dim3(256), 0, s, (const uint8_t*)img, stride, H, W, w9x64, bias, out);
}

}  // namespace rfe
