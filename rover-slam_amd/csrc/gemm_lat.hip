// gemm_lat.hip -- the LATENCY tiling of the fp32 "NT" GEMM  C[m][n] = alpha * (bias[n] + sum_k A[m][k] * W[n][k])  for LightGlue's
// Linears at the shapes the reference itself runs: ONE pair per call = 2048 token rows (src/Matchers/lightglue_onnx.cpp:168-172, batch 1).
//
// Why a second tiling.  With 2048 rows a Linear has 2048 x N outputs for 1024 SIMDs: gemm.hip's 64 x 64 tile of four 32 x 32 x 2
// accumulators gives N = 256 only 128 workgroups (half the chip, lg_ffn2 at 0.19 of the fp32 peak) and every wave one chain of K / 2
// dependent 64-cycle matrix instructions behind a single-buffered LDS tile with two barriers per 32 k.  Here:
//   * v_mfma_f32_16x16x4_f32 (32-cycle issue, 40-cycle dependent latency): 2048 x N outputs = 8 N tiles of 16 x 16, i.e. 2 / 4 / 6 tiles per
//     SIMD for N = 256 / 512 / 768 -- every SIMD owns WI x WJ independent accumulator chains, the workgroup tile (BN x BM below) is chosen so
//     that the grid is exactly 256 workgroups of 4 waves;
//   * the product is formed TRANSPOSED (matrix-A operand = weight rows, matrix-B operand = activation rows), so a lane ends up with four
//     CONSECUTIVE output columns of one row: bias / residual / rotary table / store are one 16-byte access each, and the row reductions of
//     the LayerNorm statistics are in-lane plus two shuffles;
//   * both operand tiles travel by global_load_lds_dwordx4 into a THREE-stage LDS ring of 64-k tiles (no staging registers, two stages in
//     flight under the matrix instructions of the third, one barrier per 64 k); rows are 256 B = sixteen 16-byte slots, slot c of row R
//     stored at c ^ (R & 15) -- applied on the SOURCE address, the LDS side of the copy is lane-linear -- which makes every fragment
//     read one conflict-free ds_read_b128 (lane (r, q) reads row r, k = 16 kg + 4 q .. + 3: the k order inside a 16-k group is permuted,
//     as in gemm.hip's k-permuted path; LightGlue is tolerance-checked, SuperPoint never comes here);
//   * the workgroups of one 64-row activation panel run on ONE XCD (block -> (panel, column tile) decode as in the attention kernels): the
//     activations cross the fabric once, only the (smaller) weight matrix is read by all eight L2s.
// Fusion: ROPE (qkv projection) -- LightGlue's rotary encoding of the q and k columns, (t0, t1) -> (t0 c - t1 s, t1 c + t0 s) on adjacent
//   pairs with (c, s) = rope_csn[row][pair], the same three fp32 operations the attention kernels apply on load -- so that the one-pair
//   attention kernel (lg_attention_lat_kernel) needs no rotary variant.
// Measured and NOT kept (profiles/r04_ab_notes.md): LayerNorm + GELU of ffn.3's activation applied on the fragment as it is read from LDS
//   (with per-row statistics partials from ffn.0's epilogue).  Correct, one launch fewer, and 2.4 x SLOWER than ffn.3 plus the stand-alone
//   pass: a 16-row wave tile re-evaluates the GELU of its whole 16 x 512 panel (8 column workgroups per panel = 8 x the transcendentals,
//   3200 VALU instructions per lane against 256 matrix instructions), and a wave alone on its SIMD has nobody to hide them behind.
//   Also: ring depth 3 vs 5-6 stages and "copy only the first stages" (timing ablation) change nothing -- the kernel is bound by the matrix
//   pipe, which sustains 38.6 cycles per v_mfma_f32_16x16x4_f32 from one wave per SIMD on random data (pure-MFMA ablation; nominal 32);
//   the K loop runs at 43.
// Roofline: fp32 MFMA peak 157.3 TFLOP/s, algorithmic 2 M N K FLOP.
#include "rfe_internal.h"
#include "h2_split.h"

namespace rfe {

typedef _Float16 glat_f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t glat_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* glat_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glat_gptr_t;

constexpr int LBK = 64;       // k per stage: LDS rows of 256 B

#ifdef RFE_TUNING
// in-kernel timeline (tuning build, abl & 4): per workgroup, wave 0 lane 0 records (shader clock, 100 MHz wall clock) at entry, after the
// first stage has landed, after the K loop and at exit; read back with rfe_k_dbg_timeline
__device__ unsigned long long rfe_dbg_ts[2048 * 8];
#define RFE_TS(slot) do { if ((abl & 4) && tid == 0 && blockIdx.x < 2048) { rfe_dbg_ts[blockIdx.x * 8 + (slot)] = clock64(); rfe_dbg_ts[blockIdx.x * 8 + 4 + (slot)] = wall_clock64(); } } while (0)
#else
#define RFE_TS(slot) do { } while (0)
#endif

// WI x WJ: 16-column (n) x 16-row (m) blocks per wave; WGN x WGM: waves per workgroup along n and m (WGN * WGM == 4).
// LSTAGES: depth of the LDS ring (LSTAGES - 1 stages in flight under the matrix instructions of one).
// H2 (RFE_OPT_LG_FP16X2, default off): the same tile, ring, block decode and epilogue with every product as a SPLIT product on the f16 matrix
// pipe (h2_split.h: x = hi + lo in fp16, lo hi + hi lo + hi hi with fp32 accumulation): a weight row of a stage is [64 hi | 64 lo] fp16 =
// 256 B copied from the load-time planes (GemmArgs::Bh / Bl) -- the same bytes, granules and swizzle as an fp32 row -- the activation rows
// stay fp32 and are split as they are read from LDS; one v_mfma_f32_16x16x32_f16 consumes 32 k (lane group q holds k = 8 q .. 8 q + 7 of
// both operands), three of them replace eight v_mfma_f32_16x16x4_f32.
template <int WI, int WJ, int WGN, int WGM, int LSTAGES, bool RES, bool ROPE, bool H2 = false>
__global__ __launch_bounds__(64 * WGN * WGM, 1) void gemm_lat_kernel(GemmArgs g, const float* __restrict__ rope_csn, int rope_cols, int MT, int abl) {
    constexpr int NW = WGN * WGM;                         // waves per workgroup: 4 (one per SIMD) or 8 (two per SIMD)
    static_assert(NW == 4 || NW == 8, "four or eight waves per workgroup");
    constexpr int BN = 16 * WI * WGN, BM = 16 * WJ * WGM, ROWS = BN + BM, STAGE_F = ROWS * LBK;
    constexpr int NDMA = ROWS / (4 * NW);                 // copy instructions per wave and stage (one moves 4 rows)
    static_assert(ROWS % (4 * NW) == 0, "rows per stage must split over the waves x 4 rows");
    static_assert(LSTAGES >= 2 && LSTAGES <= 6 && 4 * NDMA <= 63, "ring depth / copy count outside the counted-wait cases below");
    extern __shared__ __attribute__((aligned(16))) float lds[];   // LSTAGES stages of (BN + BM) x 64 floats

    const int tid = threadIdx.x, lane = tid & 63;
    RFE_TS(0);
    if (H2) h2_saturate_mode();                           // an activation past fp16's range saturates instead of turning its row into NaN (h2_split.h)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    // all column tiles of one activation panel on one XCD (blocks are dealt round-robin over the 8 XCDs)
    const int NT = g.N / BN;
    const int Lb = blockIdx.x, xcd = Lb & 7, t_ = Lb >> 3;
    const int nt = t_ % NT, mt = (t_ / NT) * 8 + xcd;
    if (mt >= MT) return;
    const int n0 = nt * BN, m0 = mt * BM;
    const int wn = wave % WGN, wm = wave / WGN;
    // batch entry z (the similarity matrix of pair z: A, B, C strided; no second A source / residual there), rows past m_valid[z] not stored
    const int z = blockIdx.y;
    int M = g.M;
    if (g.m_valid) { const int mv = g.m_valid[z]; M = mv < M ? mv : M; }
    if (m0 >= M) return;
    g.A += (size_t)z * g.sA; g.B += (size_t)z * g.sB; g.C += (size_t)z * g.sC;
    const int T = g.K / LBK;

    // ---- copy geometry: group G = 4 consecutive LDS rows; wave w issues groups w, w + 4, ...  Lane l fills row 4 G + (l >> 4), physical slot l & 15
    const float* src[NDMA];
    const float* src2[NDMA];
#pragma unroll
    for (int u = 0; u < NDMA; ++u) {
        const int R = (wave + NW * u) * 4 + (lane >> 4);
        const int sl = ((lane & 15) ^ (R & 15)) << 2;
        if (R < BN) {
            if (H2) {   // granules 0..7 of a row: 8 fp16 each of the hi plane, 8..15: of the lo plane; kept as a float pointer advancing k0 / 2 floats per k0
                const int gi = (lane & 15) ^ (R & 15);
                const uint16_t* pl = (gi < 8 ? g.Bh : g.Bl) + (size_t)(n0 + R) * g.ldb + (gi & 7) * 8;
                src[u] = reinterpret_cast<const float*>(pl);
            } else {
                src[u] = g.B + (size_t)(n0 + R) * g.ldb + sl;
            }
            src2[u] = src[u];
        } else {
            int m = m0 + R - BN; m = m < M ? m : M - 1;      // rows past the edge are clamped (computed, never stored)
            src[u] = g.A + (size_t)m * g.lda + sl;
            src2[u] = g.A2 ? g.A2 + (size_t)m * g.lda2 + sl - g.K1 : src[u];
        }
    }
    auto issue = [&](int t, int st) {
        const int k0 = t * LBK;
        const bool second = g.A2 && k0 >= g.K1;
#pragma unroll
        for (int u = 0; u < NDMA; ++u) {
            const bool wrow = H2 && (wave + NW * u) * 4 < BN;          // a weight-row group of the split form: fp16 planes, k0 halves
            const float* p = (second ? src2[u] : src[u]) + (wrow ? k0 / 2 : k0);
            __builtin_amdgcn_global_load_lds((glat_gptr_t)p, (glat_lds_ptr_t)(lds + st * STAGE_F + (wave + NW * u) * 4 * LBK), 16, 0, 0);
        }
    };

    // ---- accumulators start at the bias (D layout: lane (r, q) holds columns n = 16 i + 4 q .. + 3 of row m = 16 j + r)
    f32x4 acc[WI][WJ];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) bv = *reinterpret_cast<const f32x4*>(g.bias + n0 + (wn * WI + i) * 16 + 4 * q);
#pragma unroll
        for (int j = 0; j < WJ; ++j) acc[i][j] = bv;
    }

#pragma unroll
    for (int u = 0; u < LSTAGES - 1; ++u)
        if (u < T) issue(u, u);

    const float* const wfrag = lds + ((wn * WI) * 16 + r) * LBK;            // + i * 16 * LBK + stage
    const float* const afrag = lds + (BN + (wm * WJ) * 16 + r) * LBK;       // + j * 16 * LBK + stage
    for (int t = 0; t < T; ++t) {
        const int st = t % LSTAGES;
        // stage t has landed (this wave's copies: counted wait -- the stages t + 1 .. still in flight are younger; everybody's: barrier),
        // and every wave has left stage t - 1, whose buffer the next request reuses
#if defined(RFE_EXP) && (RFE_EXP & 2)   // compile-time ablations of the floor measurement (tools/kbench/build_exp.sh, wrong results): 2 = no waits / barriers after the first stage
        if (t == 0)
#endif
        {
            int younger = T - 1 - t; younger = younger < LSTAGES - 2 ? younger : LSTAGES - 2;
            if (abl & 1) younger = 0;
            if (younger >= 4) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * NDMA) : "memory");
            else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(3 * NDMA) : "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * NDMA) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (t == 0) RFE_TS(1);
        if (t + LSTAGES - 1 < T && !(abl & 1)) issue(t + LSTAGES - 1, (t + LSTAGES - 1) % LSTAGES);
        const float* const ws = wfrag + st * STAGE_F;
        const float* const as = afrag + st * STAGE_F;
        if constexpr (H2) {
            // two 32-k groups per stage.  Weights: granule 4 g + q (hi) / 8 + 4 g + q (lo) of row n = the lane's 8 k; activations: the fp32
            // granules 8 g + 2 q, + 1 of row m, split in registers.  (Requesting group 1's raw fragments ahead of group 0's products changes
            // nothing: at 16 cycles per matrix instruction these kernels wait for the copies -- 8 MB per stage over the chip at ~10 TB/s.)
#pragma unroll
            for (int g32 = 0; g32 < LBK / 32; ++g32) {
                glat_f16x8 wh[WI], wl[WI], ah[WJ], al[WJ];
#pragma unroll
                for (int i = 0; i < WI; ++i) {
                    wh[i] = *reinterpret_cast<const glat_f16x8*>(ws + i * 16 * LBK + (((4 * g32 + q) ^ r) << 2));
                    wl[i] = *reinterpret_cast<const glat_f16x8*>(ws + i * 16 * LBK + (((8 + 4 * g32 + q) ^ r) << 2));
                }
#pragma unroll
                for (int j = 0; j < WJ; ++j) {
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(as + j * 16 * LBK + (((8 * g32 + 2 * q) ^ r) << 2));
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(as + j * 16 * LBK + (((8 * g32 + 2 * q + 1) ^ r) << 2));
                    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
                    h2_split2(x0[0], x0[1], h0, l0); h2_split2(x0[2], x0[3], h1, l1);
                    h2_split2(x1[0], x1[1], h2, l2); h2_split2(x1[2], x1[3], h3, l3);
                    ah[j] = __builtin_bit_cast(glat_f16x8, glat_u32x4{h0, h1, h2, h3}); al[j] = __builtin_bit_cast(glat_f16x8, glat_u32x4{l0, l1, l2, l3});
                }
#pragma unroll
                for (int i = 0; i < WI; ++i)
#pragma unroll
                    for (int j = 0; j < WJ; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], ah[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], al[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], ah[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
        // fragments of k group kg + 1 are requested before the matrix instructions of group kg (the scheduler would otherwise sink the
        // reads to their first use and expose the LDS round trip four times per stage)
        f32x4 a4[WI], b4[WJ];
        auto frags = [&](int kg, f32x4 (&a)[WI], f32x4 (&b)[WJ]) {
            const int sl = ((4 * kg + q) ^ r) << 2;
#pragma unroll
            for (int i = 0; i < WI; ++i) a[i] = *reinterpret_cast<const f32x4*>(ws + i * 16 * LBK + sl);
#pragma unroll
            for (int j = 0; j < WJ; ++j) b[j] = *reinterpret_cast<const f32x4*>(as + j * 16 * LBK + sl);
        };
#if defined(RFE_EXP) && (RFE_EXP & 1)   // 1 = no LDS fragment reads after the first (operands stay in registers): with 2, the pure-MFMA K loop (38.6 cycles per instruction)
        if (t == 0)
#endif
        frags(0, a4, b4);
        // Order inside a k group (pinned with sched_barriers; one wave per SIMD: nobody else covers an exposed LDS round trip):
        //   first matrix instruction of group kg  -- the s_waitcnt for kg's fragments sits in front of it, and they were requested a whole
        //                                            group (>= 8 matrix instructions) earlier;
        //   request the fragments of group kg + 1 -- behind that wait, so that it does not cover them too;
        //   the remaining matrix instructions of group kg.
#pragma unroll
        for (int kg = 0; kg < LBK / 16; ++kg) {
            f32x4 an[WI], bn[WJ];
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[0][0], b4[0][0], acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#if defined(RFE_EXP) && (RFE_EXP & 1)
            if (kg + 1 < LBK / 16) {
#pragma unroll
                for (int i = 0; i < WI; ++i) an[i] = a4[i] + 1e-9f;
#pragma unroll
                for (int j = 0; j < WJ; ++j) bn[j] = b4[j] + 1e-9f;
            }
#else
            if (kg + 1 < LBK / 16) { frags(kg + 1, an, bn); __builtin_amdgcn_sched_barrier(0); }
#endif
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < WI; ++i)
#pragma unroll
                    for (int j = 0; j < WJ; ++j)
                        if (e + i + j > 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i][e], b4[j][e], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int e = 2; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < WI; ++i)
#pragma unroll
                    for (int j = 0; j < WJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i][e], b4[j][e], acc[i][j], 0, 0, 0);
            if (kg + 1 < LBK / 16) {
#pragma unroll
                for (int i = 0; i < WI; ++i) a4[i] = an[i];
#pragma unroll
                for (int j = 0; j < WJ; ++j) b4[j] = bn[j];
            }
        }
            }
    }

    RFE_TS(2);
    // ---- epilogue: one 16-byte access per (block, lane) for rotary table / residual / store
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int m = m0 + (wm * WJ + j) * 16 + r;
        const bool live = m < M;
        const int mc = live ? m : M - 1;
        f32x4 v[WI];
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int n = n0 + (wn * WI + i) * 16 + 4 * q;
            v[i] = acc[i][j] * g.alpha;
            if (g.relu) { v[i][0] = fmaxf(v[i][0], 0.f); v[i][1] = fmaxf(v[i][1], 0.f); v[i][2] = fmaxf(v[i][2], 0.f); v[i][3] = fmaxf(v[i][3], 0.f); }
            if (ROPE && n < rope_cols) {
                const f32x4 cs = *reinterpret_cast<const f32x4*>(rope_csn + (size_t)mc * 64 + (n & 63));   // (c, s) of pairs n / 2, n / 2 + 1 of this head
                const f32x4 t = v[i];
                v[i] = f32x4{t[0] * cs[0] - t[1] * cs[1], t[1] * cs[0] + t[0] * cs[1], t[2] * cs[2] - t[3] * cs[3], t[3] * cs[2] + t[2] * cs[3]};
            }
            if (RES) {
                const f32x4 rv = *reinterpret_cast<const f32x4*>(g.R + (size_t)mc * g.ldr + n);
                v[i] = rv + v[i];
            }
            if (live) *reinterpret_cast<f32x4*>(g.C + (size_t)m * g.ldc + n) = v[i];
        }
    }
    RFE_TS(3);
}

#ifdef RFE_TUNING
extern "C" int rfe_k_dbg_timeline(unsigned long long* host, int n) {   // tuning build only: copy out the first n entries of the timeline
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rfe_dbg_ts), (size_t)n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

// Shapes served: k-permuted LightGlue Linears (GemmArgs::kperm) without batch / m_valid / fused LayerNorm, K % 64 == 0 (K1 % 64 == 0), at most
// 8192 rows, N a multiple of 32 -- i.e. the Linears of a one- or few-pair forward.  Returns false when the shape is not one of them (nothing
// is launched; the caller falls back to gemm.hip's tiles).
bool launch_gemm_lat(hipStream_t s, const GemmArgs& g, const float* rope_csn, int rope_cols) {
    const int batch = g.batch > 0 ? g.batch : 1;
    if (batch > 1 && (g.A2 || g.R || rope_csn || batch > 16)) return false;
    if (!g.kperm || g.K % LBK || (g.A2 && g.K1 % LBK) || g.M > 8192 || g.M < 1 || g.stats_in || g.stats_out) return false;
    const bool h2 = g.Bh && g.Bl && !g.relu && (g.ldb % 8) == 0;   // RFE_OPT_LG_FP16X2: the split form of the same tiles (weight planes from load time)
    if ((g.lda % 4) || (g.ldb % 4) || (g.ldc % 4) || (g.A2 && (g.lda2 % 4)) || (g.R && (g.ldr % 4))) return false;
    const bool res = g.R != nullptr, rope = rope_csn != nullptr;
#ifdef RFE_TUNING
    const int abl = tune_int("RFE_GLAT_ABL", 0);         // 1 = only the first LSTAGES - 1 stages are copied (timing ablation, wrong results); 4 = record the timeline
    const int stages_env = tune_int("RFE_GLAT_STAGES", 0);
    const int w8 = tune_int("RFE_GLAT_W8", 3);           // bit 0: qkv, bit 1: residual form with EIGHT waves per workgroup (two per SIMD); 0 = four (A/B)
#else
    constexpr int abl = 0, stages_env = 0;
#endif
#define RFE_GLAT_LAUNCH(WI_, WJ_, WGN_, WGM_, LS_, RES_, ROPE_)                                                                  \
    do {                                                                                                                         \
        constexpr int BN_ = 16 * WI_ * WGN_, BM_ = 16 * WJ_ * WGM_, BYTES_ = LS_ * (BN_ + BM_) * LBK * 4;                        \
        const int MT = (g.M + BM_ - 1) / BM_;                                                                                    \
        auto kern = h2 ? gemm_lat_kernel<WI_, WJ_, WGN_, WGM_, LS_, RES_, ROPE_, true> : gemm_lat_kernel<WI_, WJ_, WGN_, WGM_, LS_, RES_, ROPE_, false>; \
        static bool ls_[2][64]; ensure_dynamic_lds((const void*)kern, BYTES_, ls_[h2 ? 1 : 0]);                                  \
        hipLaunchKernelGGL(kern, dim3((g.N / BN_) * ((MT + 7) / 8 * 8), batch), dim3(64 * WGN_ * WGM_), BYTES_, s, g, rope_csn, rope_cols, MT, abl); \
        return true;                                                                                                             \
    } while (0)
    // ring depth: as many 64-k stages as the 160 KB of LDS hold, at most 6 (measured: 3 is as fast -- the kernel is bound by the matrix
    // pipe -- the deeper ring only buys tolerance against a slow first touch of the weights); RFE_GLAT_STAGES=3 in the tuning build
#define RFE_GLAT_GO(WI_, WJ_, WGN_, WGM_, RES_, ROPE_)                                                                           \
    do {                                                                                                                         \
        constexpr int ROWS_ = 16 * WI_ * WGN_ + 16 * WJ_ * WGM_;                                                                 \
        constexpr int LSMAX_ = 160 * 1024 / (ROWS_ * LBK * 4) > 6 ? 6 : 160 * 1024 / (ROWS_ * LBK * 4);                          \
        if (stages_env == 3) RFE_GLAT_LAUNCH(WI_, WJ_, WGN_, WGM_, 3, RES_, ROPE_);                                              \
        RFE_GLAT_LAUNCH(WI_, WJ_, WGN_, WGM_, LSMAX_, RES_, ROPE_);                                                              \
    } while (0)
    // tile: the widest column tile (96 / 64 / 32 columns x 64 rows) that still gives the chip about one workgroup per CU.  The qkv and the
    // residual forms run EIGHT waves per workgroup (wave tiles 48 x 16 / 16 x 16 instead of 48 x 32 / 32 x 16): measured on one pair, qkv
    // 138.6 -> 131.2 us per forward, ffn.3 221 -> 213; the 64-wide plain tiles (ffn.0, cross-qkv) do not gain (305 -> 303) and keep four
    // (profiles/r04_ab_notes.md) -- the matrix pipe, not latency hiding, is what these kernels wait for
    const long long panels = (long long)batch * ((g.M + 63) / 64);
    if (rope) {   // qkv
        if (g.N % 96 || res || (rope_cols % 64)) return false;
#ifdef RFE_TUNING
        if (!(w8 & 1)) RFE_GLAT_GO(3, 2, 2, 2, false, true);
#endif
        RFE_GLAT_GO(3, 1, 2, 4, false, true);
    }
    if (res) {
        if (g.N % 32) return false;
#ifdef RFE_TUNING
        if (!(w8 & 2)) RFE_GLAT_GO(2, 1, 1, 4, true, false);
#endif
        RFE_GLAT_GO(1, 1, 2, 4, true, false);
    }
    if (g.N % 96 == 0 && panels * (g.N / 96) >= 224) RFE_GLAT_GO(3, 2, 2, 2, false, false);
    if (g.N % 64 == 0 && panels * (g.N / 64) >= 224) RFE_GLAT_GO(2, 2, 2, 2, false, false);
    if (g.N % 32 == 0) RFE_GLAT_GO(2, 1, 1, 4, false, false);
#undef RFE_GLAT_GO
#undef RFE_GLAT_LAUNCH
    return false;
}

}  // namespace rfe
