// gemm.hip -- fp32 "NT" GEMM on the CDNA4 matrix cores:  C[m][n] = alpha*(bias[n] + sum_k A[m][k]*B[n][k])
// (+ReLU, +residual).  Both operands are K-contiguous, i.e. B is a PyTorch Linear weight [N][K] or
// a second activation matrix (LightGlue similarity md0 . md1^T).
//
// Used for: SuperPoint 1x1 heads convPb (256->65) / convDb (256->256) and every LightGlue Linear
// (the arithmetic of superpoint.onnx / lightglue_sim.onnx that the reference runs through
// Ort::Session::Run, src/Extractors/superpoint_onnx.cc:135, src/Matchers/lightglue_onnx.cpp:213).
//
// Workgroup 256 threads = 2x2 waves, tile 128 x 256 x 32 (128 x 128 when N is not a multiple of 256);
// each wave 64 x 128 = 2x4 v_mfma_f32_32x32x2_f32 accumulators.  LDS tiles [rows][33] (odd row stride ->
// the 32 rows of a fragment hit 32 banks), single buffered so that 3 workgroups share a CU.
// Reduction order: k ascending, accumulator initialised with the bias (== oracle rfo_linear, bit-exact).
// Roofline: fp32 MFMA peak 157.3 TFLOP/s, algorithmic 2*M*N*K FLOP.
#include <stdlib.h>
#include "rfe_internal.h"

namespace rfe {

constexpr int BK = 32, LDT = BK + 1;

// GELU(erf) with a short branch-free erf: Abramowitz-Stegun 7.1.26, |erf error| <= 1.5e-7 (libm erff: <= 1 ulp = 6e-8), about a
// third of the instructions of the library erff (ffn.3 with the fused LayerNorm + GELU: 3.30 -> 3.13 ms per step).  Over the 20-case
// tolerance study the match-score deviation from the oracle is unchanged (<= 1.9e-4, lists identical): profiles/r02_ab_notes.md.
__device__ __forceinline__ float gelu_short(float t) {
    const float x = t * 0.70710678118654752f, ax = fabsf(x);
    const float k = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    const float poly = fmaf(fmaf(fmaf(fmaf(1.061405429f, k, -1.453152027f), k, 1.421413741f), k, -0.284496736f), k, 0.254829592f) * k;
    const float er = 1.0f - poly * __builtin_amdgcn_exp2f(-(ax * ax) * 1.44269504088896341f);
    return 0.5f * t * (1.0f + copysignf(er, x));
}

// NB = 32-column MFMA blocks per wave: wave tile 64 x (NB*32), workgroup tile 128 x (NB*64).
// NB = 4 (128x256 tile, 49 KB LDS, 3 workgroups/CU) measured best for N % 256 == 0
// (tools/kbench/gemm_variants.hip: 119-132 TFLOP/s vs 110-125 for 128x128 double-buffered).
// MB = 32-row MFMA blocks per wave (2 for throughput; 1 gives 64-row tiles for small, latency-bound problems:
// a single 1024-keypoint pair has only M = 2048 rows, 16 tiles of 128 x 256 would use 16 of the 256 CUs).
// RES: the residual variant reads R in the D layout (four rows of 128-B segments in flight per step) and adds it after bias /
// alpha like the oracle.  (Round 1's LDS-transposed whole-row epilogue, which also applied the LightGlue rotary to q | k, is gone:
// the rotary moved into the attention kernel's loads -- same arithmetic, same time overall, one epilogue fewer here.)
// PFT: register prefetch of the next K tile under the MFMAs (always on for the 64-row latency tiles).
// KP (k-permuted, opt-in through GemmArgs::kperm): LDS tiles [rows][32] WITHOUT padding, the eight 16-byte slots of a row XOR-swizzled
// by (row & 7): staging writes one ds_write_b128 per loaded float4 (the padded layout needs four scalar writes), a fragment read is
// one ds_read_b128 per four k-steps (lane half h reads k = 16 h + 4 g .. + 3 of its row; eight consecutive lanes = eight rows hit the
// eight different slots -> all 32 banks).  MFMA step (g, c) therefore multiplies k = 4 g + c and k = 16 + 4 g + c: every product
// of the K tile is summed, in a different order than k ascending -- LightGlue only (SuperPoint's 1x1 heads stay bit-exact on the
// padded path).  LDS instructions per K tile and wave: 36 instead of 144.
// DMAB (KP throughput tile only; not batched): the B tile (weights) never touches a register -- it is copied by global_load_lds_dwordx4
// into a DOUBLE-buffered LDS tile (2 x 32 KB; with the 16 KB A tile = 80 KB per workgroup: still two workgroups per CU,
// tools/kbench/lds_occupancy.hip), tile t + 1 requested right after tile t is published; the 16-byte-slot swizzle is applied on the SOURCE
// address (the LDS side of the copy is lane-linear: lane l of a wave instruction fills bytes 16 l .. of 1 KB = 8 rows).  Eight of the
// twelve ds_write_b128 per thread and K tile and 32 staging VGPRs go away; the barriers are raw (s_waitcnt + s_barrier) so that the
// copies stay in flight across the one that publishes the A tile.
typedef __attribute__((address_space(3))) void* gemm_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gemm_gptr_t;
// ROPE (the qkv projection on the k-permuted throughput tile): LightGlue's rotary encoding of the q | k column tiles in the epilogue, so that the attention
// kernel takes K tiles straight into LDS (lg_attention_dma_kernel for the self blocks too).  The table rows of the workgroup's 128 token rows (256 B
// each = the 32 (cos, sin) pairs every head shares) are copied by global_load_lds_dwordx4 into a 32 KB dynamic LDS tile when the workgroup STARTS --
// no registers, and the whole K loop to land -- and the epilogue rotates in place from LDS before the first store.  48 + 32 = 80 KB: still two
// workgroups per CU (tools/kbench/lds_occupancy.hip).  Forms that read the table from global memory in the epilogue measured 216 -> 288 us (row by row
// between the stores: every load waits for the stores in front of it on the in-order vector-memory counter) and 216 -> 239 us (16 loads at a time in
// front of all stores; the same whether the q | k or only the k tiles rotate: exposed round trips, not work) -- profiles/r05_ab_notes.md.
template <int MB, int NB, bool RES, bool PFT = false, bool LNA = false, bool KP = false, bool DMAB = false, bool ROPE = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
    constexpr int BM = MB * 64, BN = NB * 64;
    constexpr int LDR = KP ? BK : LDT;          // LDS row stride in words
    static_assert(!DMAB || (KP && PFT && !LNA && MB == 2), "DMAB: k-permuted throughput tile only");
    static_assert(!ROPE || (KP && PFT && !LNA && !RES && !DMAB && MB == 2 && NB == 4), "ROPE: plain k-permuted 128 x 256 tile only");
    __shared__ __attribute__((aligned(16))) float lds_static[DMAB ? 4 : (BM + BN) * LDR];   // A tile | B tile
    extern __shared__ __attribute__((aligned(16))) float lds_dynamic[];                      // DMAB: A tile | B tile x 2 = (BM + 2 BN) * LDR words
    float* const lds_ab = DMAB ? lds_dynamic : lds_static;
    float* const As = lds_ab;
    float* const Bs = lds_ab + BM * LDR;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    const int z = blockIdx.z;
    int bxt = blockIdx.x, byt = blockIdx.y;      // column tile, row panel
    if (g.xcd) {   // launch order is x fastest: linear id L -> XCD L & 7; the nct column tiles of row panel ((slot / nct) * 8 + xcd) sit in consecutive slots of that XCD
        const int L = blockIdx.y * gridDim.x + blockIdx.x, nct = gridDim.x;
        const int slot = L >> 3;
        byt = (slot / nct) * 8 + (L & 7);
        bxt = slot % nct;
    }
    const int m0 = byt * BM, n0 = bxt * BN;
    int M = g.M;
    if (g.m_valid) { M = g.m_valid[z]; if (M > g.M) M = g.M; }
    if (m0 >= M) return;

    const float* A = g.A + (size_t)z * g.sA;
    const float* A2 = g.A2 ? g.A2 + (size_t)z * g.sA2 : nullptr;
    const float* B = g.B + (size_t)z * g.sB;
    float* C = g.C + (size_t)z * g.sC;

    const bool rope = ROPE && n0 >= g.rope_c0 && n0 < g.rope_c1;   // workgroup-uniform (the bounds are multiples of the column tile)
    f32x16 acc[MB][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + (wn * NB + nb) * 32 + i;
        const float bv = (g.bias && n < g.N) ? g.bias[n] : 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = bv;
    }

    // staging: thread -> (row = tid/8 + 32*it, 4 consecutive k).  Rows past the M / N edge are CLAMPED
    // (their products land in accumulators that are never stored), so the loop has no bounds branches.
    constexpr int A_IT = BM / 32, B_IT = BN / 32;
    const int lrow = tid >> 3, lkq = tid & 7;
    // 32-bit element offsets from wave-uniform tile bases: loads use the SGPR-base + VGPR-offset form
    // (64-bit per-lane addresses cost 32 VGPRs and ~20 u64 adds per K tile)
    int mlast = M - 1 - m0; mlast = mlast < BM - 1 ? mlast : BM - 1;          // last valid row of this tile
    int nlast = g.N - 1 - n0; nlast = nlast < BN - 1 ? nlast : BN - 1;
    const float* const At = A + (size_t)m0 * g.lda;
    const float* const A2t = A2 ? A2 + (size_t)m0 * g.lda2 : nullptr;
    const float* const Bt = B + (size_t)n0 * g.ldb;
    int aoff[A_IT], a2off[A_IT], boff[B_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        int row = lrow + 32 * it; row = row < mlast ? row : mlast;
        aoff[it] = row * g.lda + lkq * 4;
        a2off[it] = row * g.lda2 + lkq * 4;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        int row = lrow + 32 * it; row = row < nlast ? row : nlast;
        boff[it] = row * g.ldb + lkq * 4;
    }
    // LNA: LayerNorm statistics of this thread's A rows from the producer's partials.  Each partial is (mean, M2 = sum of squared
    // deviations from that mean) of NPART = K / stats_p columns; they are merged with the parallel-variance formula
    //   mean = avg(mean_p),  M2 = sum(M2_p) + NPART * sum((mean_p - mean)^2),  var = M2 / K
    // -- no E[x^2] - mean^2 anywhere, so rows with |mean| >> std (a large ffn.0 bias in real weights) keep their variance
    float ln_mean[A_IT], ln_rstd[A_IT];
    if (LNA) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            int row = lrow + 32 * it; row = row < mlast ? row : mlast;
            const float* sp = g.stats_in + ((size_t)(m0 + row) + (size_t)z * g.M) * g.stats_p * 2;
            float ms = 0.f, m2 = 0.f;
            for (int p = 0; p < g.stats_p; ++p) { ms += sp[2 * p]; m2 += sp[2 * p + 1]; }
            const float mean = ms / (float)g.stats_p;
            float dev = 0.f;
            for (int p = 0; p < g.stats_p; ++p) { const float d = sp[2 * p] - mean; dev = fmaf(d, d, dev); }
            const float var = (m2 + dev * ((float)g.K / (float)g.stats_p)) / (float)g.K;
            ln_mean[it] = mean; ln_rstd[it] = 1.0f / sqrtf(var + 1e-5f);
        }
    }
    float* const da = KP ? As + lrow * LDR + ((lkq ^ (lrow & 7)) << 2) : As + lrow * LDR + lkq * 4;   // rows lrow + 32 it: same (row & 7)
    float* const db = KP ? Bs + lrow * LDR + ((lkq ^ (lrow & 7)) << 2) : Bs + lrow * LDR + lkq * 4;
    const float* const ap = KP ? As + (wm * MB * 32 + i) * LDR : As + (wm * MB * 32 + i) * LDR + h;
    const float* const bp = KP ? Bs + (wn * NB * 32 + i) * LDR : Bs + (wn * NB * 32 + i) * LDR + h;

    // Small (64-row) tiles serve latency-bound problems (one pair: M = 2048, one workgroup per CU), where every K step
    // would otherwise expose a full global-memory round trip: there the next tile's loads are issued before the MFMA
    // loop of the current one (16 VGPRs).  The throughput tiles rely on the other resident workgroups instead
    // (register prefetch measured slower there).
    constexpr bool PF = (MB == 1) || PFT;
    constexpr bool LNI = LNA && PF && MB == 2;   // LN + GELU of the PREFETCHED tile interleaved with the second half of the MFMA loop
    float4 ra[A_IT], rb[B_IT], rg, rbeta;
    auto load_tile = [&](int k0) {
        if (LNA) { rg = *reinterpret_cast<const float4*>(g.ln_g + k0 + lkq * 4); rbeta = *reinterpret_cast<const float4*>(g.ln_b + k0 + lkq * 4); }
        if (A2t && k0 >= g.K1) {
            const float* base = A2t + (k0 - g.K1);
#pragma unroll
            for (int it = 0; it < A_IT; ++it) ra[it] = *reinterpret_cast<const float4*>(base + a2off[it]);
        } else {
            const float* base = At + k0;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) ra[it] = *reinterpret_cast<const float4*>(base + aoff[it]);
        }
        if (DMAB) return;   // the B tile goes through dma_b
        const float* base = Bt + k0;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) rb[it] = *reinterpret_cast<const float4*>(base + boff[it]);
    };
    // one wave instruction copies 64 x 16 B = 1 KB = 8 rows of 128 B; wave w issues the row groups (BN / 32) w .. of the tile.  Lane l fills
    // row 8 grp + l / 8, physical slot l & 7, which holds logical slot (l & 7) ^ (row & 7) = (l & 7) ^ (l >> 3)
    auto dma_b = [&](int k0, int buf) {
        constexpr int GPW = BN / 32;   // row groups per wave
#pragma unroll
        for (int u = 0; u < GPW; ++u) {
            const int grp = wave * GPW + u;
            int row = grp * 8 + (lane >> 3); row = row < nlast ? row : nlast;
            const float* src = Bt + (size_t)row * g.ldb + k0 + 4 * ((lane & 7) ^ (lane >> 3));
            __builtin_amdgcn_global_load_lds((gemm_gptr_t)src, (gemm_lds_ptr_t)(Bs + buf * (BN * LDR) + grp * 256), 16, 0, 0);
        }
    };
    // LN + GELU of one staged element, same operation order as the stand-alone kernel: ((a - mean) * rstd) * g + b, then GELU
    auto ln_elem = [&](float a, int it, float gq, float bq) { return gelu_short((a - ln_mean[it]) * ln_rstd[it] * gq + bq); };
    auto ln_tile = [&]() {
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
            ra[it] = make_float4(ln_elem(ra[it].x, it, rg.x, rbeta.x), ln_elem(ra[it].y, it, rg.y, rbeta.y),
                                 ln_elem(ra[it].z, it, rg.z, rbeta.z), ln_elem(ra[it].w, it, rg.w, rbeta.w));
    };
#ifdef RFE_TUNING
    const int abl = g.abl;
#else
    constexpr int abl = 0;
#endif
    if (DMAB) dma_b(0, 0);
    if (PF) load_tile(0);
    if (rope) {   // behind the first tile's loads on the in-order counter: the wait for that tile does not cover these copies
        // one wave instruction = 64 x 16 B = 4 table rows; wave w copies rows 32 w .. 32 w + 31 of the tile (rows past M: the last row, never stored)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int row = m0 + wave * 32 + u * 4 + (lane >> 4);
            row = row < M ? row : M - 1;
            __builtin_amdgcn_global_load_lds((gemm_gptr_t)(g.rope_csn + (size_t)row * 64 + (lane & 15) * 4), (gemm_lds_ptr_t)(lds_dynamic + (wave * 32 + u * 4) * 64), 16, 0, 0);
        }
    }
    if (LNI) ln_tile();   // first tile: nothing to hide it under
    int bbuf = 0;
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        if (!PF) load_tile(k0);
        if (DMAB) {
            // #1: this tile's B copy and A rows have landed (requested one tile ago); every wave has finished the previous tile
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int it = 0; it < A_IT; ++it)
                *reinterpret_cast<f32x4*>(da + it * 32 * LDR) = f32x4{ra[it].x, ra[it].y, ra[it].z, ra[it].w};
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // #2: the A tile is visible
            // next tile, in flight under this tile's MFMAs.  Unconditional (the last tile re-requests itself into the idle buffer) and fenced:
            // behind a branch the compiler merges the loaded registers with copies and waits for the loads on the spot, without the fence
            // the scheduler sinks them below the MFMAs (profiles/r03_ab_notes.md)
            { const int kn = k0 + BK < g.K ? k0 + BK : k0; dma_b(kn, bbuf ^ 1); load_tile(kn); }
            __builtin_amdgcn_sched_barrier(0);
        } else
        if (!(abl & 2) || k0 == 0) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            float4 v = ra[it];
            if (LNA && !LNI) {   // same operation order as the stand-alone LayerNorm + GELU kernel: ((a - mean) * rstd) * g + b, then 0.5 v (1 + erf(v / sqrt 2))
                float e[4] = {v.x, v.y, v.z, v.w};
                const float gg[4] = {rg.x, rg.y, rg.z, rg.w}, bb[4] = {rbeta.x, rbeta.y, rbeta.z, rbeta.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t = (e[q] - ln_mean[it]) * ln_rstd[it] * gg[q] + bb[q];
                    e[q] = gelu_short(t);
                }
                v = make_float4(e[0], e[1], e[2], e[3]);
            }
            float* d = da + it * 32 * LDR;
            if (KP) *reinterpret_cast<f32x4*>(d) = f32x4{v.x, v.y, v.z, v.w};
            else { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            float* d = db + it * 32 * LDR;
            if (KP) *reinterpret_cast<f32x4*>(d) = f32x4{rb[it].x, rb[it].y, rb[it].z, rb[it].w};
            else { d[0] = rb[it].x; d[1] = rb[it].y; d[2] = rb[it].z; d[3] = rb[it].w; }
        }
        __syncthreads();
        }
        if (!DMAB && PF && k0 + BK < g.K && !(abl & 1)) load_tile(k0 + BK);
        // LNI: the tile prefetched before this loop has landed by now (>= 64 MFMAs = 4096 cycles after its loads were issued):
        // its LayerNorm + GELU -- VALU work, 2 of the 16 elements of this thread per k-step -- runs in the shadow of the
        // MFMAs of the remaining k-steps instead of on the staging path in front of the barrier
#define RFE_LN_SHADOW(S_)                                                                             \
        if (LNI && (S_) >= BK / 4) {                                                                  \
            constexpr int per = (A_IT * 4) / (BK / 4);                                                \
            _Pragma("unroll") for (int u = 0; u < per; ++u) {                                         \
                const int e = ((S_) - BK / 4) * per + u, it = e >> 2, q = e & 3;                      \
                float* comp = q == 0 ? &ra[it].x : q == 1 ? &ra[it].y : q == 2 ? &ra[it].z : &ra[it].w; \
                const float gq = q == 0 ? rg.x : q == 1 ? rg.y : q == 2 ? rg.z : rg.w;                \
                const float bq = q == 0 ? rbeta.x : q == 1 ? rbeta.y : q == 2 ? rbeta.z : rbeta.w;    \
                *comp = ln_elem(*comp, it, gq, bq);                                                   \
            }                                                                                         \
        }
        if (KP) {
            // fragments of group g+1 are requested while group g multiplies: the A fragments into a second register set, each B
            // fragment back into its own registers as soon as its last MFMA of this group has been issued (+8 VGPRs, no LDS
            // round trip exposed at the group boundaries)
            const int iswz = i & 7;
            const float* const bpt = DMAB ? bp + bbuf * (BN * LDR) : bp;   // this tile's B buffer
            f32x4 a4[MB], b4[NB];
            {
                const int slot = ((h << 2) ^ iswz) << 2;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) a4[mb] = *reinterpret_cast<const f32x4*>(ap + mb * 32 * LDR + slot);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) b4[nb] = *reinterpret_cast<const f32x4*>(bpt + nb * 32 * LDR + slot);
            }
#pragma unroll
            for (int gq4 = 0; gq4 < 4; ++gq4) {
                const int nslot = (((h << 2) + gq4 + 1) ^ iswz) << 2;
                f32x4 an[MB];
                if (gq4 < 3) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) an[mb] = *reinterpret_cast<const f32x4*>(ap + mb * 32 * LDR + nslot);
                    __builtin_amdgcn_sched_barrier(0);   // keep the requests up here (the scheduler would sink them to their first use)
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                    for (int cq = 0; cq < 4; ++cq)
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[mb][cq], b4[nb][cq], acc[mb][nb], 0, 0, 0);
                    if (gq4 < 3) { b4[nb] = *reinterpret_cast<const f32x4*>(bpt + nb * 32 * LDR + nslot); __builtin_amdgcn_sched_barrier(0); }
                    RFE_LN_SHADOW(gq4 * 4 + nb)
                }
                if (gq4 < 3) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) a4[mb] = an[mb];
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < BK / 2; ++s) {
                float a[MB], b[NB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) a[mb] = ap[mb * 32 * LDR + 2 * s];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) b[nb] = bp[nb * 32 * LDR + 2 * s];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb], b[nb], acc[mb][nb], 0, 0, 0);
                RFE_LN_SHADOW(s)
            }
        }
#undef RFE_LN_SHADOW
        if (DMAB) bbuf ^= 1;
    }

    {   // bias (+alpha, +ReLU, +residual) epilogue: 128-B coalesced accesses straight from the D layout
        const float* Rz = (RES && g.R) ? g.R + (size_t)z * g.sR : nullptr;
        if (rope) {
            // rotary epilogue (alpha = 1, no ReLU), IN PLACE and before the first store: in the D layout a lane holds column n = .. + i of 16 rows, the other
            // element of its pair (n ^ 1) sits in lane i ^ 1 of the same register -> one DPP move.  (c, s) of head dimension (nb & 1) * 32 + i from the LDS
            // table tile (blocks nb and nb + 2 = the same dimensions of two heads): the copies were requested before the K loop, whose barriers have long
            // published them.
            const float* csl = lds_dynamic + ((wm * MB) * 32 + 4 * h) * 64 + (i & ~1);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* cp = csl + (mb * 32 + (r & 3) + 8 * (r >> 2)) * 64;
                    float2 c0 = *reinterpret_cast<const float2*>(cp), c1 = *reinterpret_cast<const float2*>(cp + 32);
                    // even lane: t0 c - t1 s = t0 c + t1 (-s); odd lane: t1 c + t0 s -- one signed sine per lane, then the same mul, mul, add in both
                    // (bit-identical to the attention kernels' on-load form: x (-s) = -(x s) and a + (-b) = a - b exactly)
                    c0.y = (i & 1) ? c0.y : -c0.y; c1.y = (i & 1) ? c1.y : -c1.y;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const float v = acc[mb][nb][r];
                        const float pv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]: lane i ^ 1
                        const float2 t = (nb & 1) ? c1 : c0;
                        acc[mb][nb][r] = v * t.x + pv * t.y;
                    }
                }
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                float rv[4][NB];
                if (Rz) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        int m = m0 + (wm * MB + mb) * 32 + rr + 8 * rq + 4 * h;
                        m = m < M ? m : M - 1;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            int n = n0 + (wn * NB + nb) * 32 + i;
                            n = n < g.N ? n : g.N - 1;
                            rv[rr][nb] = Rz[(size_t)m * g.ldr + n];
                        }
                    }
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = rq * 4 + rr;
                    const int m = m0 + (wm * MB + mb) * 32 + rr + 8 * rq + 4 * h;
                    if (m >= M || ((abl & 4) && m != 0)) continue;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int n = n0 + (wn * NB + nb) * 32 + i;
                        if (n >= g.N) continue;
                        float v = acc[mb][nb][r] * g.alpha;
                        if (g.relu) v = fmaxf(v, 0.f);
                        if (Rz) v = rv[rr][nb] + v;
                        C[(size_t)m * g.ldc + n] = v;
                    }
                }
            }
    }
    if (g.stats_out) {
        // Per-row LayerNorm partials of this wave's NB*32 stored columns, as (mean, M2 = sum of squared deviations from that mean):
        // a lane first reduces its own NB values of a row (two passes over registers), then the 32 lanes of a half-wave -- the same
        // rows, 32 different columns -- are merged with the parallel-variance formula for equal counts c
        //   mean = (mean_a + mean_b) / 2,   M2 = M2_a + M2_b + (mean_b - mean_a)^2 * c / 2
        // in a butterfly that halves the rows kept per lane at every step (T/2, ..., 1 exchanges instead of 5 T): lane i ends
        // with row i's partial.  Shifted data throughout: no sum of raw squares that would cancel for |mean| >> std.
        // (launch_gemm_nt only passes stats_out when N is a whole number of column tiles: every column is a real column.)
        constexpr int T = MB * 16;
        float sm[T], s2[T];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v[NB], a1 = 0.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    v[nb] = acc[mb][nb][r] * g.alpha;
                    if (g.relu) v[nb] = fmaxf(v[nb], 0.f);
                    a1 += v[nb];
                }
                const float mu = a1 * (1.0f / NB);
                float a2 = 0.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) { const float d = v[nb] - mu; a2 = fmaf(d, d, a2); }
                sm[mb * 16 + r] = mu; s2[mb * 16 + r] = a2;
            }
        float cnt_half = 0.5f * NB;     // c / 2 of the groups being merged
#pragma unroll
        for (int o = T / 2; o >= 1; o >>= 1) {
            const bool up = (i & o) != 0;
#pragma unroll
            for (int j = 0; j < o; ++j) {
                const float km = up ? sm[j + o] : sm[j], tm = up ? sm[j] : sm[j + o];
                const float k2 = up ? s2[j + o] : s2[j], t2 = up ? s2[j] : s2[j + o];
                const float om = __shfl_xor(tm, o), o2 = __shfl_xor(t2, o);
                const float d = om - km;
                sm[j] = 0.5f * (km + om);
                s2[j] = (k2 + o2) + d * d * cnt_half;
            }
            cnt_half *= 2.0f;
        }
        if (T < 32) {   // 16 rows on 32 lanes: lane bit 4 still to fold
            const float om = __shfl_xor(sm[0], 16), o2 = __shfl_xor(s2[0], 16), d = om - sm[0];
            s2[0] = (s2[0] + o2) + d * d * cnt_half; sm[0] = 0.5f * (sm[0] + om);
        }
        // lane i (< T) of half h now holds local row t = i: mb = t / 16, r = t % 16 -> tile row (r & 3) + 8 (r >> 2) + 4 h
        if (i < T) {
            const int r = i & 15;
            const int m = m0 + (wm * MB + (i >> 4)) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < M) {
                const int P = (int)gridDim.x * 2;
                float* sp = g.stats_out + (((size_t)m + (size_t)z * g.M) * P + bxt * 2 + wn) * 2;
                sp[0] = sm[0]; sp[1] = s2[0];
            }
        }
    }
}

// the shapes whose projection can carry the rotary epilogue: what takes the plain k-permuted 128 x 256 tile below
bool gemm_nt_rope_ok(const GemmArgs& g) {
    static const bool kp_on = tune_int("RFE_GEMM_KP", 1) != 0, pft = tune_int("RFE_GEMM_PF", 1) != 0, dmab = tune_int("RFE_GEMM_DMA", 0) != 0;
    const int batch = g.batch > 0 ? g.batch : 1;
    return kp_on && pft && !dmab && g.kperm && batch == 1 && !g.m_valid && !g.R && !g.stats_in && !g.stats_out && !g.Bh && !g.relu && g.alpha == 1.0f && g.N % 256 == 0 &&
           g.rope_c0 % 256 == 0 && g.rope_c1 % 256 == 0 && (long long)((g.M + 127) / 128) * (g.N / 256) >= 256 && !gemm_latency_regime(g);
}

int launch_gemm_nt(hipStream_t s, const GemmArgs& g_in) {
    GemmArgs g = g_in;
    g.xcd = 0;
#ifdef RFE_TUNING
    g.abl = tune_int("RFE_DBG_GEMM_ABL", 0);
#endif
    const int batch = g.batch > 0 ? g.batch : 1;
    auto tiles = [&](int bm, int bn) { return (long long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * batch; };
    // RFE_OPT_LG_FP16X2: a LightGlue Linear whose weights come with fp16 (hi, lo) planes and whose shape takes the throughput tile
    if (g.rope_csn && !gemm_nt_rope_ok(g)) return -1;   // callers ask first
    if (g.Bh && g.Bl && batch == 1 && !g.m_valid && !g.relu && g.N % 256 == 0 && g.K % 32 == 0 && (!g.A2 || g.K1 % 32 == 0) && tiles(128, 256) >= 256)
        return launch_gemm_h2(s, g);
    if (!g.rope_csn && gemm_latency_regime(g)) {   // one / few pairs per call: the 16x16x4 latency tiling (gemm_lat.hip) when it serves the shape
        if (launch_gemm_lat(s, g, nullptr, 0)) return 0;
    }
    const bool res = g.R != nullptr, lna = g.stats_in != nullptr;
#ifdef RFE_TUNING
    static const bool ws_on = tune_int("RFE_GEMM_WS", 0) != 0;   // round 6, measured and not adopted: LayerNorm + GELU of the consumer on producer waves (gemm_ws.hip)
    if (lna && ws_on && res && launch_gemm_ln_ws(s, g)) return 0;
#endif
    static const bool lni = tune_int("RFE_LN_INTERLEAVE", 1) != 0;   // tuning switch

    static const bool kp_on = tune_int("RFE_GEMM_KP", 1) != 0;   // tuning switch: 0 = padded layout / k-ascending order everywhere
    const bool kp = kp_on && g.kperm != 0;
    static const bool pft = tune_int("RFE_GEMM_PF", 1) != 0;   // register prefetch of the next K tile also on the 128-row tiles (+1 % on ffn1 / ffn2, profiles/r02_ab_notes.md); RFE_GEMM_PF=0 (tuning build) disables
    static const bool dmab = tune_int("RFE_GEMM_DMA", 0) != 0;   // tuning switch, OFF in the product: B tile by LDS-DMA on the k-permuted throughput tile (ffn.0 269.5 -> 266.1 us, qkv 218.6 -> 216.3, cross-qkv unchanged: within 1.3 %, not worth a second code path under the headline -- profiles/r03_ab_notes.md)
    constexpr int kDmaLds = (128 + 2 * 256) * 32 * 4;            // 80 KB
#define RFE_GEMM_GO(MB_, NB_, GRID)                                                                  \
    do {                                                                                             \
        if (kp && MB_ == 2 && NB_ == 4 && pft && !lna && dmab && batch == 1 && !g.m_valid) {         \
            if (res) {                                                                               \
                static bool ls_[64]; ensure_dynamic_lds((const void*)gemm_nt_kernel<2, 4, true, true, false, true, true>, kDmaLds, ls_); \
                hipLaunchKernelGGL((gemm_nt_kernel<2, 4, true, true, false, true, true>), GRID, dim3(256), kDmaLds, s, g); \
            } else {                                                                                 \
                static bool ls_[64]; ensure_dynamic_lds((const void*)gemm_nt_kernel<2, 4, false, true, false, true, true>, kDmaLds, ls_); \
                hipLaunchKernelGGL((gemm_nt_kernel<2, 4, false, true, false, true, true>), GRID, dim3(256), kDmaLds, s, g); \
            }                                                                                        \
        } else                                                                                       \
        if (g.rope_csn && MB_ == 2 && NB_ == 4) {                                                    \
            static bool lr_[64]; ensure_dynamic_lds((const void*)gemm_nt_kernel<2, 4, false, true, false, true, false, true>, 128 * 64 * 4, lr_); \
            hipLaunchKernelGGL((gemm_nt_kernel<2, 4, false, true, false, true, false, true>), GRID, dim3(256), 128 * 64 * 4, s, g); \
        } else                                                                                       \
        if (kp && MB_ == 2 && pft && !lna) {                                                         \
            if (res) hipLaunchKernelGGL((gemm_nt_kernel<MB_, NB_, true, MB_ == 2, false, MB_ == 2>), GRID, dim3(256), 0, s, g); \
            else hipLaunchKernelGGL((gemm_nt_kernel<MB_, NB_, false, MB_ == 2, false, MB_ == 2>), GRID, dim3(256), 0, s, g);         \
        } else                                                                                       \
        if (lna && lni && MB_ == 2) hipLaunchKernelGGL((gemm_nt_kernel<MB_, NB_, true, MB_ == 2, true>), GRID, dim3(256), 0, s, g);   /* LN + GELU on A under the MFMAs */ \
        else if (lna) hipLaunchKernelGGL((gemm_nt_kernel<MB_, NB_, true, false, true>), GRID, dim3(256), 0, s, g);   /* LN + GELU on A at staging */ \
        else if (pft && MB_ == 2 && res) hipLaunchKernelGGL((gemm_nt_kernel<MB_, NB_, true, MB_ == 2>), GRID, dim3(256), 0, s, g);    \
        else if (pft && MB_ == 2) hipLaunchKernelGGL((gemm_nt_kernel<MB_, NB_, false, MB_ == 2>), GRID, dim3(256), 0, s, g);     \
        else if (res) hipLaunchKernelGGL((gemm_nt_kernel<MB_, NB_, true>), GRID, dim3(256), 0, s, g);    \
        else hipLaunchKernelGGL((gemm_nt_kernel<MB_, NB_, false>), GRID, dim3(256), 0, s, g);        \
    } while (0)
    // largest tile that still gives every CU a workgroup; small problems (single-pair latency) fall to 64 x 64
    // The LayerNorm statistics are only produced by the 128-row throughput tiles: in the latency regime (64-row tiles, one
    // workgroup per CU) the fused normalisation puts the erf on the critical path and measured slower than the stand-alone pass
    // (one pair: ffn.0 + ffn.3 + LN 0.74 -> 0.89 ms); the caller falls back to it when 0 comes back.
    const bool big = (g.N % 256 == 0 && tiles(128, 256) >= 256) || tiles(128, 128) >= 256 || g.M > 8192;
    if (!big || g.N % 256) g.stats_out = nullptr;   // partials cover whole column tiles only
    int ntiles;
    // XCD-aware tile decode (round 6; tuning build only, OFF in the product): only where it is a bijection and there is something to share -- several column tiles, a
    // whole number of 8 row panels, one batch.  Measured (tools/tune_sweep.py, three alternating runs, profiles/r06_ab_notes.md): qkv 1.991 -> 1.963 ms per step,
    // cross-qkv 1.333 -> 1.329, ffn.0 4.902 -> 4.918, and ffn.3 -- whose own decode does not change -- 3.113 -> 3.191: the step does not move (34.15 - 34.21 against
    // 34.11 - 34.18 ms).  The panels are not HBM-bound (440 MB in 224 us), so saving the A re-reads buys nothing.
    static const bool xcd_on = tune_int("RFE_GEMM_XCD", 0) != 0;
    auto xcd_ok = [&](int bm, int bn) { const int nct = (g.N + bn - 1) / bn, gy = (g.M + bm - 1) / bm; return xcd_on && batch == 1 && nct > 1 && gy % 8 == 0 && !g.m_valid; };
    if (g.N % 256 == 0 && tiles(128, 256) >= 256) { ntiles = g.N / 256; g.xcd = xcd_ok(128, 256) ? 1 : 0; RFE_GEMM_GO(2, 4, dim3(g.N / 256, (g.M + 127) / 128, batch)); }
    else if (tiles(128, 128) >= 256 || g.M > 8192) { ntiles = (g.N + 127) / 128; RFE_GEMM_GO(2, 2, dim3((g.N + 127) / 128, (g.M + 127) / 128, batch)); }
    else if (tiles(64, 128) >= 256) { ntiles = (g.N + 127) / 128; RFE_GEMM_GO(1, 2, dim3((g.N + 127) / 128, (g.M + 63) / 64, batch)); }
    else { ntiles = (g.N + 63) / 64; RFE_GEMM_GO(1, 1, dim3((g.N + 63) / 64, (g.M + 63) / 64, batch)); }
    return g.stats_out ? 2 * ntiles : 0;
}

}  // namespace rfe
