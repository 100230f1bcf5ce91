// gemm_h2.hip -- fp32-class "NT" GEMM on the f16 matrix pipe by operand splitting (option RFE_OPT_LG_FP16X2, default OFF):
//   C[m][n] = alpha * (bias[n] + sum_k A[m][k] * B[n][k])   with   a = a_hi + a_lo,  b = b_hi + b_lo   in fp16
//   (a_hi = fp16(a), a_lo = fp16(a - a_hi): 22 of the 24 significand bits),   a.b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi
// = 3 v_mfma_f32_32x32x16_f16 (32 cycles each, fp32 accumulation) per 32 x 32 x 16 block against 8 v_mfma_f32_32x32x2_f32 (64 cycles
// each): nominally 5.3x.  Measured stand-alone at LightGlue's Linear shapes (tools/kbench/gemm_bf16x3.hip, profiles/r03_ab_notes.md):
// 2.1-2.6x the fp32 kernels of gemm.hip, error against float64 rms 3.2e-8 of sum|a||b| (fp32 fmaf chain: 2.8e-8), maxima below the
// fp32 chain's.  What the split does NOT carry: |a| >= 65520 (fp16 overflow -> inf) and the low bits of residuals below fp16's
// subnormal spacing 2^-24 (an absolute error of <= 3e-8 per operand, irrelevant next to sum|a||b| = O(1)): LightGlue's token states,
// attention contexts and LayerNorm'd activations are O(1 .. 100).  SuperPoint never takes this path (bit-exact by fmaf-chain
// equivalence).  The reference runs these Linears inside Session::Run(lightglue_sim.onnx), src/Matchers/lightglue_onnx.cpp:210-214.
//
// Weights (B) are split ONCE at load time into two fp16 planes (rfe_api.hip: set_lg_upload); activations (A) are split while they
// are staged into LDS (3 VALU per element pair, h2_split.h; LNA: LayerNorm + GELU of ffn.3's operand is applied in the same pass, before the
// split).  Tile 128 x 256 x 32, 256 threads = 2x2 waves of 64 x 128, two workgroups per CU (as gemm.hip); LDS: per plane [rows][32
// fp16] = 64-byte rows of four 16-byte slots, slot s = 2 c + h stored at s ^ ((row >> 2) & 3) -- ds_read_b128 fragment reads and the
// staging writes are bank-conflict free.  Epilogue (bias in the accumulator, alpha, residual, LayerNorm partials) as gemm.hip.
#include "rfe_internal.h"
#include "h2_split.h"

namespace rfe {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2_f16(float x, float y, uint32_t& hi, uint32_t& lo) { h2_split2(x, y, hi, lo); }   // h2_split.h

// fp32 [n] -> fp16 planes hi [n], lo [n] (n even)
__global__ void split_f16_kernel(const float* __restrict__ x, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    uint32_t h, l;
    h2_split2_sat(x[i], i + 1 < n ? x[i + 1] : 0.f, h, l);   // load-time pass: the saturating form costs nothing here
    if (i + 1 < n) { *reinterpret_cast<uint32_t*>(hi + i) = h; *reinterpret_cast<uint32_t*>(lo + i) = l; }
    else { hi[i] = (uint16_t)h; lo[i] = (uint16_t)l; }
}
void launch_split_f16(hipStream_t s, const float* x, uint16_t* hi, uint16_t* lo, size_t n) {
    hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)((n / 2 + 256) / 256)), dim3(256), 0, s, x, hi, lo, n);
}

__device__ __forceinline__ float gelu_short_h2(float t) {   // = gemm.hip:gelu_short (Abramowitz-Stegun 7.1.26)
    const float x = t * 0.70710678118654752f, ax = fabsf(x);
    const float k = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    const float poly = fmaf(fmaf(fmaf(fmaf(1.061405429f, k, -1.453152027f), k, 1.421413741f), k, -0.284496736f), k, 0.254829592f) * k;
    const float er = 1.0f - poly * __builtin_amdgcn_exp2f(-(ax * ax) * 1.44269504088896341f);
    return 0.5f * t * (1.0f + copysignf(er, x));
}

// DMA: the weight planes never touch a register -- they are copied by global_load_lds_dwordx4 into a DOUBLE-buffered LDS tile (2 x 32 KB;
// with the 16 KB A tile = 80 KB per workgroup: exactly two workgroups per CU, tools/kbench/lds_occupancy.hip), tile t + 1 requested
// right after tile t is published, together with the A rows of tile t + 1 (16 registers); the 16-byte-slot swizzle is applied on the
// SOURCE address (the LDS side of the copy is lane-linear: lane l of a wave instruction fills bytes 16 l .. of 1 KB = 16 rows).
// Without it (register staging of all twelve loads of a tile at its top) both workgroups of a CU end up waiting for L2 / HBM together.
typedef __attribute__((address_space(3))) void* h2_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* h2_gptr_t;
template <bool RES, bool LNA, bool DMA, int ABL = 0>   // ABL: timing ablations of the DMA variant (tuning build only, wrong results): 1 A staged once, 2 no barriers, 4 weights copied once, 8 no epilogue stores
__global__ __launch_bounds__(256, 2) void gemm_h2_kernel(GemmArgs g) {
    constexpr int BM = 128, BN = 256, BK = 32, MB = 2, NB = 4;
    constexpr int BPL = BN * 64;                      // bytes of one B plane of one K tile
    __shared__ __attribute__((aligned(16))) unsigned char lds_static[DMA ? 16 : 2 * (BM + BN) * 64];
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];   // DMA: 2 * BM * 64 + 2 * 2 * BPL = 80 KB
    unsigned char* const lds = DMA ? lds_dyn : lds_static;
    unsigned char* const As = lds;                    // [2 planes][BM][64 B]
    unsigned char* const Bs = lds + 2 * BM * 64;      // [2 planes][BN][64 B] (DMA: x 2 buffers)
    const int tid = threadIdx.x, lane = tid & 63;
    // Activations are split while they are staged (h2_split2): with the default mode one |a| >= 65520 becomes inf and lo = inf - inf = NaN for
    // its whole output row.  MODE.FP16_OVFL saturates instead (h2_split.h) -- set for the variants without the fused LayerNorm + GELU, where it
    // measured free (round-3 advisor finding; the LNA variant's operand has just been through a LayerNorm, |a| is O(10))
    if (!LNA) h2_saturate_mode();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    // XCD-aware block order (blocks are dealt round-robin to the 8 XCDs): the N / 256 column tiles of one 128-row panel are consecutive
    // workgroups of ONE XCD, so the panel's second (third) read of A hits that XCD's L2 instead of HBM -- at the f16 rate these
    // kernels run within 1.5x of their HBM time, and A read once instead of N / 256 times is a third of the traffic
    const int ncol = g.N / BN;
    const int xb = blockIdx.x & 7, tb = blockIdx.x >> 3;
    const int bx = tb % ncol, by = (tb / ncol) * 8 + xb;
    if (by * BM >= g.M) return;
    const int m0 = by * BM, n0 = bx * BN;
    const int M = g.M;

    f32x16 acc[MB][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float bv = g.bias ? g.bias[n0 + (wn * NB + nb) * 32 + i] : 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = bv;
    }

    // A staging: thread -> (row = tid / 8 + 32 it, 4 consecutive k); rows past the M edge are clamped (their products land in
    // accumulators that are never stored).  B staging: thread -> (row = tid / 4 + 64 p, one 16-byte slot = 8 k) of each plane.
    const int lrow = tid >> 3, lkq = tid & 7;
    int mlast = M - 1 - m0; mlast = mlast < BM - 1 ? mlast : BM - 1;
    const float* const At = g.A + (size_t)m0 * g.lda;
    const float* const A2t = g.A2 ? g.A2 + (size_t)m0 * g.lda2 : nullptr;
    int aoff[4], a2off[4], adst[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = lrow + 32 * it, rc = row < mlast ? row : mlast;
        aoff[it] = rc * g.lda + lkq * 4;
        a2off[it] = rc * g.lda2 + lkq * 4;
        adst[it] = row * 64 + (((lkq >> 1) ^ ((row >> 2) & 3)) << 4) + ((lkq & 1) << 3);   // 4 k = half a 16-byte slot
    }
    const int srow = tid >> 2, sslot = tid & 3;
    const int sw_off = srow * 64 + ((sslot ^ ((srow >> 2) & 3)) << 4);   // rows srow + 64 p: same (row >> 2) & 3
    const uint16_t* const Bh = g.Bh + (size_t)(n0 + srow) * g.ldb + sslot * 8;
    const uint16_t* const Bl = g.Bl + (size_t)(n0 + srow) * g.ldb + sslot * 8;
    const int frag_sw = (i >> 2) & 3;

    float ln_mean[4], ln_rstd[4];
    if (LNA) {   // LayerNorm statistics of this thread's four A rows from the producer's (mean, M2) partials: gemm.hip, same merge
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            int row = lrow + 32 * it; row = row < mlast ? row : mlast;
            const float* sp = g.stats_in + (size_t)(m0 + row) * g.stats_p * 2;
            float ms = 0.f, m2 = 0.f;
            for (int p = 0; p < g.stats_p; ++p) { ms += sp[2 * p]; m2 += sp[2 * p + 1]; }
            const float mean = ms / (float)g.stats_p;
            float dev = 0.f;
            for (int p = 0; p < g.stats_p; ++p) { const float d = sp[2 * p] - mean; dev = fmaf(d, d, dev); }
            const float var = (m2 + dev * ((float)g.K / (float)g.stats_p)) / (float)g.K;
            ln_mean[it] = mean; ln_rstd[it] = 1.0f / sqrtf(var + 1e-5f);
        }
    }

    // LNA (ffn.3): the A rows of K tile t + 1 are requested right after tile t is published and land under its MFMAs -- the LayerNorm /
    // GELU / split pass then starts on data that is already there (89.5 -> 74.5 us).  The plain variants keep all twelve loads at the top
    // of the tile they belong to: with the prefetch they measured 3 % slower (ffn.0, cross-qkv), and with the weight planes prefetched as
    // well (48 more live registers) everything spills -- profiles/r03_ab_notes.md.
    float4 fa[4];
    auto fetchA = [&](int k0) {
        if (A2t && k0 >= g.K1) {
            const float* base = A2t + (k0 - g.K1);
#pragma unroll
            for (int it = 0; it < 4; ++it) fa[it] = *reinterpret_cast<const float4*>(base + a2off[it]);
        } else {
            const float* base = At + k0;
#pragma unroll
            for (int it = 0; it < 4; ++it) fa[it] = *reinterpret_cast<const float4*>(base + aoff[it]);
        }
    };
    // DMA: one wave instruction copies 64 x 16 B = 1 KB = 16 rows of one plane; wave w issues the row groups 4 w .. 4 w + 3 of both planes.
    // Lane l fills row 16 grp + l / 4, physical slot l & 3, which holds logical slot (l & 3) ^ ((row >> 2) & 3) = (l & 3) ^ (l >> 4).
    auto dmaB = [&](int k0, int b) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int grp = wave * 4 + u;
            const size_t src = (size_t)(n0 + grp * 16 + (lane >> 2)) * g.ldb + k0 + 8 * ((lane & 3) ^ (lane >> 4));
            unsigned char* const dst = Bs + b * (2 * BPL) + grp * 1024;
            __builtin_amdgcn_global_load_lds((h2_gptr_t)(g.Bh + src), (h2_lds_ptr_t)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((h2_gptr_t)(g.Bl + src), (h2_lds_ptr_t)(dst + BPL), 16, 0, 0);
        }
    };
    auto stageA = [&](int k0) {   // [LayerNorm + GELU,] split, write the A planes
        float4 rg, rbeta;
        if (LNA) { rg = *reinterpret_cast<const float4*>(g.ln_g + k0 + lkq * 4); rbeta = *reinterpret_cast<const float4*>(g.ln_b + k0 + lkq * 4); }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            float4 v = fa[it];
            if (LNA) {   // same operation order as lg_ln_gelu / gemm.hip: ((a - mean) * rstd) * g + b, then GELU
                v.x = gelu_short_h2((v.x - ln_mean[it]) * ln_rstd[it] * rg.x + rbeta.x);
                v.y = gelu_short_h2((v.y - ln_mean[it]) * ln_rstd[it] * rg.y + rbeta.y);
                v.z = gelu_short_h2((v.z - ln_mean[it]) * ln_rstd[it] * rg.z + rbeta.z);
                v.w = gelu_short_h2((v.w - ln_mean[it]) * ln_rstd[it] * rg.w + rbeta.w);
            }
            uint32_t h0, l0, h1, l1;
            split2_f16(v.x, v.y, h0, l0);
            split2_f16(v.z, v.w, h1, l1);
            *reinterpret_cast<u32x2*>(As + adst[it]) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(As + BM * 64 + adst[it]) = u32x2{l0, l1};
        }
    };
    if (DMA) { dmaB(0, 0); fetchA(0); }
    else if (LNA) fetchA(0);
    int bbuf = 0;
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        const unsigned char* Bt = Bs;                 // this tile's B planes
        if (DMA) {
            // #1: this tile's weight planes and A rows have landed (requested one tile ago); every wave has finished the previous tile
            if (ABL & 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (!(ABL & 1) || k0 == 0) stageA(k0);
            if (ABL & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // #2: the A planes are visible
            // in flight under this tile's MFMAs.  UNCONDITIONAL (the last tile re-requests itself into the idle buffer): behind a branch
            // the compiler merges the loaded registers with copies and waits for the loads right here
            { const int kn = k0 + BK < g.K ? k0 + BK : k0; if (!(ABL & 4)) dmaB(kn, bbuf ^ 1); if (!(ABL & 1)) fetchA(kn); }
            __builtin_amdgcn_sched_barrier(0);   // ... and without this fence the scheduler sinks the A loads below the MFMAs, next to the barrier that waits for them
            Bt = Bs + bbuf * (2 * BPL);
            bbuf ^= 1;
        } else {
            if (!LNA) fetchA(k0);
            u32x4 rh[4], rl[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                rh[p] = *reinterpret_cast<const u32x4*>(Bh + (size_t)64 * p * g.ldb + k0);
                rl[p] = *reinterpret_cast<const u32x4*>(Bl + (size_t)64 * p * g.ldb + k0);
            }
            __syncthreads();   // previous tile consumed
            stageA(k0);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                *reinterpret_cast<u32x4*>(Bs + p * 64 * 64 + sw_off) = rh[p];
                *reinterpret_cast<u32x4*>(Bs + BN * 64 + p * 64 * 64 + sw_off) = rl[p];
            }
            __syncthreads();
            if (LNA) { fetchA(k0 + BK < g.K ? k0 + BK : k0); __builtin_amdgcn_sched_barrier(0); }   // unconditional, fenced: see the DMA branch
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int slot = ((2 * c + h) ^ frag_sw) << 4;
            f16x8 ah[MB], al[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                ah[mb] = *reinterpret_cast<const f16x8*>(As + ((wm * MB + mb) * 32 + i) * 64 + slot);
                al[mb] = *reinterpret_cast<const f16x8*>(As + BM * 64 + ((wm * MB + mb) * 32 + i) * 64 + slot);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const f16x8 bh = *reinterpret_cast<const f16x8*>(Bt + ((wn * NB + nb) * 32 + i) * 64 + slot);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(Bt + BN * 64 + ((wn * NB + nb) * 32 + i) * 64 + slot);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {   // small terms first
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mb], bh, acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bl, acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bh, acc[mb][nb], 0, 0, 0);
                }
            }
        }
    }

    if (!(ABL & 8)) {   // alpha (+residual) epilogue: 128-B coalesced accesses straight from the D layout (gemm.hip)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                float rv[4][NB];
                if (RES) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        int m = m0 + (wm * MB + mb) * 32 + rr + 8 * rq + 4 * h;
                        m = m < M ? m : M - 1;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) rv[rr][nb] = g.R[(size_t)m * g.ldr + n0 + (wn * NB + nb) * 32 + i];
                    }
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = rq * 4 + rr;
                    const int m = m0 + (wm * MB + mb) * 32 + rr + 8 * rq + 4 * h;
                    if (m >= M) continue;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        float v = acc[mb][nb][r] * g.alpha;
                        if (RES) v = rv[rr][nb] + v;
                        g.C[(size_t)m * g.ldc + n0 + (wn * NB + nb) * 32 + i] = v;
                    }
                }
            }
    }
    if (g.stats_out) {   // per-row LayerNorm partials (mean, M2) of this wave's 128 stored columns: gemm.hip, same butterfly
        constexpr int T = MB * 16;
        float sm[T], s2[T];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v[NB], a1 = 0.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) { v[nb] = acc[mb][nb][r] * g.alpha; a1 += v[nb]; }
                const float mu = a1 * (1.0f / NB);
                float a2 = 0.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) { const float d = v[nb] - mu; a2 = fmaf(d, d, a2); }
                sm[mb * 16 + r] = mu; s2[mb * 16 + r] = a2;
            }
        float cnt_half = 0.5f * NB;
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
        {
            const int r = i & 15;
            const int m = m0 + (wm * MB + (i >> 4)) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < M) {
                const int P = ncol * 2;
                float* sp = g.stats_out + ((size_t)m * P + bx * 2 + wn) * 2;
                sp[0] = sm[0]; sp[1] = s2[0];
            }
        }
    }
}

// Called by launch_gemm_nt for the shapes it would give the 128 x 256 throughput tile (N % 256 == 0, K % 32 == 0, no batch / m_valid /
// relu): returns the number of partial-statistics pairs per row it wrote (0 without stats_out), like launch_gemm_nt.
int launch_gemm_h2(hipStream_t s, const GemmArgs& g) {
    const int panels8 = (((g.M + 127) / 128) + 7) / 8 * 8;   // row panels padded to a multiple of 8: the block -> (panel, column tile) decode stays bijective
    const dim3 grid((unsigned)(panels8 * (g.N / 256)));
    const bool res = g.R != nullptr, lna = g.stats_in != nullptr;
    static const bool dma = tune_int("RFE_H2_DMA", 1) != 0;   // tuning switch: 0 = weight planes staged through registers
    constexpr int kDmaLds = 2 * 128 * 64 + 2 * 2 * 256 * 64;   // 80 KB
#define RFE_H2_LAUNCH(RES_, LNA_)                                                                                                     \
    do {                                                                                                                              \
        if (dma) {                                                                                                                    \
            static bool lds_set[64];                                                                                                  \
            ensure_dynamic_lds((const void*)gemm_h2_kernel<RES_, LNA_, true>, kDmaLds, lds_set);                                      \
            hipLaunchKernelGGL((gemm_h2_kernel<RES_, LNA_, true>), grid, dim3(256), kDmaLds, s, g);                                   \
        } else {                                                                                                                      \
            hipLaunchKernelGGL((gemm_h2_kernel<RES_, LNA_, false>), grid, dim3(256), 0, s, g);                                        \
        }                                                                                                                             \
    } while (0)
#ifdef RFE_TUNING
    if (!lna && !res) switch (tune_int("RFE_DBG_H2_ABL", 0)) {
#define RFE_H2_ABL(n) case n: { static bool ls_[64]; ensure_dynamic_lds((const void*)gemm_h2_kernel<false, false, true, n>, kDmaLds, ls_); \
                                hipLaunchKernelGGL((gemm_h2_kernel<false, false, true, n>), grid, dim3(256), kDmaLds, s, g); return 0; }
        RFE_H2_ABL(1) RFE_H2_ABL(2) RFE_H2_ABL(3) RFE_H2_ABL(4) RFE_H2_ABL(7) RFE_H2_ABL(8) RFE_H2_ABL(15)
#undef RFE_H2_ABL
        default: break;
    }
#endif
    if (lna && res) RFE_H2_LAUNCH(true, true);
    else if (lna) RFE_H2_LAUNCH(false, true);
    else if (res) RFE_H2_LAUNCH(true, false);
    else RFE_H2_LAUNCH(false, false);
#undef RFE_H2_LAUNCH
    return g.stats_out ? 2 * (g.N / 256) : 0;
}

}  // namespace rfe
