// ffn2_lat.hip -- LightGlue's ffn.3 at the ONE-PAIR shape with LayerNorm(512) + GELU fused in:
//     C[m][n] = R[m][n] + bias[n] + sum_k gelu(LN(h[m][:]))[k] * W2[n][k]        (K = 512, N = 256, m < 2 P L rows, P = 1 .. 4 pairs)
// the shape the reference itself runs (src/Matchers/lightglue_onnx.cpp:168-172, batch 1).  It replaces, on the latency path, the stand-alone
// lg_ln_gelu pass (4.9 us x 18 per forward, an 8 MB round trip of h) + gemm_lat_kernel<1,1,2,4,6,RES> (9.6 us x 18, matrix pipe busy 0.37,
// profiles/r04_pmc_wait.md) -- VERDICT r04 item 3a.
//
// Why the round-4 fusion failed and what is different here.  Round 4 applied LN + GELU on the A fragments of gemm_lat's 64-row x 32-column
// tiles: 8 column workgroups per row panel x every wave of a workgroup re-evaluating the GELU of the whole panel = 8 x (and more) the
// transcendentals, nothing to hide them behind (17.6 us, profiles/r04_ab_notes.md).  Here the workgroup tile is 16 rows x 128 columns:
//   * the 16 x 512 activation panel (32 KB) is loaded ONCE per workgroup into LDS, its per-row statistics are computed in registers by the
//     two-pass formula (mean, then squared deviations: no cancellation for rows of large mean) -- no partial statistics from ffn.0's epilogue
//     are needed -- and LN + GELU are applied ONCE per element, cooperatively by the 512 threads, before the K loop; only 2 column workgroups
//     share a panel (2 x the transcendentals instead of >= 8 x, 32 elements per lane);
//   * the K loop then streams ONLY the weight rows (128 rows x 64 k = 32 KB per stage) through a four-stage global_load_lds ring (three
//     stages = 96 KB per CU in flight under the matrix instructions of the fourth), reading the activation fragments from the resident,
//     already normalised panel: one barrier per 64 k;
//   * 8 waves per workgroup (two per SIMD, each ONE 16 x 16 accumulator chain of v_mfma_f32_16x16x4_f32: the two chains of a SIMD cover each
//     other's 40-cycle dependent latency), grid = 2 x ceil(M / 16) = 256 workgroups for one pair -- every CU busy, where a 16 x 256 tile
//     (one workgroup per panel) would leave half the chip idle and need a 64 KB weight stage;
//   * fragment layout, slot swizzle (slot c of row R at c ^ (R & 15), applied on the source side of the weight copy), transposed product
//     (a lane ends with four consecutive output columns of one row: bias / residual / store are one 16-byte access each) and the k order
//     inside a 16-k group are gemm_lat.hip's; the GELU is lg_ln_gelu_kernel's (Abramowitz-Stegun erf, |error| <= 1.5e-7).
// LDS: 32 KB panel + 4 x 32 KB ring = 160 KB (one workgroup per CU).  Roofline: fp32 MFMA, algorithmic 2 M N K FLOP.
#include "rfe_internal.h"

namespace rfe {

typedef __attribute__((address_space(3))) void* f2l_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* f2l_gptr_t;

namespace {
constexpr int F2_K = 512, F2_BM = 16, F2_BN = 128, F2_LBK = 64, F2_ST = 4, F2_NW = 8;
constexpr int F2_PANEL_F = F2_BM * F2_K;            // 8192 floats
constexpr int F2_STAGE_F = F2_BN * F2_LBK;          // 8192 floats
constexpr int F2_NDMA = F2_BN / (4 * F2_NW);        // 4 copy instructions per wave and stage (one moves 4 rows of 256 B)
constexpr int F2_LDS_BYTES = F2_ST * F2_STAGE_F * 4;   // dynamic part (the ring); the panel is a static 32 KB array

__device__ __forceinline__ float f2_gelu(float t) {   // == lg_kernels.hip gelu_short_ / gemm.hip gelu_short: one GELU for every LightGlue tiling
    const float x = t * 0.70710678118654752f, ax = fabsf(x);
    const float k = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    const float poly = fmaf(fmaf(fmaf(fmaf(1.061405429f, k, -1.453152027f), k, 1.421413741f), k, -0.284496736f), k, 0.254829592f) * k;
    const float er = 1.0f - poly * __builtin_amdgcn_exp2f(-(ax * ax) * 1.44269504088896341f);
    return 0.5f * t * (1.0f + copysignf(er, x));
}
}  // namespace

__global__ __launch_bounds__(64 * F2_NW, 1) void ffn2_ln_lat_kernel(const float* __restrict__ h, const float* __restrict__ W, const float* __restrict__ bias,
                                                                    const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                                    const float* __restrict__ R, int ldr, float* __restrict__ C, int ldc, int M, int MT) {
    // two distinct LDS objects: the compiler's wait insertion must be able to tell the panel's ds_writes / ds_reads from the ring the copies fill
    // (one array for both makes it put s_waitcnt vmcnt(0) in front of every LDS access that follows a copy request)
    __shared__ __attribute__((aligned(16))) float panel[F2_PANEL_F];   // 16 x 512, normalised activations
    extern __shared__ __attribute__((aligned(16))) float ring[];       // 4 stages of 128 x 64 weight floats
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    // both column tiles of a row panel on one XCD (blocks are dealt round-robin over the 8 XCDs): the panel crosses the fabric once
    const int Lb = blockIdx.x, xcd = Lb & 7, t_ = Lb >> 3;
    const int nt = t_ & 1, mt = (t_ >> 1) * 8 + xcd;
    if (mt >= MT) return;
    const int n0 = nt * F2_BN, m0 = mt * F2_BM;

    // ---- every request of the prologue goes out back to back (register loads, then the first three weight stages; sched_barriers keep the
    //      scheduler from interleaving them with the statistics, which made three serial round trips out of one)
    // this thread's share of the activation panel: row tid >> 5, the four 16-byte slots c = (tid & 31) + 32 i of its 128
    const int prow = tid >> 5, psub = tid & 31;
    int pm = m0 + prow; pm = pm < M ? pm : M - 1;                 // rows past the edge are clamped (computed, never stored)
    f32x4 hv[4], gg[4], bb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) hv[i] = *reinterpret_cast<const f32x4*>(h + (size_t)pm * F2_K + 4 * (psub + 32 * i));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        gg[i] = *reinterpret_cast<const f32x4*>(ln_g + 4 * (psub + 32 * i));
        bb[i] = *reinterpret_cast<const f32x4*>(ln_b + 4 * (psub + 32 * i));
    }
    // accumulator starts at the bias (D layout: lane (r, q) holds columns n = 4 q .. + 3 of the wave's 16, row m = r)
    const int ncol = n0 + wave * 16 + 4 * q;
    f32x4 acc = *reinterpret_cast<const f32x4*>(bias + ncol);
    // ... and at the residual (requested here, consumed in the epilogue: its round trip hides under the K loop; in place on x, every lane
    // reads and later writes only its own four floats)
    const int mres = m0 + r < M ? m0 + r : M - 1;
    acc += *reinterpret_cast<const f32x4*>(R + (size_t)mres * ldr + ncol);
    __builtin_amdgcn_sched_barrier(0);

    // ---- weight ring.  Group G = 4 consecutive rows; wave w issues groups w, w + 8, ...; lane l fills row 4 G + (l >> 4), physical slot l & 15
    const float* src[F2_NDMA];
#pragma unroll
    for (int u = 0; u < F2_NDMA; ++u) {
        const int Rw = (wave + F2_NW * u) * 4 + (lane >> 4);
        src[u] = W + (size_t)(n0 + Rw) * F2_K + (((lane & 15) ^ (Rw & 15)) << 2);
    }
    auto issue = [&](int t, int st) {
#pragma unroll
        for (int u = 0; u < F2_NDMA; ++u)
            __builtin_amdgcn_global_load_lds((f2l_gptr_t)(src[u] + t * F2_LBK), (f2l_lds_ptr_t)(ring + st * F2_STAGE_F + (wave + F2_NW * u) * 4 * F2_LBK), 16, 0, 0);
    };
    constexpr int T = F2_K / F2_LBK;   // 8
#pragma unroll
    for (int u = 0; u < F2_ST - 1; ++u) issue(u, u);
    __builtin_amdgcn_sched_barrier(0);

    // ---- LayerNorm statistics of the row (two passes over the registers; the 32 lanes of a half-wave hold one row) + GELU, once per element
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) sum += (hv[i][0] + hv[i][1]) + (hv[i][2] + hv[i][3]);
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum * (1.0f / 512.0f);
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { hv[i][e] -= mean; var = fmaf(hv[i][e], hv[i][e], var); }
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) var += __shfl_xor(var, off);
    const float rs = 1.0f / sqrtf(var * (1.0f / 512.0f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = psub + 32 * i;                               // 16-byte slot of the 512-k row: k = 4 c .. 4 c + 3
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2_gelu(hv[i][e] * rs * gg[i][e] + bb[i][e]);
        // 64-k block c >> 4, slot c & 15 of row prow stored at (c & 15) ^ prow: the fragment reads below are conflict-free ds_read_b128
        *reinterpret_cast<f32x4*>(panel + (c >> 4) * (F2_BM * F2_LBK) + prow * F2_LBK + ((((c & 15) ^ prow)) << 2)) = o;
    }

    const float* const wfrag = ring + (wave * 16 + r) * F2_LBK;
    const float* const afrag = panel + r * F2_LBK;
    for (int t = 0; t < T; ++t) {
        // stage t has landed (this wave's copies: counted wait -- the younger stages stay in flight; everybody's: barrier, which at t = 0 also
        // publishes the normalised panel), and every wave has left stage t - 1, whose buffer the next request reuses
        __builtin_amdgcn_sched_barrier(0);   // nothing of stage t is read above its wait + barrier (the machine scheduler moves ds_reads across the asm statement otherwise)
        {
            int younger = T - 1 - t; younger = younger < F2_ST - 2 ? younger : F2_ST - 2;
            if (younger == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * F2_NDMA) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(F2_NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + F2_ST - 1 < T) issue(t + F2_ST - 1, (t + F2_ST - 1) % F2_ST);
        const float* const ws = wfrag + (t % F2_ST) * F2_STAGE_F;
        const float* const as = afrag + t * (F2_BM * F2_LBK);
        f32x4 a4[4], b4[4];
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            const int sl = ((4 * kg + q) ^ r) << 2;
            a4[kg] = *reinterpret_cast<const f32x4*>(ws + sl);
            b4[kg] = *reinterpret_cast<const f32x4*>(as + sl);
        }
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[kg][e], b4[kg][e], acc, 0, 0, 0);
    }

    // ---- epilogue: one 16-byte store (bias and residual went into the accumulator's start value)
    const int m = m0 + r;
    if (m < M) *reinterpret_cast<f32x4*>(C + (size_t)m * ldc + ncol) = acc;
}

// Serves: ffn.3 of a one- / few-pair forward (K = 512, N = 256, residual, LayerNorm parameters, at most 8192 rows).  h is the RAW output of
// ffn.0 (bias included, not normalised).  false = shape not served, nothing launched.
bool launch_ffn2_ln_lat(hipStream_t s, const float* h, const float* w2, const float* b2, const float* ln_g, const float* ln_b, const float* R, int ldr,
                        float* C, int ldc, int M) {
    static const bool on = tune_int("RFE_FFN2_LAT_FUSE", 1) != 0;   // tuning build: 0 = stand-alone lg_ln_gelu + gemm_lat (round 4)
    if (!on || M < 1 || M > 8192 || !h || !w2 || !b2 || !ln_g || !ln_b || !R || (ldr % 4) || (ldc % 4)) return false;
    const int MT = (M + F2_BM - 1) / F2_BM;
    static bool ls_[64];
    ensure_dynamic_lds((const void*)ffn2_ln_lat_kernel, F2_LDS_BYTES, ls_);
    hipLaunchKernelGGL(ffn2_ln_lat_kernel, dim3(2 * ((MT + 7) / 8 * 8)), dim3(64 * F2_NW), F2_LDS_BYTES, s, h, w2, b2, ln_g, ln_b, R, ldr, C, ldc, M, MT);
    return true;
}

}  // namespace rfe
