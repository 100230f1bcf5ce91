// gemm_ws.hip -- LightGlue's ffn.3 at THROUGHPUT shapes with its LayerNorm + GELU on a separate set of waves (round 6):
//     C[m][n] = R[m][n] + bias[n] + sum_k gelu(LN(h[m][:]))[k] * W2[n][k]        (K = 512, N = 256, m < 2 P L rows, tens of thousands of them)
// the Linear -> LayerNorm -> GELU -> Linear tail of every LightGlue block inside lightglue_sim.onnx (reference: Ort::Session::Run at
// src/Matchers/lightglue_onnx.cpp:210-214).
//
// gemm_nt_kernel<2, 4, RES, PFT, LNA> normalises the A tile on the waves that multiply it: 16 elements x (LayerNorm affine + erf-GELU) per thread and K tile sit on the
// issue path of the same waves' matrix instructions, and the stage runs at 0.61 of the fp32 matrix peak where the plain tiles reach 0.78 (profiles/r05_pmc_wait.md:
// matrix pipe busy 0.66, twice the VALU issue of the plain form).  Here the two jobs belong to different waves of one workgroup:
//   * waves 8 .. 11 (PRODUCERS, 256 threads): the staging code of gemm_nt_kernel -- A rows from global memory a K tile ahead, per-row (mean, rstd) from ffn.0's
//     partials, gelu(((a - mean) rstd) g + b), one ds_write_b128 per float4 into the k-permuted swizzled tile -- plus the B tile (weights) by global_load_lds_dwordx4;
//   * waves 0 .. 7 (MATH, 2 x 4, 64 x 64 outputs each -- two per SIMD, so that one wave's fragment reads and barrier waits hide behind the other's matrix
//     instructions; a first form with four math waves of 64 x 128 ran at 182 us against the fused kernel's 173): gemm_nt_kernel's k-permuted matrix loop on the tile the
//     producers finished one barrier ago;
//   * two LDS stages of (128 + 256) rows x 32 k (96 KB, one workgroup per CU), ONE workgroup barrier per K tile: behind it stage t is complete AND the math waves have left
//     stage t - 1, which the producers fill next.
// The VALU work of a K tile (about 2 400 cycles per producer wave) runs next to the 2 x 64 matrix instructions of the two math waves on the same SIMD (about 10 000 cycles).
// Arithmetic, reduction order and epilogue are gemm_nt_kernel's k-permuted path: same results as the form it replaces.
// Roofline: fp32 MFMA peak 157.3 TFLOP/s, algorithmic 2 M N K FLOP.
#include "rfe_internal.h"

namespace rfe {

namespace {
typedef __attribute__((address_space(3))) void* ws_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* ws_gptr_t;
constexpr int WS_BM = 128, WS_BN = 256, WS_BK = 32;
constexpr int WS_A_F = WS_BM * WS_BK, WS_B_F = WS_BN * WS_BK, WS_STAGE_F = WS_A_F + WS_B_F;   // 4096 + 8192 floats = 48 KB per stage

__device__ __forceinline__ float ws_gelu(float t) {   // == gemm.hip gelu_short
    const float x = t * 0.70710678118654752f, ax = fabsf(x);
    const float k = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    const float poly = fmaf(fmaf(fmaf(fmaf(1.061405429f, k, -1.453152027f), k, 1.421413741f), k, -0.284496736f), k, 0.254829592f) * k;
    const float er = 1.0f - poly * __builtin_amdgcn_exp2f(-(ax * ax) * 1.44269504088896341f);
    return 0.5f * t * (1.0f + copysignf(er, x));
}
}  // namespace

__global__ __launch_bounds__(768, 1) void gemm_ln_ws_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 stages of [A 128 x 32 | B 256 x 32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * WS_BM, n0 = blockIdx.x * WS_BN;
    const int M = g.M;
    if (m0 >= M) return;
    const int T = g.K / WS_BK;
    int mlast = M - 1 - m0; mlast = mlast < WS_BM - 1 ? mlast : WS_BM - 1;

    if (wave >= 8) {
        // ------------------------------------------------------------------------------------------------ producers
        const int pt = tid - 512, pw = wave - 8;
        const int lrow = pt >> 3, lkq = pt & 7;
        const float* const At = g.A + (size_t)m0 * g.lda;
        const float* const Bt = g.B + (size_t)n0 * g.ldb;
        int aoff[4];
        float ln_mean[4], ln_rstd[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            int row = lrow + 32 * it; row = row < mlast ? row : mlast;
            aoff[it] = row * g.lda + lkq * 4;
            const float* sp = g.stats_in + (size_t)(m0 + row) * g.stats_p * 2;
            float ms = 0.f, m2 = 0.f;
            for (int p = 0; p < g.stats_p; ++p) { ms += sp[2 * p]; m2 += sp[2 * p + 1]; }
            const float mean = ms / (float)g.stats_p;
            float dev = 0.f;
            for (int p = 0; p < g.stats_p; ++p) { const float d = sp[2 * p] - mean; dev = fmaf(d, d, dev); }
            const float var = (m2 + dev * ((float)g.K / (float)g.stats_p)) / (float)g.K;
            ln_mean[it] = mean; ln_rstd[it] = 1.0f / sqrtf(var + 1e-5f);
        }
        // B tile: one wave instruction copies 64 x 16 B = 8 rows of 128 B; producer wave w issues the row groups 8 w .. 8 w + 7.  Lane l fills row 8 grp + l / 8,
        // physical slot l & 7, which holds logical slot (l & 7) ^ (row & 7) = (l & 7) ^ (l >> 3)  (gemm.hip: dma_b)
        int nlast = g.N - 1 - n0; nlast = nlast < WS_BN - 1 ? nlast : WS_BN - 1;
        int boff[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int row = (pw * 8 + u) * 8 + (lane >> 3); row = row < nlast ? row : nlast;
            boff[u] = row * g.ldb + 4 * ((lane & 7) ^ (lane >> 3));
        }
        float4 ra[4], rg, rbeta;
        auto load_a = [&](int t) {
            const int k0 = t * WS_BK;
            rg = *reinterpret_cast<const float4*>(g.ln_g + k0 + lkq * 4); rbeta = *reinterpret_cast<const float4*>(g.ln_b + k0 + lkq * 4);
#pragma unroll
            for (int it = 0; it < 4; ++it) ra[it] = *reinterpret_cast<const float4*>(At + k0 + aoff[it]);
        };
        load_a(0);
        for (int t = 0; t < T; ++t) {
            float* const As = lds + (t & 1) * WS_STAGE_F;
            float* const Bs = As + WS_A_F;
            // weights of tile t straight into LDS ...
            {
                const float* src = Bt + t * WS_BK;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    __builtin_amdgcn_global_load_lds((ws_gptr_t)(src + boff[u]), (ws_lds_ptr_t)(Bs + (pw * 8 + u) * 256), 16, 0, 0);
            }
            // ... the A rows of tile t (requested one tile ago): LayerNorm + GELU, same operation order as the stand-alone kernel
            const float gg[4] = {rg.x, rg.y, rg.z, rg.w}, bb[4] = {rbeta.x, rbeta.y, rbeta.z, rbeta.w};
            f32x4 va[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float e[4] = {ra[it].x, ra[it].y, ra[it].z, ra[it].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) va[it][q] = ws_gelu((e[q] - ln_mean[it]) * ln_rstd[it] * gg[q] + bb[q]);
            }
            if (t + 1 < T) load_a(t + 1);                 // next tile's rows: in flight under this tile's stores and the barrier
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = lrow + 32 * it;             // (row & 7) == (lrow & 7)
                *reinterpret_cast<f32x4*>(As + row * WS_BK + ((lkq ^ (lrow & 7)) << 2)) = va[it];
            }
            // stage t complete for this wave.  A full wait: it also covers the loads of tile t + 1, which costs nothing here -- a producer wave has about a quarter of
            // the math waves' time per tile to fill, and a counted wait would depend on the compiler keeping those loads behind the copies
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- math waves
    const int wm = wave >> 2, wn = wave & 3;     // 2 x 4 waves of 64 rows x 64 columns
    const int i = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int n = n0 + (wn * 2 + nb) * 32 + i;
        const float bv = (g.bias && n < g.N) ? g.bias[n] : 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = bv;
    }
    const int iswz = i & 7;
    for (int t = 0; t < T; ++t) {
        // stage t is complete (every producer has passed its wait) and -- for the producers -- this wave has left stage t - 1
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const float* const ap = lds + (t & 1) * WS_STAGE_F + (wm * 64 + i) * WS_BK;
        const float* const bp = lds + (t & 1) * WS_STAGE_F + WS_A_F + (wn * 64 + i) * WS_BK;
        f32x4 a4[2], b4[2];
        {
            const int slot = ((h << 2) ^ iswz) << 2;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) a4[mb] = *reinterpret_cast<const f32x4*>(ap + mb * 32 * WS_BK + slot);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) b4[nb] = *reinterpret_cast<const f32x4*>(bp + nb * 32 * WS_BK + slot);
        }
#pragma unroll
        for (int gq4 = 0; gq4 < 4; ++gq4) {
            const int nslot = (((h << 2) + gq4 + 1) ^ iswz) << 2;
            f32x4 an[2];
            if (gq4 < 3) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) an[mb] = *reinterpret_cast<const f32x4*>(ap + mb * 32 * WS_BK + nslot);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
                for (int cq = 0; cq < 4; ++cq)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[mb][cq], b4[nb][cq], acc[mb][nb], 0, 0, 0);
                if (gq4 < 3) { b4[nb] = *reinterpret_cast<const f32x4*>(bp + nb * 32 * WS_BK + nslot); __builtin_amdgcn_sched_barrier(0); }
            }
            if (gq4 < 3) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) a4[mb] = an[mb];
            }
        }
        // every fragment of this stage has been READ (the matrix instructions above consumed them) before the next barrier lets the producers refill it
    }

    // ---- epilogue (gemm_nt_kernel's): alpha, residual read in the accumulator layout and added after bias like the oracle, 128-byte coalesced rows
    float* C = g.C;
    const float* Rz = g.R;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            float rv[4][2];
            if (Rz) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    int m = m0 + (wm * 2 + mb) * 32 + rr + 8 * rq + 4 * h;
                    m = m < M ? m : M - 1;
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        int n = n0 + (wn * 2 + nb) * 32 + i;
                        n = n < g.N ? n : g.N - 1;
                        rv[rr][nb] = Rz[(size_t)m * g.ldr + n];
                    }
                }
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = rq * 4 + rr;
                const int m = m0 + (wm * 2 + mb) * 32 + rr + 8 * rq + 4 * h;
                if (m >= M) continue;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int n = n0 + (wn * 2 + nb) * 32 + i;
                    if (n >= g.N) continue;
                    float v = acc[mb][nb][r] * g.alpha;
                    if (Rz) v = rv[rr][nb] + v;
                    C[(size_t)m * g.ldc + n] = v;
                }
            }
        }
}

// Serves: the LayerNorm-fused consumer (stats_in) on the k-permuted path, one batch, one A source, N a multiple of 256, K a multiple of 32, enough rows for the
// 128 x 256 tile to fill the chip.  false = not served (nothing launched; launch_gemm_nt goes on to gemm_nt_kernel<.., LNA>).
bool launch_gemm_ln_ws(hipStream_t s, const GemmArgs& g) {
    const int batch = g.batch > 0 ? g.batch : 1;
    if (!g.stats_in || !g.kperm || batch != 1 || g.m_valid || g.A2 || g.relu || g.stats_out || g.Bh || g.rope_csn || g.N % WS_BN || g.K % WS_BK || g.K < 2 * WS_BK ||
        (g.lda % 4) || (g.ldb % 4) || !g.ln_g || !g.ln_b || g.stats_p < 1)
        return false;
    if ((long long)((g.M + WS_BM - 1) / WS_BM) * (g.N / WS_BN) < 256) return false;
    constexpr int bytes = 2 * WS_STAGE_F * 4;   // 96 KB
    static bool ls_[64];
    ensure_dynamic_lds((const void*)gemm_ln_ws_kernel, bytes, ls_);
    hipLaunchKernelGGL(gemm_ln_ws_kernel, dim3(g.N / WS_BN, (g.M + WS_BM - 1) / WS_BM), dim3(768), bytes, s, g);
    return true;
}

}  // namespace rfe
