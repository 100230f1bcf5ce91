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

// NB = 32-column MFMA blocks per wave: wave tile 64 x (NB*32), workgroup tile 128 x (NB*64).
// NB = 4 (128x256 tile, 49 KB LDS, 3 workgroups/CU) measured best for N % 256 == 0
// (tools/kbench/gemm_variants.hip: 119-132 TFLOP/s vs 110-125 for 128x128 double-buffered).
// MB = 32-row MFMA blocks per wave (2 for throughput; 1 gives 64-row tiles for small, latency-bound problems:
// a single 1024-keypoint pair has only M = 2048 rows, 16 tiles of 128 x 256 would use 16 of the 256 CUs).
// TEPI: LDS-transposed whole-row epilogue (rotary variant); false = per-lane 4-byte stores.  The residual variant reads
// R in the D layout (four rows of 128-B segments in flight per step) and adds it after bias / alpha like the oracle.
template <int MB, int NB, bool TEPI, bool PFT = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
    constexpr int BM = MB * 64, BN = NB * 64;
    __shared__ float lds_ab[(BM + BN) * LDT];   // A tile | B tile; reused for the rotary tables in the epilogue
    float* const As = lds_ab;
    float* const Bs = lds_ab + BM * LDT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    const int z = blockIdx.z;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    int M = g.M;
    if (g.m_valid) { M = g.m_valid[z]; if (M > g.M) M = g.M; }
    if (m0 >= M) return;

    const float* A = g.A + (size_t)z * g.sA;
    const float* A2 = g.A2 ? g.A2 + (size_t)z * g.sA2 : nullptr;
    const float* B = g.B + (size_t)z * g.sB;
    float* C = g.C + (size_t)z * g.sC;

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
    float* const da = As + lrow * LDT + lkq * 4;
    float* const db = Bs + lrow * LDT + lkq * 4;
    const float* const ap = As + (wm * MB * 32 + i) * LDT + h;
    const float* const bp = Bs + (wn * NB * 32 + i) * LDT + h;

    // Small (64-row) tiles serve latency-bound problems (one pair: M = 2048, one workgroup per CU), where every K step
    // would otherwise expose a full global-memory round trip: there the next tile's loads are issued before the MFMA
    // loop of the current one (16 VGPRs).  The throughput tiles rely on the other resident workgroups instead
    // (register prefetch measured slower there).
    constexpr bool PF = (MB == 1) || PFT;
    float4 ra[A_IT], rb[B_IT];
    auto load_tile = [&](int k0) {
        if (A2t && k0 >= g.K1) {
            const float* base = A2t + (k0 - g.K1);
#pragma unroll
            for (int it = 0; it < A_IT; ++it) ra[it] = *reinterpret_cast<const float4*>(base + a2off[it]);
        } else {
            const float* base = At + k0;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) ra[it] = *reinterpret_cast<const float4*>(base + aoff[it]);
        }
        const float* base = Bt + k0;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) rb[it] = *reinterpret_cast<const float4*>(base + boff[it]);
    };
    if (PF) load_tile(0);
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        if (!PF) load_tile(k0);
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int it = 0; it < A_IT; ++it) { float* d = da + it * 32 * LDT; d[0] = ra[it].x; d[1] = ra[it].y; d[2] = ra[it].z; d[3] = ra[it].w; }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) { float* d = db + it * 32 * LDT; d[0] = rb[it].x; d[1] = rb[it].y; d[2] = rb[it].z; d[3] = rb[it].w; }
        __syncthreads();
        if (PF && k0 + BK < g.K) load_tile(k0 + BK);
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float a[MB], b[NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) a[mb] = ap[mb * 32 * LDT + 2 * s];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b[nb] = bp[nb * 32 * LDT + 2 * s];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb], b[nb], acc[mb][nb], 0, 0, 0);
        }
    }

    // workgroups whose column tile holds no rotary columns (the q and v thirds of the qkv projection) take the direct epilogue
    const bool wg_rope = g.rope_cs != nullptr && n0 < g.rope_ncols && n0 + BN > g.rope_n0;
    if (!TEPI || (g.rend && g.R && !g.rope_cs) || (g.rope_cs && !wg_rope && !g.R)) {   // plain bias (+alpha, +ReLU, +residual) epilogue: 128-B coalesced accesses straight from the D layout
        const float* Rz = (TEPI && g.R) ? g.R + (size_t)z * g.sR : nullptr;
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
                    if (m >= M) continue;
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
        return;
    }
    // ---- epilogue: accumulators (lane = column) are transposed through the now free A/B LDS space in 32-row
    // chunks so that residual loads and result stores are whole-row float4 accesses (4x fewer memory
    // instructions than per-lane 4-byte accesses; the residual read alone cost 25 % of ffn2 before).
    const float* R = g.R ? g.R + (size_t)z * g.sR : nullptr;
    const bool rope = g.rope_cs != nullptr;
    constexpr int CS = BN + 4;                 // chunk row stride (floats), keeps rows 16-byte aligned
    static_assert(32 * CS <= (BM + BN) * LDT, "epilogue chunk must fit in the tile LDS");
    float* const ch = lds_ab;
    float* const tab = lds_ab + 32 * CS;       // rotary tables of the chunk rows: [32][32 cos | 32 sin]
    static_assert(32 * CS + 32 * 64 <= (BM + BN) * LDT, "rotary table staging must fit");
    const bool vec_ok = (g.ldc % 4 == 0) && (g.N % 4 == 0) && (!R || g.ldr % 4 == 0);
#pragma unroll 1
    for (int c4 = 0; c4 < 2 * MB; ++c4) {      // chunk = 32 rows: wave row wm = c4 / MB, M-block mb = c4 % MB
        __syncthreads();
        if (wm == c4 / MB) {
            const int mb = c4 % MB;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    // the two M-blocks are distinct registers: select without dynamic indexing
                    const float v = (MB > 1 && mb) ? acc[MB - 1][nb][r] : acc[0][nb][r];
                    ch[row * CS + (wn * NB + nb) * 32 + i] = v * g.alpha;
                }
            }
        }
        const int mbase = m0 + c4 * 32;
        if (rope) {
            for (int idx = tid; idx < 32 * 16; idx += 256) {
                const int row = idx >> 4, q4 = idx & 15;     // 16 float4 per row: 8 cos + 8 sin
                int m = mbase + row; m = m < M ? m : M - 1;
                const float* src = (q4 < 8 ? g.rope_cs : g.rope_sn) + (size_t)m * 32 + (q4 & 7) * 4;
                *reinterpret_cast<float4*>(tab + row * 64 + q4 * 4) = *reinterpret_cast<const float4*>(src);
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 32 * (BN / 4); idx += 256) {
            const int row = idx / (BN / 4), q = idx % (BN / 4);
            const int m = mbase + row, n = n0 + q * 4;
            if (m >= M || n >= g.N) continue;
            float4 v = *reinterpret_cast<const float4*>(ch + row * CS + q * 4);
            if (rope && n < g.rope_ncols && n >= g.rope_n0) {   // LightGlue rotary: (t0,t1) -> (t0 c - t1 s, t1 c + t0 s), pairs (2f,2f+1), f = (n%64)/2
                const int f = (n & 63) >> 1;
                const float2 c2 = *reinterpret_cast<const float2*>(tab + row * 64 + f);
                const float2 s2 = *reinterpret_cast<const float2*>(tab + row * 64 + 32 + f);
                v = make_float4(v.x * c2.x - v.y * s2.x, v.y * c2.x + v.x * s2.x, v.z * c2.y - v.w * s2.y, v.w * c2.y + v.z * s2.y);
            }
            if (g.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            float* dst = C + (size_t)m * g.ldc + n;
            if (vec_ok && n + 3 < g.N) {
                if (R) { const float4 rv = *reinterpret_cast<const float4*>(R + (size_t)m * g.ldr + n); v.x = rv.x + v.x; v.y = rv.y + v.y; v.z = rv.z + v.z; v.w = rv.w + v.w; }
                *reinterpret_cast<float4*>(dst) = v;
            } else {
                const float* rp = R ? R + (size_t)m * g.ldr + n : nullptr;
                dst[0] = rp ? rp[0] + v.x : v.x;
                if (n + 1 < g.N) dst[1] = rp ? rp[1] + v.y : v.y;
                if (n + 2 < g.N) dst[2] = rp ? rp[2] + v.z : v.z;
                if (n + 3 < g.N) dst[3] = rp ? rp[3] + v.w : v.w;
            }
        }
    }
}

void launch_gemm_nt(hipStream_t s, const GemmArgs& g_in) {
    static const int rend_on = tune_int("RFE_GEMM_REND", 1);   // A/B switch, see profiles/r01_pmc.md
    GemmArgs g = g_in;
    g.rend = rend_on;
    const int batch = g.batch > 0 ? g.batch : 1;
    auto tiles = [&](int bm, int bn) { return (long long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * batch; };
    static const bool force_tepi = tune_env("RFE_GEMM_TEPI") != nullptr;   // tuning switch
    const bool tepi = force_tepi || g.R != nullptr || g.rope_cs != nullptr;
    static const bool pft = tune_int("RFE_GEMM_PF", 1) != 0;   // register prefetch of the next K tile also on the 128-row tiles (+1 % on ffn1 / ffn2, profiles/r02_pmc.md); RFE_GEMM_PF=0 (tuning build) disables
#define RFE_GEMM_GO(MB_, NB_, GRID)                                                                  \
    do {                                                                                             \
        if (pft && MB_ == 2 && tepi) hipLaunchKernelGGL((gemm_nt_kernel<MB_, NB_, true, MB_ == 2>), GRID, dim3(256), 0, s, g);    \
        else if (pft && MB_ == 2) hipLaunchKernelGGL((gemm_nt_kernel<MB_, NB_, false, MB_ == 2>), GRID, dim3(256), 0, s, g);       \
        else if (tepi) hipLaunchKernelGGL((gemm_nt_kernel<MB_, NB_, true>), GRID, dim3(256), 0, s, g);    \
        else hipLaunchKernelGGL((gemm_nt_kernel<MB_, NB_, false>), GRID, dim3(256), 0, s, g);        \
    } while (0)
    // largest tile that still gives every CU a workgroup; small problems (single-pair latency) fall to 64 x 64
    if (g.N % 256 == 0 && tiles(128, 256) >= 256) RFE_GEMM_GO(2, 4, dim3(g.N / 256, (g.M + 127) / 128, batch));
    else if (tiles(128, 128) >= 256 || g.M > 8192) RFE_GEMM_GO(2, 2, dim3((g.N + 127) / 128, (g.M + 127) / 128, batch));
    else if (tiles(64, 128) >= 256) RFE_GEMM_GO(1, 2, dim3((g.N + 127) / 128, (g.M + 63) / 64, batch));
    else RFE_GEMM_GO(1, 1, dim3((g.N + 63) / 64, (g.M + 63) / 64, batch));
}

}  // namespace rfe
