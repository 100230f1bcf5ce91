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
#include "rfe_internal.h"

namespace rfe {

constexpr int BM = 128, BK = 32, LDT = BK + 1;

// NB = 32-column MFMA blocks per wave: wave tile 64 x (NB*32), workgroup tile 128 x (NB*64).
// NB = 4 (128x256 tile, 49 KB LDS, 3 workgroups/CU) measured best for N % 256 == 0
// (tools/kbench/gemm_variants.hip: 119-132 TFLOP/s vs 110-125 for 128x128 double-buffered).
template <int NB>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
    constexpr int BN = NB * 64;
    __shared__ float lds_ab[(BM + BN) * LDT];   // A tile | B tile; reused for the rotary tables in the epilogue
    float* const As = lds_ab;
    float* const Bs = lds_ab + BM * LDT;
    static_assert((BM + BN) * LDT >= BM * 65, "rotary table staging must fit");
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

    f32x16 acc[2][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + (wn * NB + nb) * 32 + i;
        const float bv = (g.bias && n < g.N) ? g.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][nb][r] = bv; acc[1][nb][r] = bv; }
    }

    // staging: thread -> (row = tid/8 + 32*it, 4 consecutive k).  Rows past the M / N edge are CLAMPED
    // (their products land in accumulators that are never stored), so the loop has no bounds branches.
    constexpr int A_IT = BM / 32, B_IT = BN / 32;
    const int lrow = tid >> 3, lkq = tid & 7;
    size_t aoff[A_IT], a2off[A_IT], boff[B_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        int row = m0 + lrow + 32 * it; row = row < M ? row : M - 1;
        aoff[it] = (size_t)row * g.lda + lkq * 4;
        a2off[it] = (size_t)row * g.lda2 + lkq * 4;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        int row = n0 + lrow + 32 * it; row = row < g.N ? row : g.N - 1;
        boff[it] = (size_t)row * g.ldb + lkq * 4;
    }
    float* const da = As + lrow * LDT + lkq * 4;
    float* const db = Bs + lrow * LDT + lkq * 4;
    const float* const ap = As + (wm * 64 + i) * LDT + h;
    const float* const bp = Bs + (wn * NB * 32 + i) * LDT + h;

    for (int k0 = 0; k0 < g.K; k0 += BK) {
        float4 ra[A_IT], rb[B_IT];
        if (A2 && k0 >= g.K1) {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) ra[it] = *reinterpret_cast<const float4*>(A2 + a2off[it] + (k0 - g.K1));
        } else {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) ra[it] = *reinterpret_cast<const float4*>(A + aoff[it] + k0);
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) rb[it] = *reinterpret_cast<const float4*>(B + boff[it] + k0);
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int it = 0; it < A_IT; ++it) { float* d = da + it * 32 * LDT; d[0] = ra[it].x; d[1] = ra[it].y; d[2] = ra[it].z; d[3] = ra[it].w; }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) { float* d = db + it * 32 * LDT; d[0] = rb[it].x; d[1] = rb[it].y; d[2] = rb[it].z; d[3] = rb[it].w; }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const float a0 = ap[2 * s], a1 = ap[32 * LDT + 2 * s];
            float b[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b[nb] = bp[nb * 32 * LDT + 2 * s];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[nb], acc[0][nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[nb], acc[1][nb], 0, 0, 0);
        }
    }

    const float* R = g.R ? g.R + (size_t)z * g.sR : nullptr;
    const bool rope = g.rope_cs != nullptr;
    if (rope) {   // stage this tile's rotary tables [128 rows][32 cos | 32 sin] in the (now free) A/B LDS space
        __syncthreads();
        for (int idx = tid; idx < BM * 16; idx += 256) {
            const int row = idx >> 4, q4 = idx & 15;     // 16 float4 per row: 8 cos + 8 sin
            int m = m0 + row; m = m < M ? m : M - 1;
            const float* src = (q4 < 8 ? g.rope_cs : g.rope_sn) + (size_t)m * 32 + (q4 & 7) * 4;
            const float4 v4 = *reinterpret_cast<const float4*>(src);
            float* d = As + row * 65 + (q4 < 8 ? 0 : 32) + (q4 & 7) * 4;   // row stride 65: conflict-free reads below
            d[0] = v4.x; d[1] = v4.y; d[2] = v4.z; d[3] = v4.w;
        }
        __syncthreads();
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int m = m0 + ml;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int n = n0 + (wn * NB + nb) * 32 + i;
                float v = acc[mb][nb][r] * g.alpha;
                if (rope) {   // LightGlue rotary on (q,k): t' = t*cos + rot(t)*sin, rot pairs (2f,2f+1) -> (-t1, t0)
                    const float partner = __shfl_xor(v, 1);   // column n^1 of the same row
                    if (n < g.rope_ncols) {
                        const int f = (n & 63) >> 1;
                        const float c = As[ml * 65 + f], sn = As[ml * 65 + 32 + f];
                        v = (n & 1) ? v * c + partner * sn : v * c - partner * sn;
                    }
                }
                if (m >= M || n >= g.N) continue;
                if (g.relu) v = fmaxf(v, 0.f);
                if (R) v = R[(size_t)m * g.ldr + n] + v;
                C[(size_t)m * g.ldc + n] = v;
            }
        }
}

void launch_gemm_nt(hipStream_t s, const GemmArgs& g) {
    const int batch = g.batch > 0 ? g.batch : 1;
    if (g.N % 256 == 0) {
        dim3 grid(g.N / 256, (g.M + BM - 1) / BM, batch);
        hipLaunchKernelGGL(gemm_nt_kernel<4>, grid, dim3(256), 0, s, g);
    } else {
        dim3 grid((g.N + 127) / 128, (g.M + BM - 1) / BM, batch);
        hipLaunchKernelGGL(gemm_nt_kernel<2>, grid, dim3(256), 0, s, g);
    }
}

}  // namespace rfe
