// gemm.hip -- fp32 "NT" GEMM on the CDNA4 matrix cores:  C[m][n] = alpha*(bias[n] + sum_k A[m][k]*B[n][k])
// (+ReLU, +residual).  Both operands are K-contiguous, i.e. B is a PyTorch Linear weight [N][K] or
// a second activation matrix (LightGlue similarity md0 . md1^T).
//
// Used for: SuperPoint 1x1 heads convPb (256->65) / convDb (256->256) and every LightGlue Linear
// (the arithmetic of superpoint.onnx / lightglue_sim.onnx that the reference runs through
// Ort::Session::Run, src/Extractors/superpoint_onnx.cc:135, src/Matchers/lightglue_onnx.cpp:213).
//
// Workgroup 256 threads = 2x2 waves, tile 128x128x32; each wave 64x64 = 2x2 v_mfma_f32_32x32x2_f32
// accumulators.  LDS tiles [128][33] (odd row stride -> the 32 rows of a fragment hit 32 banks).
// Reduction order: k ascending, accumulator initialised with the bias (== oracle rfo_linear, bit-exact).
// Roofline: fp32 MFMA peak 157.3 TFLOP/s, algorithmic 2*M*N*K FLOP.
#include "rfe_internal.h"

namespace rfe {

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 1;

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
    __shared__ float As[BM * LDT];
    __shared__ float Bs[BN * LDT];
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

    f32x16 acc[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int n = n0 + wn * 64 + nb * 32 + i;
        const float bv = (g.bias && n < g.N) ? g.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][nb][r] = bv; acc[1][nb][r] = bv; }
    }

    const int lrow = tid >> 3, lkq = tid & 7;  // 32 rows x 8 float4 per pass
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        __syncthreads();
        const float* Asrc = A; int lda = g.lda; int kk = k0;
        if (A2 && k0 >= g.K1) { Asrc = A2; lda = g.lda2; kk = k0 - g.K1; }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = lrow + 32 * it;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
            if (m0 + row < M) va = *reinterpret_cast<const float4*>(Asrc + (size_t)(m0 + row) * lda + kk + lkq * 4);
            if (n0 + row < g.N) vb = *reinterpret_cast<const float4*>(B + (size_t)(n0 + row) * g.ldb + k0 + lkq * 4);
            float* da = As + row * LDT + lkq * 4;
            float* db = Bs + row * LDT + lkq * 4;
            da[0] = va.x; da[1] = va.y; da[2] = va.z; da[3] = va.w;
            db[0] = vb.x; db[1] = vb.y; db[2] = vb.z; db[3] = vb.w;
        }
        __syncthreads();
        const float* ap = As + (wm * 64 + i) * LDT + h;
        const float* bp = Bs + (wn * 64 + i) * LDT + h;
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const float a0 = ap[2 * s], a1 = ap[32 * LDT + 2 * s];
            const float b0 = bp[2 * s], b1 = bp[32 * LDT + 2 * s];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    const float* R = g.R ? g.R + (size_t)z * g.sR : nullptr;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= M) continue;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int n = n0 + wn * 64 + nb * 32 + i;
                if (n >= g.N) continue;
                float v = acc[mb][nb][r] * g.alpha;
                if (g.relu) v = fmaxf(v, 0.f);
                if (R) v = R[(size_t)m * g.ldr + n] + v;
                C[(size_t)m * g.ldc + n] = v;
            }
        }
}

void launch_gemm_nt(hipStream_t s, const GemmArgs& g) {
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.batch > 0 ? g.batch : 1);
    hipLaunchKernelGGL(gemm_nt_kernel, grid, dim3(256), 0, s, g);
}

}  // namespace rfe
