// kbench: bisecting why the product GEMM (112 TF) trails the bare tiling (130 TF) on the same shape.
// V bit0: clamp rows, bit1: struct kernarg (vs scalars), bit2: static LDS (vs dynamic), bit3: runtime alpha/relu/bounds
// in the epilogue, bit4: A2 split branch in the loop, bit5: offsets hoisted out of the loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct Args { const float* A; int lda; const float* A2; int lda2; int K1; const float* B; int ldb; const float* bias; const float* R; int ldr;
              float* C; int ldc; int M, N, K; float alpha; int relu; long long sA, sA2, sB, sC, sR; const int* m_valid; int batch;
              const float* rope_cs; const float* rope_sn; int rope_ncols; };
constexpr int BM = 128, BN = 256, BK = 32, LDT = 33, NB = 4, MB = 2;

template <int V>
__global__ __launch_bounds__(256, 2) void k(Args g, const float* A_, const float* B_, const float* bias_, float* C_, int M_, int N_, int K_) {
    extern __shared__ float dyn[];
    __shared__ float stat[(V & 4) ? (BM + BN) * LDT : 1];
    float* lds_ab = (V & 4) ? stat : dyn;
    float* const As = lds_ab; float* const Bs = lds_ab + BM * LDT;
    const float* A = (V & 2) ? g.A : A_; const float* B = (V & 2) ? g.B : B_; const float* bias = (V & 2) ? g.bias : bias_;
    float* C = (V & 2) ? g.C : C_; const int M = (V & 2) ? g.M : M_, N = (V & 2) ? g.N : N_, K = (V & 2) ? g.K : K_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float bv = bias[n0 + (wn * NB + nb) * 32 + i];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = bv;
    }
    const int lrow = tid >> 3, lkq = tid & 7;
    int aoff[4], boff[8];
#pragma unroll
    for (int it = 0; it < 4; ++it) { int row = lrow + 32 * it; if (V & 1) { int ml = M - 1 - m0; ml = ml < BM - 1 ? ml : BM - 1; row = row < ml ? row : ml; } aoff[it] = row * K + lkq * 4; }
#pragma unroll
    for (int it = 0; it < 8; ++it) { int row = lrow + 32 * it; if (V & 1) { int nl = N - 1 - n0; nl = nl < BN - 1 ? nl : BN - 1; row = row < nl ? row : nl; } boff[it] = row * K + lkq * 4; }
    const float* At = A + (size_t)m0 * K; const float* Bt = B + (size_t)n0 * K;
    for (int k0 = 0; k0 < K; k0 += BK) {
        float4 ra[4], rb[8];
        __syncthreads();
        if ((V & 16) && g.A2 && k0 >= g.K1) {
#pragma unroll
            for (int it = 0; it < 4; ++it) ra[it] = *reinterpret_cast<const float4*>(g.A2 + (size_t)m0 * g.lda2 + (k0 - g.K1) + aoff[it]);
        } else {
#pragma unroll
            for (int it = 0; it < 4; ++it) ra[it] = *reinterpret_cast<const float4*>(At + k0 + ((V & 32) ? aoff[it] : (lrow + 32 * it) * K + lkq * 4));
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) rb[it] = *reinterpret_cast<const float4*>(Bt + k0 + ((V & 32) ? boff[it] : (lrow + 32 * it) * K + lkq * 4));
#pragma unroll
        for (int it = 0; it < 4; ++it) { float* d = As + (lrow + 32 * it) * LDT + lkq * 4; d[0] = ra[it].x; d[1] = ra[it].y; d[2] = ra[it].z; d[3] = ra[it].w; }
#pragma unroll
        for (int it = 0; it < 8; ++it) { float* d = Bs + (lrow + 32 * it) * LDT + lkq * 4; d[0] = rb[it].x; d[1] = rb[it].y; d[2] = rb[it].z; d[3] = rb[it].w; }
        __syncthreads();
        const float* ap = As + (wm * 64 + i) * LDT + h; const float* bp = Bs + (wn * NB * 32 + i) * LDT + h;
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
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if ((V & 8) && m >= M) continue;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int n = n0 + (wn * NB + nb) * 32 + i;
                if ((V & 8) && n >= N) continue;
                float v = acc[mb][nb][r];
                if (V & 8) { v *= g.alpha; if (g.relu) v = fmaxf(v, 0.f); }
                C[(size_t)m * N + n] = v;
            }
        }
}

template <int V> static void run(const char* name, Args g) {
    const size_t lds = (V & 4) ? 0 : (size_t)(BM + BN) * LDT * 4;
    auto kern = k<V>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(g.N / BN, g.M / BM);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, g, g.A, g.B, g.bias, g.C, g.M, g.N, g.K);
    hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, g, g.A, g.B, g.bias, g.C, g.M, g.N, g.K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  V=%2d %-44s %.1f us  %.1f TF\n", V, name, ms / 20 * 1e3, 2.0 * g.M * g.N * g.K * 20 / (ms * 1e-3) / 1e12);
}
int main() {
    const int M = 65536, N = 256, K = 512;
    float *A, *B, *bias, *C;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&bias, N * 4); hipMalloc(&C, (size_t)M * N * 4);
    std::vector<float> h((size_t)M * K);
    for (size_t q = 0; q < h.size(); ++q) h[q] = (float)((q * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, h.data(), N * 4, hipMemcpyHostToDevice);
    Args g{}; g.A = A; g.lda = K; g.B = B; g.ldb = K; g.bias = bias; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K; g.alpha = 1.f; g.batch = 1;
    run<0>("bare (per-tile address math)", g);
    run<32>("offsets hoisted", g);
    run<33>("hoisted + clamp", g);
    run<34>("hoisted + struct kernarg", g);
    run<36>("hoisted + static LDS", g);
    run<40>("hoisted + runtime epilogue", g);
    run<48>("hoisted + A2 branch", g);
    run<63>("all (product-like)", g);
    run<0>("bare again", g);
    return 0;
}
