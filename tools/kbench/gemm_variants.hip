// kbench: GEMM NT tiling variants on fp32 MFMA (tuning harness, not product code).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off gemm_variants.hip -o gemm_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// pure MFMA issue-rate ceiling
__global__ __launch_bounds__(256) void mfma_peak(float* out, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}

// WG = WM x WN waves; wave tile = (MB*32) x (NB*32); K tile BK; DBUF: LDS double buffer + reg prefetch
template <int WM, int WN, int MB, int NB, int BK, bool DBUF, int MINW, int PIPE = 0>
__global__ __launch_bounds__(WM * WN * 64, MINW) void gemm_v(const float* __restrict__ A, const float* __restrict__ B,
                                                              const float* __restrict__ bias, float* __restrict__ C,
                                                              int M, int N, int K) {
    constexpr int T = WM * WN * 64, BMt = WM * MB * 32, BNt = WN * NB * 32, LDT = BK + 1;
    constexpr int NBUF = DBUF ? 2 : 1;
    extern __shared__ float lds[];
    float* As = lds;                       // [NBUF][BMt*LDT]
    float* Bs = lds + NBUF * BMt * LDT;    // [NBUF][BNt*LDT]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BMt, n0 = blockIdx.x * BNt;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float bv = bias[n0 + wn * NB * 32 + nb * 32 + i];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = bv;
    }
    constexpr int F4ROW = BK / 4;                      // float4 per tile row
    constexpr int A_IT = BMt * F4ROW / T, B_IT = BNt * F4ROW / T;
    float4 ra[A_IT], rb[B_IT];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) { const int idx = tid + it * T; ra[it] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + idx / F4ROW) * K + k0 + (idx % F4ROW) * 4); }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) { const int idx = tid + it * T; rb[it] = *reinterpret_cast<const float4*>(B + (size_t)(n0 + idx / F4ROW) * K + k0 + (idx % F4ROW) * 4); }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) { const int idx = tid + it * T; float* d = As + buf * BMt * LDT + (idx / F4ROW) * LDT + (idx % F4ROW) * 4; d[0] = ra[it].x; d[1] = ra[it].y; d[2] = ra[it].z; d[3] = ra[it].w; }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) { const int idx = tid + it * T; float* d = Bs + buf * BNt * LDT + (idx / F4ROW) * LDT + (idx % F4ROW) * 4; d[0] = rb[it].x; d[1] = rb[it].y; d[2] = rb[it].z; d[3] = rb[it].w; }
    };
    auto compute = [&](int buf) {
        const float* ap = As + buf * BMt * LDT + (wm * MB * 32 + i) * LDT + h;
        const float* bp = Bs + buf * BNt * LDT + (wn * NB * 32 + i) * LDT + h;
        if (PIPE) {   // explicit operand pipelining: LDS reads of step s+PIPE are issued before the MFMAs of step s
            float a[BK / 2][MB], b[BK / 2][NB];
#pragma unroll
            for (int s = 0; s < PIPE; ++s) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) a[s][mb] = ap[mb * 32 * LDT + 2 * s];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) b[s][nb] = bp[nb * 32 * LDT + 2 * s];
            }
#pragma unroll
            for (int s = 0; s < BK / 2; ++s) {
                if (s + PIPE < BK / 2) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) a[s + PIPE][mb] = ap[mb * 32 * LDT + 2 * (s + PIPE)];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) b[s + PIPE][nb] = bp[nb * 32 * LDT + 2 * (s + PIPE)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][mb], b[s][nb], acc[mb][nb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
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
    };
    if (DBUF) {
        fetch(0); stash(0); __syncthreads();
        int buf = 0;
        for (int k0 = 0; k0 < K; k0 += BK) {
            const bool more = k0 + BK < K;
            if (more) fetch(k0 + BK);
            compute(buf);
            if (more) stash(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    } else {
        for (int k0 = 0; k0 < K; k0 += BK) {
            __syncthreads();
            fetch(k0); stash(0);
            __syncthreads();
            compute(0);
        }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) C[(size_t)m * N + n0 + (wn * NB + nb) * 32 + i] = acc[mb][nb][r];
        }
}

// ---- direct-to-LDS (global_load_lds, 16 B per lane) variant: unpadded 128-B rows, XOR-swizzled 16-B slots
// (slot' = slot ^ ((row >> 1) & 7)) so that ds_read_b128 fragment reads are bank-conflict free; the swizzle is
// applied on the per-lane SOURCE address (the LDS destination of an LDS-DMA is lane-linear).
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
template <int MB, int NB, bool DBUF>
__global__ __launch_bounds__(256, 2) void gemm_glds(const float* __restrict__ A, const float* __restrict__ B,
                                                    const float* __restrict__ bias, float* __restrict__ C, int M, int N, int K) {
    constexpr int BMt = 2 * MB * 32, BNt = 2 * NB * 32, NBUF = DBUF ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;                         // [NBUF][BMt][32]
    float* Bs = lds + NBUF * BMt * 32;       // [NBUF][BNt][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BMt, n0 = blockIdx.x * BNt;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float bv = bias[n0 + wn * NB * 32 + nb * 32 + i];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = bv;
    }
    // DMA: each wave instruction moves 8 rows x 128 B; wave w owns rows [w*R/4, (w+1)*R/4) of each operand tile
    const int drow = lane >> 3, dp = lane & 7;
    auto stage = [&](int k0, int buf) {
#pragma unroll
        for (int it = 0; it < BMt / 32; ++it) {
            const int row0 = wave * (BMt / 4) + it * 8, row = row0 + drow;
            const float* src = A + (size_t)(m0 + row) * K + k0 + ((dp ^ ((row >> 1) & 7)) << 2);
            __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(As + (buf * BMt + row0) * 32), 16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < BNt / 32; ++it) {
            const int row0 = wave * (BNt / 4) + it * 8, row = row0 + drow;
            const float* src = B + (size_t)(n0 + row) * K + k0 + ((dp ^ ((row >> 1) & 7)) << 2);
            __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(Bs + (buf * BNt + row0) * 32), 16, 0, 0);
        }
    };
    auto compute = [&](int buf) {
        const int sw = (i >> 1) & 7;
        const float* ap = As + (buf * BMt + wm * MB * 32 + i) * 32;
        const float* bp = Bs + (buf * BNt + wn * NB * 32 + i) * 32;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float4 a4[MB], b4[NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) a4[mb] = *reinterpret_cast<const float4*>(ap + mb * 32 * 32 + ((c ^ sw) << 2));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b4[nb] = *reinterpret_cast<const float4*>(bp + nb * 32 * 32 + ((c ^ sw) << 2));
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const float av = half ? (h ? a4[mb].w : a4[mb].z) : (h ? a4[mb].y : a4[mb].x);
                        const float bv = half ? (h ? b4[nb].w : b4[nb].z) : (h ? b4[nb].y : b4[nb].x);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mb][nb], 0, 0, 0);
                    }
        }
    };
    if (DBUF) {
        stage(0, 0);
        __syncthreads();
        int buf = 0;
        for (int k0 = 0; k0 < K; k0 += 32) {
            if (k0 + 32 < K) stage(k0 + 32, buf ^ 1);
            compute(buf);
            __syncthreads();
            buf ^= 1;
        }
    } else {
        for (int k0 = 0; k0 < K; k0 += 32) {
            __syncthreads();
            stage(k0, 0);
            __syncthreads();
            compute(0);
        }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) C[(size_t)m * N + n0 + (wn * NB + nb) * 32 + i] = acc[mb][nb][r];
        }
}

template <int MB, int NB, bool DBUF>
static double run_glds(const char* name, const float* A, const float* B, const float* bias, float* C, int M, int N, int K) {
    constexpr int BMt = 2 * MB * 32, BNt = 2 * NB * 32;
    const size_t lds = (size_t)(DBUF ? 2 : 1) * (BMt + BNt) * 32 * 4;
    if (N % BNt || M % BMt) return 0;
    auto kern = gemm_glds<MB, NB, DBUF>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(N / BNt, M / BMt);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, A, B, bias, C, M, N, K);
    hipEventRecord(e0);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, A, B, bias, C, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = 2.0 * M * N * K * reps / (ms * 1e-3) / 1e12;
    printf("  %-34s M=%d N=%d K=%d lds=%zuKB  %.1f us  %.1f TF\n", name, M, N, K, lds / 1024, ms / reps * 1e3, tf);
    return tf;
}

template <int WM, int WN, int MB, int NB, int BK, bool DBUF, int MINW, int PIPE = 0>
static double run(const char* name, const float* A, const float* B, const float* bias, float* C, int M, int N, int K) {
    constexpr int BMt = WM * MB * 32, BNt = WN * NB * 32, LDT = BK + 1;
    const size_t lds = (size_t)(DBUF ? 2 : 1) * (BMt + BNt) * LDT * 4;
    if (N % BNt || M % BMt) return 0;
    auto kern = gemm_v<WM, WN, MB, NB, BK, DBUF, MINW, PIPE>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(N / BNt, M / BMt);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, 0, A, B, bias, C, M, N, K);
    hipEventRecord(e0);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, 0, A, B, bias, C, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = 2.0 * M * N * K * reps / (ms * 1e-3) / 1e12;
    printf("  %-34s M=%d N=%d K=%d lds=%zuKB  %.1f us  %.1f TF\n", name, M, N, K, lds / 1024, ms / reps * 1e3, tf);
    return tf;
}

int main() {
    const int M = 65536;
    float *A, *B, *bias, *C, *o;
    CK(hipMalloc(&A, (size_t)M * 512 * 4)); CK(hipMalloc(&B, (size_t)768 * 512 * 4)); CK(hipMalloc(&bias, 768 * 4));
    CK(hipMalloc(&C, (size_t)M * 768 * 4)); CK(hipMalloc(&o, 1 << 22));
    std::vector<float> h((size_t)M * 512);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), (size_t)768 * 512 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, h.data(), 768 * 4, hipMemcpyHostToDevice));
    {   // clock / power warm-up: the first ~0.2 s of a process run 10-15 % slower (measured with gemm_bisect: the same
        // kernel gives 115 TF as the first variant and 131 TF as the last) -- never compare variants without it
        for (int w = 0; w < 40; ++w) hipLaunchKernelGGL(mfma_peak, dim3(1024), dim3(256), 0, 0, o, 20000);
        hipDeviceSynchronize();
    }
    {   // MFMA ceiling
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 20000, blocks = 256 * 4;
        hipLaunchKernelGGL(mfma_peak, dim3(blocks), dim3(256), 0, 0, o, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_peak, dim3(blocks), dim3(256), 0, 0, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mfma_peak: %.1f TF (%.2f ms)\n", (double)blocks * 4 * iters * 4 * 4096 / (ms * 1e-3) / 1e12, ms);
    }
    const int shapes[4][2] = {{256, 256}, {512, 512}, {256, 512}, {768, 256}};  // N, K
    for (auto& s : shapes) {
        const int N = s[0], K = s[1];
        printf("N=%d K=%d\n", N, K);
        {   // correctness of the glds variant against the reference tiling (bitwise: same k order)
            std::vector<float> c0((size_t)4096 * N), c1((size_t)4096 * N);
            run<2, 2, 2, 2, 32, false, 2>("ref", A, B, bias, C, M, N, K);
            hipMemcpy(c0.data(), C, c0.size() * 4, hipMemcpyDeviceToHost);
            hipMemset(C, 0, (size_t)4096 * N * 4);
            run_glds<2, 4, false>("glds 128x256 single", A, B, bias, C, M, N, K);
            hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
            size_t bad = 0; for (size_t q = 0; q < c0.size(); ++q) bad += c0[q] != c1[q];
            printf("  glds vs ref mismatches: %zu of %zu\n", bad, c0.size());
        }
        run_glds<2, 4, true>("glds 128x256 dbuf", A, B, bias, C, M, N, K);
        run_glds<2, 2, false>("glds 128x128 single", A, B, bias, C, M, N, K);
        run_glds<2, 2, true>("glds 128x128 dbuf", A, B, bias, C, M, N, K);
        run<2, 2, 2, 2, 32, false, 2>("2x2w 64x64 BK32 single", A, B, bias, C, M, N, K);
        run<2, 2, 2, 2, 32, false, 2, 1>("2x2w 64x64 BK32 single pipe1", A, B, bias, C, M, N, K);
        run<2, 2, 2, 2, 32, false, 2, 2>("2x2w 64x64 BK32 single pipe2", A, B, bias, C, M, N, K);
        run<2, 2, 2, 2, 32, false, 2, 4>("2x2w 64x64 BK32 single pipe4", A, B, bias, C, M, N, K);
        run<2, 2, 2, 2, 32, true, 2, 2>("2x2w 64x64 BK32 dbuf pipe2", A, B, bias, C, M, N, K);
        run<2, 2, 2, 4, 32, false, 2>("2x2w 64x128 BK32 single", A, B, bias, C, M, N, K);
        run<1, 4, 2, 4, 32, false, 2>("1x4w 64x128 BK32 single (64x512)", A, B, bias, C, M, N, K);
        run<1, 4, 2, 4, 16, false, 2>("1x4w 64x128 BK16 single (64x512)", A, B, bias, C, M, N, K);
        run<2, 2, 2, 4, 32, false, 2, 1>("2x2w 64x128 BK32 single pipe1", A, B, bias, C, M, N, K);
        run<2, 2, 2, 4, 32, false, 2, 2>("2x2w 64x128 BK32 single pipe2", A, B, bias, C, M, N, K);
        run<2, 2, 2, 4, 32, true, 2, 2>("2x2w 64x128 BK32 dbuf pipe2", A, B, bias, C, M, N, K);
        run<2, 2, 4, 2, 32, false, 2, 2>("2x2w 128x64 BK32 single pipe2", A, B, bias, C, M, N, K);
        run<2, 2, 2, 2, 16, false, 3, 2>("2x2w 64x64 BK16 single pipe2 minw3", A, B, bias, C, M, N, K);
    }
    return 0;
}
