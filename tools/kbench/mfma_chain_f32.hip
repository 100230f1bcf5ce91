// kbench: issue rate of v_mfma_f32_32x32x2_f32 as a function of the number of independent accumulators a wave cycles through
// (NACC = 1: a dependent chain, every MFMA waits for the previous result -- the S = K.Q^T phase of the attention kernel) and of the
// waves per SIMD (tuning harness, not product code).   hipcc --offload-arch=gfx950 -O3 mfma_chain_f32.hip -o mfma_chain_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NACC>
__global__ void chain_kernel(float* out, int iters) {
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
static void run(float* out, int threads) {
    const int iters = 16384 / NACC;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) chain_kernel<NACC><<<256, threads>>>(out, iters);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) chain_kernel<NACC><<<256, threads>>>(out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / 5 * 1e3, n = (double)iters * NACC, wps = threads / 256.0;
    printf("  f32 32x32x2, %d accumulator(s), %g wave(s)/SIMD: %.1f cycles per MFMA per SIMD at 2.4 GHz (%.1f TF)\n", NACC, wps,
           us * 1e-6 * 2.4e9 / (n * wps), n * (threads / 64) * 256 * 4096.0 / (us * 1e-6) / 1e12);
}
int main() {
    float* o; CK(hipMalloc(&o, 256 * 1024 * 4));
    run<1>(o, 256); run<2>(o, 256); run<4>(o, 256);
    run<1>(o, 512); run<2>(o, 512);
    run<1>(o, 1024); run<2>(o, 1024);
    return 0;
}
