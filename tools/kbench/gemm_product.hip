// kbench: the PRODUCT GEMM (rfe::launch_gemm_nt from librover_fe.so) in isolation, at the LightGlue shapes and with
// the real epilogue options, to separate kernel efficiency from pipeline effects (tuning harness, not product code).
// hipcc --offload-arch=gfx950 -O3 -I../../rover-slam_amd/csrc gemm_product.hip -L../../rover-slam_amd -lrover_fe -o gemm_product
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "rfe_internal.h"
using namespace rfe;

static GemmArgs plain(const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc, int M, int N, int K) {
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.f; g.batch = 1;
    return g;
}
static void timeit(const char* name, const GemmArgs& g) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_gemm_nt(0, g);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch_gemm_nt(0, g);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  %-28s M=%d N=%d K=%d  %.1f us  %.1f TF\n", name, g.M, g.N, g.K, ms / reps * 1e3, 2.0 * g.M * g.N * g.K * reps / (ms * 1e-3) / 1e12);
}
int main() {
    const int M = 65536;
    float *x, *ctx, *h, *qkv, *w, *bias, *cs, *sn;
    hipMalloc(&x, (size_t)M * 256 * 4); hipMalloc(&ctx, (size_t)M * 256 * 4); hipMalloc(&h, (size_t)M * 512 * 4);
    hipMalloc(&qkv, (size_t)M * 768 * 4); hipMalloc(&w, (size_t)768 * 512 * 4); hipMalloc(&bias, 768 * 4);
    hipMalloc(&cs, (size_t)M * 32 * 4); hipMalloc(&sn, (size_t)M * 32 * 4);
    std::vector<float> hv((size_t)M * 512);
    for (size_t i = 0; i < hv.size(); ++i) hv[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(h, hv.data(), hv.size() * 4, hipMemcpyHostToDevice); hipMemcpy(x, hv.data(), (size_t)M * 256 * 4, hipMemcpyHostToDevice);
    hipMemcpy(ctx, hv.data(), (size_t)M * 256 * 4, hipMemcpyHostToDevice); hipMemcpy(w, hv.data(), (size_t)768 * 512 * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, hv.data(), 768 * 4, hipMemcpyHostToDevice); hipMemcpy(cs, hv.data(), (size_t)M * 32 * 4, hipMemcpyHostToDevice);
    hipMemcpy(sn, hv.data(), (size_t)M * 32 * 4, hipMemcpyHostToDevice);
    { GemmArgs g = plain(x, 256, w, 256, bias, ctx, 256, M, 256, 256); for (int i = 0; i < 3000; ++i) launch_gemm_nt(0, g); hipDeviceSynchronize(); }   // clock / power warm-up (~0.25 s)
    { GemmArgs g = plain(x, 256, w, 512, bias, h, 512, M, 512, 512); g.A2 = ctx; g.lda2 = 256; g.K1 = 256; timeit("ffn1 (split A)", g); g.kperm = 1; timeit("ffn1 (split A), k-permuted", g); }
    { GemmArgs g = plain(h, 512, w, 512, bias, x, 256, M, 256, 512); timeit("ffn2 no residual", g); g.R = x; g.ldr = 256; timeit("ffn2 + residual (in place)", g); }
    { GemmArgs g = plain(x, 256, w, 256, bias, qkv, 768, M, 768, 256); timeit("qkv", g); g.kperm = 1; timeit("qkv, k-permuted LDS path", g); }
    { GemmArgs g = plain(x, 256, w, 256, bias, qkv, 512, M, 512, 256); timeit("cross qkv", g); }
    { GemmArgs g = plain(x, 256, w, 256, bias, ctx, 256, M, 256, 256); timeit("proj 256x256", g); }
    // RFE_OPT_LG_FP16X2 (gemm_h2.hip): the same shapes with the weights' fp16 (hi, lo) planes attached, beside the fp32 k-permuted kernel
    uint16_t* wh; hipMalloc(&wh, (size_t)768 * 512 * 2 * 2);
    uint16_t* wl = wh + (size_t)768 * 512;
    launch_split_f16(0, w, wh, wl, (size_t)768 * 512);
    float* stat; hipMalloc(&stat, (size_t)M * 8 * 2 * 4);
    printf("fp32 k-permuted kernel / fp16x2 split kernel (gemm_h2.hip):\n");
    for (int h2 = 0; h2 < 2; ++h2) {
        auto with = [&](GemmArgs g) { g.kperm = 1; if (h2) { g.Bh = wh; g.Bl = wl; } return g; };
        const char* tag = h2 ? "h2 " : "f32";
        char nm[64];
        { GemmArgs g = with(plain(x, 256, w, 512, bias, h, 512, M, 512, 512)); g.A2 = ctx; g.lda2 = 256; g.K1 = 256;
          snprintf(nm, 64, "%s ffn1", tag); timeit(nm, g);
          g.stats_out = stat; snprintf(nm, 64, "%s ffn1 + LN partials", tag); timeit(nm, g); }
        { GemmArgs g = with(plain(h, 512, w, 512, bias, x, 256, M, 256, 512)); g.R = x; g.ldr = 256;
          snprintf(nm, 64, "%s ffn2 + residual", tag); timeit(nm, g);
          g.stats_in = stat; g.stats_p = 4; g.ln_g = bias; g.ln_b = bias;
          snprintf(nm, 64, "%s ffn2 + res + LN/GELU on A", tag); timeit(nm, g); }
        { GemmArgs g = with(plain(x, 256, w, 256, bias, qkv, 768, M, 768, 256)); snprintf(nm, 64, "%s qkv", tag); timeit(nm, g); }
        { GemmArgs g = with(plain(x, 256, w, 256, bias, qkv, 512, M, 512, 256)); snprintf(nm, 64, "%s cross qkv", tag); timeit(nm, g); }
    }
    return 0;
}
