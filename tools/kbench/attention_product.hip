// kbench: the PRODUCT attention (rfe::launch_lg_attention from librover_fe.so) in isolation at the bench shape -- 64 sequences of 1024,
// 4 heads of 64 -- fp32 kernels (self with rotary = lg_attention_kernel, cross = lg_attention_dma_kernel) beside RFE_OPT_LG_FP16X2's
// lg_attention_h2_kernel (tuning harness, not product code).
// hipcc --offload-arch=gfx950 -O3 -I../../rover-slam_amd/csrc attention_product.hip -L../../rover-slam_amd -lrover_fe -Wl,-rpath,'$ORIGIN/../../rover-slam_amd' -o attention_product
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "rfe_internal.h"
using namespace rfe;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int nseq = 64, L = 1024, reps = argc > 1 ? atoi(argv[1]) : 20;
    const int only = argc > 2 ? atoi(argv[2]) : -1;   // 0..3: run one variant only (PMC passes)
    const size_t rows = (size_t)nseq * L;
    float *qkv, *out, *rope; int *lens, *kvmap;
    CK(hipMalloc(&qkv, rows * 768 * 4)); CK(hipMalloc(&out, rows * 256 * 4)); CK(hipMalloc(&rope, rows * 64 * 4));
    CK(hipMalloc(&lens, nseq * 4)); CK(hipMalloc(&kvmap, nseq * 4));
    std::vector<float> h(rows * 768);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)(st >> 8) / 16777216.f - 0.5f; };
    for (auto& x : h) x = 3.0f * rnd();      // roughly unit variance
    CK(hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < rows * 32; ++i) { const float t = 6.f * rnd(); h[2 * i] = cosf(t); h[2 * i + 1] = sinf(t); }
    CK(hipMemcpy(rope, h.data(), rows * 64 * 4, hipMemcpyHostToDevice));
    std::vector<int> hl(nseq, L), hm(nseq);
    for (int i = 0; i < nseq; ++i) hm[i] = i ^ 1;
    CK(hipMemcpy(lens, hl.data(), nseq * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(kvmap, hm.data(), nseq * 4, hipMemcpyHostToDevice));
    const double flops = 4.0 * 4 * 64 * (double)nseq * L * L;
    auto run = [&](int variant) {
        const bool cross = variant & 1, h2 = variant & 2;
        if (cross) launch_lg_attention(0, qkv, qkv, qkv + 256, 512, out, nseq, L, L, lens, lens, kvmap, nullptr, nullptr, h2);
        else launch_lg_attention(0, qkv, qkv + 256, qkv + 512, 768, out, nseq, L, L, lens, lens, nullptr, nullptr, rope, h2);
    };
    const char* names[4] = {"fp32 self (rotary)", "fp32 cross (LDS-DMA)", "fp16x2 self (rotary)", "fp16x2 cross"};
    for (int i = 0; i < 200; ++i) run(0);   // clocks / power warm-up
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int v = 0; v < 4; ++v) {
        if (only >= 0 && v != only) continue;
        for (int i = 0; i < 3; ++i) run(v);
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) run(v);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %-24s %.1f us  %.1f TF (fp32-equivalent)\n", names[v], ms / reps * 1e3, flops * reps / (ms * 1e-3) / 1e12);
    }
    CK(hipGetLastError());
    return 0;
}
