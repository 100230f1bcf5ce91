// kbench: a WIDE fp32-MFMA GEMM tile for LightGlue's Linears (tuning harness, not product code).
//   256 x 256 x 32 tile, 512 threads = 4x2 waves of 64 x 128 (eight 32x32 accumulators each), ONE workgroup per CU (2 waves / SIMD).
//   B (weights, static) pre-packed at load time into the LDS image of every (column tile, K tile) -- [N/256][K/32] images of 32 KB with
//   the 16-byte slots already XOR-swizzled -- and copied by global_load_lds_dwordx4 into a DOUBLE-buffered LDS tile (no VGPRs, no
//   ds_write, no per-lane addressing); A register-prefetched one K tile ahead, single LDS buffer.  Against the product's 128 x 256 tile:
//   2/3 of the operand traffic per FLOP, a third of the LDS stores, loads a whole K tile (16 k cycles) ahead.
//   PERSIST: one workgroup per 256-row panel walks the panel's N/256 column tiles and issues the first K tile of the next column
//   tile before the epilogue stores of the current one.
// k order inside a K tile: (s, 16 + s) like the product's k-permuted path (LightGlue is tolerance-checked, not bit-exact).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off gemm_wide_f32.hip -o gemm_wide_f32
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// fp32 [N][K] -> [N/256][K/32] LDS images [256 rows][8 slots of 16 B], slot s of row r stored at s ^ ((r >> 1) & 7)
// (conflict-free for ds_read_b128's 16-lane groups {0-3,12-15,20-27} ... : the eight even / odd rows of a group get eight different slots)
__global__ void pack_b_kernel(const float* __restrict__ B, float* __restrict__ out, int N, int K) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one float4
    if (idx >= (size_t)N * K / 4) return;
    const int n = (int)(idx / (K / 4)), k = (int)(idx % (K / 4)) * 4;
    const int ct = n >> 8, r = n & 255, kt = k >> 5, slot = (k & 31) >> 2;
    const size_t base = ((size_t)ct * (K / 32) + kt) * (256 * 32);
    *reinterpret_cast<float4*>(out + base + r * 32 + ((slot ^ ((r >> 1) & 7)) << 2)) = *reinterpret_cast<const float4*>(B + (size_t)n * K + k);
}

template <bool PERSIST>
__global__ __launch_bounds__(512, 2) void gemm_f32w_kernel(const float* __restrict__ A, const float* __restrict__ Bpk, const float* __restrict__ bias,
                                                           float* __restrict__ C, int M, int N, int K) {
    constexpr int BM = 256, BN = 256, BK = 32, MB = 2, NB = 4, BIMG = BN * BK;   // floats
    __shared__ __attribute__((aligned(16))) float lds[BM * BK + 2 * BIMG];      // 32 KB + 64 KB
    float* const As = lds;
    float* const Bs = lds + BM * BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    const int nct = N / BN, nkt = K / BK;
    const int by = PERSIST ? blockIdx.x : blockIdx.y;
    const int m0 = by * BM;
    const int lrow = tid >> 3, lkq = tid & 7;                                   // staging: rows lrow + 64 it
    float* const da = As + lrow * BK + ((lkq ^ ((lrow >> 1) & 7)) << 2);
    const float* const Arow = A + (size_t)(m0 + lrow) * K + lkq * 4;
    const int iswz = (i >> 1) & 7;
    const float* const ap = As + ((wm * MB) * 32 + i) * BK;
    float4 ra[4];
    auto load_a = [&](int k0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) ra[it] = *reinterpret_cast<const float4*>(Arow + (size_t)64 * it * K + k0);
    };
    auto issue_b = [&](int ct, int kt, int buf) {   // 32 KB = 32 wave-level copies of 1 KB, four per wave
        const float* src = Bpk + ((size_t)ct * nkt + kt) * BIMG + (wave * 4) * 256 + lane * 4;
        float* dst = Bs + buf * BIMG + (wave * 4) * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(src + j * 256), (lds_ptr_t)(dst + j * 256), 16, 0, 0);
    };
    const int ct_first = PERSIST ? 0 : blockIdx.x, ct_end = PERSIST ? nct : blockIdx.x + 1;
    int buf = 0;
    issue_b(ct_first, 0, 0);
    load_a(0);
    for (int ct = ct_first; ct < ct_end; ++ct) {
        const int n0 = ct * BN;
        f32x16 acc[MB][NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float bv = bias[n0 + (wn * NB + nb) * 32 + i];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][nb][r] = bv;
        }
        for (int t = 0; t < nkt; ++t) {
#pragma unroll
            for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(da + it * 64 * BK) = f32x4{ra[it].x, ra[it].y, ra[it].z, ra[it].w};
            __syncthreads();   // A tile visible; every wave's part of this B tile has landed (vmcnt(0) in front of the barrier)
            if (t + 1 < nkt) { issue_b(ct, t + 1, buf ^ 1); load_a((t + 1) * BK); }
            else if (PERSIST && ct + 1 < ct_end) { issue_b(ct + 1, 0, buf ^ 1); load_a(0); }   // next column tile's first K tile flies under the epilogue
            const float* const bp = Bs + buf * BIMG + ((wn * NB) * 32 + i) * BK;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int slot = (((h << 2) + g) ^ iswz) << 2;
                f32x4 a4[MB], b4[NB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) a4[mb] = *reinterpret_cast<const f32x4*>(ap + mb * 32 * BK + slot);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) b4[nb] = *reinterpret_cast<const f32x4*>(bp + nb * 32 * BK + slot);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[mb][c], b4[nb][c], acc[mb][nb], 0, 0, 0);
            }
            __syncthreads();   // A tile consumed
            buf ^= 1;
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
}

template <typename F>
static float timeit(F&& launch, int reps = 20) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps * 1e3f;
}

static void run_shape(const char* name, int M, int N, int K) {
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hb(N);
    uint64_t s = 0x9E3779B97F4A7C15ull ^ (uint64_t)(N * 131 + K);
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0 - 0.5; };
    for (auto& v : hA) v = (float)(rnd() * 2);
    for (auto& v : hB) v = (float)(rnd() * 2 / std::sqrt((double)K));
    for (auto& v : hb) v = (float)(rnd() * 0.2);
    float *dA, *dB, *db, *dC, *pkB;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&pkB, hB.size() * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    pack_b_kernel<<<(unsigned)((hB.size() / 4 + 255) / 256), 256>>>(dB, pkB, N, K);
    CK(hipDeviceSynchronize());
    const double flop = 2.0 * M * N * K;
    std::vector<float> hC((size_t)M * N);
    auto report = [&](const char* what, float us) {
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int q = 0; q < 24; ++q) {
            const int m = (int)(((uint64_t)q * 2654435761u + 977) % (uint64_t)M);
            for (int n = 0; n < N; ++n) {
                double ref = hb[n];
                for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * (double)hB[(size_t)n * K + k];
                worst = std::max(worst, std::fabs((double)hC[(size_t)m * N + n] - ref));
            }
        }
        printf("  %-34s %-6s M=%d N=%d K=%d  %7.1f us %6.1f TF   max |err| vs f64 %.2e\n", what, name, M, N, K, us, flop / (us * 1e-6) / 1e12, worst);
    };
    CK(hipMemset(dC, 0, (size_t)M * N * 4));
    report("wide 256x256, LDS-DMA weights", timeit([&] { gemm_f32w_kernel<false><<<dim3(N / 256, M / 256), 512>>>(dA, pkB, db, dC, M, N, K); }));
    CK(hipMemset(dC, 0, (size_t)M * N * 4));
    report("... persistent over the row panel", timeit([&] { gemm_f32w_kernel<true><<<dim3(M / 256), 512>>>(dA, pkB, db, dC, M, N, K); }));
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(db)); CK(hipFree(dC)); CK(hipFree(pkB));
}

int main() {
    const int M = 65536;
    {   // clock / power warm-up
        float *x, *w, *c;
        CK(hipMalloc(&x, (size_t)M * 256 * 4)); CK(hipMemset(x, 0, (size_t)M * 256 * 4));
        CK(hipMalloc(&w, 256 * 256 * 4)); CK(hipMemset(w, 0, 256 * 256 * 4)); CK(hipMalloc(&c, (size_t)M * 256 * 4));
        for (int i = 0; i < 1500; ++i) gemm_f32w_kernel<false><<<dim3(1, M / 256), 512>>>(x, w, w, c, M, 256, 256);
        CK(hipDeviceSynchronize()); CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(c));
    }
    run_shape("ffn.0", M, 512, 512);
    run_shape("qkv", M, 768, 256);
    run_shape("cqkv", M, 512, 256);
    run_shape("ffn.3", M, 256, 512);
    return 0;
}
