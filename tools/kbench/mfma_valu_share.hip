// kbench (round 6): do fp32 VALU instructions of ANOTHER wave run in the shadow of a wave's fp32 matrix instructions, or do the two add up?
// Workgroup = 8 waves, two per SIMD: waves 0-3 issue v_mfma_f32_32x32x2_f32 on 4 independent accumulators (NM instructions), waves 4-7 issue a chain-free
// stream of v_fma_f32 (or v_pk_fma_f32, or v_exp_f32) -- NV instructions.  Times: matrix waves alone (NV = 0), VALU waves alone (NM = 0), both together.
// If the pipes overlap, together ~ max(alone, alone); if fp32 VALU and the fp32 matrix path share their multipliers, together ~ sum.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_share.hip -o mfma_valu_share        (tuning harness, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int KIND>   // 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_exp_f32, 3: integer (v_xor / v_lshl_add: the address arithmetic of an LDS fragment read), 4: LDS reads
__global__ __launch_bounds__(512, 1) void share_kernel(float* out, int nm, int nv) {
    const int wave = threadIdx.x >> 6;
    float res = 0.f;
    if (wave < 4) {
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
        for (int it = 0; it < nm / 4; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        res = acc[0][0] + acc[1][3] + acc[2][5] + acc[3][7];
    } else {
        // 16 independent chains so that the VALU is never waiting for a result
        float v[16]; f32x2 p[8];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 1e-4f + j;
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = f32x2{v[2 * j], v[2 * j + 1]};
        const float c = 1.0000001f, d = 1e-7f;
        for (int it = 0; it < nv / 16; ++it) {
            if (KIND == 0) {
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], c, d);
            } else if (KIND == 1) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) p[j] = __builtin_elementwise_fma(p[j], f32x2{c, c}, f32x2{d, d});
            } else if (KIND == 2) {
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_exp2f(v[j] * 1e-3f);
            } else if (KIND == 3) {
#pragma unroll
                for (int j = 0; j < 16; ++j) { unsigned u = __float_as_uint(v[j]); u = ((u ^ (unsigned)it) << 2) + (unsigned)j; v[j] = __uint_as_float(u); }
            } else if (KIND == 4) {
                extern __shared__ float sm[];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] += sm[(threadIdx.x * 16 + j * 64 + it) & 8191];
            } else {
                // KIND 5: 16 conflict-free ds_read_b32 (lane-linear), KIND 6: 16 ds_read_b128 -- no VALU at all between them, one wait per 16
                extern __shared__ float sm[];
                const unsigned addr = (unsigned)(size_t)(sm) + (threadIdx.x & 63) * (KIND == 5 ? 4 : 16);
                float4 d4;
                if (KIND == 5) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"(j * 256));
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d4) : "v"(addr), "n"(j * 1024));
                    v[0] += d4.x;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) res += v[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) res += p[j][0] + p[j][1];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

template <int KIND>
static double run(float* out, int nm, int nv) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) share_kernel<KIND><<<256, 512, 32768>>>(out, nm, nv);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) share_kernel<KIND><<<256, 512, 32768>>>(out, nm, nv);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5 * 1e3;
}
template <int KIND>
static void series(float* o, const char* name, int NV) {
    const int NM = 8192;
    const double tm = run<KIND>(o, NM, 0), tv = run<KIND>(o, 0, NV), tb = run<KIND>(o, NM, NV);
    printf("| %s | %d matrix instr alone %.1f us (%.1f cycles each at 2.4 GHz) | %d VALU instr alone %.1f us (%.2f cycles each) | together %.1f us | max %.1f, sum %.1f |\n",
           name, NM, tm, tm * 2400.0 / NM, NV, tv, tv * 2400.0 / NV, tb, tm > tv ? tm : tv, tm + tv);
}
int main() {
    float* o; CK(hipMalloc(&o, 256 * 512 * 4));
    printf("| VALU stream of waves 4-7 | waves 0-3 | waves 4-7 | both | overlap (max) or serial (sum)? |\n|---|---|---|---|---|\n");
    series<0>(o, "v_fma_f32", 32768); series<0>(o, "v_fma_f32", 65536);
    series<1>(o, "v_pk_fma_f32", 32768); series<1>(o, "v_pk_fma_f32", 65536);
    series<2>(o, "v_exp_f32 (+ v_mul)", 16384);
    series<3>(o, "integer: v_xor + v_lshl_add (2 per count)", 32768);
    series<4>(o, "ds_read_b32 + v_add_f32 (+ address VALU)", 16384);
    series<5>(o, "ds_read_b32, conflict-free, no VALU", 65536);
    series<6>(o, "ds_read_b128, conflict-free, no VALU", 32768);
    return 0;
}
