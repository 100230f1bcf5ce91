// kbench: calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 against kernels whose HBM byte count is known
// (tuning harness, not product code).  Every kernel touches each byte of a buffer larger than L2 + Infinity
// Cache exactly once:
//   stream_read16   : contiguous float4 loads (64 lanes x 16 B = 1 KB per wave instruction)        -- the guide's "wide coalesced" case
//   stream_read4    : contiguous 4-byte loads (256 B per wave instruction)
//   seg128_read4    : lg_col_kernel's pattern -- a workgroup owns a stripe of 32 columns of a [P, 1024, 1024] fp32 buffer and reads
//                     128-byte row segments 4 KB apart with 4-byte loads (two segments per wave instruction)
//   seg128_read16   : the same stripe with float4 loads (eight segments per wave instruction) = what global_load_lds_dwordx4 issues
//   stream_write16 / stream_write4 : contiguous stores
// usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- fetch_calib ; rocprofv3 --pmc WRITE_SIZE ... ; bytes per launch are printed
// hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void stream_read16(const float4* __restrict__ x, float* __restrict__ out, size_t n4) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { const float4 v = x[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void stream_read4(const float* __restrict__ x, float* __restrict__ out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += x[i];
    if (acc == 123.456f) out[0] = acc;
}
__global__ void seg128_read4(const float* __restrict__ x, float* __restrict__ out, int L) {   // grid (L / 32, P), 256 threads
    const int c = threadIdx.x % 32, rg = threadIdx.x / 32;
    const float* base = x + (size_t)blockIdx.y * L * L + blockIdx.x * 32 + c;
    float acc = 0.f;
    for (int i = rg; i < L; i += 8) acc += base[(size_t)i * L];
    if (acc == 123.456f) out[0] = acc;
}
__global__ void seg128_read16(const float* __restrict__ x, float* __restrict__ out, int L) {  // grid (L / 32, P), 256 threads
    const int cq = threadIdx.x % 8, rg = threadIdx.x / 8;
    const float* base = x + (size_t)blockIdx.y * L * L + blockIdx.x * 32 + cq * 4;
    float acc = 0.f;
    for (int i = rg; i < L; i += 32) { const float4 v = *reinterpret_cast<const float4*>(base + (size_t)i * L); acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void stream_write16(float4* __restrict__ x, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) x[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void stream_write4(float* __restrict__ x, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = (float)i;
}

int main() {
    const int P = 256, L = 1024;                       // 256 x 4 MB = 1 GiB
    const size_t n = (size_t)P * L * L;
    float *x, *out;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&out, 4));
    CK(hipMemset(x, 0, n * 4));
    for (int rep = 0; rep < 3; ++rep) {
        stream_read16<<<4096, 256>>>(reinterpret_cast<const float4*>(x), out, n / 4);
        stream_read4<<<4096, 256>>>(x, out, n);
        seg128_read4<<<dim3(L / 32, P), 256>>>(x, out, L);
        seg128_read16<<<dim3(L / 32, P), 256>>>(x, out, L);
        stream_write16<<<4096, 256>>>(reinterpret_cast<float4*>(x), n / 4);
        stream_write4<<<4096, 256>>>(x, n);
    }
    CK(hipDeviceSynchronize());
    printf("every kernel moves %zu bytes (%.1f KiB) per launch\n", n * 4, n * 4 / 1024.0);
    return 0;
}
