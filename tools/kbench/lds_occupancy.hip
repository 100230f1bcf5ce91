// kbench: how many workgroups of a given LDS footprint does a CU hold?  512 workgroups of 256 threads on 256 CUs, each spinning a fixed
// number of cycles: one round (~T) if two fit per CU, two rounds (~2T) if only one does.  hipcc --offload-arch=gfx950 -O3 lds_occupancy.hip -o lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(256, 2) void spin_kernel(long cycles, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    smem[threadIdx.x] = (unsigned char)threadIdx.x;
    __syncthreads();
    const long t0 = clock64();
    while (clock64() - t0 < cycles) { }
    if (threadIdx.x == 0) out[blockIdx.x] = (float)smem[17];
}
int main() {
    float* out; CK(hipMalloc(&out, 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int kb : {32, 64, 72, 76, 78, 79, 80, 81, 82, 96, 160}) {
        const size_t bytes = (size_t)kb * 1024;
        hipError_t e = hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) { printf("%3d KB: hipFuncSetAttribute failed (%s)\n", kb, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        hipLaunchKernelGGL(spin_kernel, dim3(512), dim3(256), bytes, 0, 1000L, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(spin_kernel, dim3(512), dim3(256), bytes, 0, 2000000L, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%3d KB per workgroup: 512 workgroups x 2e6 spin cycles: %.3f ms\n", kb, ms);
    }
    return 0;
}
