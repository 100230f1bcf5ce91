// What does a phase boundary cost on MI355X: a grid-wide barrier inside ONE persistent kernel (256 workgroups, one per CU, as the latency
// kernels of the one-pair LightGlue are launched) against a kernel boundary on a stream?  Decides whether a persistent "one kernel per
// transformer layer" form can beat the ~10 dependent launches per layer (round-3 review, item 2b).
// Every phase: each workgroup writes BYTES of its own region (its "activation tile"), [boundary], reads the region another workgroup -- on a
// different XCD -- wrote in this phase and checks it.  Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NWG = 256, NT = 256;

// flat barrier: one counter, monotonically increasing target
__device__ __forceinline__ void barrier_flat(unsigned* ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// two-level barrier: workgroups of one XCD (blockIdx % 8) meet on their own counter, the last arriver of each XCD goes to the global one,
// everybody polls one global flag word
__device__ __forceinline__ void barrier_xcd(unsigned* xcd_ctr /*[8*32]*/, unsigned* glob, unsigned round) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const int x = blockIdx.x & 7;
        const unsigned a = __hip_atomic_fetch_add(&xcd_ctr[x * 32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (a == round * (NWG / 8) + (NWG / 8 - 1)) __hip_atomic_fetch_add(glob, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(glob, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (round + 1) * 8) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

template <int MODE>   // 0 flat, 1 two-level, 2 no barrier at all (wrong, lower bound of the phase body)
__global__ __launch_bounds__(NT) void persistent(float4* buf, int vec_per_wg, int rounds, unsigned* ctr, int* bad) {
    const int b = blockIdx.x, t = threadIdx.x;
    const int peer = (b + 37) % NWG;
    int wrong = 0;
    for (int r = 0; r < rounds; ++r) {
        const float tag = (float)(r + 1);
        for (int i = t; i < vec_per_wg; i += NT) buf[(size_t)b * vec_per_wg + i] = make_float4(tag, (float)b, (float)i, tag);
        if (MODE == 0) barrier_flat(ctr, (unsigned)(r + 1) * NWG);
        else if (MODE == 1) barrier_xcd(ctr + 64, ctr, (unsigned)r);
        for (int i = t; i < vec_per_wg; i += NT) {
            const float4 v = buf[(size_t)peer * vec_per_wg + i];
            wrong += (v.x != tag) | (v.y != (float)peer);
        }
        // a second boundary so that the next round's writes cannot overtake this round's reads (as a real phase chain would have)
        if (MODE == 0) barrier_flat(ctr + 32, (unsigned)(r + 1) * NWG);
        else if (MODE == 1) barrier_xcd(ctr + 64 + 512, ctr + 32, (unsigned)r);
    }
    if (wrong && MODE != 2) atomicAdd(bad, wrong);
}

__global__ __launch_bounds__(NT) void phase_write(float4* buf, int vec_per_wg, float tag) {
    const int b = blockIdx.x, t = threadIdx.x;
    for (int i = t; i < vec_per_wg; i += NT) buf[(size_t)b * vec_per_wg + i] = make_float4(tag, (float)b, (float)i, tag);
}
__global__ __launch_bounds__(NT) void phase_read(const float4* buf, int vec_per_wg, float tag, int* bad) {
    const int b = blockIdx.x, t = threadIdx.x;
    const int peer = (b + 37) % NWG;
    int wrong = 0;
    for (int i = t; i < vec_per_wg; i += NT) {
        const float4 v = buf[(size_t)peer * vec_per_wg + i];
        wrong += (v.x != tag) | (v.y != (float)peer);
    }
    if (wrong) atomicAdd(bad, wrong);
}

int main() {
    const int rounds = 200;
    unsigned* ctr; int* bad; float4* buf;
    CK(hipMalloc(&ctr, 8192)); CK(hipMalloc(&bad, 4));
    CK(hipMalloc(&buf, (size_t)NWG * 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t s; CK(hipStreamCreate(&s));
    printf("| bytes per workgroup and phase | kernel boundaries: us per boundary | flat barrier | two-level (per-XCD) barrier | no barrier (body only) | wrong values (flat / two-level / launches) |\n|---:|---:|---:|---:|---:|---|\n");
    for (int bytes : {0, 1024, 8192, 32768}) {
        const int vec = bytes / 16;
        float ms[4]; int nbad[4] = {0, 0, 0, 0};
        for (int mode = 0; mode < 4; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {   // first repetition warms up
                CK(hipMemsetAsync(ctr, 0, 8192, s)); CK(hipMemsetAsync(bad, 0, 4, s));
                CK(hipEventRecord(e0, s));
                if (mode == 0) hipLaunchKernelGGL(persistent<0>, dim3(NWG), dim3(NT), 0, s, buf, vec, rounds, ctr, bad);
                else if (mode == 1) hipLaunchKernelGGL(persistent<1>, dim3(NWG), dim3(NT), 0, s, buf, vec, rounds, ctr, bad);
                else if (mode == 2) hipLaunchKernelGGL(persistent<2>, dim3(NWG), dim3(NT), 0, s, buf, vec, rounds, ctr, bad);
                else for (int r = 0; r < rounds; ++r) {
                    hipLaunchKernelGGL(phase_write, dim3(NWG), dim3(NT), 0, s, buf, vec, (float)(r + 1));
                    hipLaunchKernelGGL(phase_read, dim3(NWG), dim3(NT), 0, s, buf, vec, (float)(r + 1), bad);
                }
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms[mode], e0, e1));
                CK(hipMemcpy(&nbad[mode], bad, 4, hipMemcpyDeviceToHost));
            }
        }
        // two boundaries per round in every variant
        printf("| %d | %.2f | %.2f | %.2f | %.2f | %d / %d / %d |\n", bytes, ms[3] * 1e3 / (2 * rounds), ms[0] * 1e3 / (2 * rounds), ms[1] * 1e3 / (2 * rounds),
               ms[2] * 1e3 / (2 * rounds), nbad[0], nbad[1], nbad[3]);
    }
    return 0;
}
