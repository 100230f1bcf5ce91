// search.hip -- the descriptor arithmetic of the callers' classic searches:
//   * search_candidates_kernel: best / second-best scan of SPmatcher::SearchByProjection1
//     (src/Matchers/SPmatcher.cc:1218-1248; same loop in SearchByProjection :755-800 and Fuse :150-200)
//     for many map points at once.  Candidate lists stay with the CPU grid (Frame::GetFeaturesInArea) and
//     arrive as CSR; one wave per map point walks its list in order (strict '<' keeps the first minimum).
//   * distinctive_rows_kernel / distinctive_pick_kernel: MapPoint::ComputeDistinctiveDescriptors
//     (src/MapPoint.cc:438-530) for many map points at once: one wave per observed descriptor computes its
//     row of distances (symmetric: (a-b)^2 == (b-a)^2 exactly, so no n x n matrix is stored), bitonic-sorts
//     it in LDS and takes sorted[(int)(0.5*(n-1))]; a second kernel picks the first strictly smallest median.
// Distance = DescriptorDistance_sp (SPmatcher.cc:2184-2189) in the canonical order shared with stereo.hip and
// the oracle (float differences, double accumulation, lane-of-4 then xor butterfly) -> bit-exact results.
// HBM / latency bound: (1 + candidates) KB per map point.
#include "rfe_internal.h"

namespace rfe {

__device__ __forceinline__ float desc_dist_wave(const float4 a, const float4 b) {
    const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
    double p = 0.0;
    p += (double)d0 * (double)d0; p += (double)d1 * (double)d1; p += (double)d2 * (double)d2; p += (double)d3 * (double)d3;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off);
    return (float)sqrt(p);
}

__global__ __launch_bounds__(256) void search_candidates_kernel(const float* __restrict__ q, int Nq, const float* __restrict__ f, int Nf,
                                                                const int32_t* __restrict__ offsets, const int32_t* __restrict__ cand,
                                                                const uint8_t* __restrict__ skip, int32_t* __restrict__ best_idx,
                                                                float* __restrict__ best_dist, float* __restrict__ second_dist) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= Nq) return;
    const float4 a = reinterpret_cast<const float4*>(q + (size_t)i * 256)[lane];
    float bestDist = 256.f, bestDist2 = 256.f; int bestIdx = -1;
    const int e = offsets[i + 1];
    for (int c = offsets[i]; c < e; ++c) {
        const int idx = cand[c];
        if ((unsigned)idx >= (unsigned)Nf) continue;   // device-resident lists cannot be validated by the host: stay inside f
        if (skip && skip[idx]) continue;
        const float dist = desc_dist_wave(a, reinterpret_cast<const float4*>(f + (size_t)idx * 256)[lane]);
        if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = idx; }
        else if (dist < bestDist2) bestDist2 = dist;
    }
    if (lane == 0) { best_idx[i] = bestIdx; best_dist[i] = bestDist; second_dist[i] = bestDist2; }
}

void launch_search_candidates(hipStream_t s, const float* q, int Nq, const float* f, int Nf, const int32_t* offsets, const int32_t* cand,
                              const uint8_t* skip, int32_t* best_idx, float* best_dist, float* second_dist) {
    if (Nq <= 0) return;
    hipLaunchKernelGGL(search_candidates_kernel, dim3((Nq + 3) / 4), dim3(256), 0, s, q, Nq, f, Nf, offsets, cand, skip, best_idx,
                       best_dist, second_dist);
}

// one wave (= one workgroup) per observed descriptor g.  Its map point p (offsets[p] <= g < offsets[p+1]) is found by a binary
// search over the device-resident offsets (empty points share an offset: the LAST p with offsets[p] <= g is the owner), so no
// host-built descriptor -> point table is needed.  Points with more than P2 observations (the LDS row) are left to the pick kernel.
__global__ __launch_bounds__(64) void distinctive_rows_kernel(const float* __restrict__ desc, const int32_t* __restrict__ offsets,
                                                              int Np, int P2, float* __restrict__ med) {
    extern __shared__ float row[];
    const int lane = threadIdx.x;
    const int g = blockIdx.x;
    if (g >= offsets[Np]) return;          // the grid is sized for the caller's `total`, an upper bound (a larger offsets[Np]: rows >= total are
                                           // never launched, and distinctive_pick_kernel reports their points as -2)
    int lo = 0, hi = Np;                   // invariant: offsets[lo] <= g < offsets[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= g) lo = mid; else hi = mid;
    }
    const int p = lo;
    const int o = offsets[p], n = offsets[p + 1] - o;
    if (n > P2) { if (lane == 0) med[g] = __builtin_inff(); return; }
    const float4 a = reinterpret_cast<const float4*>(desc + (size_t)g * 256)[lane];
    int p2 = 1;
    while (p2 < n) p2 <<= 1;
    for (int j = 0; j < n; ++j) {
        const float d = (o + j == g) ? 0.f : desc_dist_wave(a, reinterpret_cast<const float4*>(desc + (size_t)(o + j) * 256)[lane]);
        if (lane == 0) row[j] = d;
    }
    for (int j = n + lane; j < p2; j += 64) row[j] = __builtin_inff();
    __syncthreads();
    for (int k = 2; k <= p2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < p2; t += 64) {
                const int u = t ^ j;
                if (u > t) {
                    const float x = row[t], y = row[u];
                    const bool up = (t & k) == 0;
                    if ((x > y) == up) { row[t] = y; row[u] = x; }
                }
            }
            __syncthreads();
        }
    if (lane == 0) med[g] = row[(int)(0.5 * (n - 1))];
}

__global__ void distinctive_pick_kernel(const float* __restrict__ med, const int32_t* __restrict__ offsets, int Np, int maxn, int total,
                                        int32_t* __restrict__ best, float* __restrict__ median) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= Np) return;
    const int o = offsets[p], n = offsets[p + 1] - o;
    if (n <= 0) { best[p] = -1; median[p] = 0.f; return; }
    // more observations than the caller's bound, or a list that leaves the caller's `total` (= the size of the med scratch: an
    // understated total, a non-zero offsets[0]): reported, not computed -- every read below stays inside med[0, total)
    if (n > maxn || o < 0 || o + n > total) { best[p] = -2; median[p] = 0.f; return; }
    float bm = 2147483647.0f; int bi = 0;
    for (int i = 0; i < n; ++i) {
        const float m = med[o + i];
        if (m < bm) { bm = m; bi = i; }
    }
    best[p] = bi; median[p] = bm;
}

void launch_distinctive(hipStream_t s, const float* desc, const int32_t* offsets, int total, int Np, int maxn,
                        float* med, int32_t* best, float* median) {
    if (Np <= 0) return;
    int P2 = 1;
    while (P2 < maxn) P2 <<= 1;
    if (total > 0)
        hipLaunchKernelGGL(distinctive_rows_kernel, dim3(total), dim3(64), (size_t)P2 * 4, s, desc, offsets, Np, P2, med);
    hipLaunchKernelGGL(distinctive_pick_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, med, offsets, Np, P2, total, best, median);
}

}  // namespace rfe
