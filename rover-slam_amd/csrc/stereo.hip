// stereo.hip -- sparse stereo matching on the extracted features: Frame::ComputeStereoMatches of the
// reference (src/Frame.cc:1159-1446), which runs right after the two extractor calls on every stereo frame
// (src/Frame.cc:142-171).  SURVEY.md section 8(f) row N2.  nLevels == 1 only (octave 0, scale 1), like the
// SuperPoint path itself.
//   kernel 1 (one wave per left keypoint): row-band (+-2 px) / disparity-range candidate scan over the right
//     keypoints, 256-d L2 distance (DescriptorDistance_sp, src/Matchers/SPmatcher.cc:2184) -- float
//     differences, double accumulation in a fixed lane/butterfly order shared with the oracle --, best
//     candidate under TH_HIGH, accepted under (TH_HIGH+TH_LOW)/2; then the 11x11 SAD slide over +-5 px on
//     the raw images (integer, exact), parabola sub-pixel fit, disparity -> depth.
//   kernel 2 (one workgroup): bitonic sort of (SAD, index), median, outlier cut at 1.5*1.4*median.
// HBM/latency bound: N*(1 KB descriptor + candidates * 1 KB) reads; exact integer / IEEE arithmetic ->
// bit-exact against oracle/rfe_oracle.c:rfo_stereo_match.
#include "rfe_internal.h"

namespace rfe {

// KT: keypoint coordinate type -- float (the caller's cv::KeyPoint::pt) or int32 (the extractor's own output, device
// resident); counts: optional device-side {N, Nr} (stream mode: no host round trip for the keypoint counts), the grid
// then covers the capacity and surplus waves leave at once.
template <typename KT>
__global__ __launch_bounds__(256) void stereo_match_kernel(
    const uint8_t* __restrict__ imgL, const uint8_t* __restrict__ imgR, int H, int W, int stride,
    const KT* __restrict__ kL, int N, const KT* __restrict__ kR, int Nr, const int32_t* __restrict__ counts,
    const float* __restrict__ dL, const float* __restrict__ dR, float maxD, float mbf, float* __restrict__ uRight,
    float* __restrict__ depth, int32_t* __restrict__ sadv) {
    const int lane = threadIdx.x & 63;
    const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (counts) { N = counts[0] < N ? counts[0] : N; Nr = counts[1] < Nr ? counts[1] : Nr; }
    if (iL >= N) return;
    const float TH_HIGH = 1.4f, TH_LOW = 1.2f;
    const float thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const float minD = 0.f;
    float outU = -1.0f, outZ = -1.0f; int outS = -1;
    const float uL = (float)kL[2 * iL], vL = (float)kL[2 * iL + 1];
    const float minU = uL - maxD, maxU = uL - minD;
    float bestDist = TH_HIGH; int bestIdx = -1;
    if (!(maxU < 0)) {
        const float4 a = reinterpret_cast<const float4*>(dL + (size_t)iL * 256)[lane];
        const int row = (int)vL;
        for (int base = 0; base < Nr; base += 64) {
            const int iR = base + lane;
            bool cand = false;
            if (iR < Nr) {
                const float uR = (float)kR[2 * iR], yR = (float)kR[2 * iR + 1];
                cand = !(row < (int)floorf(yR - 2.0f) || row > (int)ceilf(yR + 2.0f)) && uR >= minU && uR <= maxU;
            }
            unsigned long long mask = __ballot(cand);
            while (mask) {
                const int j = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int ic = base + j;
                const float4 b = reinterpret_cast<const float4*>(dR + (size_t)ic * 256)[lane];
                const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
                double p = 0.0;
                p += (double)d0 * (double)d0; p += (double)d1 * (double)d1; p += (double)d2 * (double)d2; p += (double)d3 * (double)d3;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off);
                const float dist = (float)sqrt(p);
                if (dist < bestDist) { bestDist = dist; bestIdx = ic; }
            }
        }
    }
    if (bestDist < thOrbDist && bestIdx >= 0) {
        const float uR0 = (float)kR[2 * bestIdx];
        const int su = (int)roundf(uL), sv = (int)roundf(vL), sr = (int)roundf(uR0);
        const int w = 5, Lh = 5;
        const bool ok = !(sr - Lh - w < 0 || sr + Lh + w + 1 >= W) && !(sv - w < 0 || sv + w >= H || su - w < 0 || su + w >= W);
        if (ok) {
            float vd[11]; float best = 2147483647.0f; int bestinc = 0;
            // this lane's two patch pixels (121 = 64 + 57)
            const int p0 = lane, p1 = lane + 64;
            const int y0 = p0 / 11 - w, x0 = p0 % 11 - w, y1 = p1 / 11 - w, x1 = p1 % 11 - w;
            const int l0 = imgL[(size_t)(sv + y0) * stride + su + x0];
            const int l1 = p1 < 121 ? imgL[(size_t)(sv + y1) * stride + su + x1] : 0;
#pragma unroll
            for (int inc = -Lh; inc <= Lh; ++inc) {
                int sad = abs(l0 - (int)imgR[(size_t)(sv + y0) * stride + sr + inc + x0]);
                if (p1 < 121) sad += abs(l1 - (int)imgR[(size_t)(sv + y1) * stride + sr + inc + x1]);
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) sad += __shfl_xor(sad, off);
                const float dist = (float)sad;
                if (dist < best) { best = dist; bestinc = inc; }
                vd[Lh + inc] = dist;
            }
            if (!(bestinc == -Lh || bestinc == Lh)) {
                float d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
                for (int t = 1; t < 10; ++t) if (t == Lh + bestinc) { d1 = vd[t - 1]; d2 = vd[t]; d3 = vd[t + 1]; }
                const float deltaR = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = 1.0f * ((float)sr + (float)bestinc + deltaR);
                    float disparity = uL - bestuR;
                    if (disparity >= minD && disparity < maxD) {
                        if (disparity <= 0) { disparity = 0.01f; bestuR = uL - 0.01f; }
                        outZ = mbf / disparity; outU = bestuR; outS = (int)best;
                    }
                }
            }
        }
    }
    if (lane == 0) { uRight[iL] = outU; depth[iL] = outZ; sadv[iL] = outS; }
}

// median outlier cut (Frame.cc:1431-1445): one workgroup, N <= 4096
__global__ __launch_bounds__(1024) void stereo_filter_kernel(int N, int P2, const int32_t* __restrict__ counts,
                                                             const int32_t* __restrict__ sadv, float* __restrict__ uRight,
                                                             float* __restrict__ depth) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    __shared__ int cnt;
    const int tid = threadIdx.x;
    if (counts) N = counts[0] < N ? counts[0] : N;
    if (tid == 0) cnt = 0;
    for (int k = tid; k < P2; k += 1024) keys[k] = ~0ull;   // padding sorts last
    __syncthreads();
    for (int i = tid; i < N; i += 1024) {
        const int d = sadv[i];
        if (d >= 0) { const int pos = atomicAdd(&cnt, 1); keys[pos] = ((unsigned long long)(unsigned int)d << 32) | (unsigned int)i; }
    }
    __syncthreads();
    const int nv = cnt;
    if (nv == 0) return;
    for (int kk = 2; kk <= P2; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < P2; t += 1024) {
                const int ixj = t ^ j;
                if (ixj > t) {
                    const unsigned long long a = keys[t], c = keys[ixj];
                    const bool asc = (t & kk) == 0;
                    if (asc ? (a > c) : (a < c)) { keys[t] = c; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    const float median = (float)(int)(keys[nv / 2] >> 32);
    const float thDist = 1.5f * 1.4f * median;
    for (int k = tid; k < nv; k += 1024) {
        const unsigned long long key = keys[k];
        if (!((float)(int)(key >> 32) < thDist)) { const int i = (int)(key & 0xffffffffu); uRight[i] = -1.0f; depth[i] = -1.0f; }
    }
}

void launch_stereo_match(hipStream_t s, const uint8_t* imgL, const uint8_t* imgR, int H, int W, int stride,
                         const float* kL, int N, const float* kR, int Nr, const float* dL, const float* dR, float mb,
                         float mbf, float* uRight, float* depth, int32_t* sadv) {
    if (N <= 0) return;
    const float maxD = mbf / mb;
    hipLaunchKernelGGL(stereo_match_kernel<float>, dim3((N + 3) / 4), dim3(256), 0, s, imgL, imgR, H, W, stride, kL, N, kR, Nr,
                       (const int32_t*)nullptr, dL, dR, maxD, mbf, uRight, depth, sadv);
    int P2 = 1;
    while (P2 < N) P2 <<= 1;
    hipLaunchKernelGGL(stereo_filter_kernel, dim3(1), dim3(1024), (size_t)P2 * 8, s, N, P2, (const int32_t*)nullptr, sadv, uRight, depth);
}

// device-resident form: integer keypoints straight from the extractor, counts = device {n_left, n_right}, capacity Kmax.
// Entries >= n_left of uRight / depth are set to -1 (no match) so the outputs are fully defined.
__global__ void stereo_fill_kernel(float* __restrict__ uRight, float* __restrict__ depth, int32_t* __restrict__ sadv, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { uRight[i] = -1.0f; depth[i] = -1.0f; sadv[i] = -1; }
}
void launch_stereo_match_counts(hipStream_t s, const uint8_t* imgL, const uint8_t* imgR, int H, int W, int stride,
                                const int32_t* kL, const int32_t* kR, int Kmax, const int32_t* counts, const float* dL,
                                const float* dR, float mb, float mbf, float* uRight, float* depth, int32_t* sadv) {
    if (Kmax <= 0) return;
    const float maxD = mbf / mb;
    hipLaunchKernelGGL(stereo_fill_kernel, dim3((Kmax + 255) / 256), dim3(256), 0, s, uRight, depth, sadv, Kmax);
    hipLaunchKernelGGL(stereo_match_kernel<int32_t>, dim3((Kmax + 3) / 4), dim3(256), 0, s, imgL, imgR, H, W, stride, kL, Kmax, kR, Kmax,
                       counts, dL, dR, maxD, mbf, uRight, depth, sadv);
    int P2 = 1;
    while (P2 < Kmax) P2 <<= 1;
    hipLaunchKernelGGL(stereo_filter_kernel, dim3(1), dim3(1024), (size_t)P2 * 8, s, Kmax, P2, counts, sadv, uRight, depth);
}

// ------------------------------------------------------------------------------------------
// all-pairs descriptor distances D[i][j] = DescriptorDistance_sp(a_i, b_j)
// (src/Matchers/SPmatcher.cc:2184-2189) for the projection / fuse searches, whose candidate lists stay
// on the CPU (SPmatcher.cc:1170-1354, 49-357): the CPU loop reads D instead of calling cv::norm.
// One wave per row i, same canonical arithmetic as the stereo kernel (bit-exact vs the oracle).
__global__ __launch_bounds__(256) void l2_matrix_kernel(const float* __restrict__ a, int M, const float* __restrict__ b, int N,
                                                        float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= M) return;
    const float4 av = reinterpret_cast<const float4*>(a + (size_t)i * 256)[lane];
    for (int j = blockIdx.y; j < N; j += gridDim.y) {
        const float4 bv = reinterpret_cast<const float4*>(b + (size_t)j * 256)[lane];
        const float d0 = av.x - bv.x, d1 = av.y - bv.y, d2 = av.z - bv.z, d3 = av.w - bv.w;
        double p = 0.0;
        p += (double)d0 * (double)d0; p += (double)d1 * (double)d1; p += (double)d2 * (double)d2; p += (double)d3 * (double)d3;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off);
        if (lane == 0) out[(size_t)i * N + j] = (float)sqrt(p);
    }
}
void launch_l2_matrix(hipStream_t s, const float* a, int M, const float* b, int N, float* out) {
    if (M <= 0 || N <= 0) return;
    hipLaunchKernelGGL(l2_matrix_kernel, dim3((M + 3) / 4, N < 64 ? N : 64), dim3(256), 0, s, a, M, b, N, out);
}

// SURVEY 8(f) N4 (GPU half): Frame::binarize_descriptors (src/Frame.cc:1034-1043): out = desc > 0 ? 1 : 0, u8 [n,256]
__global__ void binarize_kernel(const float* __restrict__ d, int64_t n, uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = reinterpret_cast<const float4*>(d)[i];
    reinterpret_cast<uchar4*>(out)[i] = make_uchar4(v.x > 0.f, v.y > 0.f, v.z > 0.f, v.w > 0.f);
}
void launch_binarize(hipStream_t s, const float* d, int64_t rows, uint8_t* out) {
    const int64_t n4 = rows * 64;
    if (n4 <= 0) return;
    hipLaunchKernelGGL(binarize_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, d, n4, out);
}

}  // namespace rfe
