// kbench: fp32-class GEMM on the bf16 matrix pipe by operand splitting (tuning harness, not product code).
//   a = a0 + a1 + a2, b = b0 + b1 + b2 (three bf16 terms each, exact: 3 x 8 significand bits), and
//   a.b ~= a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0   (the three dropped products are O(2^-24) relative, one fp32 ulp)
// = 6 v_mfma_f32_32x32x16_bf16 (32 cycles each) per 32x32x16 block against 8 v_mfma_f32_32x32x2_f32 (64 cycles each): nominal 2.67x.
// Measures, at LightGlue's Linear shapes (M = 65536 token rows): time / effective fp32 TFLOP/s and the error against float64,
// beside a k-ascending fp32 fmaf chain (what the fp32 MFMA computes).  Variants:
//   PRE  : A and B pre-split into bf16 planes in global memory (upper bound of the MFMA / LDS path)
//   SPLIT: A fp32 in global memory, split while staging (what a drop-in Linear would do; B = weights are pre-split once at load time)
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off gemm_bf16x3.hip -o gemm_bf16x3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// three-term bf16 split of two floats (round to nearest even at every step): planes p0, p1, p2 each hold (lo, hi) packed
__device__ __forceinline__ void split2(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = cvt_pk_bf16(x, y);
    const float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xffff0000u);
    p1 = cvt_pk_bf16(rx, ry);
    const float sx = rx - __uint_as_float(p1 << 16), sy = ry - __uint_as_float(p1 & 0xffff0000u);
    p2 = cvt_pk_bf16(sx, sy);
}

// two-term fp16 split of two floats: a = a0 + a1 with a0 = fp16(a) (11 significand bits, round to nearest), a1 = fp16(a - a0) (the next
// 11): 22 bits of the 24 -- the representation error is 2^-22 |a| (plus fp16's absolute floor 2^-25 for the residual of small values:
// fp16 subnormals), i.e. about 4 fp32 ulps.  Planes p0, p1 hold (lo, hi) packed.
__device__ __forceinline__ void split2_f16(float x, float y, uint32_t& p0, uint32_t& p1) {
    const _Float16 hx = (_Float16)x, hy = (_Float16)y;
    const _Float16 lx = (_Float16)(x - (float)hx), ly = (_Float16)(y - (float)hy);
    p0 = (uint32_t)__builtin_bit_cast(uint16_t, hx) | ((uint32_t)__builtin_bit_cast(uint16_t, hy) << 16);
    p1 = (uint32_t)__builtin_bit_cast(uint16_t, lx) | ((uint32_t)__builtin_bit_cast(uint16_t, ly) << 16);
}
__global__ void presplit_f16_kernel(const float* __restrict__ x, uint16_t* __restrict__ out, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    uint32_t p0, p1;
    split2_f16(x[i], x[i + 1], p0, p1);
    *reinterpret_cast<uint32_t*>(out + i) = p0;
    *reinterpret_cast<uint32_t*>(out + n + i) = p1;
}

// one-off: fp32 [rows][K] -> three bf16 planes [3][rows][K]
__global__ void presplit_kernel(const float* __restrict__ x, uint16_t* __restrict__ out, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    uint32_t p0, p1, p2;
    split2(x[i], x[i + 1], p0, p1, p2);
    *reinterpret_cast<uint32_t*>(out + i) = p0;
    *reinterpret_cast<uint32_t*>(out + n + i) = p1;
    *reinterpret_cast<uint32_t*>(out + 2 * n + i) = p2;
}

// Workgroup 256 threads = 2x2 waves, tile 128 x 256 x 32, wave 64 x 128 = 2x4 accumulators of 32x32 (as the product fp32 kernel).
// LDS: per plane [rows][32 bf16] = 64 B rows of four 16-byte slots, slot s = 2 c + h (c = 16-wide k chunk, h = lane half) stored at
// s ^ ((row >> 2) & 3): ds_read_b128 fragment reads and ds_write_b128 staging writes are bank-conflict free.
// NPROD: 6 (default) or 3 (a0b0 + a0b1 + a1b0: the "bf16x2" variant, ~2^-16 relative) or 1 (plain bf16).
// PF: the next K tile is fetched into registers before the MFMA loop of the current one.  XM: XCD-aware block mapping -- the N/256
// column tiles of one 128-row panel are consecutive workgroups of ONE XCD (block b runs on XCD b % 8), so the panel's second and
// third read hit that XCD's L2.
// F16: the planes are fp16 (two of them: split2_f16): NPROD = 3 -> a0b0 + a0b1 + a1b0, NPROD = 4 -> + a1b1.
template <bool SPLIT_A, int NPROD, bool PF = false, bool XM = false, int ORD = 0, bool F16 = false>
__global__ __launch_bounds__(256, 2) void gemm_b3_kernel(const void* __restrict__ Aany, const uint16_t* __restrict__ Bp, const float* __restrict__ bias,
                                                         float* __restrict__ C, int M, int N, int K) {
    constexpr int BM = 128, BN = 256, BK = 32, MB = 2, NB = 4;
    constexpr int NPL = F16 ? 2 : NPROD == 1 ? 1 : NPROD == 3 ? 2 : 3;          // planes needed
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * (BM + BN) * 64];
    unsigned char* const As = lds;                    // [3][BM][64 B]
    unsigned char* const Bs = lds + 3 * BM * 64;      // [3][BN][64 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    int bx = blockIdx.x, by = blockIdx.y;
    if (XM) {   // 1-D grid of (N/256) * (M/128) blocks: XCD x = b % 8 takes the row panels x, x + 8, ...; inside an XCD the column tiles of a panel are consecutive
        const int nct = N / BN, b = blockIdx.x, xcd = b & 7, q = b >> 3;
        bx = q % nct; by = (q / nct) * 8 + xcd;
    }
    const int m0 = by * BM, n0 = bx * BN;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float bv = bias[n0 + (wn * NB + nb) * 32 + i];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = bv;
    }
    const size_t planeA = (size_t)M * K, planeB = (size_t)N * K;
    const int srow = tid >> 2, sslot = tid & 3;       // staging: 64 rows x 4 slots per pass
    const int sw_off = srow * 64 + ((sslot ^ ((srow >> 2) & 3)) << 4);   // rows srow + 64 p: same (row >> 2) & 3
    const int frag_sw = (i >> 2) & 3;                 // rows base + i, base a multiple of 32
    u32x4 ra[SPLIT_A ? 1 : NPL][2], rb[NPL][4];
    float4 fa[SPLIT_A ? 4 : 1];
    auto load_tile = [&](int k0) {
        if (SPLIT_A) {
            const float* A = static_cast<const float*>(Aany);
            // thread -> (row = tid / 8 + 32 it, 4 consecutive k): the fp32 tile is 128 rows x 128 B
#pragma unroll
            for (int it = 0; it < 4; ++it) fa[it] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + (tid >> 3) + 32 * it) * K + k0 + (tid & 7) * 4);
        } else {
            const uint16_t* A = static_cast<const uint16_t*>(Aany);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int p = 0; p < 2; ++p) ra[pl][p] = *reinterpret_cast<const u32x4*>(A + pl * planeA + (size_t)(m0 + srow + 64 * p) * K + k0 + sslot * 8);
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int p = 0; p < 4; ++p) rb[pl][p] = *reinterpret_cast<const u32x4*>(Bp + pl * planeB + (size_t)(n0 + srow + 64 * p) * K + k0 + sslot * 8);
    };
    if (PF) load_tile(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        if (!PF) load_tile(k0);
        __syncthreads();   // previous tile consumed
        if (SPLIT_A) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                uint32_t p0a, p1a, p2a = 0, p0b, p1b, p2b = 0;
                if (F16) { split2_f16(fa[it].x, fa[it].y, p0a, p1a); split2_f16(fa[it].z, fa[it].w, p0b, p1b); }
                else { split2(fa[it].x, fa[it].y, p0a, p1a, p2a); split2(fa[it].z, fa[it].w, p0b, p1b, p2b); }
                const int row = (tid >> 3) + 32 * it, kq = tid & 7;          // 4 k = half a 16-byte slot: slot kq / 2, byte 8 (kq & 1)
                const int off = row * 64 + (((kq >> 1) ^ ((row >> 2) & 3)) << 4) + ((kq & 1) << 3);
                *reinterpret_cast<u32x2*>(As + off) = u32x2{p0a, p0b};
                if (NPL > 1) *reinterpret_cast<u32x2*>(As + BM * 64 + off) = u32x2{p1a, p1b};
                if (NPL > 2) *reinterpret_cast<u32x2*>(As + 2 * BM * 64 + off) = u32x2{p2a, p2b};
            }
        } else {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(As + pl * BM * 64 + p * 64 * 64 + sw_off) = ra[pl][p];
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4*>(Bs + pl * BN * 64 + p * 64 * 64 + sw_off) = rb[pl][p];
        __syncthreads();
        if (PF && k0 + BK < K) load_tile(k0 + BK);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int slot = ((2 * c + h) ^ frag_sw) << 4;
            bf16x8 a[MB][NPL];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) a[mb][pl] = *reinterpret_cast<const bf16x8*>(As + pl * BM * 64 + ((wm * MB + mb) * 32 + i) * 64 + slot);
            if (F16) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f16x8 hb[2];
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) hb[pl] = *reinterpret_cast<const f16x8*>(Bs + pl * BN * 64 + ((wn * NB + nb) * 32 + i) * 64 + slot);
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        const f16x8 ha0 = __builtin_bit_cast(f16x8, a[mb][0]), ha1 = __builtin_bit_cast(f16x8, a[mb][1]);
                        if (NPROD == 4) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha1, hb[1], acc[mb][nb], 0, 0, 0);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha1, hb[0], acc[mb][nb], 0, 0, 0);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha0, hb[1], acc[mb][nb], 0, 0, 0);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha0, hb[0], acc[mb][nb], 0, 0, 0);
                    }
                }
                continue;
            }
            if (ORD == 2 && NPROD == 6) {   // product outer, the eight accumulators inner: eight MFMAs between two uses of an accumulator
                bf16x8 b[NB][3];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) b[nb][pl] = *reinterpret_cast<const bf16x8*>(Bs + pl * BN * 64 + ((wn * NB + nb) * 32 + i) * 64 + slot);
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][PA[q]], b[nb][PB[q]], acc[mb][nb], 0, 0, 0);
                continue;
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                bf16x8 b[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) b[pl] = *reinterpret_cast<const bf16x8*>(Bs + pl * BN * 64 + ((wn * NB + nb) * 32 + i) * 64 + slot);
                if (ORD == 1 && NPROD == 6) {   // product outer, the two row blocks inner: two MFMAs between two uses of an accumulator
                    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                    for (int q = 0; q < 6; ++q)
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][PA[q]], b[PB[q]], acc[mb][nb], 0, 0, 0);
                    continue;
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    if (NPROD == 6) {
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][2], b[0], acc[mb][nb], 0, 0, 0);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][0], b[2], acc[mb][nb], 0, 0, 0);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][1], b[1], acc[mb][nb], 0, 0, 0);
                    }
                    if (NPROD >= 3) {
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][1], b[0], acc[mb][nb], 0, 0, 0);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][0], b[1], acc[mb][nb], 0, 0, 0);
                    }
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][0], b[0], acc[mb][nb], 0, 0, 0);
                }
            }
        }
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

// ---- wide variant: 256 x 256 x 32 tile, 512 threads = 4x2 waves of 64 x 128, ONE workgroup per CU (2 waves per SIMD).
// B (the weights: static) is pre-split AND pre-packed at load time into the exact LDS image of every (column tile, K tile):
// [N/256][K/32] images of 48 KB = [3 planes][256 rows][64 B], 16-byte slots already XOR-swizzled -- a workgroup's B tile is one
// linear 48 KB run in global memory, copied by global_load_lds_dwordx4 (no VGPRs, no ds_write) into a DOUBLE-buffered LDS tile:
// tile t+1 is in flight during the MFMAs of tile t.  A (activations, fp32) is register-prefetched one tile ahead and split into
// its three bf16 planes while it is staged (single LDS buffer).  Operand traffic per FLOP is 0.63x that of the 128 x 256 tile.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

__global__ void pack_b_kernel(const float* __restrict__ B, unsigned char* __restrict__ out, int N, int K) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * K / 2) return;
    const int n = (int)(idx / (K / 2)), k = (int)(idx % (K / 2)) * 2;
    uint32_t p0, p1, p2;
    split2(B[(size_t)n * K + k], B[(size_t)n * K + k + 1], p0, p1, p2);
    const int ct = n >> 8, r = n & 255, kt = k >> 5, kk = k & 31;
    const size_t base = ((size_t)ct * (K / 32) + kt) * (3 * 256 * 64);
    const int off = r * 64 + ((((kk >> 3) ^ ((r >> 2) & 3))) << 4) + (kk & 7) * 2;
    *reinterpret_cast<uint32_t*>(out + base + off) = p0;
    *reinterpret_cast<uint32_t*>(out + base + 256 * 64 + off) = p1;
    *reinterpret_cast<uint32_t*>(out + base + 2 * 256 * 64 + off) = p2;
}

template <bool XM>
__global__ __launch_bounds__(512, 2) void gemm_b3w_kernel(const float* __restrict__ A, const unsigned char* __restrict__ Bpk, const float* __restrict__ bias,
                                                          float* __restrict__ C, int M, int N, int K) {
    constexpr int BM = 256, BN = 256, BK = 32, MB = 2, NB = 4, BIMG = 3 * BN * 64;
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * BM * 64 + 2 * BIMG];   // 48 KB + 96 KB
    unsigned char* const As = lds;                    // [3][BM][64 B]
    unsigned char* const Bs = lds + 3 * BM * 64;      // [2][3][BN][64 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    int bx = blockIdx.x, by = blockIdx.y;
    if (XM) { const int nct = N / BN, b = blockIdx.x, xcd = b & 7, q = b >> 3; bx = q % nct; by = (q / nct) * 8 + xcd; }
    const int m0 = by * BM, n0 = bx * BN, nkt = K / BK;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float bv = bias[n0 + (wn * NB + nb) * 32 + i];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = bv;
    }
    const int frag_sw = (i >> 2) & 3;
    float4 fa[4];
    const float* const Arow = A + (size_t)(m0 + (tid >> 3)) * K + (tid & 7) * 4;     // rows (tid >> 3) + 64 it
    auto load_a = [&](int k0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) fa[it] = *reinterpret_cast<const float4*>(Arow + (size_t)64 * it * K + k0);
    };
    const unsigned char* const Bimg = Bpk + (size_t)bx * nkt * BIMG;
    auto issue_b = [&](int kt, int buf) {   // 48 KB = 48 wave-level copies of 1 KB, six per wave
        const unsigned char* src = Bimg + (size_t)kt * BIMG + (wave * 6) * 1024 + lane * 16;
        unsigned char* dst = Bs + buf * BIMG + (wave * 6) * 1024;
#pragma unroll
        for (int j = 0; j < 6; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(src + j * 1024), (lds_ptr_t)(dst + j * 1024), 16, 0, 0);
    };
    issue_b(0, 0);
    load_a(0);
    for (int t = 0; t < nkt; ++t) {
        const int buf = t & 1;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            uint32_t p0a, p1a, p2a, p0b, p1b, p2b;
            split2(fa[it].x, fa[it].y, p0a, p1a, p2a);
            split2(fa[it].z, fa[it].w, p0b, p1b, p2b);
            const int row = (tid >> 3) + 64 * it, kq = tid & 7;
            const int off = row * 64 + (((kq >> 1) ^ ((row >> 2) & 3)) << 4) + ((kq & 1) << 3);
            *reinterpret_cast<u32x2*>(As + off) = u32x2{p0a, p0b};
            *reinterpret_cast<u32x2*>(As + BM * 64 + off) = u32x2{p1a, p1b};
            *reinterpret_cast<u32x2*>(As + 2 * BM * 64 + off) = u32x2{p2a, p2b};
        }
        __syncthreads();   // A planes of tile t visible; every wave's part of B tile t has landed (vmcnt(0) before the barrier)
        if (t + 1 < nkt) { issue_b(t + 1, buf ^ 1); load_a((t + 1) * BK); }
        const unsigned char* const Bt = Bs + buf * BIMG;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int slot = ((2 * c + h) ^ frag_sw) << 4;
            bf16x8 a[MB][3];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[mb][pl] = *reinterpret_cast<const bf16x8*>(As + pl * BM * 64 + ((wm * MB + mb) * 32 + i) * 64 + slot);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                bf16x8 b[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const bf16x8*>(Bt + pl * BN * 64 + ((wn * NB + nb) * 32 + i) * 64 + slot);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][2], b[0], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][0], b[2], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][1], b[1], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][1], b[0], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][0], b[1], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][0], b[0], acc[mb][nb], 0, 0, 0);
                }
            }
        }
        __syncthreads();   // A planes consumed
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

// the fp32 matrix pipe on the same tile shape, padded LDS, k ascending (= the product's bit-exact path without its refinements): the yardstick
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                                                          float* __restrict__ C, int M, int N, int K) {
    constexpr int BM = 128, BN = 256, BK = 32, LDT = 33, MB = 2, NB = 4;
    __shared__ float lds[(BM + BN) * LDT];
    float* As = lds; float* Bs = lds + BM * LDT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float bv = bias[n0 + (wn * NB + nb) * 32 + i];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = bv;
    }
    for (int k0 = 0; k0 < K; k0 += BK) {
        float4 ra[4], rb[8];
#pragma unroll
        for (int it = 0; it < 4; ++it) ra[it] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + (tid >> 3) + 32 * it) * K + k0 + (tid & 7) * 4);
#pragma unroll
        for (int it = 0; it < 8; ++it) rb[it] = *reinterpret_cast<const float4*>(B + (size_t)(n0 + (tid >> 3) + 32 * it) * K + k0 + (tid & 7) * 4);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) { float* d = As + ((tid >> 3) + 32 * it) * LDT + (tid & 7) * 4; d[0] = ra[it].x; d[1] = ra[it].y; d[2] = ra[it].z; d[3] = ra[it].w; }
#pragma unroll
        for (int it = 0; it < 8; ++it) { float* d = Bs + ((tid >> 3) + 32 * it) * LDT + (tid & 7) * 4; d[0] = rb[it].x; d[1] = rb[it].y; d[2] = rb[it].z; d[3] = rb[it].w; }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float a[MB], b[NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) a[mb] = As[((wm * MB + mb) * 32 + i) * LDT + 2 * s + h];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b[nb] = Bs[((wn * NB + nb) * 32 + i) * LDT + 2 * s + h];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb], b[nb], acc[mb][nb], 0, 0, 0);
        }
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

struct Err { double max_abs = 0, max_rel_scale = 0, rms_rel_scale = 0; };
// error of C against float64 on `nrows` sampled rows; scale = sum_k |a||b| (the condition-free yardstick of a dot product)
static Err check(const std::vector<float>& C, const std::vector<float>& A, const std::vector<float>& B, const std::vector<float>& bias, int M, int N, int K, int nrows) {
    Err e; double s2 = 0; size_t cnt = 0;
    for (int q = 0; q < nrows; ++q) {
        const int m = (int)(((uint64_t)q * 2654435761u + 12345) % (uint64_t)M);
        for (int n = 0; n < N; ++n) {
            double ref = bias[n], sc = std::fabs((double)bias[n]);
            for (int k = 0; k < K; ++k) { const double p = (double)A[(size_t)m * K + k] * (double)B[(size_t)n * K + k]; ref += p; sc += std::fabs(p); }
            const double d = std::fabs((double)C[(size_t)m * N + n] - ref);
            if (d > e.max_abs) e.max_abs = d;
            const double r = d / sc;
            if (r > e.max_rel_scale) e.max_rel_scale = r;
            s2 += r * r; ++cnt;
        }
    }
    e.rms_rel_scale = std::sqrt(s2 / cnt);
    return e;
}
static Err check_chain(const std::vector<float>& A, const std::vector<float>& B, const std::vector<float>& bias, int M, int N, int K, int nrows) {
    // the k-ascending fp32 fmaf chain itself against float64 (what any fp32 MFMA GEMM gives, up to the order)
    std::vector<float> C((size_t)M * N, 0.f);
    for (int q = 0; q < nrows; ++q) {
        const int m = (int)(((uint64_t)q * 2654435761u + 12345) % (uint64_t)M);
        for (int n = 0; n < N; ++n) {
            float acc = bias[n];
            for (int k = 0; k < K; ++k) acc = fmaf(A[(size_t)m * K + k], B[(size_t)n * K + k], acc);
            C[(size_t)m * N + n] = acc;
        }
    }
    return check(C, A, B, bias, M, N, K, nrows);
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

// micro: issue rate of v_mfma_f32_32x32x16_bf16 as a function of the number of independent accumulators a wave cycles through
// (NACC = 1: every MFMA waits for the previous one's result) and of the waves per SIMD (blocks of 256 or 512 threads, one per CU)
template <int NACC>
__global__ void mfma_chain_kernel(float* out, int iters) {
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f + j); b[j] = (__bf16)(blockIdx.x * 1e-3f - j); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][j & 15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
static void chain_case(float* out, int threads) {
    const int iters = 4096 / NACC * 8;
    auto go = [&] { mfma_chain_kernel<NACC><<<256, threads>>>(out, iters); };
    const float us = timeit(go, 5);
    const double n = (double)iters * NACC;     // MFMAs per wave
    printf("  bf16 32x32x16 chain, %d accumulator(s), %d wave(s)/SIMD: %.1f cycles per MFMA per SIMD at 2.4 GHz (%.0f TF)\n", NACC, threads / 256,
           us * 1e-6 * 2.4e9 / (n * (threads / 256)), n * (threads / 64) * 256 * 32768.0 / (us * 1e-6) / 1e12);
}

static bool g_quick = false;   // `gemm_bf16x3 quick`: ffn.0 only, three kernels, no host check (for rocprofv3 --pmc passes)
static void run_shape(const char* name, int M, int N, int K, bool wide_range) {
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hb(N);
    uint64_t s = 0x9E3779B97F4A7C15ull ^ (uint64_t)(N * 131 + K);
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    auto gauss = [&]() { const double u = rnd() + 1e-12, v = rnd(); return std::sqrt(-2 * std::log(u)) * std::cos(6.283185307179586 * v); };
    // activations ~ N(0,1) (optionally x 2^U(-6,6): mixed magnitudes inside a row), weights ~ N(0, 1/sqrt(K)) like an initialised Linear
    for (auto& v : hA) v = (float)(gauss() * (wide_range ? std::exp2(rnd() * 12 - 6) : 1.0));
    for (auto& v : hB) v = (float)(gauss() / std::sqrt((double)K));
    for (auto& v : hb) v = (float)(gauss() * 0.1);
    float *dA, *dB, *db, *dC; uint16_t *pA, *pB; unsigned char* pkB;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&pA, hA.size() * 6)); CK(hipMalloc(&pB, hB.size() * 6));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    presplit_kernel<<<(unsigned)((hA.size() / 2 + 255) / 256), 256>>>(dA, pA, hA.size());
    presplit_kernel<<<(unsigned)((hB.size() / 2 + 255) / 256), 256>>>(dB, pB, hB.size());
    uint16_t* hB16; CK(hipMalloc(&hB16, hB.size() * 4));
    presplit_f16_kernel<<<(unsigned)((hB.size() / 2 + 255) / 256), 256>>>(dB, hB16, hB.size());
    CK(hipMalloc(&pkB, hB.size() * 6));
    pack_b_kernel<<<(unsigned)((hB.size() / 2 + 255) / 256), 256>>>(dB, pkB, N, K);
    CK(hipDeviceSynchronize());
    const dim3 grid(N / 256, M / 128);
    const double flop = 2.0 * M * N * K;
    std::vector<float> hC((size_t)M * N);
    const int nrows = 48;
    printf("%s  M=%d N=%d K=%d%s\n", name, M, N, K, wide_range ? "  (activations with mixed magnitudes 2^-6..2^6)" : "");
    const Err ec = g_quick ? Err{} : check_chain(hA, hB, hb, M, N, K, nrows);
    printf("  %-44s %8s %8s   err vs f64: max abs %.3e  max |d|/sum|a||b| %.3e  rms %.3e\n", "host fp32 fmaf chain (k ascending)", "", "", ec.max_abs, ec.max_rel_scale, ec.rms_rel_scale);
    auto report = [&](const char* what, float us) {
        if (g_quick) { printf("  %-44s %7.1f us %6.1f TF\n", what, us, flop / (us * 1e-6) / 1e12); return; }
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        const Err e = check(hC, hA, hB, hb, M, N, K, nrows);
        printf("  %-44s %7.1f us %6.1f TF   err vs f64: max abs %.3e  max |d|/sum|a||b| %.3e  rms %.3e\n", what, us, flop / (us * 1e-6) / 1e12, e.max_abs, e.max_rel_scale, e.rms_rel_scale);
    };
    report("fp32 MFMA 32x32x2, plain tile (yardstick)", timeit([&] { gemm_f32_kernel<<<grid, 256>>>(dA, dB, db, dC, M, N, K); }));
    if (g_quick) {
        report("bf16x3, 6 products, A split at staging", timeit([&] { gemm_b3_kernel<true, 6><<<grid, 256>>>(dA, pB, db, dC, M, N, K); }));
        report("bf16x3 wide 256x256, packed B via LDS-DMA", timeit([&] { gemm_b3w_kernel<false><<<dim3(N / 256, M / 256), 512>>>(dA, pkB, db, dC, M, N, K); }));
        return;
    }
    report("bf16x3, 6 products, A and B pre-split", timeit([&] { gemm_b3_kernel<false, 6><<<grid, 256>>>(pA, pB, db, dC, M, N, K); }));
    report("bf16x3, 6 products, A split at staging", timeit([&] { gemm_b3_kernel<true, 6><<<grid, 256>>>(dA, pB, db, dC, M, N, K); }));
    report("bf16x3, split A, MFMA order: 2 apart", timeit([&] { gemm_b3_kernel<true, 6, false, false, 1><<<grid, 256>>>(dA, pB, db, dC, M, N, K); }));
    report("bf16x3, split A, MFMA order: 8 apart", timeit([&] { gemm_b3_kernel<true, 6, false, false, 2><<<grid, 256>>>(dA, pB, db, dC, M, N, K); }));
    report("bf16x3, 6 products, split A, prefetch", timeit([&] { gemm_b3_kernel<true, 6, true><<<grid, 256>>>(dA, pB, db, dC, M, N, K); }));
    report("bf16x3, 6 products, split A, XCD map", timeit([&] { gemm_b3_kernel<true, 6, false, true><<<dim3((N / 256) * (M / 128)), 256>>>(dA, pB, db, dC, M, N, K); }));
    report("bf16x3, 6 products, split A, prefetch + XCD map", timeit([&] { gemm_b3_kernel<true, 6, true, true><<<dim3((N / 256) * (M / 128)), 256>>>(dA, pB, db, dC, M, N, K); }));
    report("bf16x3 wide 256x256, packed B via LDS-DMA", timeit([&] { gemm_b3w_kernel<false><<<dim3(N / 256, M / 256), 512>>>(dA, pkB, db, dC, M, N, K); }));
    report("bf16x3 wide 256x256, packed B, XCD map", timeit([&] { gemm_b3w_kernel<true><<<dim3((N / 256) * (M / 256)), 512>>>(dA, pkB, db, dC, M, N, K); }));
    report("fp16x2, 3 products (a0b0 + a0b1 + a1b0), A split at staging", timeit([&] { gemm_b3_kernel<true, 3, false, false, 0, true><<<grid, 256>>>(dA, hB16, db, dC, M, N, K); }));
    report("fp16x2, 4 products (+ a1b1), A split at staging", timeit([&] { gemm_b3_kernel<true, 4, false, false, 0, true><<<grid, 256>>>(dA, hB16, db, dC, M, N, K); }));
    report("bf16x2, 3 products, A split at staging", timeit([&] { gemm_b3_kernel<true, 3><<<grid, 256>>>(dA, pB, db, dC, M, N, K); }));
    report("plain bf16, 1 product, A converted at staging", timeit([&] { gemm_b3_kernel<true, 1><<<grid, 256>>>(dA, pB, db, dC, M, N, K); }));
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(db)); CK(hipFree(dC)); CK(hipFree(pA)); CK(hipFree(pB)); CK(hipFree(pkB)); CK(hipFree(hB16));
}

int main(int argc, char** argv) {
    const int M = 65536;
    g_quick = argc > 1 && argv[1][0] == 'q';
    {   // clock / power warm-up
        float* x; CK(hipMalloc(&x, (size_t)M * 256 * 4)); CK(hipMemset(x, 0, (size_t)M * 256 * 4));
        float* w; CK(hipMalloc(&w, 256 * 256 * 4)); CK(hipMemset(w, 0, 256 * 256 * 4));
        float* c; CK(hipMalloc(&c, (size_t)M * 256 * 4));
        for (int i = 0; i < 1500; ++i) gemm_f32_kernel<<<dim3(1, M / 128), 256>>>(x, w, w, c, M, 256, 256);
        CK(hipDeviceSynchronize()); CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(c));
    }
    if (g_quick) { run_shape("ffn.0", M, 512, 512, false); return 0; }
    {
        float* o; CK(hipMalloc(&o, 256 * 512 * 4));
        chain_case<1>(o, 256); chain_case<2>(o, 256); chain_case<4>(o, 256); chain_case<8>(o, 256);
        chain_case<1>(o, 512); chain_case<2>(o, 512); chain_case<4>(o, 512);
        CK(hipFree(o));
    }
    run_shape("ffn.0", M, 512, 512, false);
    run_shape("ffn.0", M, 512, 512, true);
    run_shape("qkv", M, 768, 256, false);
    run_shape("ffn.3", M, 256, 512, false);
    return 0;
}
