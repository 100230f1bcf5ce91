// lg_kernels.hip -- LightGlue on CDNA4: positional encoding, rotary, fused (flash-style) attention
// on the fp32 matrix cores, LayerNorm+GELU, and the dual-softmax assignment with mutual filtering.
// Replaces the body of lightglue_sim.onnx, which the reference executes through
// Ort::Session::Run (src/Matchers/lightglue_onnx.cpp:210-214); output contract matches0 [S,2] /
// mscores0 [S] as consumed by Matcher_PostProcess_fused (lightglue_onnx.cpp:404-409).
// fp32 throughout (descriptor tolerance 1e-4 of the north star rules out bf16 operands).
#include <stdlib.h>
#include "rfe_internal.h"

namespace rfe {

// ---------------------------------------------------------------- posenc: theta = Wr . p ; cos/sin
// table layout: [row][f] -> (cos, sin), so that the two pairs a staging thread rotates are one 16-byte load
__global__ void lg_posenc_kernel(const float* __restrict__ kn, const float* __restrict__ wr, int rows, float2* __restrict__ csn) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rows * 32) return;
    const int i = gid >> 5, f = gid & 31;
    const float th = fmaf(wr[2 * f + 1], kn[2 * i + 1], wr[2 * f] * kn[2 * i]);
    csn[gid] = make_float2(cosf(th), sinf(th));
}
void launch_lg_posenc(hipStream_t s, const float* kn, const float* wr, int rows, float* csn) {
    hipLaunchKernelGGL(lg_posenc_kernel, dim3((rows * 32 + 255) / 256), dim3(256), 0, s, kn, wr, rows, reinterpret_cast<float2*>(csn));
}

// ---------------------------------------------------------------- fused attention
// softmax(Q K^T / 8) V per (sequence, head), online softmax, never materialising the L x L matrix.
// Workgroup = 4 waves = 128 queries of one head; wave = 32 queries.  Per 64-key tile staged in LDS:
//   S^T = K . Q^T   (A = K tile rows from LDS, B = Q fragment held in 32 VGPRs)  -> lane = one query,
//         16 keys in registers: the softmax row reduction is in-lane + one xor-32 shuffle;
//   O^T += V^T . P^T  where the B operand of k-step r IS accumulator register r of S^T (the key
//         order of the reduction is permuted to the D-layout order, so P never moves).
// 64 MFMA (32x32x2 f32) per 32x32 tile, no wasted FLOPs: 4*L*L*64 per head.
constexpr int AT_Q = 128, AT_K = 64, AT_LDK = 65;
constexpr float AT_DEFER = 16.0f;   // log2 units
constexpr int AT_SPLIT_MAX = 8;               // key ranges of the split variant
constexpr size_t AT_SPLIT_MAX_ROWS = 8192;    // only problems this small are latency-bound enough to split

// Online softmax step on one 32-key x 32-query S^T block, trimmed for VALU count: on gfx950 the fp32 MFMA runs at the vector rate
// (64 FLOP / clk / SIMD) and every VALU instruction of ANY wave of the SIMD measurably costs matrix-pipe time (the softmax removed:
// +12 % with the staging gone, profiles/r01_pmc.md; LayerNorm + GELU on ffn.3's operand path: its full VALU time), so the steady
// state does as little as it can:
//   * the reference maximum is subtracted INSIDE the accumulation -- the accumulator starts at -m_run instead of 0 (at_softmax_init) --
//     so no subtraction per element;
//   * the tile maximum is a v_max3 tree per lane, compared against the deferred-rescale threshold WITHOUT combining the two half-waves
//     (__any covers both); only the rare branch (always the first tile) combines them, moves the reference and shifts the block;
//   * the row sum stays a per-half-wave partial; the halves are added once after the last tile (at_softmax_finish).
// st: in = S^T block relative to the reference, out = P^T block.  first: wave-uniform, the first tile of this wave's key range.
__device__ __forceinline__ float at_softmax_init(bool first, float m_run) { return first ? 0.f : -m_run; }
__device__ __forceinline__ void at_softmax_step(f32x16& st, bool first, float& m_run, float& l_run, f32x16& o0, f32x16& o1) {
    float mx = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, st[r]), st[r + 1]);
    mx = fmaxf(mx, st[15]);
    if (__any(first || mx > AT_DEFER)) {
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float ref = first ? 0.f : m_run;            // what the accumulator already subtracted
        const float m_new = fmaxf(m_run, mx + ref);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // m_run = -inf on the first tile -> 0
        const float d = ref - m_new;
        l_run *= alpha;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] += d; o0[r] *= alpha; o1[r] *= alpha; }
    }
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = __builtin_amdgcn_exp2f(st[r]); ps += st[r]; }
    l_run += ps;
}
__device__ __forceinline__ float at_softmax_finish(float l_run) { return l_run + __shfl_xor(l_run, 32); }

// ABL: timing ablations only (wrong results): 1 = no softmax VALU, 2 = stage only the first K/V tile, 3 = both
// SPLIT: split-key variant for latency-bound problems (one pair = 64 (sequence, head, query block) units for 256 CUs and
// a 2048-MFMA serial chain per wave): blockIdx.y selects one of gridDim.y key ranges, the workgroup writes its
// unnormalised accumulators and (reference max, sum) to `part`, and lg_attention_combine_kernel merges the ranges.
// ROPE (since round 5 the FALLBACK of the self blocks: the throughput projection rotates q | k in its epilogue -- gemm.hip, ROPE -- and the self blocks run
// lg_attention_dma_kernel like the cross blocks, 9.48 -> 9.14 ms of attention per step for +0.08 ms of projection; this form still serves RFE_OPT_LG_FP16X2's
// fp32 fallbacks and shapes the rotary epilogue does not take): the LightGlue rotary encoding of the self blocks -- (t0, t1) -> (t0 c - t1 s, t1 c + t0 s) on adjacent pairs (2f, 2f+1),
// (c, s) = rope_csn[row][f] -- is applied HERE, to the Q fragment as it is loaded and to every K tile as it is staged, with the
// same three fp32 operations round 1's projection epilogue used (bit-identical results).  The qkv projection keeps the plain
// coalesced epilogue (101 -> 122 TFLOP/s); the K rows are rotated once per staging workgroup (8 query blocks per (sequence,
// head): redundant VALU work worth ~1 % of the MFMA time, one extra 16-byte load per staged K float4; the table is 256 B per
// row and L2 resident).  Net effect on the step: none within noise (profiles/r02_ab_notes.md) -- kept because it removes the
// LDS-transposed epilogue from the GEMM.  Variants that did not pay (double-buffered LDS, register prefetch of the next tile,
// 64 queries per wave, 256-query workgroups, k rotated by the projection and q here) are recorded in profiles/r01_pmc.md / r02_ab_notes.md.
// PFK (split variant only): the next K/V tile is fetched into registers right after the current one is published, i.e. under the
// MFMAs.  The split variant runs one workgroup per CU, so nothing else hides the global round trip of every tile (the throughput
// variant has 4 co-resident workgroups and measured slower with the prefetch: 146 VGPRs -> 3 workgroups, profiles/r02_ab_notes.md).
template <int ABL = 0, bool SPLIT = false, bool ROPE = false, bool PFK = false>
__global__ __launch_bounds__(256, PFK ? 2 : 4) void lg_attention_kernel(   // 4 workgroups per CU: at most 128 VGPRs
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld, float* __restrict__ out,
    int Lq, int Lk, int nqb, const int* __restrict__ qlen, const int* __restrict__ klen, const int* __restrict__ kv_map,
    int prio, float* __restrict__ part, int nseq_total, const float* __restrict__ rope_csn) {
    constexpr bool DBUF = false;   // (the double-buffered variant measured 4 % slower; kept out of the build)
    __shared__ float Ks[1][AT_K * AT_LDK];
    __shared__ float Vs[1][AT_K * 64];
    // XCD-aware decode (blocks are dealt round-robin to the 8 XCDs): all query blocks of one
    // (sequence, head) run on the same XCD so its K/V (512 KB) is fetched into one L2 only.
    const int Lb = blockIdx.x, xcd = Lb & 7, t_ = Lb >> 3;
    const int qb = t_ % nqb, unit = (t_ / nqb) * 8 + xcd;
    if (unit >= 4 * nseq_total) return;   // the grid is padded to a multiple of 8 (sequence, head) units so that the decode stays bijective
    const int seq = unit >> 2, head = unit & 3;
    const int kvseq = kv_map ? kv_map[seq] : seq;
    const int nq = qlen ? qlen[seq] : Lq;
    const int nk = klen ? klen[kvseq] : Lk;
    const int tid = threadIdx.x, lane = tid & 63;
    if (qb * AT_Q >= nq) {  // whole block is padding: keep the padded context rows defined (zero)
        if (SPLIT) return;  // the combine kernel zeroes them
        for (int e = tid; e < AT_Q * 64; e += 256) {
            const int row = qb * AT_Q + (e >> 6);
            if (row < Lq) out[((size_t)seq * Lq + row) * 256 + head * 64 + (e & 63)] = 0.f;
        }
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int qrow = qb * AT_Q + wave * 32 + j;  // this lane's query (may be >= nq: computed, stored as 0)
    const size_t qrow_c = (size_t)seq * Lq + (qrow < Lq ? qrow : Lq - 1);
    const float* qp = q + qrow_c * ld + head * 64 + h;
    constexpr float kScale = 0.125f * 1.44269504088896341f;  // 1/sqrt(64) * log2(e): softmax in base 2, folded into Q
    float qreg[32];
    if (ROPE) {   // lane (j, h) keeps component h of every pair: both components are loaded, the rotated one is kept
        const float2* qp2 = reinterpret_cast<const float2*>(q + qrow_c * ld + head * 64);
        const float4* cp = reinterpret_cast<const float4*>(rope_csn + qrow_c * 64);
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const float4 cs = cp[s2];                      // (c, s) of pairs 2 s2 and 2 s2 + 1
            const float2 t0 = qp2[2 * s2], t1 = qp2[2 * s2 + 1];
            const float a0 = t0.x * cs.x - t0.y * cs.y, a1 = t0.y * cs.x + t0.x * cs.y;
            const float b0 = t1.x * cs.z - t1.y * cs.w, b1 = t1.y * cs.z + t1.x * cs.w;
            qreg[2 * s2] = (h ? a1 : a0) * kScale;
            qreg[2 * s2 + 1] = (h ? b1 : b0) * kScale;
        }
    } else {
#pragma unroll
        for (int s = 0; s < 32; ++s) qreg[s] = qp[2 * s] * kScale;
    }

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;   // running max in the log2 domain, running sum

    const float* kbase = k + (size_t)kvseq * Lk * ld + head * 64;
    const float* vbase = v + (size_t)kvseq * Lk * ld + head * 64;
    const int skey = tid >> 4, sdq = tid & 15;  // staging: thread -> (key, 4 dims), 4 passes of 16 keys
    float4 rk[4], rv[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int key = k0 + skey + 16 * it;
            rk[it] = make_float4(0.f, 0.f, 0.f, 0.f); rv[it] = rk[it];
            if (key < nk) {
                rk[it] = *reinterpret_cast<const float4*>(kbase + (size_t)key * ld + sdq * 4);
                rv[it] = *reinterpret_cast<const float4*>(vbase + (size_t)key * ld + sdq * 4);
                if (ROPE) {   // dims 4 sdq .. 4 sdq + 3 = pairs f = 2 sdq, 2 sdq + 1
                    const float4 cs = *reinterpret_cast<const float4*>(rope_csn + ((size_t)kvseq * Lk + key) * 64 + 4 * sdq);
                    const float4 t = rk[it];
                    rk[it] = make_float4(t.x * cs.x - t.y * cs.y, t.y * cs.x + t.x * cs.y, t.z * cs.z - t.w * cs.w, t.w * cs.z + t.z * cs.w);
                }
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int key = skey + 16 * it;
            float* dk = Ks[buf] + key * AT_LDK + sdq * 4;
            dk[0] = rk[it].x; dk[1] = rk[it].y; dk[2] = rk[it].z; dk[3] = rk[it].w;
            *reinterpret_cast<float4*>(Vs[buf] + key * 64 + sdq * 4) = rv[it];
        }
    };
    if (DBUF) {
        fetch(0);
        stash(0);
        __syncthreads();
    }
    int buf = 0;
    int kbeg = 0, kend = nk;
    if (SPLIT) {   // key range of this split: whole 64-key tiles
        const int per = ((nk + (int)gridDim.y - 1) / (int)gridDim.y + AT_K - 1) / AT_K * AT_K;
        kbeg = (int)blockIdx.y * per;
        kend = kbeg + per < nk ? kbeg + per : nk;
    }
    if (PFK && kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += AT_K) {
        const bool more = DBUF && (k0 + AT_K < kend);
        if (DBUF) {
            if (more) fetch(k0 + AT_K);
        } else if (PFK) {
            __syncthreads();
            stash(0);
            __syncthreads();
            if (k0 + AT_K < kend) fetch(k0 + AT_K);
        } else if (!(ABL & 2) || k0 == 0) {
            __syncthreads();
            fetch(k0);
            stash(0);
            __syncthreads();
        }
#pragma unroll
        for (int sub = 0; sub < AT_K / 32; ++sub) {
            if (k0 + sub * 32 >= nk) break;
            // ---- S^T[key][query] = sum_d K[key][d] * Q[query][d] (relative to the running reference maximum: at_softmax_step)
            f32x16 st;
            const bool first = k0 == kbeg && sub == 0;
            {
                const float init = (ABL & 1) ? 0.f : at_softmax_init(first, m_run);
#pragma unroll
                for (int r = 0; r < 16; ++r) st[r] = init;
            }
            const float* ka = Ks[buf] + (sub * 32 + j) * AT_LDK + h;
            if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 32; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[2 * s], qreg[s], st, 0, 0, 0);
            if (prio) __builtin_amdgcn_s_setprio(0);
            // ---- online softmax over this lane's 16 keys (+ the other half-wave's 16)
            if (!(ABL & 1)) {
            // only the last key tile can contain keys >= nk
            if (k0 + sub * 32 + 32 > nk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (key >= nk) st[r] = -INFINITY;
                }
            }
            at_softmax_step(st, first, m_run, l_run, o0, o1);
            } else { l_run = 1.f; }
            // ---- O^T[d][query] += sum_key V[key][d] * P[key][query]; k-step r uses key(r,h)
            const float* va = Vs[buf] + (sub * 32 + 4 * h) * 64 + j;
            if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = (r & 3) + 8 * (r >> 2);
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[kr * 64], st[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[kr * 64 + 32], st[r], o1, 0, 0, 0);
            }
            if (prio) __builtin_amdgcn_s_setprio(0);
        }
        if (DBUF) {
            if (more) stash(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    if (!(ABL & 1)) l_run = at_softmax_finish(l_run);
    if (SPLIT) {
        if (qrow < Lq) {
            const size_t prow = ((size_t)blockIdx.y * nseq_total + seq) * Lq + qrow;
            float* op = part + prow * 256 + head * 64;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = (r & 3) + 8 * (r >> 2) + 4 * h;
                op[d] = o0[r];
                op[d + 32] = o1[r];
            }
            if (h == 0) {
                float* ml = part + (size_t)gridDim.y * nseq_total * Lq * 256 + (prow * 4 + head) * 2;
                ml[0] = m_run; ml[1] = l_run;
            }
        }
        return;
    }
    if (qrow < Lq) {
        const float inv = (qrow < nq && l_run > 0.f) ? 1.0f / l_run : 0.f;  // padded rows -> 0
        float* op = out + ((size_t)seq * Lq + qrow) * 256 + head * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = (r & 3) + 8 * (r >> 2) + 4 * h;
            op[d] = o0[r] * inv;
            op[d + 32] = o1[r] * inv;
        }
    }
}

// ---- LDS-DMA variant of the throughput attention (cross blocks: no rotary on K).
// lg_attention_kernel fetches a K/V tile AFTER the barrier that frees the single LDS buffer, so each of the four workgroups of a CU
// sits out one global round trip per tile, and workgroups that compete for the same matrix pipes drift into step (PMC: 0.21 of the
// wave cycles at s_waitcnt / s_barrier, matrix pipe 0.83 busy).  A register prefetch costs the fourth workgroup (146 VGPRs, round 2).
// Here the tiles never touch a VGPR: 32-key tiles are copied by global_load_lds_dwordx4 into a DOUBLE-buffered LDS tile (2 x (8 + 8) KB
// = the 32 KB of the single 64-key buffer), tile t+1 is requested right after the barrier and has the whole compute phase of tile t
// to land; one barrier per 32 keys (as before: two per 64).  K image: row = key, sixteen 16-byte slots per row, slot c stored at
// c ^ (key & 15) -- the swizzle is applied on the SOURCE address (the LDS side of the copy is lane-linear) -- so that the A fragment
// of S^T = K . Q^T is ONE conflict-free ds_read_b128 per four k-steps (lane (j, h) reads K[j][8 g + 4 h .. + 3]); MFMA step (g, e)
// therefore multiplies k = 8 g + e and 8 g + 4 + e: the sum over the head dimension runs in a permuted order (tolerance-checked
// like every LightGlue kernel), Q is loaded in that order with eight float4 loads.  V image: plain [key][64].  Keys past the
// sequence length are read from the last valid row (finite) and masked in S.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
constexpr int AD_K = 32;
__global__ __launch_bounds__(256, 4) void lg_attention_dma_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld, float* __restrict__ out,
    int Lq, int Lk, int nqb, const int* __restrict__ qlen, const int* __restrict__ klen, const int* __restrict__ kv_map,
    int prio, int nseq_total, const float* __restrict__ rope_q) {
    // Dynamic LDS on purpose: with static `__shared__ float Kd[2][..]` tiles the compiler's wait-count pass cannot tell buffer buf from
    // buf ^ 1 and puts an s_waitcnt vmcnt(0) in front of the first ds_read of tile t -- i.e. AFTER tile t + 1 has just been requested, so
    // every wave sat out that round trip and nothing overlapped (round-3 advisor finding, visible in the ISA).  With one dynamic array and
    // the explicit wait + barrier below, the only vmcnt wait of the loop is the one that covers tile t.
    extern __shared__ __attribute__((aligned(16))) float ad_lds[];
    float (*Kd)[AD_K * 64] = reinterpret_cast<float (*)[AD_K * 64]>(ad_lds);
    float (*Vd)[AD_K * 64] = reinterpret_cast<float (*)[AD_K * 64]>(ad_lds + 2 * AD_K * 64);
    const int Lb = blockIdx.x, xcd = Lb & 7, t_ = Lb >> 3;
    const int qb = t_ % nqb, unit = (t_ / nqb) * 8 + xcd;
    if (unit >= 4 * nseq_total) return;
    const int seq = unit >> 2, head = unit & 3;
    const int kvseq = kv_map ? kv_map[seq] : seq;
    const int nq = qlen ? qlen[seq] : Lq;
    const int nk = klen ? klen[kvseq] : Lk;
    const int tid = threadIdx.x, lane = tid & 63;
    if (qb * AT_Q >= nq) {
        for (int e = tid; e < AT_Q * 64; e += 256) {
            const int row = qb * AT_Q + (e >> 6);
            if (row < Lq) out[((size_t)seq * Lq + row) * 256 + head * 64 + (e & 63)] = 0.f;
        }
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int qrow = qb * AT_Q + wave * 32 + j;
    const size_t qrow_c = (size_t)seq * Lq + (qrow < Lq ? qrow : Lq - 1);
    constexpr float kScale = 0.125f * 1.44269504088896341f;
    float qreg[32];   // qreg[4 g + e] = Q[query][8 g + 4 h + e] * scale
    {
        const float4* qp4 = reinterpret_cast<const float4*>(q + qrow_c * ld + head * 64) + h;
        // rope_q (self blocks at throughput shapes: the projection's epilogue has rotated K, gemm.hip): Q is rotated here, as it is loaded -- a lane's
        // float4 is two whole pairs (dims 8 g + 4 h .. + 3), (c, s) from the table row of this query; the three operations of every other rotary form
        const float4* cp4 = rope_q ? reinterpret_cast<const float4*>(rope_q + qrow_c * 64) + h : nullptr;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            float4 t = qp4[2 * g];
            if (rope_q) {
                const float4 cs = cp4[2 * g];
                t = make_float4(t.x * cs.x - t.y * cs.y, t.y * cs.x + t.x * cs.y, t.z * cs.z - t.w * cs.w, t.w * cs.z + t.z * cs.w);
            }
            qreg[4 * g] = t.x * kScale; qreg[4 * g + 1] = t.y * kScale; qreg[4 * g + 2] = t.z * kScale; qreg[4 * g + 3] = t.w * kScale;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // Q is in: from here on the vector-memory counter only sees tile copies
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    const float* kbase = k + (size_t)kvseq * Lk * ld + head * 64;
    const float* vbase = v + (size_t)kvseq * Lk * ld + head * 64;
    // copy geometry: a wave instruction moves 64 granules of 16 B = 4 rows; wave w owns rows 8 w .. 8 w + 7 of a tile (two instructions
    // for K, two for V).  Granule (row, slot') of the K image holds global chunk slot' ^ (row & 15).
    const int crow = lane >> 4, cslot = lane & 15;
    auto issue = [&](int k0, int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = wave * 8 + u * 4 + crow;
            int key = k0 + row; key = key < nk ? key : nk - 1;
            const float* ksrc = kbase + (size_t)key * ld + ((cslot ^ (row & 15)) << 2);
            const float* vsrc = vbase + (size_t)key * ld + (cslot << 2);
            __builtin_amdgcn_global_load_lds((gptr_t)ksrc, (lds_ptr_t)(Kd[buf] + (wave * 8 + u * 4) * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)vsrc, (lds_ptr_t)(Vd[buf] + (wave * 8 + u * 4) * 64), 16, 0, 0);
        }
    };
    if (nk > 0) issue(0, 0);
    int buf = 0;
    const int jsw = j & 15;
    for (int k0 = 0; k0 < nk; k0 += AD_K) {
        // tile k0 has landed (this wave's copies: the wait; everybody's: the barrier) and the other buffer is no longer read
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (k0 + AD_K < nk) issue(k0 + AD_K, buf ^ 1);
        f32x16 st;
        const bool first = k0 == 0;
        {
            const float init = at_softmax_init(first, m_run);
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = init;
        }
        const float* ka = Kd[buf] + j * 64;
        if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(ka + (((2 * g + h) ^ jsw) << 2));
#pragma unroll
            for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], qreg[4 * g + e], st, 0, 0, 0);
        }
        if (prio) __builtin_amdgcn_s_setprio(0);
        if (k0 + AD_K > nk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (key >= nk) st[r] = -INFINITY;
            }
        }
        at_softmax_step(st, first, m_run, l_run, o0, o1);
        const float* va = Vd[buf] + (4 * h) * 64 + j;
        if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = (r & 3) + 8 * (r >> 2);
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[kr * 64], st[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[kr * 64 + 32], st[r], o1, 0, 0, 0);
        }
        if (prio) __builtin_amdgcn_s_setprio(0);
        buf ^= 1;
    }
    l_run = at_softmax_finish(l_run);
    if (qrow < Lq) {
        const float inv = (qrow < nq && l_run > 0.f) ? 1.0f / l_run : 0.f;
        float* op = out + ((size_t)seq * Lq + qrow) * 256 + head * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = (r & 3) + 8 * (r >> 2) + 4 * h;
            op[d] = o0[r] * inv;
            op[d + 32] = o1[r] * inv;
        }
    }
}

// merges the key ranges of the split variant: out = sum_s o_s 2^(m_s - m) / sum_s l_s 2^(m_s - m), m = max_s m_s
__global__ __launch_bounds__(256) void lg_attention_combine_kernel(const float* __restrict__ part, int ns, int nseq, int Lq,
                                                                   const int* __restrict__ qlen, float* __restrict__ out) {
    const int gid = blockIdx.x * 256 + threadIdx.x;            // (row, head, 16 float4)
    const int d4 = gid & 15, head = (gid >> 4) & 3;
    const size_t row = (size_t)(gid >> 6);                       // seq * Lq + qrow
    if (row >= (size_t)nseq * Lq) return;
    const int seq = (int)(row / Lq), qrow = (int)(row % Lq);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qrow < (qlen ? qlen[seq] : Lq)) {
        const float* ml = part + (size_t)ns * nseq * Lq * 256;
        float m = -INFINITY;
        for (int sp = 0; sp < ns; ++sp) m = fmaxf(m, ml[(((size_t)sp * nseq * Lq + row) * 4 + head) * 2]);
        float l = 0.f;
        for (int sp = 0; sp < ns; ++sp) {
            const size_t prow = (size_t)sp * nseq * Lq + row;
            const float ms = ml[(prow * 4 + head) * 2], ls = ml[(prow * 4 + head) * 2 + 1];
            if (!(ls > 0.f)) continue;                           // empty key range
            const float w = __builtin_amdgcn_exp2f(ms - m);
            l += ls * w;
            const float4 o = *reinterpret_cast<const float4*>(part + prow * 256 + head * 64 + d4 * 4);
            acc.x += o.x * w; acc.y += o.y * w; acc.z += o.z * w; acc.w += o.w * w;
        }
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
    }
    *reinterpret_cast<float4*>(out + row * 256 + head * 64 + d4 * 4) = acc;
}

size_t lg_attention_part_bytes(int nseq, int Lq) {   // scratch of the split-key variant (0 when it would never be chosen)
    const size_t rows = (size_t)nseq * Lq;
    return rows <= AT_SPLIT_MAX_ROWS ? rows * AT_SPLIT_MAX * (256 + 8) * sizeof(float) : 0;
}

void launch_lg_attention(hipStream_t s, const float* q, const float* k, const float* v, int ld, float* out, int nseq, int Lq,
                         int Lk, const int* qlen, const int* klen, const int* kv_map, float* part, const float* rope_csn, bool fp16x2, bool k_roped) {
    if (rope_csn && k_roped) {   // K already rotated by the projection's epilogue, Q not (tuning form RFE_QKV_ROPE=2): K / V tiles by LDS-DMA, Q rotated on load
        const int nqb = (Lq + AT_Q - 1) / AT_Q, units8 = (4 * nseq + 7) / 8 * 8;
        static const int prio = tune_int("RFE_ATT_PRIO", 1);
        hipLaunchKernelGGL(lg_attention_dma_kernel, dim3(nqb * units8), dim3(256), 4 * AD_K * 64 * sizeof(float), s, q, k, v, ld, out, Lq, Lk, nqb, qlen, klen, kv_map, prio, nseq, rope_csn);
        return;
    }
    if (fp16x2 && (size_t)nseq * Lq >= 32768 && (ld % 4) == 0) {   // RFE_OPT_LG_FP16X2, throughput shapes (one / few pairs: the split form of the latency kernel below)
        launch_lg_attention_h2(s, q, k, v, ld, out, nseq, Lq, Lk, qlen, klen, kv_map, rope_csn);
        return;
    }
    // latency regime (one / few pairs, the shapes the reference itself runs), keys and queries already rotated by the projection: the
    // in-workgroup key split (lg_attention_lat.hip) -- no partial sums in HBM, no combine launch
    static const bool lat_on = tune_int("RFE_LAT", 1) != 0;
    if (lat_on && !rope_csn && (size_t)nseq * Lq <= AT_SPLIT_MAX_ROWS && launch_lg_attention_lat(s, q, k, v, ld, out, nseq, Lq, Lk, qlen, klen, kv_map, fp16x2)) return;
    const int nqb = (Lq + AT_Q - 1) / AT_Q;
    // (sequence, head) units, padded to a multiple of 8: the kernels deal their blocks round-robin over the 8 XCDs and map
    // block -> (unit, query block) by unit = (t / nqb) * 8 + xcd, which is a bijection only for a multiple of 8 units
    // (2P sequences always are; the per-frame self block of the stream mode runs on B sequences, e.g. 33)
    const int units8 = (4 * nseq + 7) / 8 * 8;
    const bool rope = rope_csn != nullptr;
    // latency regime: fewer (sequence, head, query block) units than CUs -> split the keys until the chip is covered
    static const int split_env = tune_int("RFE_ATT_SPLIT", -1);   // 0/1 = off, n = force n ranges
    if (part && (size_t)nseq * Lq <= AT_SPLIT_MAX_ROWS && split_env != 0 && split_env != 1) {
        const int units = nqb * 4 * nseq;
        int ns = 1;
        while (ns < AT_SPLIT_MAX && units * ns * 2 <= 256 && Lk / (ns * 2) >= 2 * AT_K) ns *= 2;
        if (split_env > 1) ns = split_env < AT_SPLIT_MAX ? split_env : AT_SPLIT_MAX;
        if (ns > 1) {
            static const bool pfk = tune_int("RFE_ATT_SPLIT_PF", 1) != 0;
            if (rope && pfk)
                hipLaunchKernelGGL((lg_attention_kernel<0, true, true, true>), dim3(nqb * units8, ns), dim3(256), 0, s, q, k, v, ld, out, Lq, Lk,
                                   nqb, qlen, klen, kv_map, 1, part, nseq, rope_csn);
            else if (pfk)
                hipLaunchKernelGGL((lg_attention_kernel<0, true, false, true>), dim3(nqb * units8, ns), dim3(256), 0, s, q, k, v, ld, out, Lq, Lk,
                                   nqb, qlen, klen, kv_map, 1, part, nseq, rope_csn);
            else if (rope)
                hipLaunchKernelGGL((lg_attention_kernel<0, true, true>), dim3(nqb * units8, ns), dim3(256), 0, s, q, k, v, ld, out, Lq, Lk,
                                   nqb, qlen, klen, kv_map, 1, part, nseq, rope_csn);
            else
                hipLaunchKernelGGL((lg_attention_kernel<0, true, false>), dim3(nqb * units8, ns), dim3(256), 0, s, q, k, v, ld, out, Lq, Lk,
                                   nqb, qlen, klen, kv_map, 1, part, nseq, rope_csn);
            hipLaunchKernelGGL(lg_attention_combine_kernel, dim3((unsigned)(((size_t)nseq * Lq * 64 + 255) / 256)), dim3(256), 0, s, part,
                               ns, nseq, Lq, qlen, out);
            return;
        }
    }
#ifdef RFE_TUNING
    const int abl = tune_int("RFE_DBG_ATT_ABL", 0);   // timing ablations (wrong results), tuning build only
    if (abl == 1) { hipLaunchKernelGGL((lg_attention_kernel<1>), dim3(nqb * units8), dim3(256), 0, s, q, k, v, ld, out, Lq, Lk, nqb, qlen, klen, kv_map, 0, nullptr, nseq, rope_csn); return; }
    if (abl == 2) { hipLaunchKernelGGL((lg_attention_kernel<2>), dim3(nqb * units8), dim3(256), 0, s, q, k, v, ld, out, Lq, Lk, nqb, qlen, klen, kv_map, 0, nullptr, nseq, rope_csn); return; }
    if (abl == 3) { hipLaunchKernelGGL((lg_attention_kernel<3>), dim3(nqb * units8), dim3(256), 0, s, q, k, v, ld, out, Lq, Lk, nqb, qlen, klen, kv_map, 0, nullptr, nseq, rope_csn); return; }
#endif
    static const int prio = tune_int("RFE_ATT_PRIO", 1);   // s_setprio(1) around the MFMA clusters (+0.8 %); RFE_ATT_PRIO=0 disables
    static const bool dma = tune_int("RFE_ATT_DMA", 1) != 0;   // tuning switch: 0 = register-staged tiles for the cross blocks too
    if (!rope && dma && (ld % 4) == 0)
        hipLaunchKernelGGL(lg_attention_dma_kernel, dim3(nqb * units8), dim3(256), 4 * AD_K * 64 * sizeof(float), s, q, k, v, ld, out, Lq, Lk, nqb, qlen, klen, kv_map, prio, nseq, (const float*)nullptr);
    else if (rope)
        hipLaunchKernelGGL((lg_attention_kernel<0, false, true>), dim3(nqb * units8), dim3(256), 0, s, q, k, v, ld, out, Lq, Lk, nqb, qlen, klen, kv_map, prio, nullptr, nseq, rope_csn);
    else
        hipLaunchKernelGGL((lg_attention_kernel<0, false, false>), dim3(nqb * units8), dim3(256), 0, s, q, k, v, ld, out, Lq, Lk, nqb, qlen, klen, kv_map, prio, nullptr, nseq, rope_csn);
}

// ---------------------------------------------------------------- LayerNorm(512) + GELU(erf), in place
// (the latency path of lg_ffn: one / few pairs per call.  Fusing it into ffn.3's operand path -- gemm_lat.hip, round 4 -- was measured and
// dropped: every 16-row wave tile re-evaluates the GELU of its whole 16 x 512 activation panel, 8 column workgroups per panel = 8 x the
// transcendentals, and a wave alone on its SIMD cannot hide 3200 VALU instructions behind 256 matrix instructions: 17.6 us against
// 4.5 + 7 us for this pass + the plain GEMM, profiles/r04_ab_notes.md)
__device__ __forceinline__ float gelu_short_(float t) {   // Abramowitz-Stegun 7.1.26 erf, |error| <= 1.5e-7 (== gemm.hip gelu_short)
    const float x = t * 0.70710678118654752f, ax = fabsf(x);
    const float k = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    const float poly = fmaf(fmaf(fmaf(fmaf(1.061405429f, k, -1.453152027f), k, 1.421413741f), k, -0.284496736f), k, 0.254829592f) * k;
    const float er = 1.0f - poly * __builtin_amdgcn_exp2f(-(ax * ax) * 1.44269504088896341f);
    return 0.5f * t * (1.0f + copysignf(er, x));
}
__global__ __launch_bounds__(256) void lg_ln_gelu_kernel(float* __restrict__ hbuf, const float* __restrict__ g,
                                                         const float* __restrict__ b, int64_t rows) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float4* p = reinterpret_cast<float4*>(hbuf + row * 512) + lane * 2;
    float4 a = p[0], c = p[1];
    float sum = ((a.x + a.y) + (a.z + a.w)) + ((c.x + c.y) + (c.z + c.w));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum * (1.0f / 512.0f);
    float x[8] = {a.x - mean, a.y - mean, a.z - mean, a.w - mean, c.x - mean, c.y - mean, c.z - mean, c.w - mean};
    float var = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) var = fmaf(x[e], x[e], var);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) var += __shfl_xor(var, off);
    const float rs = 1.0f / sqrtf(var * (1.0f / 512.0f) + 1e-5f);
    const float4 g0 = reinterpret_cast<const float4*>(g)[lane * 2], g1 = reinterpret_cast<const float4*>(g)[lane * 2 + 1];
    const float4 b0 = reinterpret_cast<const float4*>(b)[lane * 2], b1 = reinterpret_cast<const float4*>(b)[lane * 2 + 1];
    const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float vv = x[e] * rs * gg[e] + bb[e];
        x[e] = gelu_short_(vv);     // the GELU of the fused throughput path (gemm.hip gelu_short): one arithmetic for every LightGlue tiling
    }
    p[0] = make_float4(x[0], x[1], x[2], x[3]);
    p[1] = make_float4(x[4], x[5], x[6], x[7]);
}
void launch_lg_ln_gelu(hipStream_t s, float* h, const float* g, const float* b, int64_t rows) {
    hipLaunchKernelGGL(lg_ln_gelu_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, h, g, b, rows);
}

__device__ __forceinline__ float logsigmoid_(float z) { return z >= 0.f ? -log1pf(expf(-z)) : z - log1pf(expf(z)); }

// z[row] = logsigmoid(x[row] . w + b)   (matchability head, 256 -> 1; stored already in the log domain)
__global__ __launch_bounds__(256) void lg_matchability_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, int64_t rows,
                                                              float* __restrict__ z) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float4 a = reinterpret_cast<const float4*>(x + row * 256)[lane];
    const float4 ww = reinterpret_cast<const float4*>(w)[lane];
    float s = fmaf(a.w, ww.w, fmaf(a.z, ww.z, fmaf(a.y, ww.y, a.x * ww.x)));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) z[row] = logsigmoid_(s + b[0]);   // the assignment only ever uses log sigmoid(z)
}
void launch_lg_matchability(hipStream_t s, const float* x, const float* w, const float* b, int64_t rows, float* z) {
    hipLaunchKernelGGL(lg_matchability_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, w, b, rows, z);
}

// ---------------------------------------------------------------- assignment

// row log-sum-exp of sim[p][i][0..n) : one wave per row
__global__ __launch_bounds__(256) void lg_rowlse_kernel(const float* __restrict__ sim, int L, const int* __restrict__ m,
                                                        const int* __restrict__ n, float* __restrict__ rowlse) {
    const int p = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= m[p]) return;
    const int nn = n[p];
    const float* r = sim + ((size_t)p * L + i) * L;
    float mx = -INFINITY;
    for (int jj = lane; jj < nn; jj += 64) mx = fmaxf(mx, r[jj]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float s = 0.f;
    for (int jj = lane; jj < nn; jj += 64) s += expf(r[jj] - mx);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) rowlse[(size_t)p * L + i] = mx + logf(s);
}

// few-pair shapes: the matchability head of BOTH sides rides in the row log-sum-exp launch (blockIdx.z = 0: row i of the similarity matrix and
// token i of side 0; blockIdx.z = 1: token i of side 1) -- one 4.6 us launch fewer per forward; the arithmetic is the two kernels', wave for wave
__global__ __launch_bounds__(256) void lg_rowlse_z_kernel(const float* __restrict__ sim, int L, int P, const int* __restrict__ m, const int* __restrict__ n,
                                                          float* __restrict__ rowlse, const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ z) {
    const int p = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, side = blockIdx.z;
    if (i >= L) return;
    {   // matchability of token (side, p, i): every row of the padded layout, like lg_matchability_kernel
        const size_t row = ((size_t)side * P + p) * L + i;
        const float4 a = reinterpret_cast<const float4*>(x + row * 256)[lane];
        const float4 ww = reinterpret_cast<const float4*>(w)[lane];
        float s = fmaf(a.w, ww.w, fmaf(a.z, ww.z, fmaf(a.y, ww.y, a.x * ww.x)));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) z[row] = logsigmoid_(s + b[0]);
    }
    if (side || i >= m[p]) return;
    const int nn = n[p];
    const float* r = sim + ((size_t)p * L + i) * L;
    float mx = -INFINITY;
    for (int jj = lane; jj < nn; jj += 64) mx = fmaxf(mx, r[jj]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float s = 0.f;
    for (int jj = lane; jj < nn; jj += 64) s += expf(r[jj] - mx);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) rowlse[(size_t)p * L + i] = mx + logf(s);
}

__device__ __forceinline__ float lg_score(float sv, float lr, float lc, float l0, float l1) {
    return ((sv - lr) + (sv - lc)) + (l0 + l1);
}

// row argmax (first maximum) + optional dump of the score matrix: one wave per row
__device__ __forceinline__ void lg_rowarg_row(const float* __restrict__ sim, const float* __restrict__ z0,
                                              const float* __restrict__ z1, const float* __restrict__ rowlse,
                                              const float* __restrict__ collse, int L, const int* __restrict__ m,
                                              const int* __restrict__ n, int32_t* __restrict__ a0,
                                              float* __restrict__ mx0, float* __restrict__ scores_opt, int scores_pair) {
    const int p = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= m[p]) return;
    const int nn = n[p];
    const float* r = sim + ((size_t)p * L + i) * L;
    const float lr = rowlse[(size_t)p * L + i], l0 = z0[(size_t)p * L + i];
    float best = -INFINITY; int bi = 0x7fffffff;
    // score dump (test taps): every pair into [P,L,L] (scores_pair < 0) or one pair into [L,L]
    float* const dump = !scores_opt ? nullptr : scores_pair < 0 ? scores_opt + ((size_t)p * L + i) * L : p == scores_pair ? scores_opt + (size_t)i * L : nullptr;
    for (int jj = lane; jj < nn; jj += 64) {
        const float sc = lg_score(r[jj], lr, collse[(size_t)p * L + jj], l0, z1[(size_t)p * L + jj]);
        if (dump) dump[jj] = sc;
        if (sc > best) { best = sc; bi = jj; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { a0[(size_t)p * L + i] = bi == 0x7fffffff ? 0 : bi; mx0[(size_t)p * L + i] = best; }
}
__global__ __launch_bounds__(256) void lg_rowarg_kernel(const float* __restrict__ sim, const float* __restrict__ z0,
                                                        const float* __restrict__ z1, const float* __restrict__ rowlse,
                                                        const float* __restrict__ collse, int L, const int* __restrict__ m,
                                                        const int* __restrict__ n, int32_t* __restrict__ a0,
                                                        float* __restrict__ mx0, float* __restrict__ scores_opt, int scores_pair) {
    lg_rowarg_row(sim, z0, z1, rowlse, collse, L, m, n, a0, mx0, scores_opt, scores_pair);
}

// Column pass, fused: column log-sum-exp AND column argmax (first maximum) of the score matrix in one launch -- a workgroup owns a
// stripe of CW columns x all rows (CW * 4 B per row; 128 KB at L = 1024, CW = 32) and walks it three times (max, sum of exp, argmax
// of the scores).  With many pairs in flight the stripes do NOT stay in L2 between the walks (FETCH_SIZE = 3x the buffer once the
// counter's factor 2 is applied, calibrated in profiles/r03_fetch_calibration.md): that shape takes lg_col_lds_kernel below; this
// kernel serves the few-pair / odd-L shapes, where the whole similarity buffer fits the L2s.  Needs only rowlse; collse goes out
// for the row argmax that follows.
// <32, 8>: throughput shape (L/32 workgroups per pair).  <16, 64>: one or a few pairs -- 1024 threads, 64 workgroups per 1024
// columns, 16 rows per thread instead of 128 (single pair at K = 1024: 46 + 20 us for the two separate kernels -> one short launch).
// CACHE (round 5, the one- / few-pair shape with m <= 16 RG rows): a thread's <= 16 values of its column are read ONCE into registers (16 independent
// loads in flight instead of three dependent walks through L2) -- the arithmetic and every reduction order are those of the walking form, so the
// results are bit-identical to it; 25.6 -> ~10 us per forward at K = 1024.
template <int CW, int RG, bool CACHE = false>
__global__ __launch_bounds__(CW * RG) void lg_col_kernel(const float* __restrict__ sim, const float* __restrict__ z0,
                                                         const float* __restrict__ z1, const float* __restrict__ rowlse, int L,
                                                         const int* __restrict__ m, const int* __restrict__ n,
                                                         float* __restrict__ collse, int32_t* __restrict__ a1) {
    __shared__ float rb[RG][CW + 1];
    __shared__ int ri[RG][CW + 1];
    const int p = blockIdx.y, c = threadIdx.x % CW, rg = threadIdx.x / CW;
    const int jj = blockIdx.x * CW + c;
    const int mm = m[p], nn = n[p];
    const float* base = sim + (size_t)p * L * L;
    constexpr int NR = 16;
    float v[NR], lr[NR], l0[NR];
    if constexpr (CACHE) {
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int i = rg + u * RG;
            const bool live = jj < nn && i < mm;
            v[u] = live ? base[(size_t)i * L + jj] : -INFINITY;
            lr[u] = live ? rowlse[(size_t)p * L + i] : 0.f;
            l0[u] = live ? z0[(size_t)p * L + i] : 0.f;
        }
    }
    float mx = -INFINITY;
    if constexpr (CACHE) {
#pragma unroll
        for (int u = 0; u < NR; ++u) mx = fmaxf(mx, v[u]);
    } else {
        if (jj < nn) for (int i = rg; i < mm; i += RG) mx = fmaxf(mx, base[(size_t)i * L + jj]);
    }
    rb[rg][c] = mx;
    __syncthreads();
    float gm = rb[0][c];
#pragma unroll 8
    for (int g = 1; g < RG; ++g) gm = fmaxf(gm, rb[g][c]);
    __syncthreads();
    float sum = 0.f;
    if constexpr (CACHE) {
#pragma unroll
        for (int u = 0; u < NR; ++u) if (jj < nn && rg + u * RG < mm) sum += expf(v[u] - gm);
    } else {
        if (jj < nn) for (int i = rg; i < mm; i += RG) sum += expf(base[(size_t)i * L + jj] - gm);
    }
    rb[rg][c] = sum;
    __syncthreads();
    float t = 0.f;
#pragma unroll 8
    for (int g = 0; g < RG; ++g) t += rb[g][c];       // every thread of a column: the same sum in the same order
    const float lc = gm + logf(t);
    __syncthreads();
    float best = -INFINITY; int bi = 0x7fffffff;
    if (jj < nn) {
        if (rg == 0) collse[(size_t)p * L + jj] = lc;
        const float l1 = z1[(size_t)p * L + jj];
        if constexpr (CACHE) {
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const int i = rg + u * RG;
                if (i < mm) {
                    const float sc = lg_score(v[u], lr[u], lc, l0[u], l1);
                    if (sc > best) { best = sc; bi = i; }
                }
            }
        } else {
            for (int i = rg; i < mm; i += RG) {
                const float sc = lg_score(base[(size_t)i * L + jj], rowlse[(size_t)p * L + i], lc, z0[(size_t)p * L + i], l1);
                if (sc > best) { best = sc; bi = i; }
            }
        }
    }
    rb[rg][c] = best; ri[rg][c] = bi;
    __syncthreads();
    if (rg == 0 && jj < nn) {
        for (int g = 1; g < RG; ++g) {
            const float ob = rb[g][c]; const int oi = ri[g][c];
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        a1[(size_t)p * L + jj] = bi == 0x7fffffff ? 0 : bi;
    }
}

// The same column pass with the stripe RESIDENT IN LDS (L <= 1024, L % 32 == 0: the throughput shape).  The three-walk kernel above
// re-reads its 128 KB stripe from beyond L2 -- 1024 stripes are resident at once, 16 MB per XCD against 4 MB of L2, and rocprofv3
// counts FETCH_SIZE = 1.5-3x the similarity buffer per launch (profiles/r02_pmc.md) -- so here a workgroup of 1024 threads copies its
// 32 columns x m rows ONCE, by global_load_lds_dwordx4 (a wave instruction moves eight 128-byte row segments = 1 KB, which is also
// 1 KB of the [row][32] LDS image: a linear copy, no VGPRs), and walks LDS three times.  One workgroup per CU (128 KB + 8 KB of LDS).
// Reduction tree: 32 row groups per column (rows rg, rg + 32, ...), combined in ascending group order.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
__global__ __launch_bounds__(1024) void lg_col_lds_kernel(const float* __restrict__ sim, const float* __restrict__ z0,
                                                          const float* __restrict__ z1, const float* __restrict__ rowlse, int L,
                                                          const int* __restrict__ m, const int* __restrict__ n,
                                                          float* __restrict__ collse, int32_t* __restrict__ a1) {
    constexpr int CW = 32, RG = 32;
    __shared__ __attribute__((aligned(16))) float stripe[1024 * CW];
    __shared__ float rb[RG][CW + 1];
    __shared__ int ri[RG][CW + 1];
    __shared__ float rl[1024], zr[1024];        // rowlse / z0 of this pair: the argmax walk reads them per row (a dependent global load per row otherwise)
    const int p = blockIdx.y, tid = threadIdx.x, c = tid % CW, rg = tid / CW;
    const int j0 = blockIdx.x * CW, jj = j0 + c;
    const int mm = m[p], nn = n[p];
    if (j0 >= nn) return;                       // the whole stripe lies past this pair's keypoints (workgroup-uniform)
    if (tid < mm) { rl[tid] = rowlse[(size_t)p * L + tid]; zr[tid] = z0[(size_t)p * L + tid]; }
    const float* base = sim + (size_t)p * L * L + j0;
    {
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const float* src = base + (size_t)(lane >> 3) * L + (lane & 7) * 4;
        for (int q = wave; q * 8 < mm; q += 16)      // rows 8 q .. 8 q + 7 (rows up to L - 1 exist: L % 8 == 0)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)q * 8 * L), (lds_ptr_t)(stripe + q * 8 * CW), 16, 0, 0);
    }
    __syncthreads();                            // every wave's copies have landed (vmcnt(0) in front of the barrier)
    const float* col = stripe + c;
    float mx = -INFINITY;
    for (int i = rg; i < mm; i += RG) mx = fmaxf(mx, col[i * CW]);
    rb[rg][c] = mx;
    __syncthreads();
    float gm = rb[0][c];
#pragma unroll 8
    for (int g = 1; g < RG; ++g) gm = fmaxf(gm, rb[g][c]);
    __syncthreads();
    float sum = 0.f;
    for (int i = rg; i < mm; i += RG) sum += expf(col[i * CW] - gm);
    rb[rg][c] = sum;
    __syncthreads();
    float t = 0.f;
#pragma unroll 8
    for (int g = 0; g < RG; ++g) t += rb[g][c];       // every thread of a column: the same sum in the same order
    const float lc = gm + logf(t);
    __syncthreads();
    float best = -INFINITY; int bi = 0x7fffffff;
    if (jj < nn) {
        if (rg == 0) collse[(size_t)p * L + jj] = lc;
        const float l1 = z1[(size_t)p * L + jj];
        for (int i = rg; i < mm; i += RG) {
            const float sc = lg_score(col[i * CW], rl[i], lc, zr[i], l1);
            if (sc > best) { best = sc; bi = i; }
        }
    }
    rb[rg][c] = best; ri[rg][c] = bi;
    __syncthreads();
    if (rg == 0 && jj < nn) {
        for (int g = 1; g < RG; ++g) {
            const float ob = rb[g][c]; const int oi = ri[g][c];
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        a1[(size_t)p * L + jj] = bi == 0x7fffffff ? 0 : bi;
    }
}

// mutual check + exp + threshold + ordered compaction: one workgroup per pair
__device__ __forceinline__ void lg_mutual_body(int p, int* wave_tot, const int32_t* __restrict__ a0, const float* __restrict__ mx0,
                                               const int32_t* __restrict__ a1, int L, int cap,
                                               const int* __restrict__ m, const int* __restrict__ n, float thr,
                                               int32_t* __restrict__ S, int32_t* __restrict__ pairs,
                                               float* __restrict__ ms) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int mm = m[p], nn = n[p];
    int count = 0;
    for (int base = 0; base < mm; base += 256) {
        const int i = base + tid;
        bool ok = false; int jbest = 0; float e = 0.f;
        if (i < mm && nn > 0) {
            jbest = a0[(size_t)p * L + i];
            e = expf(mx0[(size_t)p * L + i]);
            ok = (a1[(size_t)p * L + jbest] == i) && (e > thr);
        }
        const unsigned long long bal = __ballot(ok);
        if (lane == 0) wave_tot[wv] = __popcll(bal);
        __syncthreads();
        int off = 0, tot = 0;
        for (int w = 0; w < 4; ++w) { if (w < wv) off += wave_tot[w]; tot += wave_tot[w]; }
        __syncthreads();
        if (ok) {
            const int pos = count + off + __popcll(bal & ((1ull << lane) - 1ull));
            if (pos < cap) {
                pairs[((size_t)p * cap + pos) * 2] = i; pairs[((size_t)p * cap + pos) * 2 + 1] = jbest;
                ms[(size_t)p * cap + pos] = e;
            }
        }
        count += tot;
    }
    if (tid == 0) S[p] = count < cap ? count : cap;
}

__global__ __launch_bounds__(256) void lg_mutual_kernel(const int32_t* __restrict__ a0, const float* __restrict__ mx0,
                                                        const int32_t* __restrict__ a1, int L, int cap,
                                                        const int* __restrict__ m, const int* __restrict__ n, float thr,
                                                        int32_t* __restrict__ S, int32_t* __restrict__ pairs,
                                                        float* __restrict__ ms) {
    __shared__ int wave_tot[4];
    lg_mutual_body(blockIdx.x, wave_tot, a0, mx0, a1, L, cap, m, n, thr, S, pairs, ms);
}

// few-pair shapes (one to four pairs per call, the reference's own): matchability + row log-sum-exp in one launch -- 4 launches instead of 5
// (launch_lg_matchability is then NOT called by lg_forward).  Measured and NOT kept (round 5, kernel trace of one pair): row argmax + mutual step in
// one launch, the workgroup that draws the last ticket of a pair (release fence + atomic per workgroup) running the mutual step -- 29.3 us against
// 8.4 + 5.1 us for the two kernels: the device-scope fences write the L2s back in every one of the 256 workgroups and the mutual step becomes a
// serial tail behind the slowest of them; a kernel boundary (2.4-4.3 us) is cheaper.
bool lg_assign_few_pairs(int P, int L) {
    static const bool on = tune_int("RFE_LG_ASSIGN_MERGE", 1) != 0;   // tuning build: 0 = the five separate launches
    return on && (long long)P * ((L + 31) / 32) < 128;
}
void launch_lg_assign(hipStream_t s, const float* sim, const float* z0, const float* z1, int P, int L, int cap,
                      const int* m, const int* n, float thr, float* scores_opt, float* rowlse, float* collse,
                      int32_t* a0, float* mx0, int32_t* a1, int32_t* S, int32_t* pairs, float* ms, int scores_pair,
                      const float* x, const float* wm, const float* bm, float* z) {
    if (lg_assign_few_pairs(P, L) && x) {
        hipLaunchKernelGGL(lg_rowlse_z_kernel, dim3((L + 3) / 4, P, 2), dim3(256), 0, s, sim, L, P, m, n, rowlse, x, wm, bm, z);
        if (L <= 1024) hipLaunchKernelGGL((lg_col_kernel<16, 64, true>), dim3((L + 15) / 16, P), dim3(1024), 0, s, sim, z0, z1, rowlse, L, m, n, collse, a1);   // m <= L <= 16 x 64 rows: the column in registers
        else hipLaunchKernelGGL((lg_col_kernel<16, 64>), dim3((L + 15) / 16, P), dim3(1024), 0, s, sim, z0, z1, rowlse, L, m, n, collse, a1);
        hipLaunchKernelGGL(lg_rowarg_kernel, dim3((L + 3) / 4, P), dim3(256), 0, s, sim, z0, z1, rowlse, collse, L, m, n, a0, mx0, scores_opt, scores_pair);
        hipLaunchKernelGGL(lg_mutual_kernel, dim3(P), dim3(256), 0, s, a0, mx0, a1, L, cap, m, n, thr, S, pairs, ms);
        return;
    }
    hipLaunchKernelGGL(lg_rowlse_kernel, dim3((L + 3) / 4, P), dim3(256), 0, s, sim, L, m, n, rowlse);
    static const bool col_lds = tune_int("RFE_LG_COL_LDS", 1) != 0;   // tuning switch: 0 = the three-walk kernel for every shape
    if (col_lds && L <= 1024 && L % 32 == 0 && (long long)P * (L / 32) >= 256)   // throughput shape: stripe resident in LDS, read from HBM once
        hipLaunchKernelGGL(lg_col_lds_kernel, dim3(L / 32, P), dim3(1024), 0, s, sim, z0, z1, rowlse, L, m, n, collse, a1);
    else if ((long long)P * ((L + 31) / 32) < 128)   // a few pairs: narrower stripes, 4x the threads per workgroup
        hipLaunchKernelGGL((lg_col_kernel<16, 64>), dim3((L + 15) / 16, P), dim3(1024), 0, s, sim, z0, z1, rowlse, L, m, n, collse, a1);
    else
        hipLaunchKernelGGL((lg_col_kernel<32, 8>), dim3((L + 31) / 32, P), dim3(256), 0, s, sim, z0, z1, rowlse, L, m, n, collse, a1);
    hipLaunchKernelGGL(lg_rowarg_kernel, dim3((L + 3) / 4, P), dim3(256), 0, s, sim, z0, z1, rowlse, collse, L, m, n, a0, mx0, scores_opt, scores_pair);
    hipLaunchKernelGGL(lg_mutual_kernel, dim3(P), dim3(256), 0, s, a0, mx0, a1, L, cap, m, n, thr, S, pairs, ms);
}

// ---------------------------------------------------------------- small helpers
__global__ void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void launch_copy_f32(hipStream_t s, const float* src, float* dst, int64_t n) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n);
}

// NormalizeKeypoints (reference src/Matchers/transform.cpp:19-32) on integer pixel keypoints
__global__ void normalize_kpts_kernel(const int32_t* __restrict__ kxy, int64_t n, float sx, float sy, float scale,
                                      float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[2 * i] = ((float)kxy[2 * i] - sx) / scale;
    out[2 * i + 1] = ((float)kxy[2 * i + 1] - sy) / scale;
}
void launch_normalize_kpts(hipStream_t s, const int32_t* kxy, int64_t n, int rows, int cols, float* out) {
    const float sx = (float)cols / 2, sy = (float)rows / 2, scale = (float)(rows > cols ? rows : cols) / 2;
    hipLaunchKernelGGL(normalize_kpts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, kxy, n, sx, sy, scale, out);
}

// ---------------------------------------------------------------- one-launch prologue of the stream mode (round 5: it replaced normalize_kpts +
// lg_posenc + copy_f32 + lg_setup, four 4-5 us launches in front of every one-pair forward).  One wave per token row of the B frames:
// NormalizeKeypoints (reference src/Matchers/transform.cpp:19-32) of the integer pixel keypoint -> kn, its rotary table row (the arithmetic of
// lg_posenc_kernel, operand for operand) -> csn, its descriptor -> x; workgroup 0 also clamps the lengths and writes the cross-attention map.  Requires L == Kmax (the caller's dedup path).
__global__ __launch_bounds__(256) void lg_frame_prologue_kernel(const int32_t* __restrict__ kxy, const float* __restrict__ desc, const float* __restrict__ wr,
                                                                const int32_t* __restrict__ nkp, int B, int L, float sx, float sy, float scale,
                                                                float* __restrict__ kn, float2* __restrict__ csn, float* __restrict__ x,
                                                                int32_t* __restrict__ lens, int32_t* __restrict__ kvmap) {
    const int P = B - 1;
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < 2 * P; i += 256) {
            int v = i < P ? nkp[i] : nkp[i - P + 1];
            v = v < 0 ? 0 : (v > L ? L : v);
            lens[i] = v;
            kvmap[i] = i < P ? i + P : i - P;
        }
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)B * L) return;
    const int lane = threadIdx.x & 63;
    reinterpret_cast<float4*>(x + row * 256)[lane] = reinterpret_cast<const float4*>(desc + row * 256)[lane];
    const float kx = ((float)kxy[2 * row] - sx) / scale, ky = ((float)kxy[2 * row + 1] - sy) / scale;
    if (lane == 0) reinterpret_cast<float2*>(kn)[row] = make_float2(kx, ky);
    if (lane < 32) {
        const float th = fmaf(wr[2 * lane + 1], ky, wr[2 * lane] * kx);
        csn[row * 32 + lane] = make_float2(cosf(th), sinf(th));
    }
}
void launch_lg_frame_prologue(hipStream_t s, const int32_t* kxy, const float* desc, const float* wr, const int32_t* nkp, int B, int L, int rows, int cols,
                              float* kn, float* csn, float* x, int32_t* lens, int32_t* kvmap) {
    const float sx = (float)cols / 2, sy = (float)rows / 2, scale = (float)(rows > cols ? rows : cols) / 2;
    hipLaunchKernelGGL(lg_frame_prologue_kernel, dim3((unsigned)(((int64_t)B * L + 3) / 4)), dim3(256), 0, s, kxy, desc, wr, nkp, B, L, sx, sy, scale, kn,
                       reinterpret_cast<float2*>(csn), x, lens, kvmap);
}

}  // namespace rfe
