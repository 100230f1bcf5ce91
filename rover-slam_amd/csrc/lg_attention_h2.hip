// lg_attention_h2.hip -- LightGlue's fused attention on the f16 matrix pipe by operand splitting (option RFE_OPT_LG_FP16X2, default OFF;
// the fp32 kernels of lg_kernels.hip are the product's default and the benchmark's headline).
//   softmax(Q K^T / 8) V per (sequence, head), online softmax in fp32, the L x L matrix never materialised -- as lg_attention_kernel --
// but every matrix product takes fp16 (hi, lo) operand pairs, x = x_hi + x_lo (22 of 24 significand bits, gemm_h2.hip), and three
// v_mfma_f32_32x32x16_f16 (x_lo y_hi + x_hi y_lo + x_hi y_hi, fp32 accumulation) instead of eight v_mfma_f32_32x32x2_f32 per
// 32 x 32 x 16 block: 24 matrix instructions of 32 cycles per 32 keys x 32 queries where the fp32 kernel issues 64 of 64 cycles.
// The reference runs this inside Session::Run(lightglue_sim.onnx), src/Matchers/lightglue_onnx.cpp:210-214.
//
// Workgroup = 4 waves = 256 queries of one head, wave = 64 queries (NQB = 2: two 32-query blocks b share every K / V^T fragment read and
// every staged tile serves twice the queries; the 128-query / 32-query-wave shape of the fp32 kernel -- NQB = 1, one tile buffer, three
// workgroups per CU -- measured 7 % slower, profiles/r03_ab_notes.md).  64-key tiles, double-buffered in LDS (64 KB, two
// workgroups per CU), ONE barrier per tile: tile t+1 is fetched into registers before the products of tile t and written (split into
// planes) after them.  Per tile and 32-key sub-block:
//   S^T = K . Q^T      A = K rows (hi / lo planes, [key][64 dims], 128-byte rows of eight 16-byte slots, slot c at c ^ ((key >> 1) & 7):
//                      one conflict-free ds_read_b128 per plane and 16 dims), B = the Q fragments, split once, held in 64 VGPRs;
//                      the accumulator starts at -(running reference maximum) like the fp32 kernel -> lane = one query, 16 keys in
//                      registers, softmax reduction in-lane + one xor-32 exchange (only when the reference moves);
//   P = 2^(S^T - ref)  fp32, split into (hi, lo) planes IN PLACE: accumulator registers 8 s .. 8 s + 7 of the block are the eight k
//                      values lane (query, h) owes the B operand of PV's k-step s, i.e. MFMA k index 8 h + e <-> key
//                      16 s + 8 (e >> 2) + 4 h + (e & 3): key bits 2 and 3 swapped;
//   O^T += V^T . P^T   A = V^T (hi / lo planes, [dim][64 keys] with the keys of every 16-key group stored in that swapped order, built
//                      by a 4 x 4 register transpose while staging: thread = 4 keys x 4 dims, one ds_write_b64 per dim and plane).
//                      Row d sits at physical row (d & ~1) | ((d ^ (d >> 4)) & 1), slot c at c ^ (((d & 15) ^ (d >> 4)) >> 1): the
//                      fragment reads (16 consecutive d) are bank-conflict free (the transposing 8-byte writes are not quite -- stores
//                      are served 16 lanes / 32 banks at a time: 14 % of the LDS cycles, which are 14 % of the kernel's).
// The deferred-rescale threshold is 2^11 (fp32 kernel: 2^16): P must stay below fp16's 65504.  Values below fp16's normal range
// (|x| < 6.1e-5: small P, low planes) are carried as fp16 subnormals -- absolute error <= 2^-25 per element, against a row sum >= 1.
// Operands past fp16's range saturate instead of overflowing (MODE.FP16_OVFL, h2_split.h): finite results outside the option's domain.
// Rotary (self blocks): applied to q and k with the same three fp32 operations as lg_attention_kernel, before the split.
#include "rfe_internal.h"
#include "h2_split.h"

namespace rfe {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int AH_K = 64;
constexpr float AH_DEFER = 11.0f;   // log2 units: P <= 2^11 between two moves of the reference

__device__ __forceinline__ void ah_split2(float x, float y, uint32_t& hi, uint32_t& lo) { h2_split2(x, y, hi, lo); }   // h2_split.h: 3 VALU per pair
__device__ __forceinline__ void ah_split8(const float* x, f16x8& hi, f16x8& lo) {
    u32x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { uint32_t a, b; ah_split2(x[2 * e], x[2 * e + 1], a, b); h[e] = a; l[e] = b; }
    hi = __builtin_bit_cast(f16x8, h);
    lo = __builtin_bit_cast(f16x8, l);
}

// online softmax step of one 32-key x 32-query block: lg_kernels.hip's at_softmax_step with this kernel's threshold
__device__ __forceinline__ void ah_softmax_step(f32x16& st, bool first, float& m_run, float& l_run, f32x16& o0, f32x16& o1) {
    float mx = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, st[r]), st[r + 1]);
    mx = fmaxf(mx, st[15]);
    if (__any(first || mx > AH_DEFER)) {
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float ref = first ? 0.f : m_run;
        const float m_new = fmaxf(m_run, mx + ref);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
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

// NQB: 32-query blocks per wave (workgroup = 128 NQB queries); DBUF: double-buffered tiles (one barrier per tile) or one buffer (two)
template <bool ROPE, int ABL = 0, int NQB = 2, bool DBUF = true>   // ABL: timing ablations (tuning build only, wrong results): 1 no softmax VALU, 2 stage only tile 0, 4 no MFMA
__global__ __launch_bounds__(256, NQB == 2 ? 2 : 3) void lg_attention_h2_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld, float* __restrict__ out,
    int Lq, int Lk, int nqb, const int* __restrict__ qlen, const int* __restrict__ klen, const int* __restrict__ kv_map,
    int nseq_total, const float* __restrict__ rope_csn) {
    // [buffer][K hi, K lo, V^T hi, V^T lo][64 rows x 128 B]
    constexpr int AH_Q = 128 * NQB;
    h2_saturate_mode();
    __shared__ __attribute__((aligned(16))) unsigned char lds[(DBUF ? 2 : 1) * 4 * AH_K * 128];
    // XCD-aware decode as lg_attention_kernel: the query blocks of one (sequence, head) run on one XCD
    const int Lb = blockIdx.x, xcd = Lb & 7, t_ = Lb >> 3;
    const int qb = t_ % nqb, unit = (t_ / nqb) * 8 + xcd;
    if (unit >= 4 * nseq_total) return;
    const int seq = unit >> 2, head = unit & 3;
    const int kvseq = kv_map ? kv_map[seq] : seq;
    const int nq = qlen ? qlen[seq] : Lq;
    const int nk = klen ? klen[kvseq] : Lk;
    const int tid = threadIdx.x, lane = tid & 63;
    if (qb * AH_Q >= nq || nk <= 0) {   // whole block is padding (or nothing to attend to): keep the context rows defined (zero)
        for (int e = tid; e < AH_Q * 64; e += 256) {
            const int row = qb * AH_Q + (e >> 6);
            if (row < Lq) out[((size_t)seq * Lq + row) * 256 + head * 64 + (e & 63)] = 0.f;
        }
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const bool wactive = qb * AH_Q + wave * (32 * NQB) < nq;   // wave-uniform: a wave whose 64 queries are all padding only stages tiles
    constexpr float kScale = 0.125f * 1.44269504088896341f;  // 1/sqrt(64) * log2(e): softmax in base 2, folded into Q

    // ---- Q fragments: lane (j, h) of block b holds Q[query][16 s + 8 h .. + 7], s = 0..3, as (hi, lo) planes
    f16x8 qh[NQB][4], ql[NQB][4];
#pragma unroll
    for (int b = 0; b < NQB; ++b) {
        const int qrow = qb * AH_Q + wave * (32 * NQB) + b * 32 + j;
        const size_t qrow_c = (size_t)seq * Lq + (qrow < Lq ? qrow : Lq - 1);
        const float* qp = q + qrow_c * ld + head * 64 + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float t[8];
            const float4 t0 = *reinterpret_cast<const float4*>(qp + 16 * s), t1 = *reinterpret_cast<const float4*>(qp + 16 * s + 4);
            t[0] = t0.x; t[1] = t0.y; t[2] = t0.z; t[3] = t0.w; t[4] = t1.x; t[5] = t1.y; t[6] = t1.z; t[7] = t1.w;
            if (ROPE) {   // dims 16 s + 8 h + (2 e, 2 e + 1) = pair 8 s + 4 h + e
                const float* cp = rope_csn + qrow_c * 64 + 16 * s + 8 * h;
                const float4 c0 = *reinterpret_cast<const float4*>(cp), c1 = *reinterpret_cast<const float4*>(cp + 4);
                const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = t[2 * e], bb = t[2 * e + 1], c = cs[2 * e], sn = cs[2 * e + 1];
                    t[2 * e] = a * c - bb * sn;
                    t[2 * e + 1] = bb * c + a * sn;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] *= kScale;
            ah_split8(t, qh[b][s], ql[b][s]);
        }
    }

    f32x16 o[NQB][2];
#pragma unroll
    for (int b = 0; b < NQB; ++b)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[b][db][r] = 0.f;
    float m_run[NQB], l_run[NQB];
#pragma unroll
    for (int b = 0; b < NQB; ++b) { m_run[b] = -INFINITY; l_run[b] = 0.f; }

    const float* kbase = k + (size_t)kvseq * Lk * ld + head * 64;
    const float* vbase = v + (size_t)kvseq * Lk * ld + head * 64;
    // staging geometry.  K: thread -> (key = tid / 8 + 32 it, 8 dims = one 16-byte slot per plane).  V: wave w stages the 16-key group w;
    // lane -> (dq = dim quad, p = 8-byte chunk of the group's 32-byte row segment) holds keys 4 kk .. 4 kk + 3, kk = p with its two bits swapped
    const int skey = tid >> 3, soct = tid & 7;
    const int vdq = lane & 15, vp = lane >> 4, vkk = ((vp & 1) << 1) | (vp >> 1);
    float4 rk[2][2], rv[4];
    auto fetchK = [&](int k0) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            int key = k0 + skey + 32 * it; key = key < nk ? key : nk - 1;   // rows past the end: the last valid row (finite), masked in S
            const float* kp = kbase + (size_t)key * ld + soct * 8;
            float4 a = *reinterpret_cast<const float4*>(kp), bq = *reinterpret_cast<const float4*>(kp + 4);
            if (ROPE) {
                const float* cp = rope_csn + ((size_t)kvseq * Lk + key) * 64 + soct * 8;
                const float4 c0 = *reinterpret_cast<const float4*>(cp), c1 = *reinterpret_cast<const float4*>(cp + 4);
                a = make_float4(a.x * c0.x - a.y * c0.y, a.y * c0.x + a.x * c0.y, a.z * c0.z - a.w * c0.w, a.w * c0.z + a.z * c0.w);
                bq = make_float4(bq.x * c1.x - bq.y * c1.y, bq.y * c1.x + bq.x * c1.y, bq.z * c1.z - bq.w * c1.w, bq.w * c1.z + bq.z * c1.w);
            }
            rk[it][0] = a; rk[it][1] = bq;
        }
    };
    auto fetchV = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int key = k0 + 16 * wave + 4 * vkk + i; key = key < nk ? key : nk - 1;
            rv[i] = *reinterpret_cast<const float4*>(vbase + (size_t)key * ld + vdq * 4);
        }
    };
    auto fetch = [&](int k0) { fetchK(k0); fetchV(k0); };
    auto stash = [&](int buf) {
        unsigned char* const Kh = lds + buf * (4 * AH_K * 128);
        unsigned char* const Kl = Kh + AH_K * 128;
        unsigned char* const Vh = Kh + 2 * AH_K * 128;
        unsigned char* const Vl = Kh + 3 * AH_K * 128;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = skey + 32 * it;
            const float t[8] = {rk[it][0].x, rk[it][0].y, rk[it][0].z, rk[it][0].w, rk[it][1].x, rk[it][1].y, rk[it][1].z, rk[it][1].w};
            f16x8 hi, lo;
            ah_split8(t, hi, lo);
            const int off = row * 128 + ((soct ^ ((row >> 1) & 7)) << 4);
            *reinterpret_cast<f16x8*>(Kh + off) = hi;
            *reinterpret_cast<f16x8*>(Kl + off) = lo;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int d = 4 * vdq + c;
            const int key = (d & 15) ^ (d >> 4);
            const int prow = (d & ~1) | (key & 1);
            const int off = prow * 128 + (((2 * wave + (vp >> 1)) ^ (key >> 1)) << 4) + ((vp & 1) << 3);
            const float x0 = c == 0 ? rv[0].x : c == 1 ? rv[0].y : c == 2 ? rv[0].z : rv[0].w;
            const float x1 = c == 0 ? rv[1].x : c == 1 ? rv[1].y : c == 2 ? rv[1].z : rv[1].w;
            const float x2 = c == 0 ? rv[2].x : c == 1 ? rv[2].y : c == 2 ? rv[2].z : rv[2].w;
            const float x3 = c == 0 ? rv[3].x : c == 1 ? rv[3].y : c == 2 ? rv[3].z : rv[3].w;
            uint32_t h0, l0, h1, l1;
            ah_split2(x0, x1, h0, l0);
            ah_split2(x2, x3, h1, l1);
            *reinterpret_cast<u32x2*>(Vh + off) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(Vl + off) = u32x2{l0, l1};
        }
    };

    // fragment addressing (bytes inside a plane)
    const int kfrag_sw = (j >> 1) & 7;                        // K rows 32 sub + j: (row >> 1) & 7 = (j >> 1) & 7
    int vrow_off[2], vfrag_sw[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int d = 32 * db + j, key = (d & 15) ^ (d >> 4);
        vrow_off[db] = ((d & ~1) | (key & 1)) * 128;
        vfrag_sw[db] = key >> 1;
    }

    fetch(0);
    int buf = 0;
    for (int k0 = 0; k0 < nk; k0 += AH_K) {
        if (!DBUF) __syncthreads();             // one buffer: the previous tile has been consumed by every wave
        if (!(ABL & 2) || k0 == 0) {
        stash(buf);                             // DBUF: buffer `buf` was last read for tile k0 - 128: every wave has passed the barrier of tile k0 - 64 since
        if (k0 + AH_K < nk) fetch(k0 + AH_K);   // in flight under this tile's products
        }
        __syncthreads();
        if (wactive) {
            const unsigned char* const Kh = lds + buf * (4 * AH_K * 128);
            const unsigned char* const Kl = Kh + AH_K * 128;
            const unsigned char* const Vh = Kh + 2 * AH_K * 128;
            const unsigned char* const Vl = Kh + 3 * AH_K * 128;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                if (k0 + sub * 32 >= nk) break;
                const bool first = k0 == 0 && sub == 0;
                // ---- S^T[key][query] relative to the running reference maximum, both query blocks
                f32x16 st[NQB];
#pragma unroll
                for (int b = 0; b < NQB; ++b) {
                    const float init = first ? 0.f : -m_run[b];
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[b][r] = init;
                }
                const int krow = (sub * 32 + j) * 128;
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int off = krow + (((2 * s + h) ^ kfrag_sw) << 4);
                    const f16x8 kh = *reinterpret_cast<const f16x8*>(Kh + off);
                    const f16x8 kl = *reinterpret_cast<const f16x8*>(Kl + off);
                    // small terms first; the two query blocks alternate so that no instruction waits for its predecessor's accumulator
                    if (ABL & 4) { st[0][s] += (float)kh[0] + (float)kl[1]; continue; }
#pragma unroll
                    for (int b = 0; b < NQB; ++b) st[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[b][s], st[b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < NQB; ++b) st[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[b][s], st[b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < NQB; ++b) st[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[b][s], st[b], 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(0);
                // ---- online softmax, then P as (hi, lo) planes: registers 8 s .. 8 s + 7 are the B operand of PV's k-step s
                f16x8 ph[NQB][2], pl[NQB][2];
#pragma unroll
                for (int b = 0; b < NQB; ++b) {
                    if (k0 + sub * 32 + 32 > nk) {   // only the last key block can contain keys >= nk
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = k0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            if (key >= nk) st[b][r] = -INFINITY;
                        }
                    }
                    if (ABL & 1) {
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            ph[b][s] = __builtin_bit_cast(f16x8, f32x4{st[b][8 * s], st[b][8 * s + 1], st[b][8 * s + 2], st[b][8 * s + 3]});
                            pl[b][s] = __builtin_bit_cast(f16x8, f32x4{st[b][8 * s + 4], st[b][8 * s + 5], st[b][8 * s + 6], st[b][8 * s + 7]});
                        }
                        l_run[b] = 1.f; m_run[b] = 0.f;
                        continue;
                    }
                    ah_softmax_step(st[b], first, m_run[b], l_run[b], o[b][0], o[b][1]);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        float t[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) t[e] = st[b][8 * s + e];
                        ah_split8(t, ph[b][s], pl[b][s]);
                    }
                }
                // ---- O^T[d][query] += sum_key V[key][d] P[key][query]
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const int off = vrow_off[db] + (((2 * (2 * sub + s) + h) ^ vfrag_sw[db]) << 4);
                        const f16x8 vh = *reinterpret_cast<const f16x8*>(Vh + off);
                        const f16x8 vl = *reinterpret_cast<const f16x8*>(Vl + off);
                        if (ABL & 4) { o[0][db][s] += (float)vh[0] + (float)vl[1] + (float)ph[0][s][1] + (float)pl[NQB - 1][s][2] + (float)ph[NQB - 1][s][3] + (float)pl[0][s][0]; continue; }
#pragma unroll
                        for (int b = 0; b < NQB; ++b) o[b][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[b][s], o[b][db], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < NQB; ++b) o[b][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[b][s], o[b][db], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < NQB; ++b) o[b][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[b][s], o[b][db], 0, 0, 0);
                    }
                __builtin_amdgcn_s_setprio(0);
            }
        }
        if (DBUF) buf ^= 1;
    }
    if (!wactive) {   // all 64 queries of this wave are padding: zero context rows
        for (int e = lane; e < 32 * NQB * 64; e += 64) {
            const int row = qb * AH_Q + wave * (32 * NQB) + (e >> 6);
            if (row < Lq) out[((size_t)seq * Lq + row) * 256 + head * 64 + (e & 63)] = 0.f;
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < NQB; ++b) {
        const int qrow = qb * AH_Q + wave * (32 * NQB) + b * 32 + j;
        const float l = l_run[b] + __shfl_xor(l_run[b], 32);
        if (qrow < Lq) {
            const float inv = (qrow < nq && l > 0.f) ? 1.0f / l : 0.f;  // padded rows -> 0
            float* op = out + ((size_t)seq * Lq + qrow) * 256 + head * 64;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = (r & 3) + 8 * (r >> 2) + 4 * h;
                op[d] = o[b][0][r] * inv;
                op[d + 32] = o[b][1][r] * inv;
            }
        }
    }
}

// Same contract as launch_lg_attention's throughput path (lg_kernels.hip); called by it when the option is on and the problem is large.
void launch_lg_attention_h2(hipStream_t s, const float* q, const float* k, const float* v, int ld, float* out, int nseq, int Lq, int Lk,
                            const int* qlen, const int* klen, const int* kv_map, const float* rope_csn) {
    static const int nqb_sel = tune_int("RFE_AH_NQB", 2);   // tuning switch: 1 = 128-query workgroups of 32-query waves, one tile buffer, three workgroups per CU
    const int AH_Q = 128 * (nqb_sel == 1 ? 1 : 2);
    const int nqb = (Lq + AH_Q - 1) / AH_Q;
    const int units8 = (4 * nseq + 7) / 8 * 8;
#ifdef RFE_TUNING
    switch (tune_int("RFE_DBG_AH_ABL", 0)) {   // timing ablations (cross variant), tuning build only: profiles/r03_ab_notes.md
#define RFE_AH_ABL(n) case n: hipLaunchKernelGGL((lg_attention_h2_kernel<false, n>), dim3(nqb * units8), dim3(256), 0, s, q, k, v, ld, out, Lq, Lk, nqb, qlen, klen, kv_map, nseq, rope_csn); return;
        RFE_AH_ABL(1) RFE_AH_ABL(2) RFE_AH_ABL(3) RFE_AH_ABL(4) RFE_AH_ABL(6)
#undef RFE_AH_ABL
        default: break;
    }
#endif
    if (nqb_sel == 1) {
        if (rope_csn) hipLaunchKernelGGL((lg_attention_h2_kernel<true, 0, 1, false>), dim3(nqb * units8), dim3(256), 0, s, q, k, v, ld, out, Lq, Lk, nqb, qlen, klen, kv_map, nseq, rope_csn);
        else hipLaunchKernelGGL((lg_attention_h2_kernel<false, 0, 1, false>), dim3(nqb * units8), dim3(256), 0, s, q, k, v, ld, out, Lq, Lk, nqb, qlen, klen, kv_map, nseq, rope_csn);
        return;
    }
    if (rope_csn)
        hipLaunchKernelGGL((lg_attention_h2_kernel<true>), dim3(nqb * units8), dim3(256), 0, s, q, k, v, ld, out, Lq, Lk, nqb, qlen, klen, kv_map, nseq, rope_csn);
    else
        hipLaunchKernelGGL((lg_attention_h2_kernel<false>), dim3(nqb * units8), dim3(256), 0, s, q, k, v, ld, out, Lq, Lk, nqb, qlen, klen, kv_map, nseq, rope_csn);
}

}  // namespace rfe
