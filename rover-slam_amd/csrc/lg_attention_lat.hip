// lg_attention_lat.hip -- the LATENCY form of LightGlue's fused attention, softmax(Q K^T / 8) V per (sequence, head), for the shapes the
// reference itself runs: ONE pair per call (src/Matchers/lightglue_onnx.cpp:168-172) = 2 sequences x 4 heads x <= 1024 queries.
//
// lg_attention_kernel<SPLIT> covered that regime with 64 (sequence, head, 128-query block) units x 4 key ranges = 256 workgroups whose
// unnormalised partial sums a second kernel merged: four waves shared every K/V tile through a single-buffered LDS tile (two barriers
// per 64 keys, register staging), every lane wrote 32 scalar partials, 8 MB of partials crossed HBM twice and the merge was a launch of
// its own -- 28-32 + 5 us against 13.7 us of matrix time.  Here the key split happens INSIDE the workgroup:
//   * workgroup = (sequence, head, 32-query block): 2 x 4 x 32 = 256 workgroups for a pair, NW = 4 waves, one per SIMD (NW = 8, two per
//     SIMD with an eighth of the keys each, exists in the tuning build: 14 % SLOWER -- two waves' dependent S^T chains interleave on the
//     matrix pipe and lose the accumulator forwarding exactly like a foreign instruction inside one chain, profiles/r04_ab_notes.md);
//   * wave w owns the w-th NW-th of the keys and the same 32 queries: nothing is shared in the main loop, so it has NO barrier at all --
//     every wave streams its own 32-key K / V tiles by global_load_lds_dwordx4 into a private LDS tile pair (16 KB per wave); a tile is
//     read into registers in one go (8 K fragments, 32 V values), so the SAME buffer takes the copy of tile t + 1 while tile t is computed;
//   * arithmetic per tile exactly as lg_attention_dma_kernel (S^T = K Q^T with the K image XOR-swizzled on the source address: one
//     conflict-free ds_read_b128 per four k-steps; base-2 online softmax with deferred rescale; the S^T accumulators ARE the B operand of
//     O^T += V^T P^T);
//   * the four (O, m, l) partials of a query are merged through LDS (the tile buffers, free by then) and the normalised context is
//     written ONCE with 16-byte stores -- no partials in HBM, no second launch.
// The rotary encoding is NOT applied here: the one-pair qkv projection (gemm_lat.hip, ROPE epilogue) has already rotated q and k.
// Roofline: fp32 MFMA peak; algorithmic 4 heads * 4 * 64 * sum n_q n_k FLOP.
#include "rfe_internal.h"
#include "h2_split.h"

namespace rfe {

typedef _Float16 alat_f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t alat_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* alat_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* alat_gptr_t;

constexpr int AL_K = 32;                         // keys per tile
constexpr int AL_TILE_F = AL_K * 64;             // floats of one K (or V) tile
constexpr int AL_WAVE_F = 2 * AL_TILE_F;         // per wave: one K tile | one V tile
constexpr float AL_DEFER = 16.0f;                // log2 units (lg_kernels.hip: AT_DEFER)
constexpr float AL_DEFER_H2 = 11.0f;             // split form: P = 2^(S - ref) must stay below fp16's 65504 (lg_attention_h2.hip)

// eight floats -> fp16 hi / lo operand registers of a v_mfma_f32_32x32x16_f16 (h2_split.h)
__device__ __forceinline__ void alat_split8(const float (&t)[8], alat_f16x8& hi, alat_f16x8& lo) {
    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
    h2_split2(t[0], t[1], h0, l0); h2_split2(t[2], t[3], h1, l1); h2_split2(t[4], t[5], h2, l2); h2_split2(t[6], t[7], h3, l3);
    hi = __builtin_bit_cast(alat_f16x8, alat_u32x4{h0, h1, h2, h3});
    lo = __builtin_bit_cast(alat_f16x8, alat_u32x4{l0, l1, l2, l3});
}

#ifdef RFE_TUNING
__device__ unsigned long long rfe_dbg_ts_att[2048 * 8];   // in-kernel timeline (tuning build, abl & 4), see gemm_lat.hip
#define RFE_ATS(slot) do { if ((abl & 4) && tid == 0 && blockIdx.x < 2048) { rfe_dbg_ts_att[blockIdx.x * 8 + (slot)] = clock64(); rfe_dbg_ts_att[blockIdx.x * 8 + 4 + (slot)] = wall_clock64(); } } while (0)
#else
#define RFE_ATS(slot) do { } while (0)
#endif

// H2 (RFE_OPT_LG_FP16X2, default off): both products of a tile as SPLIT products on the f16 matrix pipe -- K, Q, V and P = 2^(S - ref) as fp16
// hi + lo (h2_split.h), three v_mfma_f32_32x32x16_f16 per 32 x 32 x 16 block: 24 matrix instructions of 32 cycles per 32-key tile instead of 64
// of 64.  Same tiles, copies, softmax and merge; the accumulator registers 8 s .. 8 s + 7 of S^T ARE the eight keys a lane owes the B operand
// of PV's k-step s, and the V values are read in exactly that key order, so P never moves (lg_attention_h2.hip).
template <int NW, bool H2 = false>
__global__ __launch_bounds__(64 * NW, 1) void lg_attention_lat_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld, float* __restrict__ out,
    int Lq, int Lk, int nqb, const int* __restrict__ qlen, const int* __restrict__ klen, const int* __restrict__ kv_map, int nseq_total, int abl) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // NW waves x AL_WAVE_F floats (16 KB each); the merge needs NW x 32 x 68 floats
    constexpr int NT_ = 64 * NW;
    // all query blocks of one (sequence, head) on one XCD: its K / V (512 KB) is fetched into one L2
    const int Lb = blockIdx.x, xcd = Lb & 7, t_ = Lb >> 3;
    const int qb = t_ % nqb, unit = (t_ / nqb) * 8 + xcd;
    if (unit >= 4 * nseq_total) return;
    const int seq = unit >> 2, head = unit & 3;
    const int kvseq = kv_map ? kv_map[seq] : seq;
    const int nq = qlen ? qlen[seq] : Lq;
    const int nk = klen ? klen[kvseq] : Lk;
    const int tid = threadIdx.x, lane = tid & 63;
    RFE_ATS(0);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    if (qb * 32 >= nq || nk <= 0) {   // whole block is padding (or nothing to attend to): the padded context rows stay defined (zero)
        for (int e = tid; e < 32 * 16; e += NT_) {
            const int row = qb * 32 + (e >> 4);
            if (row < Lq) *reinterpret_cast<f32x4*>(out + ((size_t)seq * Lq + row) * 256 + head * 64 + (e & 15) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    const int qrow = qb * 32 + j;    // this lane's query (may be >= nq: computed, stored as 0)
    const size_t qrow_c = (size_t)seq * Lq + (qrow < Lq ? qrow : Lq - 1);
    constexpr float kScale = 0.125f * 1.44269504088896341f;   // 1 / sqrt(64) * log2(e): softmax in base 2, folded into Q
    float qreg[32];   // qreg[4 g + e] = Q[query][8 g + 4 h + e] * scale  (the k order of the swizzled K image, lg_attention_dma_kernel)
    alat_f16x8 qh[4], ql[4];   // H2: k-step s holds head dimensions 16 s + 8 h .. + 7 of this lane's query, as fp16 hi / lo
    if constexpr (H2) {
        h2_saturate_mode();
        const f32x4* qp4 = reinterpret_cast<const f32x4*>(q + qrow_c * ld + head * 64);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const f32x4 t0 = qp4[4 * s4 + 2 * h], t1 = qp4[4 * s4 + 2 * h + 1];
            const float t[8] = {t0[0] * kScale, t0[1] * kScale, t0[2] * kScale, t0[3] * kScale, t1[0] * kScale, t1[1] * kScale, t1[2] * kScale, t1[3] * kScale};
            alat_split8(t, qh[s4], ql[s4]);
        }
    } else {
        const f32x4* qp4 = reinterpret_cast<const f32x4*>(q + qrow_c * ld + head * 64) + h;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 t = qp4[2 * g];
            qreg[4 * g] = t[0] * kScale; qreg[4 * g + 1] = t[1] * kScale; qreg[4 * g + 2] = t[2] * kScale; qreg[4 * g + 3] = t[3] * kScale;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // from here on the vector-memory counter only sees this wave's tile copies

    // this wave's key range: the w-th NW-th, whole 32-key tiles
    const int per = ((nk + NW - 1) / NW + AL_K - 1) / AL_K * AL_K;
    const int kbeg = wave * per;
    const int kend = kbeg + per < nk ? kbeg + per : nk;
    const int ntile = kend > kbeg ? (kend - kbeg + AL_K - 1) / AL_K : 0;
    const float* kbase = k + (size_t)kvseq * Lk * ld + head * 64;
    const float* vbase = v + (size_t)kvseq * Lk * ld + head * 64;
    float* const wl = lds + wave * AL_WAVE_F;          // K tile [32 x 64] | V tile [32 x 64]
    float* const Kr = wl;
    float* const Vr = wl + AL_TILE_F;
    // copy geometry: one wave instruction moves 64 granules of 16 B = 4 rows of a tile, 8 instructions per K (or V) tile.  Granule
    // (row, slot') of the K image holds global chunk slot' ^ (row & 15).  Per-lane 32-bit offsets from a wave-uniform tile base, so
    // that a copy is one instruction with a scalar base (the 64-bit per-lane address arithmetic used to cost as much as the softmax)
    const int crow = lane >> 4, cslot = lane & 15;
    int koff[8], voff[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int row = u * 4 + crow;
        koff[u] = row * ld + ((cslot ^ (row & 15)) << 2);
        voff[u] = row * ld + (cslot << 2);
    }
    auto issue_k = [&](int t) {
        const int k0 = kbeg + t * AL_K;
        const float* base = kbase + (size_t)k0 * ld;
        if (k0 + AL_K <= nk) {
#pragma unroll
            for (int u = 0; u < 8; ++u) __builtin_amdgcn_global_load_lds((alat_gptr_t)(base + koff[u]), (alat_lds_ptr_t)(Kr + u * 256), 16, 0, 0);
        } else {   // the sequence's last, partial tile: keys past the end read the last valid row (finite) and are masked in S
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = u * 4 + crow;
                const int rc = k0 + row < nk ? row : nk - 1 - k0;
                __builtin_amdgcn_global_load_lds((alat_gptr_t)(base + rc * ld + ((cslot ^ (row & 15)) << 2)), (alat_lds_ptr_t)(Kr + u * 256), 16, 0, 0);
            }
        }
    };
    auto issue_v = [&](int t) {
        const int k0 = kbeg + t * AL_K;
        const float* base = vbase + (size_t)k0 * ld;
        if (k0 + AL_K <= nk) {
#pragma unroll
            for (int u = 0; u < 8; ++u) __builtin_amdgcn_global_load_lds((alat_gptr_t)(base + voff[u]), (alat_lds_ptr_t)(Vr + u * 256), 16, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = u * 4 + crow;
                const int rc = k0 + row < nk ? row : nk - 1 - k0;
                __builtin_amdgcn_global_load_lds((alat_gptr_t)(base + rc * ld + (cslot << 2)), (alat_lds_ptr_t)(Vr + u * 256), 16, 0, 0);
            }
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;   // running maximum (log2 domain) and per-half-wave partial row sum
    const int jsw = j & 15;

    // Loop order per 32-key tile (measured alternatives in profiles/r04_ab_notes.md):
    //   wait for tile t's copies -> the eight K fragments of S^T and the 32 V values of PV into registers -> request tile t + 1 into the
    //   SAME buffer (16 copies, one scalar-base instruction each; they land under this tile's 64 matrix instructions) ->
    //   32 matrix instructions of S^T = K Q^T STRICTLY back to back -> softmax -> 32 matrix instructions of O^T += V^T P^T on two
    //   alternating accumulators.
    // S^T is ONE dependent accumulator chain: any instruction between two of its matrix instructions breaks the accumulator forwarding
    // (+43 cycles per gap, guide "one EXTRA issue slot ... on the SAME accumulator"), so nothing is interleaved there -- a version that
    // pipelined QK(t + 1) against the exponentials of tile t was 8 % SLOWER for exactly that reason; with one wave per SIMD the matrix
    // pipe itself sustains ~82 cycles per v_mfma_f32_32x32x2_f32 on random data (a pure-MFMA ablation of this loop).
    if (ntile > 0) { issue_k(0); issue_v(0); }
    RFE_ATS(1);
    for (int t = 0; t < ntile; ++t) {
        const bool more = t + 1 < ntile && !(abl & 1);   // abl (tuning build, wrong results): 1 = only the first tile is copied
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float* const Kt = Kr;
        const float* const Vt = Vr;
        f32x4 kf[8];   // fp32: granule 2 g + h of key row j; H2: granules 4 s + 2 h, + 1 (k-step s = kf[2 s], kf[2 s + 1])
        {
            const float* ka = Kt + j * 64;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int gran = H2 ? 4 * (g >> 1) + 2 * h + (g & 1) : 2 * g + h;
                kf[g] = *reinterpret_cast<const f32x4*>(ka + ((gran ^ jsw) << 2));
            }
        }
        float vf[32];
        {
            const float* va = Vt + (4 * h) * 64 + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = (r & 3) + 8 * (r >> 2);       // k-step r of PV uses key (r & 3) + 8 (r >> 2) + 4 h
                vf[r] = va[kr * 64]; vf[16 + r] = va[kr * 64 + 32];
            }
        }
        // the tile is in registers: its buffer is free for the next one (the reads must have RETURNED before a copy may overwrite them)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (more) { issue_k(t + 1); issue_v(t + 1); }
        __builtin_amdgcn_sched_barrier(0);
        // ---- S^T[key][query] = sum_d K[key][d] Q[query][d], started at -(running maximum) so that no subtraction is needed per element
        f32x16 st;
        const bool first = t == 0;
        {
            const float init = first ? 0.f : -m_run;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = init;
        }
        if constexpr (H2) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float t[8] = {kf[2 * s4][0], kf[2 * s4][1], kf[2 * s4][2], kf[2 * s4][3], kf[2 * s4 + 1][0], kf[2 * s4 + 1][1], kf[2 * s4 + 1][2], kf[2 * s4 + 1][3]};
                alat_f16x8 kh, kl;
                alat_split8(t, kh, kl);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s4], st, 0, 0, 0);   // small terms first
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s4], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s4], st, 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[g][e], qreg[4 * g + e], st, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int k0 = kbeg + t * AL_K;
        if (k0 + AL_K > nk) {      // only the last tile of the sequence can hold keys >= nk
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (key >= nk) st[r] = -INFINITY;
            }
        }
        // ---- online softmax over this lane's 16 keys (the other half-wave holds the other 16): lg_kernels.hip at_softmax_step
        {
            float mx = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, st[r]), st[r + 1]);
            mx = fmaxf(mx, st[15]);
            if (__any(first || mx > (H2 ? AL_DEFER_H2 : AL_DEFER))) {
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float ref = first ? 0.f : m_run;
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
        // ---- O^T[d][query] += sum_key V[key][d] P[key][query]; k-step r uses key (r & 3) + 8 (r >> 2) + 4 h
        if constexpr (H2) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {      // k-step s2: keys 16 s2 + 8 (e >> 2) + 4 h + (e & 3) = accumulator registers 8 s2 + e
                alat_f16x8 ph, pl, v0h, v0l, v1h, v1l;
                {
                    const float t[8] = {st[8 * s2], st[8 * s2 + 1], st[8 * s2 + 2], st[8 * s2 + 3], st[8 * s2 + 4], st[8 * s2 + 5], st[8 * s2 + 6], st[8 * s2 + 7]};
                    alat_split8(t, ph, pl);
                }
                {
                    const float t[8] = {vf[8 * s2], vf[8 * s2 + 1], vf[8 * s2 + 2], vf[8 * s2 + 3], vf[8 * s2 + 4], vf[8 * s2 + 5], vf[8 * s2 + 6], vf[8 * s2 + 7]};
                    alat_split8(t, v0h, v0l);
                }
                {
                    const float t[8] = {vf[16 + 8 * s2], vf[17 + 8 * s2], vf[18 + 8 * s2], vf[19 + 8 * s2], vf[20 + 8 * s2], vf[21 + 8 * s2], vf[22 + 8 * s2], vf[23 + 8 * s2]};
                    alat_split8(t, v1h, v1l);
                }
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0l, ph, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1l, ph, o1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h, pl, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h, pl, o1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h, ph, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h, ph, o1, 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], st[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[16 + r], st[r], o1, 0, 0, 0);
        }
        }
    }
    l_run += __shfl_xor(l_run, 32);          // the two half-waves' partial sums
    RFE_ATS(2);

    // ---- merge the NW key ranges through LDS: out = sum_w o_w 2^(m_w - m) / sum_w l_w 2^(m_w - m), m = max_w m_w.
    // Layout: part[w][query j][68]: 64 context values + (m, l), row stride 68 floats = 272 B (16-byte aligned, conflict-light)
    __syncthreads();                          // every wave is done with its tiles: the region is reused
    float* const part = lds;
    {
        float* pr = part + (wave * 32 + j) * 68;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {      // accumulator registers 4 rq .. 4 rq + 3 are dims 8 rq + 4 h .. + 3
            *reinterpret_cast<f32x4*>(pr + 8 * rq + 4 * h) = f32x4{o0[4 * rq], o0[4 * rq + 1], o0[4 * rq + 2], o0[4 * rq + 3]};
            *reinterpret_cast<f32x4*>(pr + 32 + 8 * rq + 4 * h) = f32x4{o1[4 * rq], o1[4 * rq + 1], o1[4 * rq + 2], o1[4 * rq + 3]};
        }
        if (h == 0) { pr[64] = m_run; pr[65] = l_run; }
    }
    __syncthreads();
    // thread -> (query, DPT consecutive dims): 32 queries x 64 dims over 64 NW threads
    {
        constexpr int TPQ = NT_ / 32, DPT = 64 / TPQ;            // threads per query (8 / 16), dims per thread (8 / 4)
        const int qj = tid / TPQ, dq = (tid % TPQ) * DPT;
        const int row = qb * 32 + qj;
        float mw[NW], lw[NW], m = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) { mw[w] = part[(w * 32 + qj) * 68 + 64]; lw[w] = part[(w * 32 + qj) * 68 + 65]; m = fmaxf(m, mw[w]); }
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        float l = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            if (!(lw[w] > 0.f)) continue;     // empty key range
            const float wgt = __builtin_amdgcn_exp2f(mw[w] - m);
            l += lw[w] * wgt;
            const float* pr = part + (w * 32 + qj) * 68 + dq;
            a0 += *reinterpret_cast<const f32x4*>(pr) * wgt;
            if (DPT == 8) a1 += *reinterpret_cast<const f32x4*>(pr + 4) * wgt;
        }
        const float inv = (row < nq && l > 0.f) ? 1.0f / l : 0.f;    // padded rows -> 0
        if (row < Lq) {
            float* op = out + ((size_t)seq * Lq + row) * 256 + head * 64 + dq;
            *reinterpret_cast<f32x4*>(op) = a0 * inv;
            if (DPT == 8) *reinterpret_cast<f32x4*>(op + 4) = a1 * inv;
        }
    }
    RFE_ATS(3);
}


#ifdef RFE_TUNING   // measured and not adopted (profiles/r06_ab_notes.md 2e): the tuning build keeps it for A/B runs, the product library does not contain it
// ------------------------------------------------------------------------------------------------------------------------------------------------
// PIPE form (round 6, fp32): the same tiles, arithmetic and merge with the NEXT tile's operands read from LDS in the shadow of this tile's PV products.
// In lg_attention_lat_kernel a wave reads the eight K fragments and 32 V values of tile t, waits for them, and only then starts tile t's 64 matrix
// instructions: with one wave per SIMD nobody covers those 40 LDS round trips (-3 700 of 51 700 cycles when they are ablated, profiles/r04_ab_notes.md).
// The Sᵀ chain must stay 32 dependent matrix instructions strictly back to back (anything between two of them costs the accumulator forwarding), but the
// PV products alternate between TWO accumulators: an instruction between o0's and o1's costs nothing.  So:
//   * two tile buffers per wave (2 x 16 KB; 128 KB per workgroup): tile t + 1 -- copied two tiles ahead -- is read by ds_reads issued BETWEEN the PV matrix
//     instructions of tile t (two or three per o0 / o1 pair) INTO THE REGISTERS THOSE INSTRUCTIONS HAVE JUST CONSUMED (the K fragments died with the Sᵀ
//     chain, a V value dies with its two products: no second register set -- a first form with two sets needed 340 registers and paid for them in
//     accumulator-file copies), waited for once, after the last of them;
//   * the copy of tile t + 3 then goes into the buffer tile t + 1 has just left.
// Everything else (block decode, copy geometry and swizzle, online softmax with deferred rescale, merge through LDS) is the kernel above, line for line.
template <int NW>
__global__ __launch_bounds__(64 * NW, 1) void lg_attention_lat_pipe_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld, float* __restrict__ out,
    int Lq, int Lk, int nqb, const int* __restrict__ qlen, const int* __restrict__ klen, const int* __restrict__ kv_map, int nseq_total) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // NW waves x 2 x AL_WAVE_F floats; the merge needs NW x 32 x 68 floats
    constexpr int NT_ = 64 * NW;
    const int Lb = blockIdx.x, xcd = Lb & 7, t_ = Lb >> 3;
    const int qb = t_ % nqb, unit = (t_ / nqb) * 8 + xcd;
    if (unit >= 4 * nseq_total) return;
    const int seq = unit >> 2, head = unit & 3;
    const int kvseq = kv_map ? kv_map[seq] : seq;
    const int nq = qlen ? qlen[seq] : Lq;
    const int nk = klen ? klen[kvseq] : Lk;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    if (qb * 32 >= nq || nk <= 0) {
        for (int e = tid; e < 32 * 16; e += NT_) {
            const int row = qb * 32 + (e >> 4);
            if (row < Lq) *reinterpret_cast<f32x4*>(out + ((size_t)seq * Lq + row) * 256 + head * 64 + (e & 15) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    const int qrow = qb * 32 + j;
    const size_t qrow_c = (size_t)seq * Lq + (qrow < Lq ? qrow : Lq - 1);
    constexpr float kScale = 0.125f * 1.44269504088896341f;
    float qreg[32];
    {
        const f32x4* qp4 = reinterpret_cast<const f32x4*>(q + qrow_c * ld + head * 64) + h;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 t = qp4[2 * g];
            qreg[4 * g] = t[0] * kScale; qreg[4 * g + 1] = t[1] * kScale; qreg[4 * g + 2] = t[2] * kScale; qreg[4 * g + 3] = t[3] * kScale;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // from here on the vector-memory counter only sees this wave's tile copies

    const int per = ((nk + NW - 1) / NW + AL_K - 1) / AL_K * AL_K;
    const int kbeg = wave * per;
    const int kend = kbeg + per < nk ? kbeg + per : nk;
    const int ntile = kend > kbeg ? (kend - kbeg + AL_K - 1) / AL_K : 0;
    const float* kbase = k + (size_t)kvseq * Lk * ld + head * 64;
    const float* vbase = v + (size_t)kvseq * Lk * ld + head * 64;
    float* const wl = lds + wave * (2 * AL_WAVE_F);     // buffer b: K tile at wl + b * AL_WAVE_F, V tile behind it
    const int crow = lane >> 4, cslot = lane & 15;
    int koff[8], voff[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int row = u * 4 + crow;
        koff[u] = row * ld + ((cslot ^ (row & 15)) << 2);
        voff[u] = row * ld + (cslot << 2);
    }
    // 16 copy instructions per tile (8 K + 8 V), always: the counted waits below rely on it
    auto issue = [&](int t, int b) {
        const int k0 = kbeg + t * AL_K;
        const float* kb = kbase + (size_t)k0 * ld;
        const float* vb = vbase + (size_t)k0 * ld;
        float* Kd = wl + b * AL_WAVE_F;
        float* Vd = Kd + AL_TILE_F;
        if (k0 + AL_K <= nk) {
#pragma unroll
            for (int u = 0; u < 8; ++u) __builtin_amdgcn_global_load_lds((alat_gptr_t)(kb + koff[u]), (alat_lds_ptr_t)(Kd + u * 256), 16, 0, 0);
#pragma unroll
            for (int u = 0; u < 8; ++u) __builtin_amdgcn_global_load_lds((alat_gptr_t)(vb + voff[u]), (alat_lds_ptr_t)(Vd + u * 256), 16, 0, 0);
        } else {   // the sequence's last, partial tile: keys past the end read the last valid row (finite) and are masked in S
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = u * 4 + crow;
                const int rc = k0 + row < nk ? row : nk - 1 - k0;
                __builtin_amdgcn_global_load_lds((alat_gptr_t)(kb + rc * ld + ((cslot ^ (row & 15)) << 2)), (alat_lds_ptr_t)(Kd + u * 256), 16, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = u * 4 + crow;
                const int rc = k0 + row < nk ? row : nk - 1 - k0;
                __builtin_amdgcn_global_load_lds((alat_gptr_t)(vb + rc * ld + (cslot << 2)), (alat_lds_ptr_t)(Vd + u * 256), 16, 0, 0);
            }
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const int jsw = j & 15;
    f32x4 kf[8];
    float vf[32];
    // LDS addresses of this lane's fragments inside a buffer (relative to the buffer's K tile)
    const float* const ka0 = wl + j * 64;                           // + ((2 g + h) ^ jsw) * 4
    const float* const va0 = wl + AL_TILE_F + (4 * h) * 64 + j;     // + kr * 64 (+ 32)

    if (ntile > 0) {
        issue(0, 0);
        if (ntile > 1) issue(1, 1);
        if (ntile > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < 8; ++g) kf[g] = *reinterpret_cast<const f32x4*>(ka0 + (((2 * g + h) ^ jsw) << 2));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = (r & 3) + 8 * (r >> 2);
            vf[r] = va0[kr * 64]; vf[16 + r] = va0[kr * 64 + 32];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (ntile > 2) issue(2, 0);
        __builtin_amdgcn_sched_barrier(0);
    }

    for (int t = 0; t < ntile; ++t) {
        const int NXT = (t + 1) & 1;
        // ---- S^T = K Q^T, 32 dependent matrix instructions back to back
        f32x16 st;
        const bool first = t == 0;
        {
            const float init = first ? 0.f : -m_run;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = init;
        }
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[g][e], qreg[4 * g + e], st, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int k0 = kbeg + t * AL_K;
        if (k0 + AL_K > nk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (key >= nk) st[r] = -INFINITY;
            }
        }
        {
            float mx = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, st[r]), st[r + 1]);
            mx = fmaxf(mx, st[15]);
            if (__any(first || mx > AL_DEFER)) {
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
        // ---- O^T += V^T P^T on two alternating accumulators, with tile t + 1's operands read in between
        const bool nxt = t + 1 < ntile;
        if (nxt) {
            // tile t + 1 has landed; the only younger copies are tile t + 2's 16 (requested when tile t's reads had returned)
            if (t + 2 < ntile) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float* const ka = ka0 + NXT * AL_WAVE_F;
            const float* const va = va0 + NXT * AL_WAVE_F;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], st[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[16 + r], st[r], o1, 0, 0, 0);
                const int kr = (r & 3) + 8 * (r >> 2);
                vf[r] = va[kr * 64]; vf[16 + r] = va[kr * 64 + 32];           // tile t + 1's values into the registers the two products above have read
                if (r < 8) kf[r] = *reinterpret_cast<const f32x4*>(ka + (((2 * r + h) ^ jsw) << 2));
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads have RETURNED: tile t + 1's buffer may take the copy of tile t + 3
            __builtin_amdgcn_sched_barrier(0);
            if (t + 3 < ntile) issue(t + 3, NXT);
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], st[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[16 + r], st[r], o1, 0, 0, 0);
            }
        }
    }
    l_run += __shfl_xor(l_run, 32);

    // ---- merge the NW key ranges through LDS (as lg_attention_lat_kernel)
    __syncthreads();
    float* const part = lds;
    {
        float* pr = part + (wave * 32 + j) * 68;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            *reinterpret_cast<f32x4*>(pr + 8 * rq + 4 * h) = f32x4{o0[4 * rq], o0[4 * rq + 1], o0[4 * rq + 2], o0[4 * rq + 3]};
            *reinterpret_cast<f32x4*>(pr + 32 + 8 * rq + 4 * h) = f32x4{o1[4 * rq], o1[4 * rq + 1], o1[4 * rq + 2], o1[4 * rq + 3]};
        }
        if (h == 0) { pr[64] = m_run; pr[65] = l_run; }
    }
    __syncthreads();
    {
        constexpr int TPQ = NT_ / 32, DPT = 64 / TPQ;
        const int qj = tid / TPQ, dq = (tid % TPQ) * DPT;
        const int row = qb * 32 + qj;
        float mw[NW], lw[NW], m = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) { mw[w] = part[(w * 32 + qj) * 68 + 64]; lw[w] = part[(w * 32 + qj) * 68 + 65]; m = fmaxf(m, mw[w]); }
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        float l = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            if (!(lw[w] > 0.f)) continue;
            const float wgt = __builtin_amdgcn_exp2f(mw[w] - m);
            l += lw[w] * wgt;
            const float* pr = part + (w * 32 + qj) * 68 + dq;
            a0 += *reinterpret_cast<const f32x4*>(pr) * wgt;
            if (DPT == 8) a1 += *reinterpret_cast<const f32x4*>(pr + 4) * wgt;
        }
        const float inv = (row < nq && l > 0.f) ? 1.0f / l : 0.f;
        if (row < Lq) {
            float* op = out + ((size_t)seq * Lq + row) * 256 + head * 64 + dq;
            *reinterpret_cast<f32x4*>(op) = a0 * inv;
            if (DPT == 8) *reinterpret_cast<f32x4*>(op + 4) = a1 * inv;
        }
    }
}

#endif

#ifdef RFE_TUNING
extern "C" int rfe_k_dbg_timeline_att(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rfe_dbg_ts_att), (size_t)n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

// one (or a few) pairs: at most 8192 query rows, 16-byte aligned rows, no rotary (applied by the projection).  Returns false when the
// shape is not served (the caller falls back to lg_kernels.hip).
bool launch_lg_attention_lat(hipStream_t s, const float* q, const float* k, const float* v, int ld, float* out, int nseq, int Lq, int Lk,
                             const int* qlen, const int* klen, const int* kv_map, bool h2) {
    if ((size_t)nseq * Lq > 8192 || (ld % 4) || Lq < 1 || Lk < 1) return false;
    const int nqb = (Lq + 31) / 32;
    const int units8 = (4 * nseq + 7) / 8 * 8;    // (sequence, head) units padded to a multiple of 8: the block decode stays bijective
#ifdef RFE_TUNING
    const int abl = tune_int("RFE_ALAT_ABL", 0);
    static const int nw = tune_int("RFE_ALAT_WAVES", 4);   // 8 = two waves per SIMD (A/B: slower)
#else
    constexpr int abl = 0;
#endif
#define RFE_ALAT_GO(NW_, H2_)                                                                                                             \
    do {                                                                                                                                  \
        constexpr int bytes = (NW_ * AL_WAVE_F > NW_ * 32 * 68 ? NW_ * AL_WAVE_F : NW_ * 32 * 68) * 4;   /* tiles, later the merge */        \
        static bool ls_[64];                                                                                                              \
        ensure_dynamic_lds((const void*)lg_attention_lat_kernel<NW_, H2_>, bytes, ls_);                                                   \
        hipLaunchKernelGGL((lg_attention_lat_kernel<NW_, H2_>), dim3(nqb * units8), dim3(64 * NW_), bytes, s, q, k, v, ld, out, Lq, Lk, nqb, qlen, klen, kv_map, nseq, abl); \
    } while (0)
#ifdef RFE_TUNING
    if (nw == 8) { RFE_ALAT_GO(8, false); return true; }
#endif
    if (h2) { RFE_ALAT_GO(4, true); return true; }
#ifdef RFE_TUNING
    static const int pipe = tune_int("RFE_ALAT_PIPE", 0);   // round 6: next tile's operands read in the shadow of the PV products (two tile buffers per wave)
    if (pipe) {
        constexpr int bytes = 4 * 2 * AL_WAVE_F * 4;         // 128 KB (the merge needs 34.8 KB of it)
        static bool lp_[64];
        ensure_dynamic_lds((const void*)lg_attention_lat_pipe_kernel<4>, bytes, lp_);
        hipLaunchKernelGGL((lg_attention_lat_pipe_kernel<4>), dim3(nqb * units8), dim3(256), bytes, s, q, k, v, ld, out, Lq, Lk, nqb, qlen, klen, kv_map, nseq);
        return true;
    }
#endif
    RFE_ALAT_GO(4, false);
#undef RFE_ALAT_GO
    return true;
}

}  // namespace rfe
