// h2_split.h -- the operand split of RFE_OPT_LG_FP16X2 (gemm_h2.hip, lg_attention_h2.hip): x = hi + lo with hi = fp16(x) and
// lo = fp16(x - hi), both round-to-nearest: 22 of x's 24 significand bits (values below fp16's normal range are carried as subnormals:
// absolute error <= 2^-25).  Three instructions per PAIR of floats: v_cvt_pk_f16_f32 packs the two high parts; v_fma_mixlo_f16 /
// v_fma_mixhi_f16 compute hi * (-1) + x in fp32 (exact) from the packed halves and round it into the low / high half of the second word
// -- the compiler's own lowering of the same expression is six (two unpacking v_cvt_f32_f16, two v_sub_f32, two packs).
#ifndef RFE_H2_SPLIT_H
#define RFE_H2_SPLIT_H
#include <stdint.h>

namespace rfe {

typedef _Float16 h2_f16x2 __attribute__((ext_vector_type(2)));

// (x, y) -> hi = (fp16 x, fp16 y) packed (x in the low half), lo = the packed residuals
__device__ __forceinline__ void h2_split2(float x, float y, uint32_t& hi, uint32_t& lo) {
    const h2_f16x2 hv = {(_Float16)x, (_Float16)y};
    hi = __builtin_bit_cast(uint32_t, hv);
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(l) : "v"(hi), "v"(x), "v"(y));
    lo = l;
}

// The same split, saturating in software (two v_med3_f32 more per float): x is clamped to +-131008 and its high part to +-65504, so both
// halves stay finite whatever comes in -- x = hi + lo still holds up to |x| = 131008 (to lo's 11 bits).  Used where it is free (the
// load-time split of the weights).  gemm_h2's activation staging keeps the plain split: both this form and the mode bit below measured
// 8-10 % on that kernel (ffn.0 94 -> 104 us), so its documented domain stays |activation| < 65504.
__device__ __forceinline__ void h2_split2_sat(float x, float y, uint32_t& hi, uint32_t& lo) {
    const float xc = __builtin_amdgcn_fmed3f(x, -131008.0f, 131008.0f), yc = __builtin_amdgcn_fmed3f(y, -131008.0f, 131008.0f);
    const h2_f16x2 hv = {(_Float16)__builtin_amdgcn_fmed3f(xc, -65504.0f, 65504.0f), (_Float16)__builtin_amdgcn_fmed3f(yc, -65504.0f, 65504.0f)};
    hi = __builtin_bit_cast(uint32_t, hv);
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(l) : "v"(hi), "v"(xc), "v"(yc));
    lo = l;
}

// MODE.FP16_OVFL = 1 for the rest of the wave (per-wave state, set at the top of the split kernels): an fp16 result past 65504 is clamped
// to +-65504 instead of becoming infinite.  The split then degrades gracefully outside its domain -- hi saturates, lo = fp16(x - hi) takes
// up the rest, x = hi + lo holds up to |x| = 131008 (to lo's 11 bits), beyond that the operand saturates, finite -- where the default
// mode would turn one large activation into inf - inf = NaN for its whole row (tools/kbench/fp16_ovfl.hip).
// (Used by the attention kernel, which it does not slow down.)
__device__ __forceinline__ void h2_saturate_mode() { __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1); }

}  // namespace rfe
#endif
