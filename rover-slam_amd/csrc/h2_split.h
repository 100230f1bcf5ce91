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

}  // namespace rfe
#endif
