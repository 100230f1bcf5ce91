// kbench: does MODE.FP16_OVFL make the operand split of h2_split.h saturate instead of overflowing to infinity?  (x = hi + lo for |x| up to 2 x 65504)
// hipcc --offload-arch=gfx950 -O3 fp16_ovfl.hip -o fp16_ovfl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, unsigned* o, int mode) {
    if (mode) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);   // MODE.FP16_OVFL = 1
    const float a = x[threadIdx.x * 2], b = x[threadIdx.x * 2 + 1];
    const f16x2 hv = {(_Float16)a, (_Float16)b};
    unsigned hi = __builtin_bit_cast(unsigned, hv), l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(l) : "v"(hi), "v"(a), "v"(b));
    o[threadIdx.x * 2] = hi; o[threadIdx.x * 2 + 1] = l;
}
int main() {
    float hx[4] = {1.0e5f, -7.0e4f, 3.0e5f, 123.456f};
    float* dx; unsigned* dout; hipMalloc(&dx, 16); hipMalloc(&dout, 16); hipMemcpy(dx, hx, 16, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        k<<<1, 2>>>(dx, dout, mode); unsigned ho[4]; hipMemcpy(ho, dout, 16, hipMemcpyDeviceToHost);
        auto h2f = [](unsigned short h) { _Float16 f; __builtin_memcpy(&f, &h, 2); return (float)f; };
        printf("FP16_OVFL=%d:", mode);
        for (int t = 0; t < 2; ++t) printf("  [%g -> hi %g lo %g | %g -> hi %g lo %g]", hx[2*t], h2f(ho[2*t] & 0xffff), h2f(ho[2*t+1] & 0xffff), hx[2*t+1], h2f(ho[2*t] >> 16), h2f(ho[2*t+1] >> 16));
        printf("\n");
    }
    return 0;
}
