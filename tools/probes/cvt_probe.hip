// Does hipcc lower float -> __bf16 pairs to v_cvt_pk_bf16_f32 on gfx950?  (compile with -S and read the ISA)
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__global__ void k_cast(const float* a, unsigned* o) {
    const int i = threadIdx.x;
    f32x2 v = {a[2 * i], a[2 * i + 1]};
    bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    o[i] = __builtin_bit_cast(unsigned, b);
}
__global__ void k_asm(const float* a, unsigned* o) {
    const int i = threadIdx.x;
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a[2 * i]), "v"(a[2 * i + 1]));
    o[i] = r;
}
