// MINIMAL stand-alone reproducer (MI355X / gfx950): a packed fp32 instruction with CROSS-HALF operand selection
//     v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]        (d.lo = a.lo * b.hi, d.hi = a.hi * b.lo)
// returns +0 in the 16 lanes 48-63 of a wave when ANOTHER wave on the same SIMD issues VALU instructions while its own MFMAs are
// in flight.  No library code, ~100 lines:  hipcc --offload-arch=gfx950 -O3 tools/erratum/pk_min.hip -o pk_min && ./pk_min
//   victim  (stream 1): every thread loads a pair x, applies 64 packed operations that must give x back, stores it; out != in = wrong.
//   trigger (stream 2): independent MFMAs with `v_mov_b64 v[n:n+1], 0` between them (VALU = 0: MFMAs only), buffers of its own.
// Found while isolating why this repository's BatchNorm backward stored wrong values when built with packed fp32 operations
// (the library is built with -packed-fp32-ops for that reason).  Output of a run: tools/erratum/profiles/r03_pk_min.txt.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
#define R8(op) op op op op op op op op
#define R64(op) R8(op) R8(op) R8(op) R8(op) R8(op) R8(op) R8(op) R8(op)

// FORM 0: cross-half multiply by (1, 1); 1: the plain multiply; 2: cross-half ADD of (0, 0); 3: cross-half FMA a * b + 0;
// 4: the half swap v_pk_mov_b32 d, d, d op_sel:[1,0]; 5 / 6: the multiply with ONE operand half broadcast (a.lo for both results:
// op_sel_hi:[0,1]; a.hi for both: op_sel:[1,0]); 7: form 0 with the swapped operand as src0.
// BIG: the wave owns 254 VGPRs (one victim wave + one trigger wave per SIMD), else whatever the few registers below need.
template <int FORM, int BIG>
__global__ __launch_bounds__(256) void victim(const float2* __restrict__ src, float2* __restrict__ dst, long long n) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
        const float2 x = src[i];
        float2 y;
        if (BIG) asm volatile("" ::: "v253");
        asm volatile("v_mov_b32 v28, %[x0]\n\tv_mov_b32 v29, %[x1]\n\tv_mov_b32 v20, 1.0\n\tv_mov_b32 v21, 1.0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\ts_nop 4\n\t"
                     ".if %c[f] == 0\n\t" R64("v_pk_mul_f32 v[28:29], v[20:21], v[28:29] op_sel:[0,1] op_sel_hi:[1,0]\n\t") ".endif\n\t"
                     ".if %c[f] == 1\n\t" R64("v_pk_mul_f32 v[28:29], v[20:21], v[28:29]\n\t") ".endif\n\t"
                     ".if %c[f] == 2\n\t" R64("v_pk_add_f32 v[28:29], v[22:23], v[28:29] op_sel:[0,1] op_sel_hi:[1,0]\n\t") ".endif\n\t"
                     ".if %c[f] == 3\n\t" R64("v_pk_fma_f32 v[28:29], v[20:21], v[28:29], v[22:23] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t") ".endif\n\t"
                     ".if %c[f] == 4\n\t" R64("v_pk_mov_b32 v[28:29], v[28:29], v[28:29] op_sel:[1,0]\n\t") ".endif\n\t"
                     ".if %c[f] == 5\n\t" R64("v_pk_mul_f32 v[28:29], v[20:21], v[28:29] op_sel_hi:[0,1]\n\t") ".endif\n\t"
                     ".if %c[f] == 6\n\t" R64("v_pk_mul_f32 v[28:29], v[20:21], v[28:29] op_sel:[1,0]\n\t") ".endif\n\t"
                     ".if %c[f] == 7\n\t" R64("v_pk_mul_f32 v[28:29], v[28:29], v[20:21] op_sel:[1,0] op_sel_hi:[0,1]\n\t") ".endif\n\t"
                     "s_nop 4\n\tv_mov_b32 %[y0], v28\n\tv_mov_b32 %[y1], v29\n\t"
                     : [y0] "=v"(y.x), [y1] "=v"(y.y) : [x0] "v"(x.x), [x1] "v"(x.y), [f] "n"(FORM)
                     : "v20", "v21", "v22", "v23", "v28", "v29");
        dst[i] = y;
    }
}

template <int VALU>
__global__ __launch_bounds__(256) void trigger(float* __restrict__ out, int iters) {
    f16v acc[4] = {};
    asm volatile("" ::: "v147");                               // (148 + 64 registers: shares a SIMD with a 254-register wave)
    uint4 a = make_uint4(0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u), b = a;
    unsigned zsum = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc[k], 0, 0, 0);
            if (VALU) { unsigned long long z; asm volatile("v_mov_b64 %0, 0" : "=v"(z)); zsum += (unsigned)z; }
        }
    float s = 0.f;
    for (int k = 0; k < 4; ++k) s += acc[k][threadIdx.x & 15];
    if (s == 12345.678f || zsum == 77u) out[0] = s;
}

int main() {
    const long long n = 4LL << 20;
    std::vector<float> h(n * 2);
    unsigned long long s = 0x2545f4914f6cdd1dull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (float)((double)(s >> 11) / 9007199254740992.0 * 8.0 - 4.0) + 0.0078125f; }
    float2 *src, *dst; float* tout;
    CK(hipMalloc(&src, n * 8)); CK(hipMalloc(&dst, n * 8)); CK(hipMalloc(&tout, 4096));
    CK(hipMemcpy(src, h.data(), n * 8, hipMemcpyHostToDevice));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    std::vector<unsigned> out(n * 2), in(n * 2);
    memcpy(in.data(), h.data(), n * 8);
    const char* vn[10] = {"v_pk_mul_f32 cross-half, 254-register wave", "v_pk_mul_f32 plain,      254-register wave", "v_pk_add_f32 cross-half, 254-register wave",
                         "v_pk_fma_f32 cross-half, 254-register wave", "v_pk_mul_f32 cross-half, small wave", "v_pk_mul_f32 plain,      small wave",
                         "v_pk_mov_b32 op_sel:[1,0] (half swap), 254 reg.", "v_pk_mul_f32 op_sel_hi:[0,1] (a.lo twice)", "v_pk_mul_f32 op_sel:[1,0] (a.hi twice)",
                          "v_pk_mul_f32 cross-half on src0"};
    const char* tn[3] = {"alone", "beside MFMAs only", "beside MFMAs + v_mov_b64"};
    int bad_cross = 0, bad_other = 0;
    for (int v = 0; v < 10; ++v)
        for (int t = 0; t < 3; ++t) {
            long long bad = 0, l48 = 0, zeros = 0;
            for (int r = 0; r < 5; ++r) {
                CK(hipMemsetAsync(dst, 0xff, n * 8, s1)); CK(hipDeviceSynchronize());
                if (t == 1) trigger<0><<<4096, 256, 0, s2>>>(tout, 3000);
                if (t == 2) trigger<1><<<4096, 256, 0, s2>>>(tout, 3000);
                if (v == 0) victim<0, 1><<<2048, 256, 0, s1>>>(src, dst, n);
                if (v == 1) victim<1, 1><<<2048, 256, 0, s1>>>(src, dst, n);
                if (v == 2) victim<2, 1><<<2048, 256, 0, s1>>>(src, dst, n);
                if (v == 3) victim<3, 1><<<2048, 256, 0, s1>>>(src, dst, n);
                if (v == 4) victim<0, 0><<<2048, 256, 0, s1>>>(src, dst, n);
                if (v == 5) victim<1, 0><<<2048, 256, 0, s1>>>(src, dst, n);
                if (v == 6) victim<4, 1><<<2048, 256, 0, s1>>>(src, dst, n);
                if (v == 7) victim<5, 1><<<2048, 256, 0, s1>>>(src, dst, n);
                if (v == 8) victim<6, 1><<<2048, 256, 0, s1>>>(src, dst, n);
                if (v == 9) victim<7, 1><<<2048, 256, 0, s1>>>(src, dst, n);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(out.data(), dst, n * 8, hipMemcpyDeviceToHost));
                for (long long i = 0; i < n * 2; ++i)
                    if (out[i] != in[i]) { ++bad; zeros += (out[i] & 0x7fffffffu) == 0; l48 += ((i / 2) % 64) >= 48; }
            }
            printf("%-44s %-26s wrong values %8lld of %lld (zeros %lld, in lanes 48-63 %lld)\n", vn[v], tn[t], bad, 5 * n * 2, zeros, l48);
            ((v == 0 || v == 2 || v == 3 || v == 4) ? bad_cross : bad_other) += bad != 0;
        }
    printf(bad_cross && !bad_other ? "REPRODUCED: only packed arithmetic with src1's halves swapped (mul / add / fma), only beside the MFMA + VALU co-runner\n" : bad_cross ? "REPRODUCED (see the table)\n" : "NOT REPRODUCED\n");
    return 0;
}
