// Shared device/host helpers for libyolo_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/yolo_amd.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
// 2-byte element tag for kernel templates.  (Not __bf16 itself: rocprofv3's demangler garbles kernel names
// that carry the DF16b mangling, which makes profiles unreadable.)
struct bf16_t { uint16_t bits; };

// hipGetLastError() is per-thread sticky state shared with every other HIP user in the process
// (torch): YOLO_LAUNCH() drops whatever was pending so YOLO_LAUNCH_CHECK() reports only ours.
#define YOLO_LAUNCH(...)                                     \
    do {                                                     \
        (void)hipGetLastError();                             \
        hipLaunchKernelGGL(__VA_ARGS__);                     \
    } while (0)
#define YOLO_LAUNCH_CHECK()                                  \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

// Lab knobs (A/B switches and ablation sweeps of tools/): the shipped library does not read its caller's environment.  `make lab`
// builds _lab/libyolo_amd_lab.so with -DYOLO_LAB, where yolo_lab_env(name, dflt) is getenv; in the default build it IS the
// default (tests/test_host.py checks that no YOLO_* environment name is left in libyolo_amd.so).
#ifdef YOLO_LAB
#include <stdlib.h>
#define YOLO_LAB_ENV(name, dflt) (getenv(name) ? atoll(getenv(name)) : (long long)(dflt))
#define YOLO_LAB_SET(name) (getenv(name) != nullptr)
#else
#define YOLO_LAB_ENV(name, dflt) ((long long)(dflt))
#define YOLO_LAB_SET(name) (false)
#endif

__host__ __device__ inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
__host__ __device__ inline long long round_up_ll(long long a, long long b) { return (a + b - 1) / b * b; }

// float -> bf16 bits, round-to-nearest-even (matches torch .to(bfloat16) for finite values).
// float -> bf16, round-to-nearest-even: gfx950's v_cvt_pk_bf16_f32 (one VALU op per PAIR; the bit-twiddled
// version costs ~20 and was a third of the conv epilogue's instruction count).
typedef __attribute__((ext_vector_type(2))) float yolo_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 yolo_bf16x2;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const yolo_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, yolo_bf16x2));
}
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) { return pack_bf16x2(f, 0.f) & 0xffffu; }
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
// LeakyReLU for 0 <= slope <= 1 (the dispatchers reject other slopes): max(t, t*slope), two VALU ops
__device__ __forceinline__ float leaky(float t, float slope) { return fmaxf(t, t * slope); }

// IEEE half (YOLO_F16): same storage as bf16_t, another tag.  gfx950 converts pairs with v_cvt_pk_f16_f32 (round-to-nearest-even)
// and unpacks with v_cvt_f32_f16 (the high half through SDWA): one VALU operation each, like the bf16 forms.
struct f16_t { uint16_t bits; };
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 yolo_f16x2;
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    const yolo_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, yolo_f16x2));
}
// What the kernels need to know about a 2-byte activation type: its name in kernel signatures, how two fp32 values round into a
// packed pair and how a pair unpacks.  (bf16's forms are the expressions the kernels had inline before round 5.)
template <typename T> struct Elem;
template <> struct Elem<bf16_t> {
    static constexpr const char* name = "bf16_t";
    static constexpr int dtype = YOLO_BF16;
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
    static __device__ __forceinline__ float lo(uint32_t w) { return bf16_bits_to_f32(w & 0xffffu); }
    static __device__ __forceinline__ float hi(uint32_t w) { return bf16_bits_to_f32(w >> 16); }
};
template <> struct Elem<f16_t> {
    static constexpr const char* name = "f16_t";
    static constexpr int dtype = YOLO_F16;
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_f16x2(lo, hi); }
    static __device__ __forceinline__ float lo(uint32_t w) { return (float)__builtin_bit_cast(yolo_f16x2, w)[0]; }
    static __device__ __forceinline__ float hi(uint32_t w) { return (float)__builtin_bit_cast(yolo_f16x2, w)[1]; }
};
template <> struct Elem<float> {
    static constexpr const char* name = "float";
    static constexpr int dtype = YOLO_F32;
    // (never called: the 4-byte paths are separate branches; present so that `if constexpr (ES == 2)` bodies parse)
    static __device__ __forceinline__ uint32_t pack2(float, float) { return 0; }
    static __device__ __forceinline__ float lo(uint32_t) { return 0.f; }
    static __device__ __forceinline__ float hi(uint32_t) { return 0.f; }
};
// SPLIT type (round 6; YOLO_BF16X3): a value v is carried as TWO 2-byte numbers, hi = round(v) and lo = round(v - hi)
// -- 16 significant bits -- and a product w * x is taken as w_hi x_hi + w_hi x_lo + w_lo x_hi on the 2-byte MFMA
// with fp32 accumulation (the w_lo x_lo term, 2^-18 of the product, is dropped): three MFMAs per product instead of the
// sixteen-times-slower v_mfma_f32_32x32x2f32, decoded boxes inside the north-star 1e-3 (3e-4 max on the D53 random-BN nets).  Storage: a pixel holds its C hi values, then -- `lo offset` elements further -- its C lo
// values; the convolution kernels see 3 * C / 32 K-chunks [x_hi | x_lo | x_hi] against the weight image [w_hi | w_hi | w_lo].
struct bf16x3_t { uint16_t bits; };
template <> struct Elem<bf16x3_t> : Elem<bf16_t> {
    static constexpr const char* name = "bf16x3_t";
    static constexpr int dtype = YOLO_BF16X3;
};
struct f16x3_t { uint16_t bits; };        // (YOLO_F16X3: the same scheme on IEEE-half pairs, 22 significant bits)
template <> struct Elem<f16x3_t> : Elem<f16_t> {
    static constexpr const char* name = "f16x3_t";
    static constexpr int dtype = YOLO_F16X3;
};
template <typename T> struct IsSplit { static constexpr bool value = false; };
template <> struct IsSplit<bf16x3_t> { static constexpr bool value = true; };
template <> struct IsSplit<f16x3_t> { static constexpr bool value = true; };
// one 32x32x16 MFMA step on 2-byte operands (A, B: 8 elements per lane as a uint4)
template <typename T> __device__ __forceinline__ f32x16 mfma16(const uint4& a, const uint4& b, const f32x16& c);
template <> __device__ __forceinline__ f32x16 mfma16<bf16_t>(const uint4& a, const uint4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mfma16<f16_t>(const uint4& a, const uint4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mfma16<bf16x3_t>(const uint4& a, const uint4& b, const f32x16& c) { return mfma16<bf16_t>(a, b, c); }
template <> __device__ __forceinline__ f32x16 mfma16<f16x3_t>(const uint4& a, const uint4& b, const f32x16& c) { return mfma16<f16_t>(a, b, c); }
template <typename T> struct IsBf16 { static constexpr bool value = false; };
template <> struct IsBf16<bf16_t> { static constexpr bool value = true; };

// bytes of one stored element; 0 for an unknown dtype (every entry point checks dtype_valid first: a garbage dtype must not be
// sized as a 2-byte type)
static inline int elem_size(int dtype) { return dtype == YOLO_F32 ? 4 : (dtype == YOLO_BF16 || dtype == YOLO_F16 || dtype == YOLO_BF16X3 || dtype == YOLO_F16X3) ? 2 : 0; }
static inline bool dtype_valid(int dtype) { return dtype == YOLO_F32 || dtype == YOLO_BF16 || dtype == YOLO_F16 || dtype == YOLO_BF16X3 || dtype == YOLO_F16X3; }
// the single-plane dtypes (what the training / fused / streaming entries take)
static inline bool dtype_plain(int dtype) { return dtype == YOLO_F32 || dtype == YOLO_BF16 || dtype == YOLO_F16; }
static inline bool dtype_split(int dtype) { return dtype == YOLO_BF16X3 || dtype == YOLO_F16X3; }
// stored planes per activation element (split types: hi + lo) and K passes of a convolution over the input channels
static inline int dtype_planes(int dtype) { return dtype_split(dtype) ? 2 : 1; }
static inline int dtype_kpasses(int dtype) { return dtype_split(dtype) ? 3 : 1; }
// channels held by one 64-byte K-chunk
static inline int chunk_channels(int dtype) { return 64 / elem_size(dtype); }
// packed weights / scale / bias are padded to a multiple of this many output channels
#define YOLO_COUT_PAD 256
// The lanes of a wave exchange data through the wave's LDS scratch (the transpose, the offset tables, the statistics
// columns) with no barrier: LDS operations of one wave execute in order.  The COMPILER does not know that -- to it a lane
// that did not store to a table still holds what it loaded from it before, and it forwarded slab 0's output offsets to lanes
// 32-63 of slab 2 in the fused-tail kernels (found by test_conv_tail_1x1_fused_is_bit_identical, round 4).  This fence emits
// no instruction; it makes every LDS write before it visible to the loads after it as far as the optimiser is concerned.
__device__ __forceinline__ void wave_lds_fence() { asm volatile("" ::: "memory"); }

