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

__host__ __device__ inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
__host__ __device__ inline long long round_up_ll(long long a, long long b) { return (a + b - 1) / b * b; }

// float -> bf16 bits, round-to-nearest-even (matches torch .to(bfloat16) for finite values).
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return f32_to_bf16_bits(lo) | (f32_to_bf16_bits(hi) << 16);
}

static inline int elem_size(int dtype) { return dtype == YOLO_BF16 ? 2 : 4; }
// channels held by one 64-byte K-chunk
static inline int chunk_channels(int dtype) { return 64 / elem_size(dtype); }
// packed weights / scale / bias are padded to a multiple of this many output channels
#define YOLO_COUT_PAD 256
