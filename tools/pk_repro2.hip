// ONE-BINARY reproducer of the packed-fp32 corruption (round 3): no Python, no torch, no library.
//   victim : this repository's BatchNorm backward (yolo_bn_train_bwd_pp: bn_reduce_kernel + bn_apply_kernel<bf16,1,1>), compiled
//            into this binary WITH the packed fp32 operations (no -packed-fp32-ops target feature) and with the paired set-up of
//            its per-channel factors (-DYOLO_BN_PAIRED_FACTORS: the form the library shipped until round 3), on stream 1;
//   trigger: a synthetic kernel on stream 2, buffers of its own -- an MFMA loop in the shape of the generic convolution's K loop
//            (four accumulators, four ds_read_b128 per four v_mfma_f32_32x32x16_bf16, two barriers per step) with the same
//            footprint (48 KB LDS, 148 + 64 registers: ONE of its waves and ONE 254-register victim wave share a SIMD), and,
//            for --trigger 10, two `v_mov_b64 v[n:n+1], 0` per group (the zero fill of a padded 16-byte unit that every K loop
//            of the library's convolutions and weight gradients has).  --trigger 9 is the same loop without the 64-bit moves.
// Every dy of a co-run is compared bit for bit with the same call run alone.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DYOLO_BN_PAIRED_FACTORS -I yolo_amd/csrc -I include \
//         tools/pk_repro2.hip -o tools/_build/pk_repro2 && tools/_build/pk_repro2 [rounds]
// Measured (MI355X, ROCm 7.0.2 runtime, hipcc 7.2): trigger 10 corrupts 19-20 of 20 launches (exact zeros, lanes 48-63 of a wave),
// trigger 9 none, alone none; the same binary built with -Xclang -target-feature -Xclang -packed-fp32-ops: none anywhere.
#include "../yolo_amd/csrc/train.hip"
// (train.hip's weight-gradient entry points call into wgrad_walk.hip; nothing here uses them)
int wgrad_walk_dispatch(const void*, const void*, float*, int, int, int, int, int, long long, int, hipStream_t) { return YOLO_EUNSUPPORTED; }
int wgrad_gemm_dispatch(const void*, const void*, float*, long long, int, int, long long, int, hipStream_t) { return YOLO_EUNSUPPORTED; }
#include <stdio.h>
#include <vector>
#include <string.h>

typedef float f16v_ __attribute__((ext_vector_type(16)));
typedef __bf16 bf8_ __attribute__((ext_vector_type(8)));

template <int MOV64>
__global__ __launch_bounds__(256) void trigger_kernel(float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) uint4 lds[3072];
    for (int i = threadIdx.x; i < 3072; i += 256) lds[i] = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    asm volatile("" ::: "v147");
    __syncthreads();
    f16v_ acc[4];
    unsigned zsum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    const int l31 = threadIdx.x & 31, h = (threadIdx.x & 63) >> 5;
    const int base = l31 * 64 + ((h ^ ((l31 >> 2) & 3)) << 4) + (threadIdx.x >> 6) * 2048;
    const char* L = reinterpret_cast<const char*>(lds);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 18; ++g) {
            const int o = ((g * 4096) & 16383) ^ ((g & 1) * 32);
            const uint4 a0 = *(const uint4*)(L + ((base + o) & 32767)), a1 = *(const uint4*)(L + ((base + o + 2048) & 32767));
            const uint4 b0 = *(const uint4*)(L + ((base + o + 8192) & 32767)), b1 = *(const uint4*)(L + ((base + o + 10240) & 32767));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8_, a0), __builtin_bit_cast(bf8_, b0), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8_, a0), __builtin_bit_cast(bf8_, b1), acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8_, a1), __builtin_bit_cast(bf8_, b0), acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8_, a1), __builtin_bit_cast(bf8_, b1), acc[3], 0, 0, 0);
            if (MOV64) {
                unsigned long long z0, z1;
                asm volatile("v_mov_b64 %0, 0\n\tv_mov_b64 %1, 0" : "=v"(z0), "=v"(z1));
                zsum += (unsigned)z0 + (unsigned)(z1 >> 32);
            }
        }
        __syncthreads();
        __syncthreads();
    }
    float sacc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) sacc += acc[k][threadIdx.x & 15];
    if (sacc == 12345.678f || zsum == 77u) out[0] = sacc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20;
    const int N = 64, H = 52, W = 52, C = 256;
    const long long npix = (long long)N * H * W, n = npix * C;
    std::vector<unsigned short> hy(n), hdz(n);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((double)(s >> 11) / 9007199254740992.0) * 2.f - 1.f; };
    for (long long i = 0; i < n; ++i) { hy[i] = f2bf(rnd() * 1.7f); hdz[i] = f2bf(rnd() * 0.01f); }
    std::vector<float> hm(C), hi(C), hg(C), hb(C);
    for (int c = 0; c < C; ++c) { hm[c] = 0.01f * (c % 7); hi[c] = 1.f + 0.001f * c; hg[c] = 0.5f + 0.003f * c; hb[c] = 0.1f * ((c % 5) - 2); }
    void *y, *dz, *dy, *ref;
    float *mean, *invstd, *gamma, *beta, *dgam, *dbet, *tout;
    double* ws[2];
    CK(hipMalloc(&y, n * 2)); CK(hipMalloc(&dz, n * 2)); CK(hipMalloc(&dy, n * 2)); CK(hipMalloc(&ref, n * 2));
    CK(hipMalloc(&mean, C * 4)); CK(hipMalloc(&invstd, C * 4)); CK(hipMalloc(&gamma, C * 4)); CK(hipMalloc(&beta, C * 4));
    CK(hipMalloc(&dgam, C * 4)); CK(hipMalloc(&dbet, C * 4)); CK(hipMalloc(&tout, 1 << 18));
    for (int k = 0; k < 2; ++k) { CK(hipMalloc(&ws[k], 4096 * 8)); CK(hipMemset(ws[k], 0, 4096 * 8)); }
    CK(hipMemcpy(y, hy.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dz, hdz.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(mean, hm.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(invstd, hi.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(gamma, hg.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(beta, hb.data(), C * 4, hipMemcpyHostToDevice));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    int call = 0;
    auto bn = [&](void* out) {
        CK(hipMemsetAsync(ws[call & 1], 0, 4096 * 8, s1));
        const int rc = yolo_bn_train_bwd_pp(dz, y, mean, invstd, gamma, beta, out, dgam, dbet, ws[call & 1], ws[(call & 1) ^ 1], 4096, npix, C, 0.1f,
                                            YOLO_BF16, s1);
        if (rc) { printf("yolo_bn_train_bwd_pp: %d\n", rc); exit(2); }
        ++call;
    };
    bn(ref);
    CK(hipDeviceSynchronize());
    std::vector<unsigned short> href(n), hout(n);
    CK(hipMemcpy(href.data(), ref, n * 2, hipMemcpyDeviceToHost));
    const char* names[3] = {"alone", "beside the MFMA loop without 64-bit moves (trigger 9)", "beside the MFMA loop with v_mov_b64 v[n:n+1], 0 (trigger 10)"};
    int any10 = 0, anyother = 0;
    for (int mode = 0; mode < 3; ++mode) {
        long long bad = 0, zeros = 0, l48 = 0;
        int events = 0;
        for (int r = 0; r < rounds; ++r) {
            CK(hipMemsetAsync(dy, 0xff, n * 2, s1));
            CK(hipDeviceSynchronize());
            if (mode == 1) trigger_kernel<0><<<676 * 6, 256, 0, s2>>>(tout, 12);
            if (mode == 2) trigger_kernel<1><<<676 * 6, 256, 0, s2>>>(tout, 12);
            bn(dy);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hout.data(), dy, n * 2, hipMemcpyDeviceToHost));
            long long k = 0;
            for (long long i = 0; i < n; ++i)
                if (hout[i] != href[i]) { ++k; zeros += (hout[i] & 0x7fff) == 0; l48 += ((i / 8) % 64) >= 48; }
            if (k) { ++events; bad += k; }
        }
        printf("%-66s launches with a mismatch %3d / %d, elements %7lld (exact zeros %lld, in lanes 48-63 %lld)\n", names[mode], events, rounds, bad, zeros, l48);
        if (mode == 2) any10 = events; else anyother += events;
    }
    printf(any10 && !anyother ? "REPRODUCED: only beside the co-runner with the 64-bit zero moves\n" : any10 ? "REPRODUCED (see the table)\n" : "NOT REPRODUCED\n");
    return 0;
}
