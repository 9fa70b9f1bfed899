// 1x1 convolution with the K range split ACROSS THE WAVES OF ONE BLOCK (round 3): the small-map 1x1 layers of the net
// (13x13 / 26x26 maps at batch 32: 5408 / 21632 pixels, K = 512 ... 2048) launch one round of at most 1.3 blocks per CU,
// and a block of the pipelined kernel walks its K range as a chain of 16-32 dependent phases (DMA wait -> barrier -> LDS
// reads -> 4 MFMAs) with one or two waves per SIMD: ~600 cycles per phase for 128 cycles of matrix work, 17 us per layer
// whatever its size.  More blocks do not help (every pixel tile re-reads the whole weight matrix from
// L2: the tile shape is what bounds the traffic) and a split across BLOCKS needs atomics or a device-scope fence per tile
// (measured 4-20x slower).  Here the block itself is KG groups of WAVES_P x WAVES_C waves; group g owns the K chunks
// [g * nchunks / KG, (g + 1) * nchunks / KG) of the SAME output tile with an LDS ring of its own, so a CU holds KG times the
// waves (each SIMD always has another group's wave to issue from) and the dependent chain is KG times shorter.  The partial
// accumulators meet in LDS (group 0 adds groups 1 .. KG-1 in that order: deterministic) and group 0 runs the shared
// epilogue.  Same packed weight image, LDS layouts (64-byte K-chunk slots, XOR-swizzled 16-byte units), LDS-DMA staging,
// counted s_waitcnt vmcnt + one raw s_barrier per phase and epilogue as conv_pipe.hip's 1x1 unit.
#include "common.h"
#include "conv_args.h"
#include "conv_epilogue.h"
#include <stdio.h>

namespace {
__device__ __attribute__((aligned(64))) unsigned int sk_zero_page[16];
typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ void glds16_m0(const void* gsrc, uint32_t lds_dst_uniform) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T> struct FragS;
template <> struct FragS<bf16_t> {
    static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct FragS<f16_t> {
    static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct FragS<float> {
    static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};
// bijective XCD remap (8 XCDs, block b runs on XCD b % 8): logical ids are contiguous per XCD
__device__ __forceinline__ int sk_xcd_remap(int bid, int nblk) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
}  // namespace

template <typename T, int KG, int WAVES_P, int WAVES_C, int MI, int NI, int R>
__global__ __launch_bounds__(KG* WAVES_P* WAVES_C * 64) void conv_sk_kernel(ConvArgs a) {
    constexpr int NWG = WAVES_P * WAVES_C, NTG = NWG * 64;              // waves / threads of one K group
    constexpr int BP = WAVES_P * NI * 32, BC = WAVES_C * MI * 32;
    constexpr int X_STAGE = BP * 64, W_STAGE = BC * 64;                 // bytes of one phase (one 64-byte K chunk)
    constexpr int XL = X_STAGE / (NTG * 16), WL = W_STAGE / (NTG * 16); // LDS-DMAs per thread per phase
    static_assert(X_STAGE % (NTG * 16) == 0 && W_STAGE % (NTG * 16) == 0 && XL >= 1 && WL >= 1, "whole DMAs per phase");
    constexpr int W_OFF = R * X_STAGE;
    constexpr int GROUP_BYTES = R * (X_STAGE + W_STAGE);
    constexpr int ACC_BYTES = MI * NI * 16 * 64 * 4;                    // one wave's accumulators
    constexpr int RED_BYTES = (KG - 1) * NWG * ACC_BYTES;
    constexpr int EPI_BYTES = NWG * YOLO_EPI_WAVE_BYTES_MI(MI);
    constexpr int M1 = KG * GROUP_BYTES > RED_BYTES ? KG * GROUP_BYTES : RED_BYTES;
    constexpr int SMEM = M1 > EPI_BYTES ? M1 : EPI_BYTES;
    static_assert(SMEM <= 163840, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)smem;

    const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x / NTG);
    const int tid = (int)threadIdx.x - grp * NTG;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wave_p = wave % WAVES_P, wave_c = wave / WAVES_P;

    const int lid = sk_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_p = fdiv(lid, a.d_tc);
    const int tile_c = lid - tile_p * a.tiles_c;
    const int i0 = tile_p * BP, co0 = tile_c * BC;
    const int row_bytes = a.x_ps * (int)sizeof(T);
    const int nph = a.nchunks / KG;                          // phases of this group (host: nchunks % KG == 0, nph >= R - 1)
    const int chunk0 = grp * nph;

    char* gsm = smem + grp * GROUP_BYTES;
    const uint32_t wave_lds = lds0 + grp * GROUP_BYTES + wave * 1024;
    const long long wplane = (long long)a.Cout_pad * 64;
    const char* wsrc = a.wp + (long long)co0 * 64 + tid * 16 + (long long)chunk0 * wplane;
    const char* xsrc[XL];
    unsigned xinc[XL];
#pragma unroll
    for (int j = 0; j < XL; ++j) {
        const int u = tid + j * NTG;
        const int pos = u >> 2, part = u & 3;
        const int i = i0 + pos;
        const bool ok = i < a.total_i;
        const int lp = (part ^ ((pos >> 2) & 3)) * 16;
        xsrc[j] = ok ? a.x + ((size_t)i * (size_t)row_bytes + (size_t)lp + (size_t)chunk0 * 64) : (const char*)sk_zero_page;
        xinc[j] = ok ? 64u : 0u;
    }
    // DMAs of the group's phase p (clamped: the tail re-loads the last phase into a dead slot so the counted waits stay exact)
    auto issue = [&](int p) {
        const int pc = p < nph ? p : nph - 1;
        const int slot = p % R;
#pragma unroll
        for (int j = 0; j < XL; ++j) glds16_m0(xsrc[j] + (size_t)pc * xinc[j], wave_lds + slot * X_STAGE + j * NTG * 16);
#pragma unroll
        for (int j = 0; j < WL; ++j) glds16_m0(wsrc + (long long)pc * wplane + (long long)j * NTG * 16, wave_lds + W_OFF + slot * W_STAGE + j * NTG * 16);
    };
#pragma unroll
    for (int p = 0; p < R - 1; ++p) issue(p);

    int bx[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int slot = (wave_p * NI + ni) * 32 + l31;
        bx[ni] = slot * 64 + ((h ^ ((slot >> 2) & 3)) << 4);
    }
    const int aoff0 = (wave_c * MI * 32 + l31) * 64 + ((h ^ ((l31 >> 2) & 3)) << 4);
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    wait_vmcnt<(R - 2) * (XL + WL)>();
    __builtin_amdgcn_s_barrier();
    int slot = 0;
    for (int p = 0; p < nph; ++p) {
        issue(p + R - 1);                                    // into the slot read in phase p - 1 (everyone is past its barrier)
        const char* Xl = gsm + slot * X_STAGE;
        const char* Wl = gsm + W_OFF + slot * W_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 af[MI], bf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[mi] = *(const uint4*)(Wl + mi * 2048 + (aoff0 ^ (ks * 32)));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bf[ni] = *(const uint4*)(Xl + (bx[ni] ^ (ks * 32)));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) FragS<T>::mma(af[mi], bf[ni], acc[mi][ni]);
        }
        slot = slot + 1 == R ? 0 : slot + 1;
        wait_vmcnt<(R - 2) * (XL + WL)>();                   // phase p + 1 has landed
        __builtin_amdgcn_s_barrier();
    }
    wait_vmcnt<0>();                                         // the tail's dead DMAs
    __builtin_amdgcn_s_barrier();                            // every group is done with the rings

    // ---- the groups' partial sums meet in LDS; group 0 adds them in group order -------------------------------------------
    if (KG > 1) {
        if (grp > 0) {
            char* dst = smem + ((grp - 1) * NWG + wave) * ACC_BYTES + lane * 16;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                        *(f32x4*)(dst + ((mi * NI + ni) * 4 + q) * 1024) = v;
                    }
        }
        __builtin_amdgcn_s_barrier();
        if (grp == 0) {
#pragma unroll
            for (int g = 1; g < KG; ++g) {
                const char* src = smem + ((g - 1) * NWG + wave) * ACC_BYTES + lane * 16;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = *(const f32x4*)(src + ((mi * NI + ni) * 4 + q) * 1024);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[mi][ni][4 * q + e] += v[e];
                        }
            }
        }
        __builtin_amdgcn_s_barrier();                        // the epilogue's scratch overlays the partial sums
    }
    if (grp != 0) return;

    long long yoff[NI], roff[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int i = i0 + (wave_p * NI + ni) * 32 + l31;
        const int ic = min(i, a.total_i - 1);
        const int n = fdiv(ic, a.d_HoWo);
        int pix = ic - n * (a.Ho * a.Wo);
        if (a.up2) {
            const int oy = fdiv(pix, a.d_TWt);
            pix = oy * 4 * a.Wo + 2 * (pix - oy * a.Wo);
        }
        yoff[ni] = (i < a.total_i) ? (long long)n * a.y_bs + (long long)pix * a.y_ps : -1;
        roff[ni] = (long long)n * a.r_bs + (long long)pix * a.r_ps;
    }
    conv_epilogue<T, MI, NI>(acc, yoff, smem + wave * YOLO_EPI_WAVE_BYTES_MI(MI), a, co0 + wave_c * MI * 32, lane, roff, nullptr);
}

template <typename T, int KG, int WAVES_P, int WAVES_C, int MI, int NI, int R>
static int launch_sk(ConvArgs& a, hipStream_t st, const NameOut* nm) {
    constexpr int BP = WAVES_P * NI * 32, BC = WAVES_C * MI * 32;
    if (a.stats || a.d2s) return YOLO_EUNSUPPORTED;
    if (a.nchunks % KG || a.nchunks / KG < R - 1) return YOLO_EUNSUPPORTED;
    a.TWt = a.Wo;
    a.PW = a.Wo;
    a.nstrips = 1;
    const long long tot = (long long)a.N * a.Ho * a.Wo;
    if (tot > 0x7fffffffLL) return YOLO_EUNSUPPORTED;
    a.total_i = (int)tot;
    a.tiles_per_strip = (a.total_i + BP - 1) / BP;
    a.tiles_c = (a.Cout + BC - 1) / BC;
    const long long grid = (long long)a.tiles_per_strip * a.tiles_c;
    if (grid > 0x7fffffffLL) return YOLO_EUNSUPPORTED;
    conv_args_fastdiv(a);
    if (nm) {
        snprintf(nm->buf, nm->len, "void conv_sk_kernel<%s, %d, %d, %d, %d, %d, %d>(ConvArgs)", Elem<T>::name, KG,
                 WAVES_P, WAVES_C, MI, NI, R);
        if (nm->stats_rows) *nm->stats_rows = -1;
        return YOLO_OK;
    }
    YOLO_LAUNCH((conv_sk_kernel<T, KG, WAVES_P, WAVES_C, MI, NI, R>), dim3((unsigned)grid), dim3(KG * WAVES_P * WAVES_C * 64), 0, st, a);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// algo ids 30-33, 35 (1x1 stride 1 only):
//   30: 4 K groups x 4 waves, 64 px x 128 cout     31: 2 K groups x 4 waves, 64 px x 128 cout (4-slot rings)
//   32: 2 K groups x 4 waves, 64 px x 256 cout     33: 2 K groups x 4 waves, 128 px x 128 cout
//   35: 4 K groups x 4 waves, 64 px x 64 cout      (2 groups x 8 waves, 128 px x 256 cout: 58-145 spilled registers; not built)
template <typename T>
static int sk_dispatch_t(ConvArgs& a, int algo, hipStream_t st, const NameOut* nm) {
    switch (algo) {
        case 30: return launch_sk<T, 4, 2, 2, 2, 1, 3>(a, st, nm);
        case 31: return launch_sk<T, 2, 2, 2, 2, 1, 4>(a, st, nm);
        case 32: return launch_sk<T, 2, 1, 4, 2, 2, 3>(a, st, nm);
        case 33: return launch_sk<T, 2, 2, 2, 2, 2, 4>(a, st, nm);
        case 35: return launch_sk<T, 4, 2, 2, 1, 1, 4>(a, st, nm);
    }
    return YOLO_EUNSUPPORTED;
}

int conv_sk_dispatch(ConvArgs& a, int ks, int stride, int dtype, int algo, hipStream_t st, const NameOut* nm) {
    if (ks != 1 || stride != 1 || !dtype_plain(dtype)) return YOLO_EUNSUPPORTED;
    if ((a.Cin * elem_size(dtype)) % 64) return YOLO_EUNSUPPORTED;
    if ((long long)a.N * a.H * a.W * a.x_ps * elem_size(dtype) >= 0xffffff00LL) return YOLO_EUNSUPPORTED;
    if (dtype == YOLO_BF16) return sk_dispatch_t<bf16_t>(a, algo, st, nm);
    if (dtype == YOLO_F16) return sk_dispatch_t<f16_t>(a, algo, st, nm);
    return sk_dispatch_t<float>(a, algo, st, nm);
}
