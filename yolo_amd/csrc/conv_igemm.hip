// Implicit-GEMM convolution for gfx950 (MI355X): Conv(k=1|3, stride 1|2, pad k/2) + folded BN +
// LeakyReLU (+ residual), NHWC activations, MFMA 32x32x16 bf16 / 32x32x2 f32.
//
// Replaces gluoncv _conv2d = Convolution + BatchNorm + LeakyReLU (call sites
// yolo_modules/basic_yolo.py:20,24,26,118,121), DarknetBasicBlockV3's residual add, and
// YOLOOutput's Conv2D + bias + transpose + reshape (basic_yolo.py:98-103).
//
// GEMM view: D[cout][pixel] = sum_k Wp[cout][k] * X[k][pixel], k = (tap, cin).
//   A operand = packed weights, B operand = activations, both K-contiguous in LDS as rows of
//   one 64-byte K-chunk (32 bf16 / 16 f32 channels), 16-byte units XOR-swizzled by
//   (row>>2)&3 so ds_read_b128 of 32 consecutive rows is bank-conflict free.
//   D rows = cout, so each lane ends with 4 consecutive output channels of one pixel
//   -> 8/16-byte NHWC stores.
// No im2col: for a 3x3 conv the block stages ONE zero-padded halo tile of the input per K-chunk
//   ("padded strip" image: rows of the stacked batch with a zero row between images and zero
//   columns at the image edges) and the 9 taps read it at uniform slot offsets kh*PW+kw.
#include "common.h"
#include "conv_args.h"
#include "conv_epilogue.h"
#include <stdio.h>
#include <stdlib.h>


template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
    static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                    __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Frag<f16_t> {
    static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Frag<bf16x3_t> : Frag<bf16_t> {};      // (split bf16: the bf16 MFMA over three K passes, common.h)
template <> struct Frag<f16x3_t> : Frag<f16_t> {};
template <> struct Frag<float> {
    // 16 bytes = 4 f32 per lane half -> four 32x32x2 steps; step s contracts k = {s, 4+s} of
    // the 8 channels in this 32-byte sub-chunk (same mapping on A and B, so any order is exact).
    static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

// KS: kernel size; S: stride; block = WAVES_P x WAVES_C waves (=4), wave tile = NI*32 pixels x
// MI*32 couts; XSLOTS: LDS input slots (64 B each) per plane; TS: K-steps ("taps") per barrier
// round: the 3 kw taps of one kh row for KS=3, TS consecutive K-chunks for KS=1.
template <typename T, int KS, int S, int WAVES_P, int WAVES_C, int MI, int NI, int XSLOTS, int TS, int STATS = 0>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    static_assert(WAVES_P * WAVES_C == 4, "256 threads");
    static_assert(KS == 1 || TS == 3, "3x3: one kh row per step");
    constexpr int BP = WAVES_P * NI * 32;
    constexpr int BC = WAVES_C * MI * 32;
    constexpr int XPLANES = (KS == 1) ? TS : 1;
    constexpr int X_UNITS = XPLANES * XSLOTS * 4;
    constexpr int W_UNITS = TS * BC * 4;
    constexpr int XP = (X_UNITS + 255) / 256;
    constexpr int WP = (W_UNITS + 255) / 256;
    constexpr int KSTEPS = (KS == 3) ? 3 : 1;
    constexpr int NTAP = (KS == 3) ? 9 : 1;
    static_assert(KS == 3 || XSLOTS == BP, "1x1: one slot per pixel");

    constexpr int STAGE_BYTES = (X_UNITS + W_UNITS) * 16;
    constexpr int EPI_BYTES = 4 * YOLO_EPI_WAVE_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES];
    char* Xl = smem;
    char* Wl = smem + X_UNITS * 16;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wave_p = wave % WAVES_P, wave_c = wave / WAVES_P;

    const int bid = blockIdx.x;
    const int tile_c = bid % a.tiles_c;
    const int tile_p = bid / a.tiles_c;
    const int strip = tile_p / a.tiles_per_strip;
    const int i0 = (tile_p - strip * a.tiles_per_strip) * a.tile_px;       // (tile_px = BP except for row-limited tiles, launch_cfg)
    const int co0 = tile_c * BC;
    const int H = a.H, W = a.W, Ho = a.Ho, Wo = a.Wo, TWt = a.TWt, PW = a.PW;
    const int row_bytes = a.Cin * (int)sizeof(T);           // channel extent of a pixel
    const int row_pitch = a.x_ps * (int)sizeof(T);          // pitch of an input pixel

    // ---- geometry of the staged input tile ----------------------------------------------
    int Rin_lo = 0, HS = XSLOTS, x0 = 0;
    if constexpr (KS == 3) {
        const int i_last = min(i0 + a.tile_px, a.total_i) - 1;
        const int r_first = i0 / TWt, r_last = i_last / TWt;
        const int n_f = r_first / Ho, n_l = r_last / Ho;
        Rin_lo = n_f * (H + 1) + (r_first - n_f * Ho) * S;
        const int Rin_hi = n_l * (H + 1) + (r_last - n_l * Ho) * S + 2;
        HS = (Rin_hi - Rin_lo + 1) * PW;
        x0 = strip * TWt * S - 1;
    }

    // ---- per-thread staging descriptors -------------------------------------------------
    long long xoff[XP];
    int xlp[XP];
#pragma unroll
    for (int j = 0; j < XP; ++j) {
        const int u = tid + j * 256;
        if constexpr (KS == 3) {
            // (stride 1: the conflict-free halo layout of conv_pipe.hip launch_pipe -- pitch TWt + 4, unit swizzle ((slot >> 2) - halo
            //  row) & 3; a.row_swz == 0: the plain pitch and the slot swizzle.  Stride 2 keeps the plain layout: its instantiation sits at 256 registers, and the
            //  address terms of conv_pipe's stride-2 layout took it to 268 = one wave per SIMD: the 64 -> 128 layer 157 us instead of 113)
            const int pos = u >> 2, part = u & 3;
            const int slot = pos;
            bool valid = slot < HS && u < X_UNITS;
            const int rr = slot / PW;
            const int cc = slot - rr * PW;
            if constexpr (S == 1) valid = valid && cc < TWt + 2;          // (beyond: the pitch padding)
            const int Rr = Rin_lo + rr;
            const int n = Rr / (H + 1);
            const int yy = Rr - n * (H + 1) - 1;
            const int xx = x0 + cc;
            valid = valid && yy >= 0 && n < a.N && xx >= 0 && xx < W;
            xoff[j] = valid ? ((long long)(n * H + yy) * W + xx) * row_pitch : -1;
            xlp[j] = (part ^ (((pos >> 2) - (S == 1 ? rr * a.row_swz : 0)) & 3)) * 16;
        } else {
            const int plane = u / (XSLOTS * 4);
            const int rem = u - plane * (XSLOTS * 4);
            const int slot = rem >> 2, part = rem & 3;
            const int i = i0 + slot;
            const bool valid = (i < a.total_i) && (u < X_UNITS);
            xoff[j] = valid ? (long long)i * row_pitch : -1;
            xlp[j] = plane * 64 + (part ^ ((slot >> 2) & 3)) * 16;
        }
    }

    // ---- per-lane MFMA operand addresses --------------------------------------------------
    int addrX[NI][NTAP];
    long long yoff[NI], roff[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int i = i0 + (wave_p * NI + ni) * 32 + l31;
        bool ok = i < a.total_i && i - i0 < a.tile_px;
        const int ii = ok ? i : i0;
        int n, pix;
        if constexpr (KS == 3) {
            const int r = ii / TWt;
            const int tx = ii - r * TWt;
            if (strip * TWt + tx >= Wo) ok = false;            // (a ragged last strip: launch_cfg's fallback for widths no strip divides)
            n = r / Ho;
            const int oy = r - n * Ho;
            const int hrow = n * (H + 1) + oy * S - Rin_lo;
            const int slot00 = hrow * PW + tx * S;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int slot = slot00 + (t / 3) * PW + (t % 3);
                if constexpr (S == 1) addrX[ni][t] = slot * 64 + ((h ^ (((slot >> 2) - (hrow + t / 3) * a.row_swz) & 3)) << 4);
                else addrX[ni][t] = slot * 64 + ((h ^ ((slot >> 2) & 3)) << 4);
            }
            pix = oy * Wo + strip * TWt + tx;
            if (a.up2) pix = oy * 4 * Wo + 2 * (strip * TWt + tx);
        } else {
            const int slot = ii - i0;
            addrX[ni][0] = slot * 64 + ((h ^ ((slot >> 2) & 3)) << 4);
            n = ii / (Ho * Wo);
            pix = ii - n * (Ho * Wo);
            if (a.up2) {
                const int oy = pix / Wo;
                pix = oy * 4 * Wo + 2 * (pix - oy * Wo);
            }
        }
        yoff[ni] = ok ? (long long)n * a.y_bs + (long long)pix * a.y_ps : -1;
        roff[ni] = (long long)n * a.r_bs + (long long)pix * a.r_ps;
    }
    const int aoff0 = (wave_c * MI * 32 + l31) * 64 + ((h ^ ((l31 >> 2) & 3)) << 4);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    uint4 xr[XP], wr[WP];
    // split type: K-chunk c of the three passes [x_hi | x_lo | x_hi] sits at c * 64 + (c >= n ? adj1 : 0) + (c >= 2n ? adj2 : 0) bytes
    // inside the pixel (conv_args.h); its planes are padded to whole chunks, so no ragged-K test (TS == 1 for the 1x1: launch_shape)
    static_assert(!IsSplit<T>::value || KS == 3 || TS == 1, "split type: one K chunk per step");
    auto chunk_byte = [&](int c) -> int {
        if constexpr (IsSplit<T>::value) return c * 64 + (c >= a.x3_n ? a.x3_adj1 : 0) + (c >= 2 * a.x3_n ? a.x3_adj2 : 0);
        else return c * 64;
    };
#define LOAD_X(chunk_base)                                                                      \
    _Pragma("unroll") for (int j = 0; j < XP; ++j) {                                            \
        const int byte = chunk_byte(chunk_base) + xlp[j];                                       \
        uint4 v = make_uint4(0, 0, 0, 0);                                                       \
        if (xoff[j] >= 0 && (IsSplit<T>::value || byte < row_bytes)) v = *(const uint4*)(a.x + xoff[j] + byte); \
        xr[j] = v;                                                                              \
    }
    // plane0 = first (chunk*taps + tap) plane of the packed weights for the step
    const char* wsrc = a.wp + (long long)co0 * 64 + tid * 16;
    const long long wplane = (long long)a.Cout_pad * 64;
#define LOAD_W(plane0)                                                                          \
    _Pragma("unroll") for (int j = 0; j < WP; ++j) {                                            \
        const int u = tid + j * 256;                                                            \
        const int t = (j * 256) / (BC * 4);                                                     \
        uint4 v = make_uint4(0, 0, 0, 0);                                                       \
        if (W_UNITS % 256 == 0 || u < W_UNITS)                                                  \
            v = *(const uint4*)(wsrc + ((plane0) + t) * wplane + (j * 256 - t * BC * 4) * 16);  \
        wr[j] = v;                                                                              \
    }
    static_assert((BC * 4) % 256 == 0 || 256 % (BC * 4) == 0, "tap index must be uniform per pass");

    int nouter = (KS == 3) ? a.nchunks : a.nchunks / TS;
    LOAD_X(0);
    LOAD_W(0);
    for (int c = 0; c < nouter; ++c) {
#pragma unroll
        for (int kh = 0; kh < KSTEPS; ++kh) {
            __syncthreads();
            if (kh == 0) {
#pragma unroll
                for (int j = 0; j < XP; ++j) {
                    const int u = tid + j * 256;
                    if ((X_UNITS % 256 == 0 || u < X_UNITS)) *(uint4*)(Xl + u * 16) = xr[j];
                }
            }
#pragma unroll
            for (int j = 0; j < WP; ++j) {
                const int u = tid + j * 256;
                if ((W_UNITS % 256 == 0 || u < W_UNITS)) *(uint4*)(Wl + u * 16) = wr[j];
            }
            __syncthreads();
            // prefetch the next step into registers while this one computes
            if (kh + 1 < KSTEPS) {
                LOAD_W((c * KSTEPS + kh + 1) * TS);
            } else if (c + 1 < nouter) {
                LOAD_W((c + 1) * KSTEPS * TS);
                LOAD_X((KS == 3) ? (c + 1) : (c + 1) * TS);
            }
#pragma unroll
            for (int t = 0; t < TS; ++t) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    uint4 af[MI], bf[NI];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        af[mi] = *(const uint4*)(Wl + t * BC * 64 + mi * 2048 + (aoff0 ^ (ks * 32)));
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const int ax = (KS == 3) ? addrX[ni][kh * 3 + t] : addrX[ni][0] + t * XSLOTS * 64;
                        bf[ni] = *(const uint4*)(Xl + (ax ^ (ks * 32)));
                    }
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            Frag<T>::mma(af[mi], bf[ni], acc[mi][ni]);
                        }
                }
            }
        }
    }

    // ---- epilogue (conv_epilogue.h): every wave transposes its slab through its own LDS scratch ------
    __syncthreads();                         // all waves are done reading the staged tiles
    float* srow = STATS ? a.stats + ((size_t)tile_p * WAVES_P + wave_p) * 2 * a.Cout_pad : nullptr;     // (conv_epilogue.h)
    conv_epilogue<T, MI, NI, STATS>(acc, yoff, smem + wave * YOLO_EPI_WAVE_BYTES, a, co0 + wave_c * MI * 32, lane, roff, srow);
}

// ------------------------------------------------------------------------------------------
// Host side: tile selection and launch
// ------------------------------------------------------------------------------------------
constexpr int XSLOTS_S1 = 384;
constexpr int XSLOTS_S2 = 832;


template <typename T, int KS, int S, int WAVES_P, int WAVES_C, int MI, int NI, int XSLOTS, int TS>
static int launch_cfg(ConvArgs& a, hipStream_t st, const NameOut* name) {
    constexpr int BP = WAVES_P * NI * 32, BC = WAVES_C * MI * 32;
    a.row_swz = 0;
    a.TP = 0;
    if (KS == 3) {
        int best = -1, best_hs = 1 << 30, best_pad = 0;
        static const int rsw_off = (int)YOLO_LAB_ENV("YOLO_NO_ROW_SWZ", 0);       // (lab A/B)
        for (int pass = (rsw_off || S == 2) ? 1 : 0; pass < 2 && best < 0; ++pass) {        // (the padded pitch of the conflict-free layout where it fits, else the plain one)
            for (int d = 1; d <= a.Wo; ++d) {
                if (a.Wo % d) continue;
                const int pad = pass ? 0 : 2;
                const int hs = conv_halo_slots(BP, d, a.Ho, a.H, S, (long long)a.N * a.Ho, 3, pad);
                if (hs <= XSLOTS && hs <= best_hs) { best = d; best_hs = hs; best_pad = pad; }
            }
            if (best >= 0) a.row_swz = pass ? 0 : 1;
        }
        a.tile_px = BP;
        if (best < 0) {
            // No divisor of the width gives strips whose 128-pixel tiles fit the staged halo (prime widths on tall batches, maps a
            // pixel or two wide): strips of ANY width -- the last one ragged, its missing columns masked in the kernel -- and tiles
            // limited to whole rows of the strip, as many as the halo holds.  The widest strip that works wins.
            for (int d = a.Wo < BP ? a.Wo : BP; d >= 1 && best < 0; --d)
                for (int rows = BP / d; rows >= 1; --rows)
                    if (conv_halo_slots(rows * d, d, a.Ho, a.H, S, (long long)a.N * a.Ho) <= XSLOTS) {
                        best = d;
                        a.tile_px = rows * d;
                        break;
                    }
            if (best < 0) return YOLO_EUNSUPPORTED;
        }
        a.TWt = best;
        a.PW = (best - 1) * S + 3 + (a.row_swz ? best_pad : 0);
    } else {
        a.TWt = a.Wo;
        a.PW = a.Wo;
        a.tile_px = BP;
    }
    a.nstrips = (a.Wo + a.TWt - 1) / a.TWt;
    const long long tot = (long long)a.N * a.Ho * a.TWt;
    if (tot > 0x7fffffffLL) return YOLO_EUNSUPPORTED;
    a.total_i = (int)tot;
    a.tiles_per_strip = (a.total_i + a.tile_px - 1) / a.tile_px;
    a.tiles_c = (a.Cout + BC - 1) / BC;
    const long long grid = (long long)a.nstrips * a.tiles_per_strip * a.tiles_c;
    if (grid > 0x7fffffffLL) return YOLO_EUNSUPPORTED;
    // BatchNorm forward sums in the epilogue (stats_mode 1 only here; conv_epilogue.h)
    constexpr bool kStats = IsBf16<T>::value;
    const bool stats_ok = kStats && a.stats_mode == 1 && !a.out_f32 && !a.d2s && !a.up2 && (a.Cout % 8) == 0 && (a.y_ps % 8) == 0 &&
                          (a.y_bs % 8) == 0;
    if (a.stats && !stats_ok) return YOLO_EUNSUPPORTED;
    if (name) {
        if (a.stats)
            snprintf(name->buf, name->len, "void conv_igemm_kernel<%s, %d, %d, %d, %d, %d, %d, %d, %d, 1>(ConvArgs)",
                     Elem<T>::name, KS, S, WAVES_P, WAVES_C, MI, NI, XSLOTS, TS);
        else
            snprintf(name->buf, name->len, "void conv_igemm_kernel<%s, %d, %d, %d, %d, %d, %d, %d, %d>(ConvArgs)",
                     Elem<T>::name, KS, S, WAVES_P, WAVES_C, MI, NI, XSLOTS, TS);
        if (name->stats_rows) *name->stats_rows = a.stats ? a.nstrips * a.tiles_per_strip * WAVES_P : -1;
        return YOLO_OK;
    }
    if constexpr (kStats) {
        if (a.stats) {
            YOLO_LAUNCH((conv_igemm_kernel<T, KS, S, WAVES_P, WAVES_C, MI, NI, XSLOTS, TS, 1>), dim3((unsigned)grid), dim3(256), 0, st, a);
            YOLO_LAUNCH_CHECK();
            return YOLO_OK;
        }
    }
    YOLO_LAUNCH((conv_igemm_kernel<T, KS, S, WAVES_P, WAVES_C, MI, NI, XSLOTS, TS>), dim3((unsigned)grid),
                       dim3(256), 0, st, a);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

template <typename T, int WAVES_P, int WAVES_C, int MI, int NI>
static int launch_shape(ConvArgs& a, int ks, int stride, hipStream_t st, const NameOut* nm) {
    constexpr int BP = WAVES_P * NI * 32;
    if (ks == 3 && stride == 1) return launch_cfg<T, 3, 1, WAVES_P, WAVES_C, MI, NI, XSLOTS_S1, 3>(a, st, nm);
    if (ks == 3 && stride == 2) return launch_cfg<T, 3, 2, WAVES_P, WAVES_C, MI, NI, XSLOTS_S2, 3>(a, st, nm);
    if (ks == 1 && stride == 1) {
        if constexpr (IsSplit<T>::value) {
            return launch_cfg<T, 1, 1, WAVES_P, WAVES_C, MI, NI, BP, 1>(a, st, nm);      // (one chunk per step: conv_igemm_kernel)
        } else {
            if (a.nchunks % 4 == 0) return launch_cfg<T, 1, 1, WAVES_P, WAVES_C, MI, NI, BP, 4>(a, st, nm);
            if (a.nchunks % 2 == 0) return launch_cfg<T, 1, 1, WAVES_P, WAVES_C, MI, NI, BP, 2>(a, st, nm);
            return launch_cfg<T, 1, 1, WAVES_P, WAVES_C, MI, NI, BP, 1>(a, st, nm);
        }
    }
    return YOLO_EUNSUPPORTED;
}

template <typename T>
static int launch_dtype(ConvArgs& a, int ks, int stride, hipStream_t st, const NameOut* nm) {
    if (a.Cout > 64) return launch_shape<T, 2, 2, 2, 2>(a, ks, stride, st, nm);   // 128 px x 128 cout
    return launch_shape<T, 2, 2, 1, 2>(a, ks, stride, st, nm);                    // 128 px x  64 cout
}

// conv_epilogue.h form: 1 = unconditional buffer accesses (the default), 0 = the branching form (also what extents beyond 31 bits
// get).  Lab knob: YOLO_NO_BUF32 -> 0.
static int conv_buf32_form() {
    static const int v = YOLO_LAB_SET("YOLO_NO_BUF32") ? 0 : 1;
    return v;
}

static int conv_dispatch(const yolo_conv_desc* d, void* stream, const NameOut* nm);

// Heuristic choice of the pipelined variant for algo == 0 (1 = stay on the generic kernel): minimise
// rounds x tile work, where rounds = ceil(tiles / (256 CUs x resident blocks per CU)) -- at the
// reference's batch sizes tile quantisation on the 13x13 / 26x26 maps costs more than any difference
// between the variants' inner loops.  The per-variant factors are measured (tools/conv_bench.py).
static int conv_auto_algo(const ConvArgs& a, int ks, int stride, int dtype, bool px_heavy = true) {
    if ((!dtype_split(dtype) && (a.Cin * elem_size(dtype)) % 64) || (ks == 1 && a.nchunks < 2)) return 1;      // (split planes are padded to whole chunks)
    if (dtype_split(dtype)) {
        // split types (round 6, measured at 416x416 bs 32, tools/layer_times.py --dtype bf16x3): the first stages' narrow layers are not
        // covered by fused / streaming kernels there -- the 64-cout tiles for them (1x1 64 -> 32 at 208^2: 129 us against 567 on a
        // 256-cout tile), the generic kernel for the large-map stride-2 layers (32 -> 64 at 416^2: 432 us against 602)
        if (ks == 1 && a.Cout <= 64) return 40;
        if (ks == 3 && stride == 1 && a.Cout <= 64) return 41;
        if (ks == 3 && stride == 2 && a.Cin <= 64) return 1;
    }
    if (stride == 2) return ks == 3 ? (a.Cout > 128 ? 18 : 9) : 1;      // (18 = 10 with the 4-slot weight ring)
    struct V { int algo, bp, bc, bpc; float f; bool k1; };
    // (round 5: the pixel-heavy 3x3 tiles 27 / 28 -- ~40 % less L2 -> LDS weight stream per output, measured 0.5-3.4 % faster than
    //  6 / 2 wherever they fill the chip, tools/ab_tiles.sh)
    static const V vs[] = {{2, 256, 256, 1, 1.00f, true}, {3, 256, 128, 1, 1.10f, true}, {4, 128, 128, 2, 1.05f, true},
                           {6, 192, 256, 1, 1.00f, false}, {8, 192, 128, 2, 1.05f, true},
                           {27, 384, 128, 1, 0.98f, false}, {28, 512, 128, 1, 0.97f, false}};
    const long long px = (long long)a.N * a.Ho * a.Wo;
    int best = 1;
    double best_cost = 1e30;
    for (const V& v : vs) {
        if (ks == 1 && !v.k1) continue;
        if (!px_heavy && v.algo >= 27) continue;
        const long long tiles = ((px + v.bp - 1) / v.bp) * ((a.Cout + v.bc - 1) / v.bc);
        const long long slots = 256LL * v.bpc;
        // full rounds keep every CU's bpc block slots busy (the co-resident blocks share the CU: bpc block times); the last,
        // partial round puts ceil(rest / 256) blocks on the busiest CU -- a launch of fewer tiles than CUs runs one block per CU
        // at full speed whatever bpc is (13^2 512 -> 1024 at batch 32: 232 tiles of the two-per-CU 192 x 128 tile beat 120 of
        // the 384 x 128 one, 50.7 against 64.2 us)
        const long long full = tiles / slots, rest = tiles - full * slots;
        const double cost = (double)(full * v.bpc + (rest + 255) / 256) * v.bp * v.bc * v.f;
        if (cost < best_cost) { best_cost = cost; best = v.algo; }
    }
    return best;
}

extern "C" int yolo_conv_fwd(const yolo_conv_desc* d, void* stream) { return conv_dispatch(d, stream, nullptr); }

extern "C" int yolo_conv_kernel_name(const yolo_conv_desc* d, char* buf, int len) {
    if (!buf || len < 16) return YOLO_EINVAL;
    NameOut nm{buf, len, nullptr};
    return conv_dispatch(d, nullptr, &nm);
}

// Number of partial rows ([rows][2][yolo_padded_channels(Cout)] float32) yolo_conv_fwd writes into d->stats for this
// descriptor, or YOLO_EUNSUPPORTED when the kernel it would run has no statistics epilogue.  Host-only, no launch
// (d->stats only has to be non-NULL).
extern "C" int yolo_conv_stats_rows(const yolo_conv_desc* d) {
    if (!d || !d->stats) return YOLO_EINVAL;
    char buf[256];
    int rows = -1;
    NameOut nm{buf, (int)sizeof(buf), &rows};
    const int rc = conv_dispatch(d, nullptr, &nm);
    if (rc != YOLO_OK) return rc;
    return rows > 0 ? rows : YOLO_EUNSUPPORTED;
}

static int conv_dispatch(const yolo_conv_desc* d, void* stream, const NameOut* nm) {
    if (!d || !d->x || !d->w_packed || !d->y || (!d->scale != !d->bias)) return YOLO_EINVAL;      // (both NULL: identity epilogue)
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) return YOLO_EINVAL;
    if (d->ksize != 1 && d->ksize != 3) return YOLO_EUNSUPPORTED;
    if (d->stride != 1 && d->stride != 2) return YOLO_EUNSUPPORTED;
    if (d->ksize == 1 && d->stride != 1) return YOLO_EUNSUPPORTED;
    if (!dtype_valid(d->dtype)) return YOLO_EINVAL;
    if (!(d->slope >= 0.f && d->slope <= 1.f)) return YOLO_EINVAL;       // LeakyReLU is computed as max(t, t*slope)
    const int es = elem_size(d->dtype);
    const bool split = dtype_split(d->dtype);                         // (YOLO_BF16X3: common.h)
    const int planes = dtype_planes(d->dtype);
    if ((d->Cin * es) % 16) return YOLO_EUNSUPPORTED;                 // 16-byte K units
    if (!d->out_f32 && (d->Cout % 4)) return YOLO_EUNSUPPORTED;       // 4-channel store groups
    if (split && (d->stats || d->tail_w_packed || (!d->out_f32 && (d->Cout % 8)))) return YOLO_EUNSUPPORTED;
    // split tensors: each plane of a pixel is padded to whole 32-channel K-chunks (include/yolo_amd.h); the pad channels read zeros
    const int cin_p = split ? round_up(d->Cin, 32) : d->Cin, cout_p = (split && !d->out_f32) ? round_up(d->Cout, 32) : d->Cout;
    const int pad = d->ksize / 2;
    ConvArgs a;
    a.x = (const char*)d->x;
    a.wp = (const char*)d->w_packed;
    a.scale = d->scale;
    a.bias = d->bias;
    a.res = (const char*)d->residual;
    a.y = (char*)d->y;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
    a.Ho = (d->H + 2 * pad - d->ksize) / d->stride + 1;
    a.Wo = (d->W + 2 * pad - d->ksize) / d->stride + 1;
    a.Cout_pad = round_up(d->Cout, YOLO_COUT_PAD);
    a.nchunks = (cin_p * es + 63) / 64 * dtype_kpasses(d->dtype);
    a.out_f32 = d->out_f32;
    a.d2s = 0;
    a.halo_strict = 0;
    a.up2 = d->upsample2x ? 1 : 0;
    if (d->x_pixel_stride < 0 || (d->x_pixel_stride && d->x_pixel_stride < (long long)cin_p * planes) || d->x_pixel_stride > 0x7fffffffLL) return YOLO_EINVAL;
    a.x_ps = d->x_pixel_stride ? (int)d->x_pixel_stride : cin_p * planes;
    a.x3_n = 0; a.x3_adj1 = 0; a.x3_adj2 = 0; a.y_lo = 0; a.r_lo = 0;
    if (split) {
        const long long xlo = d->x_lo_offset ? d->x_lo_offset : cin_p;
        const long long ylo = d->y_lo_offset ? d->y_lo_offset : cout_p;
        if (xlo < cin_p || xlo + d->Cin > a.x_ps || (xlo * es) % 16 || xlo * es > 0x3fffffffLL) return YOLO_EINVAL;      // (the plane offsets are 32-bit byte adjustments in the kernels)
        if (!d->out_f32 && (ylo < d->Cout || (ylo * es) % 16)) return YOLO_EINVAL;      // (fp32 logits are not split: y_lo unused)
        a.x3_n = cin_p * es / 64;
        a.x3_adj1 = (int)(xlo * es) - a.x3_n * 64;
        a.x3_adj2 = -a.x3_n * 64 - (int)(xlo * es);
        a.y_lo = ylo; a.r_lo = cout_p;
    }
    if ((a.x_ps * es) % 16) return YOLO_EUNSUPPORTED;                 // 16-byte aligned pixel rows
    a.slope = d->slope;
    const int oplanes = d->out_f32 ? 1 : planes;                          // (fp32 logits are not split)
    a.y_ps = d->y_pixel_stride ? d->y_pixel_stride : cout_p * oplanes;
    if (split && !d->out_f32 && a.y_lo + d->Cout > a.y_ps) return YOLO_EINVAL;
    a.y_bs = d->y_batch_stride ? d->y_batch_stride : (long long)a.Ho * a.Wo * a.y_ps * (a.up2 ? 4 : 1);
    a.r_ps = cout_p * planes; a.r_bs = (long long)a.Ho * a.Wo * a.r_ps;  // the residual is dense
    if (a.res && d->out_f32) return YOLO_EUNSUPPORTED;
    if (a.up2 && (a.res || d->out_f32)) return YOLO_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (d->algo < 0) return YOLO_EINVAL;
    a.stats = (float*)d->stats; a.stats_mode = d->stats_mode;
    a.s_y = (const char*)d->stats_y; a.s_mean = d->stats_mean; a.s_invstd = d->stats_invstd;
    a.s_gamma = d->stats_gamma; a.s_beta = d->stats_beta; a.s_slope = d->stats_slope;
    a.t_wp = (const char*)d->tail_w_packed; a.t_scale = d->tail_scale; a.t_bias = d->tail_bias; a.t_y = (char*)d->tail_y;
    a.t_cout = d->tail_cout; a.t_out_f32 = d->tail_out_f32; a.t_slope = d->tail_slope;
    a.t_y_ps = d->tail_y_pixel_stride ? d->tail_y_pixel_stride : d->tail_cout;
    a.t_y_bs = d->tail_y_batch_stride ? d->tail_y_batch_stride : (long long)a.Ho * a.Wo * a.t_y_ps;
    {
        // extents in bytes of everything the epilogue addresses (conv_epilogue.h buf32)
        const long long lim = 0x7fffffffLL;
        const long long yb = (long long)a.N * a.y_bs * (a.out_f32 ? 4 : es);
        const long long rbytes = a.res ? (long long)a.N * a.r_bs * es : 0;
        const long long tb = a.t_wp ? (long long)a.N * a.t_y_bs * (a.t_out_f32 ? 4 : es) : 0;
        a.buf32 = (yb < lim && rbytes < lim && tb < lim) ? conv_buf32_form() : 0;
    }
    { static const int lab = (int)YOLO_LAB_ENV("YOLO_EPI_AB", 0); a.lab = lab; }
    if (a.t_wp) {
        // fused tail 1x1: pipelined 3x3 variants with 256-cout tiles only (conv_pipe.hip)
        if (!d->tail_y || (!d->tail_scale != !d->tail_bias) || d->tail_cout <= 0 || !(d->tail_slope >= 0.f && d->tail_slope <= 1.f)) return YOLO_EINVAL;
        if (a.stats || elem_size(d->dtype) != 2 || d->ksize != 3) return YOLO_EUNSUPPORTED;
        if (!d->tail_out_f32 && (d->tail_cout % 4)) return YOLO_EUNSUPPORTED;
        if (d->algo) return conv_pipe_dispatch(a, d->ksize, d->stride, d->dtype, d->algo, st, nm);
        // (tiles that hold every channel of a pixel: 128-cout tiles first when Cout <= 128, else the 256-cout ones)
        static const int s1a[] = {7, 6, 2}, s2a[] = {17, 9, 18, 16, 10}, s1b[] = {6, 2, 0}, s2b[] = {18, 16, 10, 0, 0};
        const bool small = d->Cout <= 128;
        const int* cand = d->stride == 2 ? (small ? s2a : s2b) : (small ? s1a : s1b);
        const int ncand = d->stride == 2 ? (small ? 5 : 3) : (small ? 3 : 2);
        for (int k = 0; k < ncand; ++k) {
            ConvArgs b = a;
            const int rc = conv_pipe_dispatch(b, d->ksize, d->stride, d->dtype, cand[k], st, nm);
            if (rc != YOLO_EUNSUPPORTED) return rc;
        }
        return YOLO_EUNSUPPORTED;
    }
    if (a.stats) {
        // BatchNorm statistics in the epilogue: the pipelined kernels only (yolo_conv_stats_rows tells the caller beforehand)
        if (a.stats_mode != 1 && a.stats_mode != 2) return YOLO_EINVAL;
        if (a.stats_mode == 2 && (!a.s_y || !a.s_mean || !a.s_invstd || !a.s_gamma || !a.s_beta)) return YOLO_EINVAL;
        int algo = d->algo;
        if (algo == 13 || algo == 14) return conv_stream_dispatch(a, d->ksize, d->stride, d->dtype, algo, st, nm);
        if (algo == 0) {
            if (d->ksize == 3 && d->Cin <= 64 && !(d->stride == 2 && d->Cin == 64)) {          // (as in the plain path below)
                const int rc = conv_stream_dispatch(a, d->ksize, d->stride, d->dtype, 13, st, nm);
                if (rc != YOLO_EUNSUPPORTED) return rc;
            }
            algo = conv_auto_algo(a, d->ksize, d->stride, d->dtype);
        }
        if (algo >= 2) {
            ConvArgs b = a;
            int rc = conv_pipe_dispatch(b, d->ksize, d->stride, d->dtype, algo, st, nm);
            if (rc == YOLO_EUNSUPPORTED && !d->algo && algo >= 27) {       // (halo of the pixel-heavy tile does not fit: the next best)
                b = a;
                rc = conv_pipe_dispatch(b, d->ksize, d->stride, d->dtype, conv_auto_algo(a, d->ksize, d->stride, d->dtype, false), st, nm);
            }
            if (rc != YOLO_EUNSUPPORTED || d->algo) return rc;
        }
        if (a.stats_mode != 1 || d->dtype != YOLO_BF16) return YOLO_EUNSUPPORTED;   // generic kernel: forward sums, bf16
        return launch_dtype<bf16_t>(a, d->ksize, d->stride, st, nm);
    }
    if (d->algo == 13 || d->algo == 14) return conv_stream_dispatch(a, d->ksize, d->stride, d->dtype, d->algo, st, nm);
    if (d->algo >= 30 && d->algo <= 35) return conv_sk_dispatch(a, d->ksize, d->stride, d->dtype, d->algo, st, nm);
    if (d->algo >= 2) return conv_pipe_dispatch(a, d->ksize, d->stride, d->dtype, d->algo, st, nm);
    if (d->algo == 0) {
        // small-channel 3x3 layers: the streaming kernel (measured 1.2-1.45x the generic one; the 1x1s and the
        // 64->128 stride-2 layer are a wash and stay where they were)
        if (!split && d->ksize == 3 && d->Cin <= 64 && !(d->stride == 2 && d->Cin == 64)) {
            const int rc = conv_stream_dispatch(a, d->ksize, d->stride, d->dtype, 13, st, nm);
            if (rc != YOLO_EUNSUPPORTED) return rc;
        }
        const int pick = conv_auto_algo(a, d->ksize, d->stride, d->dtype);
        if (pick >= 2) {
            ConvArgs b = a;
            int rc = conv_pipe_dispatch(b, d->ksize, d->stride, d->dtype, pick, st, nm);
            if (rc == YOLO_EUNSUPPORTED && pick == 18) {
                b = a;
                rc = conv_pipe_dispatch(b, d->ksize, d->stride, d->dtype, 10, st, nm);
            }
            if (rc == YOLO_EUNSUPPORTED && pick >= 27) {       // (the halo of a 384 / 512-pixel tile does not fit every map: the next best)
                b = a;
                rc = conv_pipe_dispatch(b, d->ksize, d->stride, d->dtype, conv_auto_algo(a, d->ksize, d->stride, d->dtype, false), st, nm);
            }
            if (rc != YOLO_EUNSUPPORTED) return rc;
        }
    }
    if (d->dtype == YOLO_BF16) return launch_dtype<bf16_t>(a, d->ksize, d->stride, st, nm);
    if (d->dtype == YOLO_F16) return launch_dtype<f16_t>(a, d->ksize, d->stride, st, nm);
    if (d->dtype == YOLO_BF16X3) return launch_dtype<bf16x3_t>(a, d->ksize, d->stride, st, nm);      // (ragged widths, maps a pixel or two wide: the generic kernel)
    if (d->dtype == YOLO_F16X3) return launch_dtype<f16x3_t>(a, d->ksize, d->stride, st, nm);
    return launch_dtype<float>(a, d->ksize, d->stride, st, nm);
}

// Data gradient of a 3x3 stride-2 pad-1 convolution without the zero-dilated dy: the four sub-pixel phases of dx are
// four output-channel blocks of ONE 2x2-window convolution over dy (16 instead of 36 tap-products per dy pixel) whose
// epilogue stores depth-to-space.  Even H/W of dx only (2H x 2W); bf16; pipelined variants only -- callers fall back
// to yolo_dilate2x + yolo_conv_fwd on YOLO_EUNSUPPORTED.
extern "C" int yolo_conv_dgrad_s2(const yolo_conv_desc* d, void* stream) {
    if (!d || !d->x || !d->w_packed || !d->y || (!d->scale != !d->bias)) return YOLO_EINVAL;      // (both NULL: identity epilogue)
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->algo < 0) return YOLO_EINVAL;
    if (!(d->slope >= 0.f && d->slope <= 1.f)) return YOLO_EINVAL;
    if (d->dtype != YOLO_BF16 || d->out_f32 || d->y_pixel_stride || d->y_batch_stride) return YOLO_EUNSUPPORTED;
    if ((d->Cout % 32) || (d->Cin * 2) % 64) return YOLO_EUNSUPPORTED;      // a lane's 8 couts stay inside one phase
    ConvArgs a;
    a.x = (const char*)d->x; a.wp = (const char*)d->w_packed; a.scale = d->scale; a.bias = d->bias;
    a.res = (const char*)d->residual; a.y = (char*)d->y;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
    a.Ho = d->H; a.Wo = d->W;
    a.Cout_pad = round_up(d->Cout, YOLO_COUT_PAD);
    a.nchunks = (d->Cin * 2 + 63) / 64;
    a.out_f32 = 0; a.d2s = 1; a.up2 = 0; a.x_ps = d->Cin; a.slope = d->slope;
    a.halo_strict = 0;
    a.stats = nullptr; a.stats_mode = 0;
    a.t_wp = nullptr;
    a.x3_n = 0; a.x3_adj1 = 0; a.x3_adj2 = 0; a.y_lo = 0; a.r_lo = 0;
    a.lab = 0;
    a.buf32 = ((long long)a.N * a.Ho * a.Wo * d->Cout * 2 < 0x7fffffffLL) ? conv_buf32_form() : 0;      // (dx: N x 2Ho x 2Wo x Cout/4, residual = dx)
    if (d->x_pixel_stride || d->upsample2x || d->stats) return YOLO_EUNSUPPORTED;
    a.y_ps = d->Cout / 4;
    a.y_bs = (long long)a.Ho * a.Wo * d->Cout;
    a.r_ps = a.y_ps; a.r_bs = a.y_bs;                    // (accumulation into dx: same depth-to-space addressing)
    hipStream_t st = (hipStream_t)stream;
    if (d->algo) return conv_pipe_dispatch(a, 2, 1, d->dtype, d->algo, st, nullptr);
    // rounds x tile cost as in conv_auto_algo, then the first variant whose halo fits
    struct V { int algo, bp, bc, bpc; };
    static const V vs[] = {{2, 256, 256, 1}, {6, 192, 256, 1}, {10, 128, 256, 1}, {4, 128, 128, 2}};      // (algo 4: large regular maps only, conv_pipe.hip)
    const long long px = (long long)a.N * a.Ho * a.Wo;
    double cost[4];
    for (int i = 0; i < 4; ++i) {
        const long long tiles = ((px + vs[i].bp - 1) / vs[i].bp) * ((a.Cout + vs[i].bc - 1) / vs[i].bc);
        const long long slots = 256LL * vs[i].bpc;
        cost[i] = (double)((tiles + slots - 1) / slots) * vs[i].bp * vs[i].bc * vs[i].bpc;
    }
    bool used[4] = {false, false, false, false};
    for (int k = 0; k < 4; ++k) {
        int b = -1;
        for (int i = 0; i < 4; ++i)
            if (!used[i] && (b < 0 || cost[i] < cost[b])) b = i;
        used[b] = true;
        ConvArgs t = a;
        const int rc = conv_pipe_dispatch(t, 2, 1, d->dtype, vs[b].algo, st, nullptr);
        if (rc != YOLO_EUNSUPPORTED) return rc;
    }
    return YOLO_EUNSUPPORTED;
}

// ------------------------------------------------------------------------------------------
// Weight packing: OIHW f32 -> [chunk][tap][Cout_pad][64 B], 16-byte units XOR-swizzled
// ------------------------------------------------------------------------------------------
extern "C" long long yolo_packed_weight_bytes(int Cout, int Cin, int ksize, int dtype);

// one element of the packed image: idx over [chunk][tap][Cout_pad][64 B / sizeof(T)]
template <typename T>
__device__ __forceinline__ void pack_one(const float* __restrict__ w, T* __restrict__ out, long long idx, int Cout, int Cin,
                                         int ks, int Cout_pad, int dgrad) {
    constexpr int CH = 64 / sizeof(T);       // channels per chunk
    constexpr int UE = 16 / sizeof(T);       // elements per 16-byte unit
    const int e = (int)(idx % CH);
    long long r = idx / CH;
    const int co = (int)(r % Cout_pad);
    r /= Cout_pad;
    const int tap = (int)(r % (ks * ks));
    int chunk = (int)(r / (ks * ks));
    // split types: three passes over the input channels -- chunks [w_hi | w_hi | w_lo] against the kernel's [x_hi | x_lo | x_hi]
    int pass = 0;
    if constexpr (IsSplit<T>::value) {
        const int n = (Cin + CH - 1) / CH;   // (chunks per pass: the planes are padded to whole chunks, the pad channels' weights are zero)
        pass = chunk / n;
        chunk -= pass * n;
    }
    const int punit = e / UE, within = e % UE;
    const int lunit = punit ^ ((co >> 2) & 3);            // physical unit holds this logical unit
    const int ci = chunk * CH + lunit * UE + within;
    float v = 0.f;
    if (co < Cout && ci < Cin) {
        if (!dgrad) {
            v = w[((long long)(co * Cin + ci) * ks + tap / ks) * ks + tap % ks];
        } else if (dgrad == 1) {
            // data-gradient image: rows = forward INPUT channels, K = forward OUTPUT channels, taps flipped
            // (w is the forward OIHW tensor with O = Cin here, I = Cout here)
            v = w[((long long)(ci * Cout + co) * ks + (ks - 1 - tap / ks)) * ks + (ks - 1 - tap % ks)];
        } else {
            // sub-pixel data gradient of a 3x3 stride-2 pad-1 conv (ks == 2, Cout = 4 x Cin_f, Cin = Cout_f):
            // dx[2m+a][2n+b][c] = sum_{ty,tx,k} W'[(a,b,c)][k][ty][tx] dy[m+ty][n+tx][k].  From y[o] = sum_t w[t] x[2o+t-1]:
            // the even position (a = 0) only meets tap 1 of output m; the odd one meets tap 2 of m and tap 0 of m+1.
            const int Cf = Cout >> 2, ph = co / Cf, cf = co - ph * Cf;
            const int ty = tap >> 1, tx = tap & 1;
            const int ky = (ph >> 1) == 0 ? (ty == 0 ? 1 : -1) : (ty == 0 ? 2 : 0);
            const int kx = (ph & 1) == 0 ? (tx == 0 ? 1 : -1) : (tx == 0 ? 2 : 0);
            if (ky >= 0 && kx >= 0) v = w[((long long)(ci * Cf + cf) * 3 + ky) * 3 + kx];
        }
    }
    if constexpr (IsSplit<T>::value) {
        const uint32_t hi = Elem<T>::pack2(v, 0.f) & 0xffffu;
        ((uint16_t*)out)[idx] = (uint16_t)(pass < 2 ? hi : (Elem<T>::pack2(v - Elem<T>::lo(hi), 0.f) & 0xffffu));
    } else if constexpr (sizeof(T) == 2)
        ((uint16_t*)out)[idx] = (uint16_t)(Elem<T>::pack2(v, 0.f) & 0xffffu);
    else
        out[idx] = v;
}

template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin,
                                    int ks, int Cout_pad, int nchunks, int dgrad) {
    constexpr int CH = 64 / sizeof(T);
    const long long total = (long long)nchunks * ks * ks * Cout_pad * CH;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x)
        pack_one<T>(w, out, idx, Cout, Cin, ks, Cout_pad, dgrad);
}

// every conv of a network in ONE launch (the training step re-packs ~150 images after each optimiser update: as
// separate launches that was 1.3 ms of mostly launch overhead).  Block b belongs to the item i with
// first_block[i] <= b < first_block[i+1] and packs elements [4096*(b - first_block[i]), +4096) of it.
struct PackItem { const float* w; void* packed; int Cout, Cin, ksize, dgrad; };
static_assert(sizeof(PackItem) == 32, "yolo_pack_item layout");
constexpr int PACK_BLOCK_ELEMS = 4096;

template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const PackItem* __restrict__ items,
                                                                 const long long* __restrict__ first_block, int n) {
    constexpr int CH = 64 / sizeof(T);
    int lo = 0, hi = n;                                   // first_block[lo] <= blockIdx.x < first_block[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (first_block[mid] <= (long long)blockIdx.x) lo = mid; else hi = mid;
    }
    const PackItem it = items[lo];
    const int Cout_pad = round_up(it.Cout, YOLO_COUT_PAD);
    const int nchunks = (it.Cin * (int)sizeof(T) + 63) / 64;
    const long long total = (long long)nchunks * it.ksize * it.ksize * Cout_pad * CH;
    const long long base = ((long long)blockIdx.x - first_block[lo]) * PACK_BLOCK_ELEMS;
#pragma unroll 4
    for (int k = 0; k < PACK_BLOCK_ELEMS / 256; ++k) {
        const long long idx = base + k * 256 + threadIdx.x;
        if (idx < total) pack_one<T>(it.w, (T*)it.packed, idx, it.Cout, it.Cin, it.ksize, Cout_pad, it.dgrad);
    }
}

extern "C" long long yolo_pack_batch_blocks(int Cout, int Cin, int ksize, int dtype) {
    if (!dtype_plain(dtype)) return dtype_valid(dtype) ? YOLO_EUNSUPPORTED : YOLO_EINVAL;      // (the training step's batched re-pack)
    const long long bytes = yolo_packed_weight_bytes(Cout, Cin, ksize, dtype);
    if (bytes < 0) return bytes;
    return (bytes / elem_size(dtype) + PACK_BLOCK_ELEMS - 1) / PACK_BLOCK_ELEMS;
}

extern "C" int yolo_pack_conv_weights_batch(const void* items_device, const long long* first_block_device, int n_items,
                                            long long total_blocks, int dtype, void* stream) {
    if (!items_device || !first_block_device || n_items <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffLL)
        return YOLO_EINVAL;
    if (dtype == YOLO_BF16)
        YOLO_LAUNCH(pack_weights_batch_kernel<bf16_t>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                    (const PackItem*)items_device, first_block_device, n_items);
    else if (dtype == YOLO_F16)
        YOLO_LAUNCH(pack_weights_batch_kernel<f16_t>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                    (const PackItem*)items_device, first_block_device, n_items);
    else if (dtype == YOLO_F32)
        YOLO_LAUNCH(pack_weights_batch_kernel<float>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                    (const PackItem*)items_device, first_block_device, n_items);
    else
        return YOLO_EINVAL;
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// The forward AND the data-gradient image of a conv from ONE read of its weights (bf16; Cout, Cin multiples of 32; 1x1 and
// 3x3): the element-per-thread kernel above is bound by its stride-9 4-byte gathers and 2-byte stores (0.77 ms per
// training step for the D53 net, 1.3 TB/s).  Here a block owns 32 cout x 32 cin x taps: the 32 contiguous runs of
// 32 * taps floats are read with 16-byte loads, rounded to bf16 into LDS, and both images leave as whole 16-byte units
// (2 KiB contiguous per tap: 32 rows x 64 B).  Padding rows of the images are never written: the caller zero-fills the
// buffers once.  Bit-identical to yolo_pack_conv_weights + yolo_pack_conv_weights_dgrad.
struct PackPair { const float* w; void* fwd; void* dgrad; int Cout, Cin, ksize, pad_; };
static_assert(sizeof(PackPair) == 40, "yolo_pack_pair layout");

__global__ __launch_bounds__(256) void pack_pairs_kernel(const PackPair* __restrict__ items, const long long* __restrict__ first_block, int n) {
    __shared__ __attribute__((aligned(16))) uint16_t t[32 * 32 * 9];
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (first_block[mid] <= (long long)blockIdx.x) lo = mid; else hi = mid;
    }
    const PackPair it = items[lo];
    const int T = it.ksize * it.ksize;
    const int tiles_ci = it.Cin >> 5;
    const int b = (int)((long long)blockIdx.x - first_block[lo]);
    const int tco = b / tiles_ci, tci = b - tco * tiles_ci;
    const int co0 = tco * 32, ci0 = tci * 32;
    const int run4 = 8 * T;                                  // float4s per cout row of the tile
    for (int i = threadIdx.x; i < 32 * run4; i += 256) {
        const int r = i / run4, q = i - r * run4;
        const float4 v = *(const float4*)(it.w + ((long long)(co0 + r) * it.Cin + ci0) * T + q * 4);
        uint2 pk;
        pk.x = pack_bf16x2(v.x, v.y); pk.y = pack_bf16x2(v.z, v.w);
        *(uint2*)(t + r * 32 * T + q * 4) = pk;              // LDS image = the global one: [cout row][cin][tap]
    }
    __syncthreads();
    const int Cout_pad = round_up(it.Cout, YOLO_COUT_PAD), Cin_pad = round_up(it.Cin, YOLO_COUT_PAD);
    for (int u = threadIdx.x; u < T * 128; u += 256) {
        const int p = u & 3, r = (u >> 2) & 31, tap = u >> 7;
        {   // forward image: row = cout, K = cin (chunk tci), unit p holds logical unit p ^ ((cout >> 2) & 3)
            const int co = co0 + r, lu = p ^ ((co >> 2) & 3);
            const uint16_t* src = t + (r * 32 + lu * 8) * T + tap;
            uint4 o;
            o.x = src[0] | ((uint32_t)src[T] << 16); o.y = src[2 * T] | ((uint32_t)src[3 * T] << 16);
            o.z = src[4 * T] | ((uint32_t)src[5 * T] << 16); o.w = src[6 * T] | ((uint32_t)src[7 * T] << 16);
            *(uint4*)((char*)it.fwd + (((long long)tci * T + tap) * Cout_pad + co) * 64 + p * 16) = o;
        }
        {   // data-gradient image: row = cin, K = cout (chunk tco), taps flipped
            const int ci = ci0 + r, lu = p ^ ((ci >> 2) & 3);
            const uint16_t* src = t + (lu * 8 * 32 + r) * T + tap;
            const int S = 32 * T;
            uint4 o;
            o.x = src[0] | ((uint32_t)src[S] << 16); o.y = src[2 * S] | ((uint32_t)src[3 * S] << 16);
            o.z = src[4 * S] | ((uint32_t)src[5 * S] << 16); o.w = src[6 * S] | ((uint32_t)src[7 * S] << 16);
            *(uint4*)((char*)it.dgrad + (((long long)tco * T + (T - 1 - tap)) * Cin_pad + ci) * 64 + p * 16) = o;
        }
    }
}

extern "C" long long yolo_pack_pair_blocks(int Cout, int Cin, int ksize) {
    if (Cout <= 0 || Cin <= 0) return YOLO_EINVAL;
    if ((Cout % 32) || (Cin % 32) || (ksize != 1 && ksize != 3)) return YOLO_EUNSUPPORTED;
    return (long long)(Cout / 32) * (Cin / 32);
}

extern "C" int yolo_pack_conv_weights_pairs(const void* items_device, const long long* first_block_device, int n_items,
                                            long long total_blocks, void* stream) {
    if (!items_device || !first_block_device || n_items <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffLL)
        return YOLO_EINVAL;
    YOLO_LAUNCH(pack_pairs_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                (const PackPair*)items_device, first_block_device, n_items);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

extern "C" long long yolo_packed_weight_bytes(int Cout, int Cin, int ksize, int dtype) {
    if (Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 2 && ksize != 3)) return YOLO_EINVAL;
    if (!dtype_valid(dtype)) return YOLO_EINVAL;
    if (dtype_split(dtype) && ((Cin % 8) || ksize == 2)) return YOLO_EUNSUPPORTED;
    const int nchunks = (Cin * elem_size(dtype) + 63) / 64 * dtype_kpasses(dtype);
    return (long long)nchunks * ksize * ksize * round_up(Cout, YOLO_COUT_PAD) * 64;
}

static int pack_impl(const float* w_oihw, void* packed, int Cout, int Cin, int ksize, int dtype, int dgrad,
                     void* stream);

extern "C" int yolo_pack_conv_weights(const float* w_oihw, void* packed, int Cout, int Cin, int ksize,
                                      int dtype, void* stream) {
    return pack_impl(w_oihw, packed, Cout, Cin, ksize, dtype, 0, stream);
}

// Weight image of the DATA-GRADIENT convolution of a forward conv (Cout_f, Cin_f, k): a stride-1 conv with
// Cin_f output channels over Cout_f input channels, W'[ci][co][a][b] = W[co][ci][k-1-a][k-1-b].
// Size = yolo_packed_weight_bytes(Cin_f, Cout_f, k, dtype).
extern "C" int yolo_pack_conv_weights_dgrad(const float* w_oihw, void* packed, int Cout_f, int Cin_f, int ksize,
                                            int dtype, void* stream) {
    return pack_impl(w_oihw, packed, Cin_f, Cout_f, ksize, dtype, 1, stream);
}

// Weight image of the SUB-PIXEL data gradient of a 3x3 stride-2 forward conv (Cout_f, Cin_f): a 2x2-window conv with
// 4 x Cin_f output channels (phase-major) over Cout_f input channels, consumed by yolo_conv_dgrad_s2.
// Size = yolo_packed_weight_bytes(4 * Cin_f, Cout_f, 2, dtype).
extern "C" int yolo_pack_conv_weights_dgrad_s2(const float* w_oihw, void* packed, int Cout_f, int Cin_f, int dtype,
                                               void* stream) {
    if (Cin_f <= 0 || Cin_f > (1 << 28)) return YOLO_EINVAL;
    return pack_impl(w_oihw, packed, 4 * Cin_f, Cout_f, 2, dtype, 2, stream);
}

static int pack_impl(const float* w_oihw, void* packed, int Cout, int Cin, int ksize, int dtype, int dgrad,
                     void* stream) {
    if (!w_oihw || !packed) return YOLO_EINVAL;
    if ((ksize == 2) != (dgrad == 2)) return YOLO_EINVAL;
    const long long bytes = yolo_packed_weight_bytes(Cout, Cin, ksize, dtype);
    if (bytes < 0) return (int)bytes;
    const int Cout_pad = round_up(Cout, YOLO_COUT_PAD);
    if (dtype_split(dtype) && dgrad) return YOLO_EUNSUPPORTED;
    const int nchunks = (Cin * elem_size(dtype) + 63) / 64 * dtype_kpasses(dtype);
    const long long total = bytes / elem_size(dtype);
    const int grid = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
    if (dtype == YOLO_BF16)
        YOLO_LAUNCH(pack_weights_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw,
                           (bf16_t*)packed, Cout, Cin, ksize, Cout_pad, nchunks, dgrad);
    else if (dtype == YOLO_F16)
        YOLO_LAUNCH(pack_weights_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw,
                           (f16_t*)packed, Cout, Cin, ksize, Cout_pad, nchunks, dgrad);
    else if (dtype == YOLO_BF16X3)
        YOLO_LAUNCH(pack_weights_kernel<bf16x3_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw,
                           (bf16x3_t*)packed, Cout, Cin, ksize, Cout_pad, nchunks, dgrad);
    else if (dtype == YOLO_F16X3)
        YOLO_LAUNCH(pack_weights_kernel<f16x3_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw,
                           (f16x3_t*)packed, Cout, Cin, ksize, Cout_pad, nchunks, dgrad);
    else
        YOLO_LAUNCH(pack_weights_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw,
                           (float*)packed, Cout, Cin, ksize, Cout_pad, nchunks, dgrad);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}
