// Streaming convolution for the small-channel layers at the top of the network (bf16; 3x3 with Cin <= 64,
// 1x1 with Cin <= 128; Cout in {32, 64, 128}).  These layers are HBM-bound (a few hundred MB of activations,
// K <= 576): what matters is that every byte is read once, in long coalesced runs, with enough loads in flight,
// not the MFMA rate.  The tiled implicit-GEMM kernels spend their time in per-tile set-up / drain there.
//
//   * WEIGHTS-STATIONARY: a wave keeps the MFMA A-fragments of its 32 output channels for the whole K
//     (<= 36 fragments = 144 VGPRs) in registers for the lifetime of the block -- read once from the packed image.
//   * a block (4 waves = COUT/32 channel slices x 128/COUT pixel groups) owns a column strip of one image and walks
//     down its output rows; input rows live in a rolling LDS window (each input byte is fetched once per strip),
//     the next row's loads are in flight (registers) while the current row is multiplied and stored.
//   * epilogue as in conv_epilogue.h (transpose through a private LDS scratch, 16-byte row stores) with the
//     residual prefetched before the MFMAs so its latency hides under them.
// Same packed-weight image, operand order and rounding points as the other convolution kernels.
#include "common.h"
#include "conv_args.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

struct StreamArgs {
    const char* x;
    const char* wp;
    const float* scale;
    const float* bias;
    const char* res;
    char* y;
    int N, H, W, Ho, Wo, Cout_pad;
    int nstrips, strip_w, rows_per_slice;
    float slope;
    float* stats;           // STATS: partial rows [row][2][Cout_pad] of BatchNorm's forward sums (conv_epilogue.h), row =
                            // (slice * blocks_x + block_x) * WAVES_P + pixel wave
};

template <int KS, int S, int CIN, int COUT, int NI, int STATS = 0, typename T = bf16_t>
__global__ __launch_bounds__(256) void conv_stream_kernel(StreamArgs a) {
    constexpr int WAVES_C = COUT / 32, WAVES_P = 4 / WAVES_C;
    constexpr int TW = WAVES_P * NI * 32;               // output pixels per step (strip width capacity)
    constexpr int NTAP = KS * KS, KC = CIN / 16, KSTEPS = NTAP * KC;
    constexpr int ROWB = CIN * 2, PARTS = ROWB / 16, PITCH = ROWB + 16;
    constexpr int PAD = KS / 2;
    constexpr int XW = (TW - 1) * S + KS;
    constexpr int INUSE = KS, NEW = S, RING = INUSE + NEW;
    constexpr int XROW = XW * PITCH;
    // stride 2: a ring row holds its even pixels first, then the odd ones -- the fragment reads (every other pixel) then have a lane
    // stride of PITCH, an odd number of 16-byte units, instead of 2 * PITCH (two-way bank conflicts; stem_down.hip)
    constexpr int XHALF = (XW + 1) / 2;
    auto pxoff = [](int px) { return S == 2 ? ((px & 1) * XHALF + (px >> 1)) * PITCH : px * PITCH; };
    constexpr int XU = (NEW * XW * PARTS + 255) / 256;
    constexpr int SCR = 32 * 144;                        // per-wave epilogue scratch: 32 pixel rows x (32 f32 + pad)
    __shared__ __attribute__((aligned(16))) char smem[RING * XROW + 4 * SCR];
    char* xl = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wave_c = wave % WAVES_C, wave_p = wave / WAVES_C;
    char* scr = smem + RING * XROW + wave * SCR;

    int b = blockIdx.x;
    const int strip = b % a.nstrips;
    const int n = b / a.nstrips;
    const int ox0 = strip * a.strip_w;
    const int ox_end = min(ox0 + a.strip_w, a.Wo);
    const int oy0 = blockIdx.y * a.rows_per_slice;
    const int oy1 = min(oy0 + a.rows_per_slice, a.Ho);
    if (oy0 >= oy1) return;
    const int H = a.H, W = a.W, Ho = a.Ho, Wo = a.Wo;
    const int ix0 = ox0 * S - PAD;

    // ---- the wave's weights: A-fragments of its 32 couts for every K-step, straight from the packed image ----
    uint4 A[KSTEPS];
    {
        const int row = wave_c * 32 + l31;
        const int swz = (l31 >> 2) & 3;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int tap = ks / KC, kc = ks % KC;
            const int chunk = kc >> 1, unit = ((kc & 1) * 2 + h) ^ swz;
            A[ks] = *(const uint4*)(a.wp + ((long long)(chunk * NTAP + tap) * a.Cout_pad + row) * 64 + unit * 16);
        }
    }
    // ---- per-lane epilogue constants (after the transpose a lane owns 8 couts of a pixel row) -------------------
    const int ecol = lane & 3, erow0 = lane >> 2;
    const int eco = wave_c * 32 + ecol * 8;
    float sc[8], bi[8];
    const bool ident = a.scale == nullptr;               // identity epilogue (conv_epilogue.h)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
        if (!ident) { s4 = *(const f32x4*)(a.scale + eco + 4 * q); b4 = *(const f32x4*)(a.bias + eco + 4 * q); }
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[4 * q + e] = s4[e]; bi[4 * q + e] = b4[e]; }
    }
    const float slope = a.slope;
    const bool has_res = a.res != nullptr;
    float ssum[8], qsum[8];                              // STATS: sums of the stored values of this lane's 8 channels
#pragma unroll
    for (int e = 0; e < 8; ++e) ssum[e] = qsum[e] = 0.f;

    // ---- input staging (registers; loads run one step ahead) ----------------------------------------------------
    // (two register sets, requested UNCONDITIONALLY two steps ahead: behind a uniform branch the compiler's s_waitcnt for the
    // other set's stores covers the no-load path and waits for every load in flight -- which is what made the first
    // two-set attempt of round 1 look slower than one set)
    uint4 xr[2][XU];                                    // two sets: the rows of step k + 2 are requested at the top of step k
    // per-thread staging units, loop invariant: byte offset from the step's (block-uniform) base and the row inside the step,
    // or -1 for a unit that never loads.  The loads are unconditional range-checked buffer loads -- a unit that must read
    // zeros gets an out-of-range offset -- instead of predicated ones (zero-initialise, exec branch, load: ~20 instructions
    // each, and the compiler waits for every outstanding load before each zero-initialisation)
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    int x_off[XU], x_row[XU];
#pragma unroll
    for (int j = 0; j < XU; ++j) {
        const int u = tid + j * 256;
        const int r = u / (XW * PARTS), rem = u - r * (XW * PARTS), px = rem / PARTS, part = rem % PARTS;
        const int ix = ix0 + px;
        x_row[j] = (r < NEW && ix >= 0 && ix < W) ? r : -1;
        x_off[j] = (r * W + px) * ROWB + part * 16;
    }
    auto load_x = [&](auto set_c, int iy_first) {
        constexpr int SET = decltype(set_c)::value;
        const char* base = a.x + (((long long)n * H + iy_first) * W + ix0) * ROWB;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < XU; ++j) {
            const int iy = iy_first + x_row[j];
            const bool ok = x_row[j] >= 0 && iy >= 0 && iy < H;
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? x_off[j] : -1, 0, 0);
            xr[SET][j] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    auto store_x = [&](auto set_c, int slot_first) {
        constexpr int SET = decltype(set_c)::value;
#pragma unroll
        for (int j = 0; j < XU; ++j) {
            const int u = tid + j * 256;
            const int r = u / (XW * PARTS), rem = u - r * (XW * PARTS), px = rem / PARTS, part = rem % PARTS;
            if (r < NEW) {
                int slot = slot_first + r;
                if (slot >= RING) slot -= RING;
                *(uint4*)(xl + slot * XROW + pxoff(px) + part * 16) = xr[SET][j];
            }
        }
    };

    // prologue: the KS input rows of the first output row
    const int iyb0 = oy0 * S - PAD;
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
#pragma unroll
    for (int r0 = 0; r0 < INUSE; r0 += NEW) {
        load_x(Set0{}, iyb0 + r0);
        store_x(Set0{}, r0);
    }
    load_x(Set1{}, oy0 * S - PAD + INUSE);              // the rows step oy0 stores at its bottom
    __syncthreads();

    const int b_off = (wave_p * NI * 32 + l31) * PITCH + h * 16;        // (stride 2: pixel 2 j + kw = half kw & 1, index j + (kw >> 1))
    int slot0 = 0;
    // step k = output row oy; its parity names the set that is free at its top (it receives the rows of step k + 2); the
    // other set holds the rows step k stores at its bottom
    auto step = [&](auto par_c, int oy) {
        constexpr int PAR = decltype(par_c)::value;
        using Mine = std::integral_constant<int, PAR>;
        using Other = std::integral_constant<int, PAR ^ 1>;
        const bool more = oy + 1 < oy1;
        load_x(Mine{}, (oy + 1) * S - PAD + INUSE);
        // residual prefetch (same addresses as this lane's stores)
        long long yo[NI][2];
        uint4 rv[NI][2];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int ox = ox0 + (wave_p * NI + ni) * 32 + erow0 + 16 * k;
                yo[ni][k] = ox < ox_end ? ((((long long)n * Ho + oy) * Wo + ox) * COUT + eco) * 2 : -1;
                rv[ni][k] = make_uint4(0, 0, 0, 0);
                if (has_res && yo[ni][k] >= 0) rv[ni][k] = *(const uint4*)(a.res + yo[ni][k]);
            }

        f32x16 acc[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
            int slot = slot0 + kh;
            if (slot >= RING) slot -= RING;
            const char* rowp = xl + slot * XROW + b_off;
#pragma unroll
            for (int kw = 0; kw < KS; ++kw)
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const uint4 bf = *(const uint4*)(rowp + (S == 2 ? ni * 32 * PITCH + pxoff(kw) : (ni * 32 + kw) * PITCH) + kc * 32);
                        acc[ni] = mfma16<T>(A[(kh * KS + kw) * KC + kc], bf, acc[ni]);
                    }
                }
        }

        // ---- epilogue: folded BN, LeakyReLU, residual, one rounding, 16-byte row stores ---------------------------
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[ni][4 * g], acc[ni][4 * g + 1], acc[ni][4 * g + 2], acc[ni][4 * g + 3]};
                *(f32x4*)(scr + l31 * 144 + (8 * g + 4 * h) * 4) = v;
            }
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 t4 = *(const f32x4*)(scr + (erow0 + 16 * k) * 144 + (ecol * 8 + 4 * q) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * q + e] = t4[e];
                }
                if (!ident) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float t = v[e] * sc[e] + bi[e];
                        v[e] = leaky(t, slope);
                    }
                }
                if (has_res) {
                    const uint32_t w[4] = {rv[ni][k].x, rv[ni][k].y, rv[ni][k].z, rv[ni][k].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[2 * q] += Elem<T>::lo(w[q]);
                        v[2 * q + 1] += Elem<T>::hi(w[q]);
                    }
                }
                if (yo[ni][k] >= 0) {
                    const uint4 o = make_uint4(Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3]), Elem<T>::pack2(v[4], v[5]),
                                               Elem<T>::pack2(v[6], v[7]));
                    *(uint4*)(a.y + yo[ni][k]) = o;
                    if constexpr (STATS) {
                        const uint32_t ow[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float t = ((e & 1) ? Elem<T>::hi(ow[e >> 1]) : Elem<T>::lo(ow[e >> 1]));
                            ssum[e] += t; qsum[e] += t * t;
                        }
                    }
                }
            }
        }

        if (more) {
            int ns = slot0 + INUSE;
            if (ns >= RING) ns -= RING;
            store_x(Other{}, ns);
        }
        slot0 += NEW;
        if (slot0 >= RING) slot0 -= RING;
        __syncthreads();
    };
    for (int oy = oy0; oy < oy1; oy += 2) {
        step(Set0{}, oy);
        if (oy + 1 < oy1) step(Set1{}, oy + 1);
    }
    if constexpr (STATS) {
        // the 16 lanes that share a channel octet (erow0 = 0..15) combine through the wave's scratch: one partial row
        // (this wave's 32 channels of it) per (block, pixel wave)
        float* sc4 = (float*)scr;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc4[lane * 16 + e] = ssum[e]; sc4[lane * 16 + 8 + e] = qsum[e]; }
        wave_lds_fence();
        float* prow = a.stats + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * WAVES_P + wave_p) * 2 * a.Cout_pad;
        const int cx = lane >> 4, val = lane & 15;       // 4 octets x 16 values = 64 outputs, one per lane
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += sc4[(r * 4 + cx) * 16 + val];
        prow[(val >> 3) * a.Cout_pad + wave_c * 32 + cx * 8 + (val & 7)] = t;
    }
}

template <int KS, int S, int CIN, int COUT, int NI, typename T = bf16_t>
static int launch_stream(const ConvArgs& c, hipStream_t st, const NameOut* nm) {
    constexpr int TW = (4 / (COUT / 32)) * NI * 32;
    if (c.stats && (c.stats_mode != 1 || c.res)) return YOLO_EUNSUPPORTED;      // forward sums only
    StreamArgs a;
    a.x = c.x; a.wp = c.wp; a.scale = c.scale; a.bias = c.bias; a.res = c.res; a.y = c.y;
    a.N = c.N; a.H = c.H; a.W = c.W; a.Ho = c.Ho; a.Wo = c.Wo; a.Cout_pad = c.Cout_pad;
    a.slope = c.slope;
    a.nstrips = (c.Wo + TW - 1) / TW;
    a.strip_w = (c.Wo + a.nstrips - 1) / a.nstrips;              // balanced strips (<= TW)
    const long long bx = (long long)c.N * a.nstrips;
    const long long target = 3072;                                // (768..12288 measured: flat within noise)
    long long slices = (target + bx - 1) / bx;                      // ~12 blocks per CU over the launch (several rounds: small tail)
    if (slices > c.Ho / 8) slices = c.Ho / 8;
    if (slices < 1) slices = 1;
    a.rows_per_slice = (int)((c.Ho + slices - 1) / slices);
    slices = (c.Ho + a.rows_per_slice - 1) / a.rows_per_slice;
    a.stats = c.stats;
    if (nm) {
        snprintf(nm->buf, nm->len, c.stats ? "void conv_stream_kernel<%d, %d, %d, %d, %d, 1, %s>(StreamArgs)"
                                           : "void conv_stream_kernel<%d, %d, %d, %d, %d, 0, %s>(StreamArgs)", KS, S, CIN, COUT, NI, Elem<T>::name);
        if (nm->stats_rows) *nm->stats_rows = c.stats ? (int)(bx * slices * (4 / (COUT / 32))) : -1;
        return YOLO_OK;
    }
    if constexpr (IsBf16<T>::value) {
        if (c.stats) {
            YOLO_LAUNCH((conv_stream_kernel<KS, S, CIN, COUT, NI, 1, T>), dim3((unsigned)bx, (unsigned)slices), dim3(256), 0, st, a);
            YOLO_LAUNCH_CHECK();
            return YOLO_OK;
        }
    } else if (c.stats) {
        return YOLO_EUNSUPPORTED;                                   // (the statistics epilogue belongs to the bf16 training step)
    }
    YOLO_LAUNCH((conv_stream_kernel<KS, S, CIN, COUT, NI, 0, T>), dim3((unsigned)bx, (unsigned)slices), dim3(256), 0, st, a);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// algo 13 (NI = 1) / 14 (NI = 2: twice the strip width per step)
int conv_stream_dispatch(const ConvArgs& c, int ks, int stride, int dtype, int algo, hipStream_t st, const NameOut* nm) {
    if ((dtype != YOLO_BF16 && dtype != YOLO_F16) || c.out_f32 || c.up2 || c.x_ps != c.Cin) return YOLO_EUNSUPPORTED;
    const bool half = dtype == YOLO_F16;
    if (c.y_ps != c.Cout || c.y_bs != (long long)c.Ho * c.Wo * c.Cout) return YOLO_EUNSUPPORTED;
    const int ni = algo == 14 ? 2 : 1;
#define STREAM_CASE(KS_, S_, CIN_, COUT_)                                                        \
    if (ks == KS_ && stride == S_ && c.Cin == CIN_ && c.Cout == COUT_)                           \
        return half ? (ni == 2 ? launch_stream<KS_, S_, CIN_, COUT_, 2, f16_t>(c, st, nm) : launch_stream<KS_, S_, CIN_, COUT_, 1, f16_t>(c, st, nm)) \
                    : (ni == 2 ? launch_stream<KS_, S_, CIN_, COUT_, 2>(c, st, nm) : launch_stream<KS_, S_, CIN_, COUT_, 1>(c, st, nm));
#define STREAM_CASE1(KS_, S_, CIN_, COUT_)                                                       \
    if (ks == KS_ && stride == S_ && c.Cin == CIN_ && c.Cout == COUT_ && ni == 1)                \
        return half ? launch_stream<KS_, S_, CIN_, COUT_, 1, f16_t>(c, st, nm) : launch_stream<KS_, S_, CIN_, COUT_, 1>(c, st, nm);
    STREAM_CASE(3, 1, 32, 64)
    STREAM_CASE(3, 1, 64, 128)
    STREAM_CASE1(3, 1, 64, 32)
    STREAM_CASE1(3, 1, 32, 32)
    STREAM_CASE(3, 1, 64, 64)
    STREAM_CASE1(3, 2, 32, 64)
    STREAM_CASE1(3, 2, 64, 128)
    STREAM_CASE1(1, 1, 64, 32)
    STREAM_CASE(1, 1, 32, 64)
    STREAM_CASE(1, 1, 128, 64)
    STREAM_CASE(1, 1, 64, 128)
    STREAM_CASE(1, 1, 64, 64)
    STREAM_CASE(1, 1, 128, 128)
#undef STREAM_CASE
#undef STREAM_CASE1
    return YOLO_EUNSUPPORTED;
}
