// One DarknetBasicBlockV3 of the first two stages as ONE inference kernel (basic_yolo.py:26; gluoncv darknet.py:
// x + conv3x3(2c)(conv1x1(c)(x)), each conv with folded BN + LeakyReLU, no activation after the add), bf16 NHWC,
// for C = 2c in {64, 128}.
//
// Run as two kernels these blocks are HBM-bound three times over: the 1x1 reads x and writes the half-width map, the
// 3x3 reads that map and x again (the residual) and writes the output -- 2.5x the bytes of "read x, write out", at the
// largest maps of the network (15 % of the 608x608 pass).  Here the half-width map never leaves LDS and x is read once:
//   * a block (4 waves) owns a column strip (<= 62 output pixels) of one image and walks down its output rows, as
//     conv_stream_kernel / stem_down_kernel do; BOTH layers' weights stay in registers for the block's lifetime
//     (C = 64: 4 + 18 A-fragments per wave; C = 128: 8 + 36);
//   * input rows live in a 3-row rolling LDS ring (row oy: the residual of the output row; row oy + 1: the operand of the
//     1x1 for the mid row the 3x3 needs next); the loads of the NEXT D - 1 rows are in flight in registers meanwhile (a
//     row-step is a dependent chain load -> LDS -> 1x1 -> LDS -> 3x3 -> store of ~1.5k cycles, shorter than one HBM round
//     trip under load: with one row in flight the kernel ran at the memory latency, 2.7 us per row-step);
//   * the 1x1's output (mid) lives in a 3-row rolling LDS ring in the layout the 3x3 reads its B fragments from; it is
//     COMPUTED there one row per step.  Mid pixels outside the image are written as zeros: they are the 3x3's zero
//     padding, not conv1x1(padding);
//   * the 3x3 epilogue (conv_stream's: transpose through a per-wave LDS scratch, 16-byte row stores) takes its residual
//     from the input ring instead of reading x a second time.
// Measured (tools/rb_probe.py with ablation builds, 32 x 208 x 208 x 64: 144 us against 164 us for the two layers; HBM floor
// 56 us): the parts add up instead of overlapping -- 3x3 MFMAs 36 us, output stores 40, mid row 26, input loads 13,
// skeleton (barriers, ring fill, weights, transposes) 36 -- two waves per SIMD do not cover each other's latencies, and
// deeper register prefetch (D) changes nothing.  The C = 128 instantiation (one wave per SIMD, 36 A-fragments in
// AGPRs) runs its MFMA phase 6x below the matrix rate and loses to the two separate layers: the tuner does not pick it.
// Same packed weight images, operand rounding points (mid is rounded to bf16 once; the residual is added in fp32 before
// the output's one rounding) and K order ((kh, kw, channel)) as conv_stream_kernel.
#include "common.h"
#include <stdlib.h>
#include <stdio.h>

namespace {
constexpr int MW = 64;                   // mid / input row width in pixels (= 2 MFMA column groups); strip width <= MW - 2
constexpr int SW_MAX = MW - 2;
constexpr int SCR = 32 * 144;            // per-wave epilogue scratch: 32 pixel rows x (32 f32 + pad)

struct ResArgs {
    const char* x;
    const char* wp1;
    const float* scale1;
    const float* bias1;
    const char* wp2;
    const float* scale2;
    const float* bias2;
    char* y;
    int N, H, W, Cpad1, Cpad2;
    int nstrips, strip_w, rows_per_slice;
    float slope;
};
}  // namespace

template <int C, int D, typename T = bf16_t>
__global__ __launch_bounds__(256) void res_block_kernel(ResArgs a) {
    constexpr int CM = C / 2;
    constexpr int WAVES_C = C / 32, WAVES_P = 4 / WAVES_C, NI = 2 / WAVES_P;   // 3x3: wave = (cout slice, pixel group)
    constexpr int MT1 = CM / 32;                                              // 1x1: cout slices of the mid row
    constexpr int K1 = C / 16, K2 = CM / 16;                                  // k-steps of 16 channels
    constexpr int PX = C * 2 + 16, PM = CM * 2 + 16;                          // pixel pitch in the rings (+16: conflict-free reads)
    constexpr int XROW = MW * PX, MROW = (MW + 2) * PM;
    constexpr int XU = MW * (C * 2 / 16) / 256;                               // 16-byte input loads per thread per row
    static_assert(MW * (C * 2 / 16) % 256 == 0, "input row = whole passes of the block");
    __shared__ __attribute__((aligned(16))) char smem[3 * XROW + 3 * MROW + 4 * SCR];
    char* xl = smem;
    char* ml = smem + 3 * XROW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wave_c = wave % WAVES_C, wave_p = wave / WAVES_C;
    char* scr = smem + 3 * XROW + 3 * MROW + wave * SCR;

    const int strip = blockIdx.x % a.nstrips;
    const int n = blockIdx.x / a.nstrips;
    const int ox0 = strip * a.strip_w;
    const int ox_end = min(ox0 + a.strip_w, a.W);
    const int oy0 = blockIdx.y * a.rows_per_slice;
    const int oy1 = min(oy0 + a.rows_per_slice, a.H);
    if (oy0 >= oy1) return;
    const int H = a.H, W = a.W;
    const int mx0 = ox0 - 1;                  // image column of ring pixel 0

    // ---- hygiene: ring pixels that are never written are only read by lanes whose results are discarded ----------
    for (int i = tid; i < (3 * XROW + 3 * MROW) / 16; i += 256) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);

    // ---- weights: A-fragments from the packed images (conv_stream's addressing) ----------------------------------
    const int swz = (l31 >> 2) & 3;
    uint4 A2[9 * K2];
    {
        const int row = wave_c * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 9 * K2; ++ks) {
            const int tap = ks / K2, kc = ks % K2;
            const int chunk = kc >> 1, unit = ((kc & 1) * 2 + h) ^ swz;
            A2[ks] = *(const uint4*)(a.wp2 + ((long long)(chunk * 9 + tap) * a.Cpad2 + row) * 64 + unit * 16);
        }
    }
    // 1x1: wave w produces mid tile (cout slice w % MT1, pixel group w / MT1); with MT1 == 1 only waves 0, 1 have one
    const int m1 = wave % MT1, p1 = wave / MT1;
    const bool has1 = p1 < 2;
    uint4 A1[K1];
    {
        const int row = m1 * 32 + l31;
#pragma unroll
        for (int kc = 0; kc < K1; ++kc) {
            const int chunk = kc >> 1, unit = ((kc & 1) * 2 + h) ^ swz;
            A1[kc] = *(const uint4*)(a.wp1 + ((long long)chunk * a.Cpad1 + row) * 64 + unit * 16);
        }
    }
    // folded BN of the 1x1 for the 16 mid channels a lane holds after its MFMAs (rows 8g + 4h + e of the slice)
    float sc1[16], bi1[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 s4 = *(const f32x4*)(a.scale1 + m1 * 32 + 8 * g + 4 * h), b4 = *(const f32x4*)(a.bias1 + m1 * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc1[4 * g + e] = s4[e]; bi1[4 * g + e] = b4[e]; }
    }
    // 3x3 epilogue constants (after the transpose a lane owns 8 couts of a pixel row)
    const int ecol = lane & 3, erow0 = lane >> 2;
    const int eco = wave_c * 32 + ecol * 8;
    float sc2[8], bi2[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x4 s4 = *(const f32x4*)(a.scale2 + eco + 4 * q), b4 = *(const f32x4*)(a.bias2 + eco + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc2[4 * q + e] = s4[e]; bi2[4 * q + e] = b4[e]; }
    }
    const float slope = a.slope;

    // ---- input rows: registers -> ring slot (row mod 3) ------------------------------------------------------------
    uint4 xr[D][XU];                                        // register sets: rows oy+2 .. oy+D in flight (static indices only)
    // unconditional range-checked buffer loads (a unit outside the image gets an out-of-range offset and reads zeros): a
    // predicated load is a zero-initialisation + an exec branch, and the compiler waits for EVERY load in flight before each
    // zero-initialisation -- which serialised the D register sets this kernel keeps in flight
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    int x_off[XU];
#pragma unroll
    for (int j = 0; j < XU; ++j) {
        const int u = tid + j * 256;
        const int px = u / (C * 2 / 16), part = u % (C * 2 / 16);
        const int ix = mx0 + px;
        x_off[j] = (ix >= 0 && ix < W) ? px * (C * 2) + part * 16 : -1;
    }
    auto load_x = [&](uint4 (&r)[XU], int iy) {
        const char* base = a.x + (((long long)n * H + iy) * W + mx0) * (C * 2);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
        const bool row_ok = iy >= 0 && iy < H;
#pragma unroll
        for (int j = 0; j < XU; ++j) {
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, row_ok ? x_off[j] : -1, 0, 0);
            r[j] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    auto store_x = [&](const uint4 (&r)[XU], int slot) {
#pragma unroll
        for (int j = 0; j < XU; ++j) {
            const int u = tid + j * 256;
            const int px = u / (C * 2 / 16), part = u % (C * 2 / 16);
            *(uint4*)(xl + slot * XROW + px * PX + part * 16) = r[j];
        }
    };
    auto slot3 = [](int row) { return (row + 3) % 3; };       // rows >= -1
    // ---- mid row my from the input ring: this wave's tile, BN + LeakyReLU, one rounding, zeros outside the image ---
    auto mid_row = [&](int my) {
        if (!has1) return;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const char* xp = xl + slot3(my) * XROW + (p1 * 32 + l31) * PX + h * 16;
#pragma unroll
        for (int kc = 0; kc < K1; ++kc) {
            const uint4 bf = *(const uint4*)(xp + kc * 32);
            acc = mfma16<T>(A1[kc], bf, acc);
        }
        const int px = p1 * 32 + l31;
        const int mx = mx0 + px;
        const bool inside = my >= 0 && my < H && mx >= 0 && mx < W;
        char* dst = ml + slot3(my) * MROW + px * PM + m1 * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = leaky(acc[4 * g + e] * sc1[4 * g + e] + bi1[4 * g + e], slope);
            *(uint2*)(dst + (8 * g + 4 * h) * 2) = inside ? make_uint2(Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3])) : make_uint2(0u, 0u);
        }
    };

    __syncthreads();                                        // the zero fill
    // ---- prologue: input rows oy0-1, oy0, oy0+1; mid rows oy0-1, oy0 --------------------------------------------------
    load_x(xr[0], oy0 - 1);
    load_x(xr[1 % D], oy0);
    store_x(xr[0], slot3(oy0 - 1));
    store_x(xr[1 % D], slot3(oy0));
    load_x(xr[D - 1], oy0 + 1);
    // rows oy0+2 .. oy0+D go into sets 0 .. D-2 (set of row r = (r - oy0 - 2) % D); step oy stores set (oy - oy0) % D and
    // re-loads the set the previous step stored
#pragma unroll
    for (int k = 0; k < D - 1; ++k) load_x(xr[k], oy0 + 2 + k);   // (unconditional: see the step loop)
    __syncthreads();
    mid_row(oy0 - 1);
    mid_row(oy0);
    store_x(xr[D - 1], slot3(oy0 + 1));
    __syncthreads();

    for (int oyb = oy0; oyb < oy1; oyb += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
        const int oy = oyb + u;
        if (oy >= oy1) break;
        const bool more = oy + 1 < oy1;
        // D - 1 rows ahead of the row stored below.  UNCONDITIONAL (rows past the slice are read and dropped, rows past the
        // image get out-of-range offsets): behind a uniform branch the compiler's s_waitcnt for the stores of the other sets
        // must assume the no-load path, i.e. wait for every load in flight -- D sets then behave like one
        load_x(xr[(u + D - 1) % D], oy + D + 1);
        mid_row(oy + 1);
        __syncthreads();

        // ---- 3x3 over mid rows oy-1 .. oy+1: 9 taps x K2 k-steps, NI pixel groups per wave ----------------------------
        f32x16 acc[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const char* rowp = ml + slot3(oy - 1 + kh) * MROW + (wave_p * NI * 32 + l31) * PM + h * 16;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int kc = 0; kc < K2; ++kc)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const uint4 bf = *(const uint4*)(rowp + (ni * 32 + kw) * PM + kc * 32);
                        acc[ni] = mfma16<T>(A2[(kh * 3 + kw) * K2 + kc], bf, acc[ni]);
                    }
        }
        // ---- epilogue: folded BN, LeakyReLU, + x (from the input ring, fp32), one rounding, 16-byte row stores ---------
        const char* xres = xl + slot3(oy) * XROW + eco * 2;
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
                const int px = (wave_p * NI + ni) * 32 + erow0 + 16 * k;          // output pixel of the strip
                float v[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 t4 = *(const f32x4*)(scr + (erow0 + 16 * k) * 144 + (ecol * 8 + 4 * q) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * q + e] = t4[e];
                }
                const uint4 rv = *(const uint4*)(xres + (px + 1) * PX);           // ring pixel px + 1 = image column ox0 + px
                const uint32_t w[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = leaky(v[e] * sc2[e] + bi2[e], slope);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[2 * q] += Elem<T>::lo(w[q]);
                    v[2 * q + 1] += Elem<T>::hi(w[q]);
                }
                const int ox = ox0 + px;
                if (ox < ox_end)
                    *(uint4*)(a.y + ((((long long)n * H + oy) * W + ox) * C + eco) * 2) =
                        make_uint4(Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3]), Elem<T>::pack2(v[4], v[5]), Elem<T>::pack2(v[6], v[7]));
            }
        }
        if (more) store_x(xr[u], slot3(oy + 2));
        __syncthreads();
    }
    }
}

// ---- two output rows per step (round 3) -------------------------------------------------------------------------------
// The kernel above is latency-bound (tools/pmc_kernel.sh: an instruction active 35 % of the wave cycles, MFMA busy 34 %, VALU
// 35 %, LDS 20 %): one or two waves per SIMD walk a dependent chain per row step and nothing fills the gaps.  Here a step
// produces TWO output rows: two independent accumulator chains per wave, every B fragment of the two shared mid rows feeds
// two MFMAs (4 mid rows x 3 x K2 fragment reads for 2 x 9 x K2 MFMAs: a third less LDS traffic per MFMA), the two mid rows
// of a step occupy all four waves at C = 64, and the barriers, ring bookkeeping and loop overhead are paid once per two
// rows.  Input ring: 5 rows (oy, oy+1: residuals; oy+1, oy+2: operands of the step's mid rows; oy+3, oy+4: arriving); mid ring:
// 4 rows (oy-1 .. oy+2).  Same operands, accumulation order (kernel row, kernel column, channel) and rounding points.
template <int C, int D, typename T = bf16_t>
__global__ __launch_bounds__(256) void res_block2_kernel(ResArgs a) {
    constexpr int CM = C / 2;
    constexpr int WAVES_C = C / 32, WAVES_P = 4 / WAVES_C, NI = 2 / WAVES_P;
    constexpr int MT1 = CM / 32;
    constexpr int K1 = C / 16, K2 = CM / 16;
    constexpr int PX = C * 2 + 16, PM = CM * 2 + 16;
    constexpr int XROW = MW * PX, MROW = (MW + 2) * PM;
    constexpr int XU = MW * (C * 2 / 16) / 256;
    constexpr int XR = 5, MR = 4;
    static_assert(MW * (C * 2 / 16) % 256 == 0, "input row = whole passes of the block");
    static_assert(XR * XROW + MR * MROW + 4 * SCR <= 163840, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[XR * XROW + MR * MROW + 4 * SCR];
    char* xl = smem;
    char* ml = smem + XR * XROW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wave_c = wave % WAVES_C, wave_p = wave / WAVES_C;
    char* scr = smem + XR * XROW + MR * MROW + wave * SCR;

    const int strip = blockIdx.x % a.nstrips;
    const int n = blockIdx.x / a.nstrips;
    const int ox0 = strip * a.strip_w;
    const int ox_end = min(ox0 + a.strip_w, a.W);
    const int oy0 = blockIdx.y * a.rows_per_slice;
    const int oy1 = min(oy0 + a.rows_per_slice, a.H);
    if (oy0 >= oy1) return;
    const int H = a.H, W = a.W;
    const int mx0 = ox0 - 1;

    for (int i = tid; i < (XR * XROW + MR * MROW) / 16; i += 256) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);

    const int swz = (l31 >> 2) & 3;
    uint4 A2[9 * K2];
    {
        const int row = wave_c * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 9 * K2; ++ks) {
            const int tap = ks / K2, kc = ks % K2;
            const int chunk = kc >> 1, unit = ((kc & 1) * 2 + h) ^ swz;
            A2[ks] = *(const uint4*)(a.wp2 + ((long long)(chunk * 9 + tap) * a.Cpad2 + row) * 64 + unit * 16);
        }
    }
    // 1x1: a step makes two mid rows of MT1 x 2 tiles (cout slice, 32-pixel group).  MT1 == 2 (C = 128): wave = (slice, group),
    // both rows; MT1 == 1 (C = 64): wave = (group, row) -- all four waves busy
    const int m1 = MT1 == 2 ? wave % 2 : 0;
    const int p1 = MT1 == 2 ? wave / 2 : wave % 2;
    const int j1 = MT1 == 2 ? 0 : wave / 2;                  // (MT1 == 1: the one row of the pair this wave makes)
    uint4 A1[K1];
    {
        const int row = m1 * 32 + l31;
#pragma unroll
        for (int kc = 0; kc < K1; ++kc) {
            const int chunk = kc >> 1, unit = ((kc & 1) * 2 + h) ^ swz;
            A1[kc] = *(const uint4*)(a.wp1 + ((long long)chunk * a.Cpad1 + row) * 64 + unit * 16);
        }
    }
    float sc1[16], bi1[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 s4 = *(const f32x4*)(a.scale1 + m1 * 32 + 8 * g + 4 * h), b4 = *(const f32x4*)(a.bias1 + m1 * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc1[4 * g + e] = s4[e]; bi1[4 * g + e] = b4[e]; }
    }
    const int ecol = lane & 3, erow0 = lane >> 2;
    const int eco = wave_c * 32 + ecol * 8;
    float sc2[8], bi2[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x4 s4 = *(const f32x4*)(a.scale2 + eco + 4 * q), b4 = *(const f32x4*)(a.bias2 + eco + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc2[4 * q + e] = s4[e]; bi2[4 * q + e] = b4[e]; }
    }
    const float slope = a.slope;

    uint4 xr[D][2][XU];                                     // register sets: a set = the two rows one step stores
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    int x_off[XU];
#pragma unroll
    for (int j = 0; j < XU; ++j) {
        const int u = tid + j * 256;
        const int px = u / (C * 2 / 16), part = u % (C * 2 / 16);
        const int ix = mx0 + px;
        x_off[j] = (ix >= 0 && ix < W) ? px * (C * 2) + part * 16 : -1;
    }
    auto load_x = [&](uint4 (&r)[XU], int iy) {
        const char* base = a.x + (((long long)n * H + iy) * W + mx0) * (C * 2);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
        const bool row_ok = iy >= 0 && iy < H;
#pragma unroll
        for (int j = 0; j < XU; ++j) {
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, row_ok ? x_off[j] : -1, 0, 0);
            r[j] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    auto xslot = [](int row) { return (row + XR) % XR; };    // rows >= -1
    auto mslot = [](int row) { return (row + MR) & (MR - 1); };
    auto store_x = [&](const uint4 (&r)[XU], int row) {
        char* dst = xl + xslot(row) * XROW;
#pragma unroll
        for (int j = 0; j < XU; ++j) {
            const int u = tid + j * 256;
            const int px = u / (C * 2 / 16), part = u % (C * 2 / 16);
            *(uint4*)(dst + px * PX + part * 16) = r[j];
        }
    };
    auto mid_row = [&](int my) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const char* xp = xl + xslot(my) * XROW + (p1 * 32 + l31) * PX + h * 16;
#pragma unroll
        for (int kc = 0; kc < K1; ++kc) {
            const uint4 bf = *(const uint4*)(xp + kc * 32);
            acc = mfma16<T>(A1[kc], bf, acc);
        }
        const int px = p1 * 32 + l31;
        const int mx = mx0 + px;
        const bool inside = my >= 0 && my < H && mx >= 0 && mx < W;
        char* dst = ml + mslot(my) * MROW + px * PM + m1 * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = leaky(acc[4 * g + e] * sc1[4 * g + e] + bi1[4 * g + e], slope);
            *(uint2*)(dst + (8 * g + 4 * h) * 2) = inside ? make_uint2(Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3])) : make_uint2(0u, 0u);
        }
    };
    // the two mid rows my0, my0 + 1 of a step
    auto mid_pair = [&](int my0) {
        if (MT1 == 2) { mid_row(my0); mid_row(my0 + 1); }
        else mid_row(my0 + j1);
    };

    __syncthreads();                                        // the zero fill
    // ---- prologue: input rows oy0-1 .. oy0+2 into the ring, mid rows oy0-1, oy0 ------------------------------------------
    load_x(xr[0][0], oy0 - 1);
    load_x(xr[0][1], oy0);
    store_x(xr[0][0], oy0 - 1);
    store_x(xr[0][1], oy0);
    load_x(xr[0][0], oy0 + 1);
    load_x(xr[0][1], oy0 + 2);
    store_x(xr[0][0], oy0 + 1);
    store_x(xr[0][1], oy0 + 2);
    // step k (output rows oy0 + 2k, + 1) stores rows oy0 + 2k + 3, + 4 from set k % D; sets 0 .. D-2 are requested here, set
    // (k - 1) % D is re-requested at the top of step k with the rows of step k + D - 1 (unconditionally: see the kernel above)
#pragma unroll
    for (int k = 0; k < D - 1; ++k) { load_x(xr[k][0], oy0 + 2 * k + 3); load_x(xr[k][1], oy0 + 2 * k + 4); }
    __syncthreads();
    mid_pair(oy0 - 1);
    __syncthreads();

    for (int oyb = oy0; oyb < oy1; oyb += 2 * D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
        const int oy = oyb + 2 * u;
        if (oy >= oy1) break;
        load_x(xr[(u + D - 1) % D][0], oy + 2 * D + 1);
        load_x(xr[(u + D - 1) % D][1], oy + 2 * D + 2);
        mid_pair(oy + 1);                                    // mid rows oy + 1, oy + 2 from input rows oy + 1, oy + 2
        __syncthreads();

        // ---- 3x3 of output rows oy, oy + 1 over mid rows oy - 1 .. oy + 2: every fragment of the shared rows feeds both ------
        f32x16 acc[2][NI];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[r][ni][e] = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const char* rowp = ml + mslot(oy - 1 + m) * MROW + (wave_p * NI * 32 + l31) * PM + h * 16;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int kc = 0; kc < K2; ++kc)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const uint4 bf = *(const uint4*)(rowp + (ni * 32 + kw) * PM + kc * 32);
                        if (m <= 2)
                            acc[0][ni] = mfma16<T>(A2[(m * 3 + kw) * K2 + kc], bf, acc[0][ni]);
                        if (m >= 1)
                            acc[1][ni] = mfma16<T>(A2[((m - 1) * 3 + kw) * K2 + kc], bf, acc[1][ni]);
                    }
        }
        // ---- epilogue of both rows: folded BN, LeakyReLU, + x (input ring, fp32), one rounding, 16-byte row stores ----------
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (oy + r >= oy1) break;
            const char* xres = xl + xslot(oy + r) * XROW + eco * 2;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[r][ni][4 * g], acc[r][ni][4 * g + 1], acc[r][ni][4 * g + 2], acc[r][ni][4 * g + 3]};
                    *(f32x4*)(scr + l31 * 144 + (8 * g + 4 * h) * 4) = v;
                }
                wave_lds_fence();
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int px = (wave_p * NI + ni) * 32 + erow0 + 16 * k;
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x4 t4 = *(const f32x4*)(scr + (erow0 + 16 * k) * 144 + (ecol * 8 + 4 * q) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * q + e] = t4[e];
                    }
                    const uint4 rv = *(const uint4*)(xres + (px + 1) * PX);
                    const uint32_t w[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = leaky(v[e] * sc2[e] + bi2[e], slope);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[2 * q] += Elem<T>::lo(w[q]);
                        v[2 * q + 1] += Elem<T>::hi(w[q]);
                    }
                    const int ox = ox0 + px;
                    if (ox < ox_end)
                        *(uint4*)(a.y + ((((long long)n * H + oy + r) * W + ox) * C + eco) * 2) =
                            make_uint4(Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3]), Elem<T>::pack2(v[4], v[5]), Elem<T>::pack2(v[6], v[7]));
                }
            }
        }
        if (oy + 2 < oy1) { store_x(xr[u][0], oy + 3); store_x(xr[u][1], oy + 4); }
        __syncthreads();
    }
    }
}

template <int C, int D, typename T = bf16_t>
static int launch_res_block2(ResArgs& a, hipStream_t st) {
    a.nstrips = (a.W + SW_MAX - 1) / SW_MAX;
    a.strip_w = (a.W + a.nstrips - 1) / a.nstrips;
    const long long bx = (long long)a.N * a.nstrips;
    if (bx > 0x7fffffffLL) return YOLO_EUNSUPPORTED;
    const long long slots = 256;                                 // one block per CU (85-143 KB of LDS)
    long long best_s = 1;
    double best_cost = 1e30;
    for (long long sl = 1; sl <= (a.H >= 16 ? a.H / 8 : 1); ++sl) {
        const long long rows = (a.H + sl - 1) / sl, nsl = (a.H + rows - 1) / rows;
        const double cost = (double)((bx * nsl + slots - 1) / slots) * (double)(rows + 4.5);
        if (cost < best_cost - 1e-9) { best_cost = cost; best_s = nsl; }
    }
    long long rows = (a.H + best_s - 1) / best_s;
    rows += rows & 1;                                            // even slices: no half-used step in the middle of the image
    a.rows_per_slice = (int)rows;
    const long long slices = (a.H + a.rows_per_slice - 1) / a.rows_per_slice;
    YOLO_LAUNCH((res_block2_kernel<C, D, T>), dim3((unsigned)bx, (unsigned)slices), dim3(256), 0, st, a);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

template <int C, int D, typename T = bf16_t>
static int launch_res_block(ResArgs& a, hipStream_t st) {
    a.nstrips = (a.W + SW_MAX - 1) / SW_MAX;
    a.strip_w = (a.W + a.nstrips - 1) / a.nstrips;               // balanced strips (<= 62)
    const long long bx = (long long)a.N * a.nstrips;
    if (bx > 0x7fffffffLL) return YOLO_EUNSUPPORTED;
    // row slices: a slice costs ~3.5 row-steps of prologue (ring fill, two extra mid rows, weights) and the launch runs in
    // rounds of `slots` resident blocks (2 per CU for C = 64, 1 for C = 128): pick the slice count with the least
    // rounds x (rows per slice + prologue)
    const long long slots = (C == 64 ? 2 : 1) * 256;
    long long best_s = 1;
    double best_cost = 1e30;
    for (long long sl = 1; sl <= (a.H >= 16 ? a.H / 8 : 1); ++sl) {
        const long long rows = (a.H + sl - 1) / sl, nsl = (a.H + rows - 1) / rows;
        const double cost = (double)((bx * nsl + slots - 1) / slots) * (double)(rows + 3.5);
        if (cost < best_cost - 1e-9) { best_cost = cost; best_s = nsl; }
    }
    a.rows_per_slice = (int)((a.H + best_s - 1) / best_s);
    const long long slices = (a.H + a.rows_per_slice - 1) / a.rows_per_slice;
    YOLO_LAUNCH((res_block_kernel<C, D, T>), dim3((unsigned)bx, (unsigned)slices), dim3(256), 0, st, a);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// Fused residual block (inference): y = x + lrelu(bn2(conv3x3(lrelu(bn1(conv1x1(x)))))); x, y (N,H,W,C) bf16 NHWC;
// w1_packed / w2_packed: yolo_pack_conv_weights images of the (C/2, C, 1, 1) and (C, C/2, 3, 3) convs; scale / bias:
// yolo_fold_bn of each layer.  bf16, C in {64, 128} only (YOLO_EUNSUPPORTED otherwise: run the two layers separately).
extern "C" int yolo_res_block_fwd(const void* x, const void* w1_packed, const float* scale1, const float* bias1,
                                  const void* w2_packed, const float* scale2, const float* bias2, void* y, int N, int H,
                                  int W, int C, int dtype, float slope, void* stream) {
    if (!x || !w1_packed || !scale1 || !bias1 || !w2_packed || !scale2 || !bias2 || !y || N <= 0 || H <= 0 || W <= 0)
        return YOLO_EINVAL;
    if (!(slope >= 0.f && slope <= 1.f)) return YOLO_EINVAL;
    if ((dtype != YOLO_BF16 && dtype != YOLO_F16) || (C != 64 && C != 128)) return YOLO_EUNSUPPORTED;
    ResArgs a;
    a.x = (const char*)x; a.wp1 = (const char*)w1_packed; a.scale1 = scale1; a.bias1 = bias1;
    a.wp2 = (const char*)w2_packed; a.scale2 = scale2; a.bias2 = bias2; a.y = (char*)y;
    a.N = N; a.H = H; a.W = W; a.slope = slope;
    a.Cpad1 = round_up(C / 2, YOLO_COUT_PAD); a.Cpad2 = round_up(C, YOLO_COUT_PAD);
    static const int dknob = (int)YOLO_LAB_ENV("YOLO_RB_D", 0);          // (ablation knob: rows in flight)
    // two output rows per step: C = 128 by default (102 against 117 us at 32 x 104 x 104, 474 against 507 at 64 x 152 x 152, same-box
    // probes; C = 64 would drop from two blocks per CU to one: 141 against 122 us).  (A/B knob YOLO_RB_R2: 0 = never, 2 / 3 = both
    // widths with that many register sets)
    static const int r2knob = (int)YOLO_LAB_ENV("YOLO_RB_R2", -1);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YOLO_F16)                                               // (the defaults; the knobs below are bf16 ablations)
        return C == 128 ? launch_res_block2<128, 2, f16_t>(a, st) : launch_res_block<64, 3, f16_t>(a, st);
    if (r2knob == 2) return C == 64 ? launch_res_block2<64, 2>(a, st) : launch_res_block2<128, 2>(a, st);
    if (r2knob == 3) return C == 64 ? launch_res_block2<64, 3>(a, st) : launch_res_block2<128, 3>(a, st);
    if (r2knob < 0 && C == 128 && !dknob) return launch_res_block2<128, 2>(a, st);
    if (dknob == 2) return C == 64 ? launch_res_block<64, 2>(a, st) : launch_res_block<128, 2>(a, st);
    if (dknob == 4) return C == 64 ? launch_res_block<64, 4>(a, st) : launch_res_block<128, 4>(a, st);
    if (dknob == 6) return C == 64 ? launch_res_block<64, 6>(a, st) : launch_res_block<128, 6>(a, st);
    return C == 64 ? launch_res_block<64, 3>(a, st) : launch_res_block<128, 3>(a, st);
}
