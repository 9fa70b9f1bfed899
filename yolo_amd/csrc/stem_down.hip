// The network's first two layers as ONE inference kernel: the stem _conv2d (basic_yolo.py:20: Conv3x3 s1 p1, 3 -> 32)
// and the first stage's down-sampling _conv2d (basic_yolo.py:24: Conv3x3 s2 p1, 32 -> 64), each with folded BN +
// LeakyReLU, from the (N,3,H,W) float32 NCHW image to the (N,H/2,W/2,64) bf16 NHWC map.
//
// Run separately (stem.hip + conv_stream.hip) both are HBM-bound on the full-resolution 32-channel map between them:
// written once and read once it is 3.0 GB of the 3.3 GB the two kernels move at 608x608 bs 64 (1.0 ms of a 15.8 ms
// forward pass).  Here it never leaves LDS:
//   * a block (4 waves) owns a column strip (<= 62 output pixels) of one image and walks down its output rows, as
//     conv_stream_kernel does; the down conv's weights stay in registers (18 A-fragments per wave);
//   * the stem output lives in a 5-row rolling LDS ring in exactly the layout conv_stream stages its input in, but it
//     is COMPUTED there: each step the four waves produce the two new stem rows (3 MFMAs per 32 pixels, K = one kernel
//     row of the NHWC4 image as in stem.hip) from an 8-row rolling image window, while the image rows of the next
//     step are in flight in registers;
//   * stem pixels outside the image are written as zeros (they are the down conv's zero padding, NOT stem(padding)).
// Same operand order, accumulation order and rounding points as the two separate kernels: results are bit-identical.
#include "common.h"
#include <stdio.h>
#include <type_traits>

namespace {
constexpr int C1 = 32, C2 = 64;
// a step covers 64 output pixels (MFMA columns); the usable strip width is <= 62 so that a stem row is 4 groups of 32
constexpr int SW_MAX = 62;
constexpr int XW = 129;                  // stem ring row capacity (pixel 128 is only read by discarded lanes)
constexpr int PITCH = C1 * 2 + 16;       // bytes per stem pixel in the ring (+16: conflict-free 16-byte reads)
constexpr int XROW = XW * PITCH;
// A ring row holds its even pixels first, then the odd ones (pixel p at ((p & 1) * XHALF + (p >> 1)) * PITCH): the stride-2 down conv
// reads every other pixel, and at a lane stride of 2 * PITCH = 160 bytes a 16-lane ds_read_b128 group covers only 8 of the 16 bank
// quads (two-way conflicts on every fragment read: SQ_LDS_BANK_CONFLICT was 34 % of the LDS-active cycles); at PITCH it covers all 16.
constexpr int XHALF = (XW + 1) / 2;
constexpr int RING = 5;                  // 3 stem rows in use + 2 being produced
constexpr int IPW = 136;                 // image window row pitch in pixels (8 B each: NHWC4 bf16)
constexpr int IROWS = 8;                 // image window rows (4 in use + 2 arriving, power of two)
constexpr int SCR = 32 * 144;            // per-wave epilogue scratch
constexpr int KSTEPS = 18;               // 9 taps x 2 k-steps of 16 channels

struct Args {
    const float* x;
    const float* w1;
    const float* scale1;
    const float* bias1;
    const char* wp2;
    const float* scale2;
    const float* bias2;
    char* y;
    int N, H, W, Ho, Wo, Cout_pad2;
    int nstrips, strip_w, rows_per_slice;
    float slope;
};
}  // namespace

template <typename T>
__global__ __launch_bounds__(256) void stem_down_kernel(Args a) {
    __shared__ __attribute__((aligned(16))) char smem[RING * XROW + IROWS * IPW * 8 + 4 * SCR];
    char* xl = smem;
    uint2* img = (uint2*)(smem + RING * XROW);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wave_c = wave & 1, wave_p = wave >> 1;
    char* scr = smem + RING * XROW + IROWS * IPW * 8 + wave * SCR;

    const int strip = blockIdx.x % a.nstrips;
    const int n = blockIdx.x / a.nstrips;
    const int ox0 = strip * a.strip_w;
    const int ox_end = min(ox0 + a.strip_w, a.Wo);
    const int oy0 = blockIdx.y * a.rows_per_slice;
    const int oy1 = min(oy0 + a.rows_per_slice, a.Ho);
    if (oy0 >= oy1) return;
    const int H = a.H, W = a.W, Ho = a.Ho, Wo = a.Wo;
    const long long HW = (long long)H * W;
    const float* xn = a.x + (long long)n * 3 * HW;
    const int sx0 = 2 * ox0 - 1;              // image column of stem pixel 0
    const int icol0 = sx0 - 1;                // image column of window pixel 0
    const int iyb = 2 * oy0 - 2;              // image row held in window slot 0 at the start

    // ---- hygiene: pixels of both LDS arrays that are never written are only read by discarded lanes ---------------
    for (int i = tid; i < (RING * XROW + IROWS * IPW * 8) / 16; i += 256) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);

    // ---- down conv: the wave's A-fragments (32 couts x K = 288) from the packed image, as conv_stream does --------
    uint4 A[KSTEPS];
    {
        const int row = wave_c * 32 + l31;
        const int swz = (l31 >> 2) & 3;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int tap = ks >> 1, kc = ks & 1;
            const int unit = (kc * 2 + h) ^ swz;
            A[ks] = *(const uint4*)(a.wp2 + ((long long)tap * a.Cout_pad2 + row) * 64 + unit * 16);
        }
    }
    // ---- stem: weight fragments (cout = l31, k-half h), one per kernel row (stem.hip's K order) -------------------
    uint4 wf[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        float v[2][3];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const int kw = 2 * h + q;          // h=0: kw 0,1 ; h=1: kw 2,(3 = padding)
                v[q][ci] = kw < 3 ? a.w1[((l31 * 3 + ci) * 3 + kh) * 3 + min(kw, 2)] : 0.f;
            }
        wf[kh] = make_uint4(Elem<T>::pack2(v[0][0], v[0][1]), Elem<T>::pack2(v[0][2], 0.f), Elem<T>::pack2(v[1][0], v[1][1]),
                            Elem<T>::pack2(v[1][2], 0.f));
    }
    // ---- epilogue constants of the down conv (after the transpose a lane owns 8 couts of a pixel row) -------------
    const int ecol = lane & 3, erow0 = lane >> 2;
    const int eco = wave_c * 32 + ecol * 8;
    float sc2[8], bi2[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x4 s4 = *(const f32x4*)(a.scale2 + eco + 4 * q);
        const f32x4 b4 = *(const f32x4*)(a.bias2 + eco + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc2[4 * q + e] = s4[e]; bi2[4 * q + e] = b4[e]; }
    }
    const float slope = a.slope;
    // folded BN of the stem for the 16 couts a lane holds after its MFMAs (rows 8g + 4h + e)
    float sc1[16], bi1[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 s4 = *(const f32x4*)(a.scale1 + 8 * g + 4 * h), b4 = *(const f32x4*)(a.bias1 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc1[4 * g + e] = s4[e]; bi1[4 * g + e] = b4[e]; }
    }

    // ---- image window: one pixel (3 planes) per thread per pass; rows [first, first+nrows) -------------------------
    auto load_img = [&](int iy_first, int r, float (&c)[3], bool& ok) {
        const int q = tid & 127;
        const int iy = iy_first + r * 2 + (tid >> 7), ix = icol0 + q;
        ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const long long o = (long long)min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 1);
        c[0] = xn[o]; c[1] = xn[HW + o]; c[2] = xn[2 * HW + o];
    };
    auto store_img = [&](int iy_first, int r, const float (&c)[3], bool ok) {
        const int q = tid & 127;
        const int iy = iy_first + r * 2 + (tid >> 7);
        img[((iy - iyb) & (IROWS - 1)) * IPW + q] =
            ok ? make_uint2(Elem<T>::pack2(c[0], c[1]), Elem<T>::pack2(c[2], 0.f)) : make_uint2(0u, 0u);
    };
    // ---- one group of 32 stem pixels of stem row sy: 3 MFMAs, BN + LeakyReLU, one rounding, into ring slot `slot` --
    auto stem_group = [&](int sy, int slot, int pxg) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int s = ((sy - 1 + kh - iyb) & (IROWS - 1)) * IPW + pxg * 32 + l31 + 2 * h;
            const uint2 lo = img[s], hi = img[s + 1];
            acc = mfma16<T>(wf[kh], make_uint4(lo.x, lo.y, hi.x, hi.y), acc);
        }
        const int px = pxg * 32 + l31;
        const int sx = sx0 + px;
        const bool inside = sy >= 0 && sy < H && sx >= 0 && sx < W;
        char* dst = xl + slot * XROW + ((px & 1) * XHALF + (px >> 1)) * PITCH;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = 8 * g + 4 * h;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = leaky(acc[4 * g + e] * sc1[4 * g + e] + bi1[4 * g + e], slope);
            *(uint2*)(dst + co * 2) = inside ? make_uint2(Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3])) : make_uint2(0u, 0u);
        }
    };

    __syncthreads();                                        // the zero fill
    // ---- prologue: image rows iyb .. iyb+6, then stem rows 2 oy0 - 1 .. 2 oy0 + 1 into ring slots 0..2 ------------
    {
        float c[4][3];
        bool ok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) load_img(iyb, r, c[r], ok[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < 3 || tid < 128) store_img(iyb, r, c[r], ok[r]);      // 7 rows: the 8th slot belongs to row iyb + 7
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 3; ++r) stem_group(2 * oy0 - 1 + r, r, wave);
    __syncthreads();

    const int b_off = (wave_p * 32 + l31) * PITCH + h * 16;          // (pixel 2 j + kw: half kw & 1, index j + (kw >> 1))
    int slot0 = 0;
    // image rows 2oy+5, 2oy+6 are used by step oy+1's stem rows and stored at the bottom of step oy; they are requested TWO
    // steps ahead (top of step oy-1) into one of two register sets, unconditionally (clamped addresses: always readable) --
    // a uniform branch around the request would make the compiler wait for every load in flight before the other set's store
    float cimg[2][3];
    bool okimg[2] = {false, false};
    load_img(2 * oy0 + 5, 0, cimg[0], okimg[0]);
    auto step = [&](auto par_c, int oy) {
        constexpr int PAR = decltype(par_c)::value;
        const bool more = oy + 1 < oy1;
        load_img(2 * oy + 7, 0, cimg[PAR ^ 1], okimg[PAR ^ 1]);
        float (&c)[3] = cimg[PAR];
        const bool ok = okimg[PAR];
        long long yo[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ox = ox0 + wave_p * 32 + erow0 + 16 * k;
            yo[k] = ox < ox_end ? ((((long long)n * Ho + oy) * Wo + ox) * C2 + eco) * 2 : -1;
        }
        // ---- down conv of output row oy: 9 taps x 2 k-steps over the stem ring ----------------------------------
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            int slot = slot0 + kh;
            if (slot >= RING) slot -= RING;
            const char* rowp = xl + slot * XROW + b_off;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const uint4 bf = *(const uint4*)(rowp + ((kw & 1) * XHALF + (kw >> 1)) * PITCH + kc * 32);
                    acc = mfma16<T>(A[(kh * 3 + kw) * 2 + kc], bf, acc);
                }
        }
        // ---- the two stem rows the next output row adds (2oy+2, 2oy+3): this wave's pixel group of each ----------
        if (more) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                int slot = slot0 + 3 + r;
                if (slot >= RING) slot -= RING;
                stem_group(2 * oy + 2 + r, slot, wave);
            }
        }
        // ---- epilogue of the down conv: folded BN, LeakyReLU, one rounding, 16-byte row stores --------------------
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
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
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = leaky(v[e] * sc2[e] + bi2[e], slope);
            if (yo[k] >= 0)
                *(uint4*)(a.y + yo[k]) = make_uint4(Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3]),
                                                    Elem<T>::pack2(v[4], v[5]), Elem<T>::pack2(v[6], v[7]));
        }
        if (more) store_img(2 * oy + 5, 0, c, ok);
        slot0 += 2;
        if (slot0 >= RING) slot0 -= RING;
        __syncthreads();
    };
    for (int oy = oy0; oy < oy1; oy += 2) {
        step(std::integral_constant<int, 0>{}, oy);
        if (oy + 1 < oy1) step(std::integral_constant<int, 1>{}, oy + 1);
    }
}

// Fused stem + first down-sampling conv (inference).  w1_oihw (32,3,3,3) float32; w2_packed: yolo_pack_conv_weights image
// of the (64,32,3,3) conv; scale / bias: folded BN of each layer.  bf16 only; C1 == 32 and C2 == 64 only.
extern "C" int yolo_stem_down_fwd(const float* x_nchw, const float* w1_oihw, const float* scale1, const float* bias1,
                                  const void* w2_packed, const float* scale2, const float* bias2, void* y, int N, int H,
                                  int W, int C1_, int C2_, int dtype, float slope, void* stream) {
    if (!x_nchw || !w1_oihw || !scale1 || !bias1 || !w2_packed || !scale2 || !bias2 || !y || N <= 0 || H <= 0 || W <= 0)
        return YOLO_EINVAL;
    if (!(slope >= 0.f && slope <= 1.f)) return YOLO_EINVAL;
    if (C1_ != C1 || C2_ != C2 || (dtype != YOLO_BF16 && dtype != YOLO_F16)) return YOLO_EUNSUPPORTED;
    Args a;
    a.x = x_nchw; a.w1 = w1_oihw; a.scale1 = scale1; a.bias1 = bias1;
    a.wp2 = (const char*)w2_packed; a.scale2 = scale2; a.bias2 = bias2; a.y = (char*)y;
    a.N = N; a.H = H; a.W = W; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.Cout_pad2 = round_up(C2, YOLO_COUT_PAD);
    a.slope = slope;
    a.nstrips = (a.Wo + SW_MAX - 1) / SW_MAX;
    a.strip_w = (a.Wo + a.nstrips - 1) / a.nstrips;
    const long long bx = (long long)N * a.nstrips;
    if (bx > 0x7fffffffLL) return YOLO_EUNSUPPORTED;
    long long slices = (1536 + bx - 1) / bx;                 // a slice re-computes 3 stem rows: keep them >= 16 rows
    if (slices > a.Ho / 16) slices = a.Ho / 16;
    if (slices < 1) slices = 1;
    a.rows_per_slice = (int)((a.Ho + slices - 1) / slices);
    slices = (a.Ho + a.rows_per_slice - 1) / a.rows_per_slice;
    if (dtype == YOLO_F16)
        YOLO_LAUNCH(stem_down_kernel<f16_t>, dim3((unsigned)bx, (unsigned)slices), dim3(256), 0, (hipStream_t)stream, a);
    else
        YOLO_LAUNCH(stem_down_kernel<bf16_t>, dim3((unsigned)bx, (unsigned)slices), dim3(256), 0, (hipStream_t)stream, a);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}
