// Stem: the network's first _conv2d (basic_yolo.py:20) -- Conv3x3 s1 p1 over the (N,3,H,W) float32
// NCHW image + folded BN + LeakyReLU -> (N,H,W,Cout) bf16 NHWC -- as one HBM-bound kernel.
//
// Cin = 3 makes K = 27, too thin for the generic implicit GEMM (one 64-byte K-chunk would be 3/32
// full).  Here the block stages a 6-row x 68-column halo tile of the image into LDS as NHWC4 bf16
// (channel 3 = 0; NCHW reads are coalesced along W in each plane), and K is re-ordered (kh | kw, c4):
// one MFMA k-step of 16 = one kernel row = [kw0 c0..3 | kw1 c0..3 || kw2 c0..3 | 0000], which for a
// pixel is 16 + 8 contiguous LDS bytes -- no gather.  3 MFMA 32x32x16 per 32 pixels x 32 couts.
// D rows = cout, so a lane ends with 4 consecutive channels of one pixel (8-byte NHWC stores).
#include "common.h"

constexpr int STEM_TW = 64;            // output pixels per tile row
constexpr int STEM_RW = 4;             // output rows per wave
constexpr int STEM_TH = 4 * STEM_RW;   // output rows per tile (4 waves)
constexpr int STEM_PW = STEM_TW + 4;   // LDS row pitch in pixels (halo + over-read slack)

// STATS (training): per-channel sum / sum of squares of the STORED bf16 outputs, one partial row [2][Cout] per wave
// (row = block * 4 + wave), for yolo_bn_train_fwd_partials -- BatchNorm's batch statistics without a pass over the
// 416 x 416 x 32 map (the largest reduction of the step).  Needs the 16-byte store path (Cout % 8 == 0, 64 % (Cout / 8) == 0).
template <int MI, int STATS = 0, typename T = bf16_t>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ bias, uint16_t* __restrict__ y,
                                                        int N, int H, int W, int Cout, float slope, int tiles_x,
                                                        int tiles_y, float* __restrict__ part = nullptr) {
    __shared__ __attribute__((aligned(16))) uint2 tile[(STEM_TH + 2) * STEM_PW];
    __shared__ __attribute__((aligned(16))) uint2 obuf[4 * 64 * (MI * 32 * 2 + 16) / 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int x0 = tx * STEM_TW, y0 = ty * STEM_TH;
    const long long HW = (long long)H * W;
    const float* xn = x + (long long)n * 3 * HW;

    // ---- stage the halo tile: NCHW f32 -> LDS NHWC4 bf16 (all loads of the block issued before the first
    //      LDS write: one HBM latency per tile, not one per pass) -------------------------------------------
    constexpr int NSLOT = (STEM_TH + 2) * STEM_PW;
    constexpr int NPASS = (NSLOT + 255) / 256;
    float c0[NPASS], c1[NPASS], c2[NPASS];
    bool okp[NPASS];
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
        const int sl = min(tid + j * 256, NSLOT - 1);
        const int r = sl / STEM_PW, c = sl - r * STEM_PW;
        const int iy = y0 + r - 1, ix = x0 + c - 1;
        okp[j] = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const long long o = (long long)min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 1);
        c0[j] = xn[o]; c1[j] = xn[HW + o]; c2[j] = xn[2 * HW + o];
    }
    // ---- weight fragments: lane (cout = mi*32 + l31, k-half h), one per kernel row -----------------
    uint4 wf[3][MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int co = mi * 32 + l31;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            float v[2][3];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const int kw = 2 * h + q;          // h=0: kw 0,1 ; h=1: kw 2,(3 = padding)
                    v[q][ci] = (co < Cout && kw < 3) ? w[((co * 3 + ci) * 3 + kh) * 3 + min(kw, 2)] : 0.f;
                }
            wf[kh][mi] = make_uint4(Elem<T>::pack2(v[0][0], v[0][1]), Elem<T>::pack2(v[0][2], 0.f),
                                    Elem<T>::pack2(v[1][0], v[1][1]), Elem<T>::pack2(v[1][2], 0.f));
        }
    }
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
        const int sl = tid + j * 256;
        if (sl < NSLOT)
            tile[sl] = okp[j] ? make_uint2(Elem<T>::pack2(c0[j], c1[j]), Elem<T>::pack2(c2[j], 0.f)) : make_uint2(0u, 0u);
    }
    __syncthreads();

    constexpr int OP = (MI * 32 * 2 + 16);             // LDS pitch per output pixel: bf16 couts + 16 B pad (16-byte aligned rows)
    char* ot = (char*)obuf + wave * (64 * OP);
    const int upp = Cout / 4;                          // 8-byte units per output pixel
    const int npx = min(STEM_TW, W - x0);
    float ssum[8], qsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ssum[e] = qsum[e] = 0.f;
#pragma unroll 1
    for (int rw = 0; rw < STEM_RW; ++rw) {
        const int lr = wave * STEM_RW + rw;            // output row within the tile
        const int oy = y0 + lr;
        if (oy >= H) break;
        f32x16 acc[MI][2];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int s = (lr + kh) * STEM_PW + ni * 32 + l31 + 2 * h;
                const uint2 lo = tile[s], hi = tile[s + 1];
                const uint4 bf = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[mi][ni] = mfma16<T>(wf[kh][mi], bf, acc[mi][ni]);
            }
        }
        // BN + LeakyReLU, then transpose through this wave's LDS scratch so each lane stores 8 contiguous
        // bytes of a contiguous NHWC run (a wave's 64 pixels x Cout channels are one run of the output row)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = mi * 32 + 8 * g + 4 * h;
                f32x4 sc = {0.f, 0.f, 0.f, 0.f}, bi = {0.f, 0.f, 0.f, 0.f};
                if (co < Cout) { sc = *(const f32x4*)(scale + co); bi = *(const f32x4*)(bias + co); }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = acc[mi][ni][4 * g + e] * sc[e] + bi[e];
                        v[e] = leaky(t, slope);
                    }
                    *(uint2*)(ot + (ni * 32 + l31) * OP + co * 2) =
                        make_uint2(Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3]));
                }
            }
        }
        uint16_t* yrow = y + (((long long)n * H + oy) * W + x0) * Cout;
        wave_lds_fence();
        // same-wave LDS write -> read -> (next row's) write: in order.  16-byte stores where the row allows it
        // (half the store instructions of the 8-byte version: the kernel is store-issue bound)
        if ((Cout & 7) == 0) {
            const int upp16 = Cout / 8;
            for (int u = lane; u < npx * upp16; u += 64) {
                const int px = u / upp16, q = u - px * upp16;
                const uint4 o = *(const uint4*)(ot + px * OP + q * 16);
                *(uint4*)(yrow + (long long)px * Cout + q * 8) = o;
                if constexpr (STATS) {                 // (q = lane % upp16 for every u of this lane: 64 % upp16 == 0)
                    const uint32_t ow[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float t = ((e & 1) ? Elem<T>::hi(ow[e >> 1]) : Elem<T>::lo(ow[e >> 1]));
                        ssum[e] += t; qsum[e] += t * t;
                    }
                }
            }
        } else {
            for (int u = lane; u < npx * upp; u += 64) {
                const int px = u / upp, q = u - px * upp;
                *(uint2*)(yrow + (long long)px * Cout + q * 4) = *(const uint2*)(ot + px * OP + q * 8);
            }
        }
    }
    if constexpr (STATS) {
        // combine the 64 / upp16 lanes that own the same 8 channels through the wave's scratch, one partial row per wave
        const int upp16 = Cout / 8, nl = 64 / upp16;
        float* sc4 = (float*)ot;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc4[lane * 16 + e] = ssum[e]; sc4[lane * 16 + 8 + e] = qsum[e]; }
        wave_lds_fence();
        float* prow = part + ((long long)blockIdx.x * 4 + wave) * 2 * Cout;
        for (int o = lane; o < upp16 * 16; o += 64) {
            const int qx = o >> 4, val = o & 15;
            float t = 0.f;
            for (int r = 0; r < nl; ++r) t += sc4[(r * upp16 + qx) * 16 + val];
            prow[(val >> 3) * Cout + qx * 8 + (val & 7)] = t;
        }
    }
}

extern "C" int yolo_stem_stats_rows(int N, int H, int W, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0) return YOLO_EINVAL;
    if ((Cout % 8) || Cout > 64 || (64 % (Cout / 8))) return YOLO_EUNSUPPORTED;
    const long long rows = (long long)N * ((W + STEM_TW - 1) / STEM_TW) * ((H + STEM_TH - 1) / STEM_TH) * 4;
    return rows > 0x7fffffffLL ? YOLO_EUNSUPPORTED : (int)rows;
}

extern "C" int yolo_stem_conv_fwd_stats(const float* x_nchw, const float* w_oihw, const float* scale, const float* bias,
                                        void* y, int N, int H, int W, int Cin, int Cout, int dtype, float slope,
                                        float* partials, void* stream) {
    if (!x_nchw || !w_oihw || !scale || !bias || !y || !partials || N <= 0 || H <= 0 || W <= 0) return YOLO_EINVAL;
    if (!(slope >= 0.f && slope <= 1.f)) return YOLO_EINVAL;
    if (Cin != 3 || dtype != YOLO_BF16 || yolo_stem_stats_rows(N, H, W, Cout) <= 0) return YOLO_EUNSUPPORTED;
    const int tiles_x = (W + STEM_TW - 1) / STEM_TW, tiles_y = (H + STEM_TH - 1) / STEM_TH;
    const long long grid = (long long)N * tiles_x * tiles_y;
    if (Cout <= 32)
        YOLO_LAUNCH((stem_mfma_kernel<1, 1>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x_nchw, w_oihw, scale,
                    bias, (uint16_t*)y, N, H, W, Cout, slope, tiles_x, tiles_y, partials);
    else
        YOLO_LAUNCH((stem_mfma_kernel<2, 1>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x_nchw, w_oihw, scale,
                    bias, (uint16_t*)y, N, H, W, Cout, slope, tiles_x, tiles_y, partials);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ---- split types (YOLO_BF16X3): the stem as a DIRECT fp32 convolution on the vector pipe -------------------------------------
// K = 27: three MFMA passes over a K padded to 32 would buy nothing, and exact fp32 products cost the stem ~0.1 ms at 416x416 bs 32
// against the ~12 ms of the split pass.  A thread owns 8 couts of one pixel (Cout / 8 adjacent lanes share a pixel: their image
// loads are one request), the 27 x Cout weights sit in LDS as [tap][cout], and the output leaves split: hi = round(v),
// lo = round(v - hi) one padded plane (round_up(Cout, 32) elements) further (dense split layout: a wave writes whole 16-byte pieces of
// consecutive pixels).
template <typename T>
__global__ __launch_bounds__(256) void stem_split_kernel(const float* __restrict__ x, const float* __restrict__ w_oihw,
                                                         const float* __restrict__ scale, const float* __restrict__ bias,
                                                         uint16_t* __restrict__ y, int H, int W, int Cout, int tpp_shift, float slope,
                                                         int tiles_x, int tiles_y) {
    // A block = 16 columns x TH rows of one image (256 threads = 256 >> tpp_shift pixels x Cout / 8 threads per pixel).  The input
    // tile + halo is staged in LDS once (the first version read its 27 inputs per thread from global memory: 9.3 M wave-wide load
    // instructions per bs-32 pass through the address unit -- 700 us, 6 % of the split pass; staged: one or two loads per thread).
    constexpr int TW = 16, XP = TW + 2 + 1;                 // padded pitch
    __shared__ __attribute__((aligned(16))) float wl[27 * 64];
    __shared__ float xs[3 * 18 * XP];
    const int TH = 16 >> tpp_shift;
    const int tid = threadIdx.x;
    for (int i = tid; i < 27 * Cout; i += 256) {
        const int tap = i / Cout, co = i - tap * Cout;
        wl[i] = w_oihw[co * 27 + tap];                      // OIHW: [co][c][kh][kw] -> [c * 9 + kh * 3 + kw][co]
    }
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const long long n = b / tiles_y;
    const long long HW = (long long)H * W;
    const float* xn = x + n * 3 * HW;
    const int y0 = ty * TH - 1, x0 = tx * TW - 1;
    for (int i = tid; i < 3 * (TH + 2) * (TW + 2); i += 256) {
        const int c = i / ((TH + 2) * (TW + 2)), r_ = i - c * (TH + 2) * (TW + 2);
        const int r = r_ / (TW + 2), col = r_ - r * (TW + 2);
        const int iy = y0 + r, ix = x0 + col;
        const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
        // (unconditional load from a clamped address, masked afterwards: a predicated load serialises on vmcnt(0), NOTES 4.2)
        const float v = xn[c * HW + (long long)min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 1)];
        xs[(c * 18 + r) * XP + col] = in ? v : 0.f;
    }
    __syncthreads();
    const int lp = tid >> tpp_shift, q = tid & ((1 << tpp_shift) - 1);
    const int ly = lp >> 4, lx = lp & 15;
    const int yy = ty * TH + ly, xx = tx * TW + lx;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float v = xs[(c * 18 + ly + kh) * XP + lx + kw];
                const float* wr = wl + (c * 9 + kh * 3 + kw) * Cout + q * 8;
                const f32x4 w0 = *(const f32x4*)wr, w1 = *(const f32x4*)(wr + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[e] = fmaf(v, w0[e], acc[e]); acc[4 + e] = fmaf(v, w1[e], acc[4 + e]); }
            }
    if (yy >= H || xx >= W) return;
    const int co = q * 8;
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a0 = leaky(acc[2 * e] * scale[co + 2 * e] + bias[co + 2 * e], slope);
        const float a1 = leaky(acc[2 * e + 1] * scale[co + 2 * e + 1] + bias[co + 2 * e + 1], slope);
        hi[e] = Elem<T>::pack2(a0, a1);
        lo[e] = Elem<T>::pack2(a0 - Elem<T>::lo(hi[e]), a1 - Elem<T>::hi(hi[e]));
    }
    const int Cp = round_up(Cout, 32);                      // (a split plane is padded to whole 32-channel chunks; the pad stays as the caller zeroed it)
    uint16_t* yp = y + ((n * H + yy) * W + xx) * 2 * Cp + co;
    *(uint4*)yp = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *(uint4*)(yp + Cp) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

extern "C" int yolo_stem_conv_fwd(const float* x_nchw, const float* w_oihw, const float* scale, const float* bias,
                                  void* y, int N, int H, int W, int Cin, int Cout, int dtype, float slope,
                                  void* stream) {
    if (!x_nchw || !w_oihw || !scale || !bias || !y || N <= 0 || H <= 0 || W <= 0) return YOLO_EINVAL;
    if (!(slope >= 0.f && slope <= 1.f)) return YOLO_EINVAL;
    if (Cin != 3 || Cout <= 0 || (Cout % 4) || Cout > 64) return YOLO_EUNSUPPORTED;
    if (dtype == YOLO_BF16X3 || dtype == YOLO_F16X3) {
        if (Cout != 8 && Cout != 16 && Cout != 32 && Cout != 64) return YOLO_EUNSUPPORTED;
        const int sh = Cout == 8 ? 0 : Cout == 16 ? 1 : Cout == 32 ? 2 : 3;
        const int th = 16 >> sh, tiles_x_ = (W + 15) / 16, tiles_y_ = (H + th - 1) / th;
        const long long nblk = (long long)N * tiles_x_ * tiles_y_;
        if (nblk > 0x7fffffffLL) return YOLO_EUNSUPPORTED;
        if (dtype == YOLO_BF16X3)
            YOLO_LAUNCH(stem_split_kernel<bf16x3_t>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, x_nchw, w_oihw,
                        scale, bias, (uint16_t*)y, H, W, Cout, sh, slope, tiles_x_, tiles_y_);
        else
            YOLO_LAUNCH(stem_split_kernel<f16x3_t>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, x_nchw, w_oihw,
                        scale, bias, (uint16_t*)y, H, W, Cout, sh, slope, tiles_x_, tiles_y_);
        YOLO_LAUNCH_CHECK();
        return YOLO_OK;
    }
    if (dtype != YOLO_BF16 && dtype != YOLO_F16) return YOLO_EUNSUPPORTED;      // the fp32 path goes through the generic kernel
    const int tiles_x = (W + STEM_TW - 1) / STEM_TW, tiles_y = (H + STEM_TH - 1) / STEM_TH;
    const long long grid = (long long)N * tiles_x * tiles_y;
    if (grid > 0x7fffffffLL) return YOLO_EUNSUPPORTED;
    if (dtype == YOLO_F16) {
        if (Cout <= 32)
            YOLO_LAUNCH((stem_mfma_kernel<1, 0, f16_t>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x_nchw, w_oihw, scale,
                        bias, (uint16_t*)y, N, H, W, Cout, slope, tiles_x, tiles_y);
        else
            YOLO_LAUNCH((stem_mfma_kernel<2, 0, f16_t>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x_nchw, w_oihw, scale,
                        bias, (uint16_t*)y, N, H, W, Cout, slope, tiles_x, tiles_y);
    } else if (Cout <= 32)
        YOLO_LAUNCH((stem_mfma_kernel<1, 0>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x_nchw, w_oihw, scale,
                    bias, (uint16_t*)y, N, H, W, Cout, slope, tiles_x, tiles_y);
    else
        YOLO_LAUNCH((stem_mfma_kernel<2, 0>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x_nchw, w_oihw, scale,
                    bias, (uint16_t*)y, N, H, W, Cout, slope, tiles_x, tiles_y);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}
