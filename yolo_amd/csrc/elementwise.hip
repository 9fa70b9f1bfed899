// HBM-bound plumbing kernels: layout conversion, BN folding, up-sample + concat.
#include "common.h"

extern "C" int yolo_version(void) { return YOLO_ABI_VERSION; }

extern "C" int yolo_padded_channels(int C) { return round_up(C, YOLO_COUT_PAD); }

// ---- BatchNorm folding (inference) ---------------------------------------------------------
__global__ void fold_bn_kernel(const float* gamma, const float* beta, const float* mean, const float* var,
                               float eps, float* scale, float* bias, int C, int Cpad) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Cpad) return;
    float s = 0.f, b = 0.f;
    if (c < C) {
        if (gamma) {
            s = gamma[c] / sqrtf(var[c] + eps);
            b = beta[c] - mean[c] * s;
        } else {
            s = 1.f;
            b = beta ? beta[c] : 0.f;
        }
    }
    scale[c] = s;
    bias[c] = b;
}

extern "C" int yolo_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var,
                            float eps, float* scale, float* bias, int C, void* stream) {
    if (!scale || !bias || C <= 0) return YOLO_EINVAL;
    if (gamma && (!beta || !mean || !var)) return YOLO_EINVAL;
    const int Cpad = round_up(C, YOLO_COUT_PAD);
    YOLO_LAUNCH(fold_bn_kernel, dim3((Cpad + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       mean, var, eps, scale, bias, C, Cpad);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ---- NCHW f32 -> NHWC(Cpad) dtype ----------------------------------------------------------
// One thread per pixel: reads are coalesced along W in every channel plane (NCHW), the write is
// one contiguous Cpad-channel vector per pixel.
template <typename T, int CPAD>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int C, long long HW,
                                    long long total) {
    const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (p >= total) return;
    const long long n = p / HW, hw = p - n * HW;
    float v[CPAD];
#pragma unroll
    for (int c = 0; c < CPAD; ++c) v[c] = (c < C) ? x[(n * C + c) * HW + hw] : 0.f;
    if constexpr (sizeof(T) == 2) {
        uint32_t w[CPAD / 2];
#pragma unroll
        for (int c = 0; c < CPAD / 2; ++c) w[c] = Elem<T>::pack2(v[2 * c], v[2 * c + 1]);
        uint4* dst = (uint4*)((uint16_t*)y + p * CPAD);
#pragma unroll
        for (int q = 0; q < CPAD / 8; ++q) dst[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    } else {
        f32x4* dst = (f32x4*)((float*)y + p * CPAD);
#pragma unroll
        for (int q = 0; q < CPAD / 4; ++q) {
            f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            dst[q] = o;
        }
    }
}

extern "C" int yolo_nchw_to_nhwc(const float* x, void* y, int N, int C, int H, int W, int Cpad, int dtype,
                                 void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0) return YOLO_EINVAL;
    if (Cpad != 8 || C > 8) return YOLO_EUNSUPPORTED;
    const long long HW = (long long)H * W, total = HW * N;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (dtype == YOLO_BF16)
        YOLO_LAUNCH((nchw_to_nhwc_kernel<bf16_t, 8>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x,
                           (bf16_t*)y, C, HW, total);
    else if (dtype == YOLO_F16)
        YOLO_LAUNCH((nchw_to_nhwc_kernel<f16_t, 8>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x,
                           (f16_t*)y, C, HW, total);
    else if (dtype == YOLO_F32)
        YOLO_LAUNCH((nchw_to_nhwc_kernel<float, 8>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x,
                           (float*)y, C, HW, total);
    else
        return YOLO_EINVAL;
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ---- (N,H,W,C) u8 -> (N,C,H,W) f32 / 255  (cv_img_2_ndarray, yolo_gluon.py:335-357) ----------
__global__ void image_u8_to_nchw_kernel(const unsigned char* __restrict__ img, float* __restrict__ y, int C,
                                        long long HW, long long total) {
    const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (p >= total) return;
    const long long n = p / HW, hw = p - n * HW;
    for (int c = 0; c < C; ++c) y[(n * C + c) * HW + hw] = (float)img[p * C + c] / 255.f;
}

extern "C" int yolo_image_u8_to_nchw(const unsigned char* img, float* y, int N, int H, int W, int C,
                                     void* stream) {
    if (!img || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) return YOLO_EINVAL;
    const long long HW = (long long)H * W, total = HW * N;
    YOLO_LAUNCH(image_u8_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, img, y, C, HW, total);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ---- NHWC dtype -> NCHW f32 (debug / parity taps) ------------------------------------------
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int C, long long HW,
                                    long long total) {
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;   // over N*C*HW (NCHW order)
    if (idx >= total) return;
    const long long hw = idx % HW;
    const long long nc = idx / HW;
    const long long n = nc / C, c = nc - n * C;
    const long long src = (n * HW + hw) * C + c;
    if constexpr (sizeof(T) == 2)
        y[idx] = Elem<T>::lo(((const uint16_t*)x)[src]);
    else
        y[idx] = x[src];
}

extern "C" int yolo_nhwc_to_nchw(const void* x, float* y, int N, int C, int H, int W, int dtype, void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0) return YOLO_EINVAL;
    const long long HW = (long long)H * W, total = HW * N * C;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (dtype == YOLO_BF16)
        YOLO_LAUNCH(nhwc_to_nchw_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, y, C, HW, total);
    else if (dtype == YOLO_F16)
        YOLO_LAUNCH(nhwc_to_nchw_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const f16_t*)x, y, C, HW, total);
    else if (dtype == YOLO_F32)
        YOLO_LAUNCH(nhwc_to_nchw_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, y, C, HW, total);
    else
        return YOLO_EINVAL;
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ---- 2x nearest up-sample + channel concat (car/utils.py:92-93) ----------------------------
// 16-byte units; unit q of output pixel (n,y,x): q < U1 -> up[n, y/2, x/2], else route[n, y, x].
__global__ void upsample_concat_kernel(const uint4* __restrict__ up, const uint4* __restrict__ route,
                                       uint4* __restrict__ y, int H, int W, int U1, int U2, long long total) {
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int U = U1 + U2;
    const long long pix = idx / U;
    const int q = (int)(idx - pix * U);
    if (q >= U1) {
        y[idx] = route[pix * U2 + (q - U1)];
    } else {
        const long long HW = (long long)H * W;
        const long long n = pix / HW;
        const int hw = (int)(pix - n * HW);
        const int yy = hw / W, xx = hw - yy * W;
        const long long sp = (n * (H / 2) + (yy >> 1)) * (W / 2) + (xx >> 1);
        y[idx] = up[sp * U1 + q];
    }
}

extern "C" int yolo_upsample2x_concat(const void* up, const void* route, void* y, int N, int H, int W, int C1,
                                      int C2, int dtype, void* stream) {
    if (!up || !route || !y || N <= 0 || H <= 0 || W <= 0 || C1 <= 0 || C2 <= 0) return YOLO_EINVAL;
    if ((H & 1) || (W & 1) || !dtype_plain(dtype)) return YOLO_EINVAL;
    const int es = elem_size(dtype);
    if ((C1 * es) % 16 || (C2 * es) % 16) return YOLO_EUNSUPPORTED;
    const int U1 = C1 * es / 16, U2 = C2 * es / 16;
    const long long total = (long long)N * H * W * (U1 + U2);
    YOLO_LAUNCH(upsample_concat_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const uint4*)up, (const uint4*)route, (uint4*)y, H, W, U1, U2, total);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ---- synthetic-target compositing (RenderCar.render, car/render_car.py:135-137) -------------------------------
// out = clip((bg / 255) * (1 - mask) + fg * mask, 0, 1), all (B,3,H,W) float32; 4 elements per thread.
// UNIT: bg is already 0..1 (LPGenerator.add pastes onto RenderCar's output, licence_plate_render/__init__.py:163).
template <bool UNIT>
__global__ void composite_kernel(const f32x4* __restrict__ bg, const f32x4* __restrict__ fg, const f32x4* __restrict__ mask,
                                 f32x4* __restrict__ out, long long n4) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 b = bg[i], f = fg[i], m = mask[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float v = (UNIT ? b[e] : b[e] / 255.f) * (1.f - m[e]) + f[e] * m[e];
        o[e] = fminf(fmaxf(v, 0.f), 1.f);
    }
    out[i] = o;
}

extern "C" int yolo_composite(const float* bg, const float* fg, const float* mask, float* out, long long n, void* stream) {
    if (!bg || !fg || !mask || !out || n <= 0) return YOLO_EINVAL;
    if (n % 4) return YOLO_EUNSUPPORTED;
    const long long n4 = n / 4;
    YOLO_LAUNCH(composite_kernel<false>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)bg,
                (const f32x4*)fg, (const f32x4*)mask, (f32x4*)out, n4);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

extern "C" int yolo_composite_unit(const float* bg, const float* fg, const float* mask, float* out, long long n, void* stream) {
    if (!bg || !fg || !mask || !out || n <= 0) return YOLO_EINVAL;
    if (n % 4) return YOLO_EUNSUPPORTED;
    const long long n4 = n / 4;
    YOLO_LAUNCH(composite_kernel<true>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)bg,
                (const f32x4*)fg, (const f32x4*)mask, (f32x4*)out, n4);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

