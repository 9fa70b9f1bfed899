// Training-step kernels for gfx950 (fp32 parity path): train-mode BatchNorm forward/backward,
// LeakyReLU backward, weight gradient (MFMA 32x32x2 f32), bias gradient, 2x dilation (stride-2 dgrad),
// up-sample/concat backward, target assignment, losses + d(loss)/d(logits), MXNet Adam.
// The data gradient re-uses the forward implicit-GEMM kernels on a flipped/transposed weight image
// (yolo_pack_conv_weights_dgrad).  Reference: car/YOLO.py:350-498 (_train_batch, _find_best, _loss_mask,
// _score_weight, _get_loss) + the mxnet/gluon operators they call.
#include "common.h"
#include <float.h>

// ------------------------------------------------------------------------------------------------
// BatchNorm (train): per-channel batch statistics over (N,H,W) of an NHWC tensor
// ------------------------------------------------------------------------------------------------
// sums[0..C) = sum(y), sums[C..2C) = sum(y*y), accumulated in double (caller zero-fills).
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ y, double* __restrict__ sums,
                                                       int C, long long npix, int pix_per_block) {
    const long long p0 = (long long)blockIdx.x * pix_per_block;
    const long long p1 = min(p0 + pix_per_block, npix);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {           // coalesced along C for every pixel
        float s = 0.f, q = 0.f;
        for (long long p = p0; p < p1; ++p) {
            const float v = y[p * C + c];
            s += v;
            q += v * v;
        }
        atomicAdd(&sums[c], (double)s);
        atomicAdd(&sums[C + c], (double)q);
    }
}

// mean / invstd (biased variance, eps) + running-stat update (momentum m: r = m*r + (1-m)*batch;
// running_var takes the BIASED batch variance -- the MXNet CPU convention).
__global__ void bn_finalize_kernel(const double* __restrict__ sums, float* __restrict__ mean,
                                   float* __restrict__ invstd, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, int C, double inv_n, float eps, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = sums[c] * inv_n;
    double v = sums[C + c] * inv_n - m * m;
    if (v < 0) v = 0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(v + (double)eps));
    if (running_mean) {
        running_mean[c] = momentum * running_mean[c] + (1.f - momentum) * (float)m;
        running_var[c] = momentum * running_var[c] + (1.f - momentum) * (float)v;
    }
}

// z = lrelu(gamma*(y-mean)*invstd + beta) (+ residual)
__global__ void bn_act_fwd_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                  const float* __restrict__ invstd, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, const float* __restrict__ res,
                                  float* __restrict__ z, int C, long long total, float slope) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    float a = gamma[c] * ((y[i] - mean[c]) * invstd[c]) + beta[c];
    a = a > 0.f ? a : a * slope;
    if (res) a += res[i];
    z[i] = a;
}

// backward reductions: sums[0..C) = sum(da), sums[C..2C) = sum(da * xhat), da = dz * lrelu'(a)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, double* __restrict__ sums,
                                                            int C, long long npix, int pix_per_block, float slope) {
    const long long p0 = (long long)blockIdx.x * pix_per_block;
    const long long p1 = min(p0 + pix_per_block, npix);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float mu = mean[c], is = invstd[c], g = gamma[c], b = beta[c];
        float s = 0.f, q = 0.f;
        for (long long p = p0; p < p1; ++p) {
            const float xh = (y[p * C + c] - mu) * is;
            const float a = g * xh + b;
            const float da = dz[p * C + c] * (a > 0.f ? 1.f : slope);
            s += da;
            q += da * xh;
        }
        atomicAdd(&sums[c], (double)s);
        atomicAdd(&sums[C + c], (double)q);
    }
}

// dy = gamma*invstd * (da - mean(da) - xhat*mean(da*xhat));  dgamma = sum(da*xhat), dbeta = sum(da)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const double* __restrict__ sums, float* __restrict__ dy, int C, long long total,
                                    double inv_n, float slope) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const float xh = (y[i] - mean[c]) * invstd[c];
    const float a = gamma[c] * xh + beta[c];
    const float da = dz[i] * (a > 0.f ? 1.f : slope);
    const float m1 = (float)(sums[c] * inv_n), m2 = (float)(sums[C + c] * inv_n);
    dy[i] = gamma[c] * invstd[c] * (da - m1 - xh * m2);
}

__global__ void bn_param_grad_kernel(const double* __restrict__ sums, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    dbeta[c] = (float)sums[c];
    dgamma[c] = (float)sums[C + c];
}

extern "C" int yolo_bn_train_fwd(const float* y, const float* gamma, const float* beta, const float* residual,
                                 float* z, float* mean, float* invstd, float* running_mean, float* running_var,
                                 double* workspace, long long npix, int C, float eps, float momentum, float slope,
                                 void* stream) {
    if (!y || !gamma || !beta || !z || !mean || !invstd || !workspace || npix <= 0 || C <= 0) return YOLO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
    (void)hipMemsetAsync(workspace, 0, sizeof(double) * 2 * C, st);
    const int ppb = (int)((npix + 2047) / 2048 > 64 ? (npix + 2047) / 2048 : 64);
    const unsigned nb = (unsigned)((npix + ppb - 1) / ppb);
    YOLO_LAUNCH(bn_stats_kernel, dim3(nb), dim3(256), 0, st, y, workspace, C, npix, ppb);
    YOLO_LAUNCH(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, workspace, mean, invstd, running_mean,
                running_var, C, 1.0 / (double)npix, eps, momentum);
    const long long total = npix * C;
    YOLO_LAUNCH(bn_act_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, y, mean, invstd, gamma,
                beta, residual, z, C, total, slope);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

extern "C" int yolo_bn_train_bwd(const float* dz, const float* y, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, float* dy, float* dgamma, float* dbeta,
                                 double* workspace, long long npix, int C, float slope, void* stream) {
    if (!dz || !y || !mean || !invstd || !gamma || !beta || !dy || !dgamma || !dbeta || !workspace) return YOLO_EINVAL;
    if (npix <= 0 || C <= 0) return YOLO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
    (void)hipMemsetAsync(workspace, 0, sizeof(double) * 2 * C, st);
    const int ppb = (int)((npix + 2047) / 2048 > 64 ? (npix + 2047) / 2048 : 64);
    const unsigned nb = (unsigned)((npix + ppb - 1) / ppb);
    YOLO_LAUNCH(bn_bwd_reduce_kernel, dim3(nb), dim3(256), 0, st, dz, y, mean, invstd, gamma, beta, workspace, C, npix,
                ppb, slope);
    const long long total = npix * C;
    YOLO_LAUNCH(bn_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dz, y, mean, invstd,
                gamma, beta, workspace, dy, C, total, 1.0 / (double)npix, slope);
    YOLO_LAUNCH(bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, st, workspace, dgamma, dbeta, C);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient: dW[co][ci][kh][kw] += sum_p dy[p][co] * x[p @ tap][ci]   (fp32, MFMA 32x32x2)
// ------------------------------------------------------------------------------------------------
// One wave = one (32 cout x 32 cin) tile of one tap over a slice of the stacked output rows; the MFMA
// contracts 2 output pixels per step (lane half h = pixel parity).  D[i = cout][j = cin].
// Partial sums are added atomically (caller zero-fills dW).
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        float* __restrict__ dw, int N, int H, int W, int Cin, int Ho,
                                                        int Wo, int Cout, int ks, int stride, long long dy_ps,
                                                        int tiles_ci, int rows_per_slice) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x;
    const int tci = tile % tiles_ci, tco = tile / tiles_ci;
    const int tap = blockIdx.y;
    const int kh = tap / ks, kw = tap - kh * ks;
    const int pad = ks / 2;
    const int co = tco * 32 + l31, ci = tci * 32 + l31;
    const bool co_ok = co < Cout, ci_ok = ci < Cin;
    const long long slice = (long long)blockIdx.z * 4 + wave;
    const long long r0 = slice * rows_per_slice;
    const long long r1 = min(r0 + rows_per_slice, (long long)N * Ho);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (long long r = r0; r < r1; ++r) {
        const int n = (int)(r / Ho), oy = (int)(r - (long long)n * Ho);
        const int iy = oy * stride + kh - pad;
        if (iy < 0 || iy >= H) continue;                                   // wave-uniform
        const float* dyr = dy + r * Wo * dy_ps;
        const float* xr = x + ((long long)n * H + iy) * W * Cin;
        for (int ox0 = 0; ox0 < Wo; ox0 += 2) {
            const int ox = ox0 + h;
            const int ix = ox * stride + kw - pad;
            const bool ok = ox < Wo && ix >= 0 && ix < W;
            const float a = (ok && co_ok) ? dyr[(long long)ox * dy_ps + co] : 0.f;
            const float b = (ok && ci_ok) ? xr[(long long)ix * Cin + ci] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    if (!ci_ok) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int oc = tco * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (oc < Cout) atomicAdd(&dw[(((long long)oc * Cin + ci) * ks + kh) * ks + kw], acc[r]);
    }
}

extern "C" int yolo_conv_wgrad_f32(const float* dy, const float* x, float* dw_oihw, int N, int H, int W, int Cin,
                                   int Cout, int ksize, int stride, long long dy_pixel_stride, void* stream) {
    if (!dy || !x || !dw_oihw || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return YOLO_EINVAL;
    if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return YOLO_EUNSUPPORTED;
    const int pad = ksize / 2;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const int tiles_ci = (Cin + 31) / 32, tiles_co = (Cout + 31) / 32;
    const long long rows = (long long)N * Ho;
    // enough K-slices to fill the chip: ~2048 waves in flight
    const long long tiles = (long long)tiles_ci * tiles_co * ksize * ksize;
    long long slices = (4096 + tiles - 1) / tiles;
    if (slices < 1) slices = 1;
    if (slices > rows) slices = rows;
    slices = (slices + 3) / 4 * 4;
    const int rps = (int)((rows + slices - 1) / slices);
    const long long ps = dy_pixel_stride ? dy_pixel_stride : Cout;
    YOLO_LAUNCH(wgrad_f32_kernel, dim3((unsigned)(tiles_ci * tiles_co), ksize * ksize, (unsigned)(slices / 4)),
                dim3(256), 0, (hipStream_t)stream, dy, x, dw_oihw, N, H, W, Cin, Ho, Wo, Cout, ksize, stride, ps,
                tiles_ci, rps);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// column sums: db[c] = sum_p dy[p*ps + c]   (bias gradient of YOLOOutput's conv)
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db, int C,
                                                        long long npix, long long ps, int pix_per_block) {
    const long long p0 = (long long)blockIdx.x * pix_per_block;
    const long long p1 = min(p0 + pix_per_block, npix);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (long long p = p0; p < p1; ++p) s += dy[p * ps + c];
        atomicAdd(&db[c], s);
    }
}

extern "C" int yolo_bias_grad(const float* dy, float* db, long long npix, int C, long long pixel_stride,
                              void* stream) {
    if (!dy || !db || npix <= 0 || C <= 0) return YOLO_EINVAL;
    const long long ps = pixel_stride ? pixel_stride : C;
    const int ppb = 64;
    YOLO_LAUNCH(bias_grad_kernel, dim3((unsigned)((npix + ppb - 1) / ppb)), dim3(256), 0, (hipStream_t)stream, dy, db,
                C, npix, ps, ppb);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ------------------------------------------------------------------------------------------------
// strided copy (N rows of C floats, source row stride ps) -> dense (N, Cpad) with zero padding
// ------------------------------------------------------------------------------------------------
__global__ void gather_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int Cpad,
                                   long long src_batch_stride, long long rows_per_batch, long long ps,
                                   long long total) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % Cpad);
    const long long row = i / Cpad;
    const long long b = row / rows_per_batch, r = row - b * rows_per_batch;
    dst[i] = c < C ? src[b * src_batch_stride + r * ps + c] : 0.f;
}

extern "C" int yolo_gather_rows(const float* src, float* dst, int B, long long rows_per_batch, int C, int Cpad,
                                long long src_batch_stride, long long src_row_stride, void* stream) {
    if (!src || !dst || B <= 0 || rows_per_batch <= 0 || C <= 0 || Cpad < C) return YOLO_EINVAL;
    const long long total = (long long)B * rows_per_batch * Cpad;
    YOLO_LAUNCH(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst,
                C, Cpad, src_batch_stride, rows_per_batch, src_row_stride, total);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ------------------------------------------------------------------------------------------------
// 2x zero-dilation (stride-2 dgrad): D[n, 2y, 2x, :] = dy[n, y, x, :], zeros elsewhere; D is (N,H,W,C)
// ------------------------------------------------------------------------------------------------
__global__ void dilate2_kernel(const float* __restrict__ dy, float* __restrict__ d, int H, int W, int Ho, int Wo,
                               int C, long long total) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    long long p = i / C;
    const int xx = (int)(p % W); p /= W;
    const int yy = (int)(p % H);
    const long long n = p / H;
    float v = 0.f;
    if (!(yy & 1) && !(xx & 1) && (yy >> 1) < Ho && (xx >> 1) < Wo)
        v = dy[((n * Ho + (yy >> 1)) * Wo + (xx >> 1)) * C + c];
    d[i] = v;
}

extern "C" int yolo_dilate2x(const float* dy, float* d, int N, int H, int W, int Ho, int Wo, int C, void* stream) {
    if (!dy || !d || N <= 0 || H <= 0 || W <= 0 || C <= 0) return YOLO_EINVAL;
    const long long total = (long long)N * H * W * C;
    YOLO_LAUNCH(dilate2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, d, H, W,
                Ho, Wo, C, total);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ------------------------------------------------------------------------------------------------
// backward of 2x nearest up-sample + concat: d_up[n,y,x,:] (+)= sum of the 2x2 block of dcat[..., :C1];
// d_route (+)= dcat[..., C1:]
// ------------------------------------------------------------------------------------------------
__global__ void upcat_bwd_kernel(const float* __restrict__ dcat, float* __restrict__ dup, float* __restrict__ droute,
                                 int H, int W, int C1, int C2, int acc_up, int acc_route, long long total_up,
                                 long long total_route) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const int C = C1 + C2;
    if (i < total_up) {
        const int c = (int)(i % C1);
        long long p = i / C1;
        const int xx = (int)(p % (W / 2)); p /= (W / 2);
        const int yy = (int)(p % (H / 2));
        const long long n = p / (H / 2);
        float s = 0.f;
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) s += dcat[((n * H + 2 * yy + dy) * W + 2 * xx + dx) * C + c];
        dup[i] = acc_up ? dup[i] + s : s;
    } else if (i < total_up + total_route) {
        const long long j = i - total_up;
        const int c = (int)(j % C2);
        const long long p = j / C2;
        const float v = dcat[p * C + C1 + c];
        droute[j] = acc_route ? droute[j] + v : v;
    }
}

extern "C" int yolo_upsample2x_concat_bwd(const float* dcat, float* dup, float* droute, int N, int H, int W, int C1,
                                          int C2, int accumulate_up, int accumulate_route, void* stream) {
    if (!dcat || !dup || !droute || N <= 0 || (H & 1) || (W & 1) || C1 <= 0 || C2 <= 0) return YOLO_EINVAL;
    const long long tu = (long long)N * (H / 2) * (W / 2) * C1, tr = (long long)N * H * W * C2;
    YOLO_LAUNCH(upcat_bwd_kernel, dim3((unsigned)((tu + tr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dcat, dup,
                droute, H, W, C1, C2, accumulate_up, accumulate_route, tu, tr);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// y = a + b (elementwise, gradient fan-in)
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                           long long n) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] + b[i];
}
extern "C" int yolo_add(const float* a, const float* b, float* y, long long n, void* stream) {
    if (!a || !b || !y || n <= 0) return YOLO_EINVAL;
    YOLO_LAUNCH(add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, y, n);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ------------------------------------------------------------------------------------------------
// MXNet Adam: g = rescale*grad; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// w -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)       (epsilon OUTSIDE the bias correction)
// ------------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr_t, float b1, float b2, float eps,
                            float rescale) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gr = g[i] * rescale;
    const float mi = b1 * m[i] + (1.f - b1) * gr;
    const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi;
    v[i] = vi;
    w[i] = w[i] - lr_t * mi / (sqrtf(vi) + eps);
}

extern "C" int yolo_adam_step(float* w, const float* grad, float* m, float* v, long long n, int t, float lr,
                              float beta1, float beta2, float eps, float rescale, void* stream) {
    if (!w || !grad || !m || !v || n <= 0 || t < 1) return YOLO_EINVAL;
    const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
    YOLO_LAUNCH(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, grad, m, v, n,
                lr_t, beta1, beta2, eps, rescale);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}
