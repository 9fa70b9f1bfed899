// Detection post-processing for gfx950: anchor decode, per-image arg-max (top-1), IoU, greedy NMS.
// Compiled with -ffp-contract=off so fp32 arithmetic follows the reference's op-by-op NDArray
// evaluation (no fused multiply-add), which is what makes index parity exact.
#include "common.h"
#include <atomic>
#include <float.h>
#include <string.h>

struct GridDev {
    int nscale, A, img_h, img_w;
    int gw[4], step[4], cum[5];   // cum[i] = first box index of scale i
    float ah[4][8], aw[4][8];
};

static int make_grid(const yolo_grid_desc* g, GridDev& d, int* nbox) {
    if (!g || g->nscale < 1 || g->nscale > 4 || g->A < 1 || g->A > 8) return YOLO_EINVAL;
    d.nscale = g->nscale; d.A = g->A; d.img_h = g->img_h; d.img_w = g->img_w;
    int cum = 0;
    for (int i = 0; i < 4; ++i) {
        d.cum[i] = cum;
        if (i < g->nscale) {
            if (g->gh[i] <= 0 || g->gw[i] <= 0 || g->step[i] <= 0) return YOLO_EINVAL;
            d.gw[i] = g->gw[i]; d.step[i] = g->step[i];
            cum += g->gh[i] * g->gw[i] * g->A;
            for (int a = 0; a < g->A; ++a) {
                d.ah[i][a] = g->anchors_hw[(i * g->A + a) * 2 + 0];
                d.aw[i][a] = g->anchors_hw[(i * g->A + a) * 2 + 1];
            }
        } else {
            d.gw[i] = 1; d.step[i] = 1;
        }
    }
    d.cum[4] = cum;
    for (int i = g->nscale; i < 4; ++i) d.cum[i] = cum;
    *nbox = cum;
    return YOLO_OK;
}

__device__ __forceinline__ float sigmoidf_ref(float x) { return 1.f / (1.f + expf(-x)); }

// Per-box constants of car/YOLO.py:123-155: s (stride px), y, x (cell origin px), h, w (anchor).
__device__ __forceinline__ void box_consts(const GridDev& g, int k, float& s, float& y, float& x, float& h,
                                           float& w) {
    int i = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q < g.nscale && k >= g.cum[q]) i = q;
    const int rel = k - g.cum[i];
    const int cell = rel / g.A, a = rel - cell * g.A;
    const int row = cell / g.gw[i], col = cell - row * g.gw[i];
    s = (float)g.step[i];
    y = (float)(row * g.step[i]);
    x = (float)(col * g.step[i]);
    h = g.ah[i][a];
    w = g.aw[i][a];
}

// _yxhw_to_ltrb, car/YOLO.py:552-566 (one coordinate pair at a time).
__device__ __forceinline__ void decode_axis(float tc, float tsz, float s, float origin, float img, float anchor,
                                            float& lo, float& hi) {
    const float c = (sigmoidf_ref(tc) * s + origin) / img;
    const float sz = expf(tsz) * anchor;
    const float half = sz / 2.f;
    lo = c - half;
    hi = c + half;
}

// ---- decode: (B,N,A,C) logits -> rows [sigmoid(obj), l,t,r,b, rot, cls...] -------------------
__global__ void decode_kernel(const float* __restrict__ out, float* __restrict__ rows, int C, int nbox,
                              long long total, GridDev g) {
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long long box = idx / C;
    const int c = (int)(idx - box * C);
    float v = out[idx];
    if (c == 0) {
        v = sigmoidf_ref(v);
    } else if (c <= 4) {
        const int k = (int)(box % nbox);
        float s, y, x, h, w;
        box_consts(g, k, s, y, x, h, w);
        const float* p = out + box * C;
        float lo, hi;
        if (c == 1 || c == 3) decode_axis(p[2], p[4], s, x, (float)g.img_w, w, lo, hi);   // l / r from tx, tw
        else                  decode_axis(p[1], p[3], s, y, (float)g.img_h, h, lo, hi);   // t / b from ty, th
        v = (c <= 2) ? lo : hi;
    }
    rows[idx] = v;
}

extern "C" int yolo_decode(const float* out, float* rows, int B, int C, const yolo_grid_desc* g, void* stream) {
    if (!out || !rows || B <= 0 || C < 6) return YOLO_EINVAL;
    GridDev d; int nbox;
    int rc = make_grid(g, d, &nbox);
    if (rc) return rc;
    const long long total = (long long)B * nbox * C;
    YOLO_LAUNCH(decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out,
                       rows, C, nbox, total, d);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ---- top-1: arg-max of sigmoid(obj), first index among ties (car/YOLO.py:581-597) ------------
__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

__global__ __launch_bounds__(1024) void predict_top1_kernel(const float* __restrict__ out, float* __restrict__ pred,
                                                            int* __restrict__ best_idx, int C, int nbox, GridDev g) {
    const int b = blockIdx.x;
    const float* o = out + (long long)b * nbox * C;
    float bv = -FLT_MAX; int bi = 0x7fffffff;
    for (int k = threadIdx.x; k < nbox; k += blockDim.x) {
        const float s = sigmoidf_ref(o[(long long)k * C]);
        if (s > bv) { bv = s; bi = k; }          // ascending k per thread: first max wins
    }
    // wavefront (64-lane) reduction, then one value per wave through LDS
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        argmax_combine(bv, bi, ov, oi);
    }
    __shared__ float sv[16];
    __shared__ int si[16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (wave == 0) {
        const int nw = blockDim.x >> 6;
        bv = lane < nw ? sv[lane] : -FLT_MAX;
        bi = lane < nw ? si[lane] : 0x7fffffff;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            argmax_combine(bv, bi, ov, oi);
        }
        if (lane == 0) { si[0] = bi == 0x7fffffff ? 0 : bi; sv[0] = bv; }   // all-NaN scores: index 0, as mxnet's argmax
    }
    __syncthreads();
    const int k = si[0];
    const float* p = o + (long long)k * C;
    float* q = pred + (long long)b * C;
    if (threadIdx.x == 0) {
        best_idx[b] = k;
        float s, y, x, h, w;
        box_consts(g, k, s, y, x, h, w);
        float l, r, t, bt;
        decode_axis(p[2], p[4], s, x, (float)g.img_w, w, l, r);
        decode_axis(p[1], p[3], s, y, (float)g.img_h, h, t, bt);
        q[0] = sv[0];
        q[1] = (t + bt) / 2.f;     // y
        q[2] = (l + r) / 2.f;      // x
        q[3] = bt - t;             // h
        q[4] = r - l;              // w
    }
    for (int c = 5 + threadIdx.x; c < C; c += blockDim.x) q[c] = p[c];
}

extern "C" int yolo_predict_top1(const float* out, float* pred, int* best_idx, int B, int C,
                                 const yolo_grid_desc* g, void* stream) {
    if (!out || !pred || !best_idx || B <= 0 || C < 6) return YOLO_EINVAL;
    GridDev d; int nbox;
    int rc = make_grid(g, d, &nbox);
    if (rc) return rc;
    YOLO_LAUNCH(predict_top1_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, out, pred, best_idx, C,
                       nbox, d);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ---- get_iou(predict, target, mode), yolo_gluon.py:127-168 ------------------------------------
// MODE 2: target = [c, y, x, h, w] (the hot path: car/YOLO.py:403,525).  MODE 1 (the reference's default): target =
// [c, l, t, r, b] -- including its target_area = target[3] * target[4] (yolo_gluon.py:166), i.e. r2 * b2 in this mode.
template <int MODE>
__global__ void iou_kernel(const float* __restrict__ boxes, const float* __restrict__ target,
                           float* __restrict__ iou, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float t1 = target[1], t2_ = target[2], t3 = target[3], t4 = target[4];
    float l2, t2, r2, b2;
    if (MODE == 1) { l2 = t1; t2 = t2_; r2 = t3; b2 = t4; }
    else { l2 = t2_ - t4 / 2.f; t2 = t1 - t3 / 2.f; r2 = t2_ + t4 / 2.f; b2 = t1 + t3 / 2.f; }
    const float4 p = ((const float4*)boxes)[i];     // l,t,r,b
    const float iw = fmaxf(fminf(r2, p.z) - fmaxf(l2, p.x), 0.f);
    const float ih = fmaxf(fminf(b2, p.w) - fmaxf(t2, p.y), 0.f);
    const float inter = iw * ih;
    const float pa = (p.z - p.x) * (p.w - p.y);
    const float ta = t3 * t4;
    iou[i] = inter / (pa + ta - inter);
}

extern "C" int yolo_iou_ltrb_vs_yxhw(const float* boxes, const float* target, float* iou, int n, void* stream) {
    if (!boxes || !target || !iou || n <= 0) return YOLO_EINVAL;
    YOLO_LAUNCH(iou_kernel<2>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, boxes, target, iou, n);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

extern "C" int yolo_iou_ltrb_vs_cltrb(const float* boxes, const float* target, float* iou, int n, void* stream) {
    if (!boxes || !target || !iou || n <= 0) return YOLO_EINVAL;
    YOLO_LAUNCH(iou_kernel<1>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, boxes, target, iou, n);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ---- NMS scores ------------------------------------------------------------------------------
// mode 0: scores (B, nbox) = sigmoid(obj) column of rows.  mode 1: scores (B, nbox*ncls) =
// sigmoid(obj) * softmax(cls)_c.  One thread per box.
template <int NB>                                        // boxes (= threads) per block
__global__ __launch_bounds__(NB) void nms_scores_kernel(const float* __restrict__ rows, float* __restrict__ scores, int C,
                                                        int ncls, int mode, long long nboxes) {
    // NB boxes per block, staged through LDS so that both the row reads and the score writes are coalesced
    // (one thread per box walking its own 120-byte row ran at 1/18 of the HBM rate)
    extern __shared__ float sm[];                        // NB*C floats of rows, then NB*ncls of scores
    const long long k0 = (long long)blockIdx.x * NB;
    const int nb = (int)min((long long)NB, nboxes - k0);
    if (mode == 0) {
        if ((int)threadIdx.x < nb) scores[k0 + threadIdx.x] = rows[(k0 + threadIdx.x) * C];
        return;
    }
    for (int i = threadIdx.x; i < nb * C; i += NB) sm[i] = rows[k0 * C + i];
    __syncthreads();
    float* so = sm + NB * C;
    if ((int)threadIdx.x < nb) {
        const float* p = sm + threadIdx.x * C;
        float m = -FLT_MAX;
        for (int c = 0; c < ncls; ++c) m = fmaxf(m, p[6 + c]);
        float sum = 0.f;
        for (int c = 0; c < ncls; ++c) sum += expf(p[6 + c] - m);
        const float obj = p[0];
        for (int c = 0; c < ncls; ++c) so[threadIdx.x * ncls + c] = obj * (expf(p[6 + c] - m) / sum);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb * ncls; i += NB) scores[k0 * ncls + i] = so[i];
}

extern "C" int yolo_nms_scores(const float* rows, float* scores, int B, int nbox, int C, int mode, void* stream) {
    if (!rows || !scores || B <= 0 || nbox <= 0 || C < 6) return YOLO_EINVAL;
    if (mode == 1 && C <= 6) return YOLO_EINVAL;
    const long long nboxes = (long long)B * nbox;
    if (C > 96) return YOLO_EUNSUPPORTED;                      // LDS staging: boxes x (C + ncls) floats
    // 256 boxes per block while their staged rows + scores fit the 64 KiB a block gets without opting in (C <= 35), else 64
    if ((size_t)256 * (C + (C - 6)) * sizeof(float) <= 65536)
        YOLO_LAUNCH(nms_scores_kernel<256>, dim3((unsigned)((nboxes + 255) / 256)), dim3(256),
                    (size_t)256 * (C + (C - 6)) * sizeof(float), (hipStream_t)stream, rows, scores, C, C - 6, mode, nboxes);
    else
        YOLO_LAUNCH(nms_scores_kernel<64>, dim3((unsigned)((nboxes + 63) / 64)), dim3(64),
                    (size_t)64 * (C + (C - 6)) * sizeof(float), (hipStream_t)stream, rows, scores, C, C - 6, mode, nboxes);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ---- decode + NMS scores in one pass over the logits ------------------------------------------------
// yolo_decode followed by yolo_nms_scores reads the logits, writes the rows, reads the rows again and writes the
// scores (665 MB at 608x608 bs 64, 380 us); here a block stages 128 boxes through LDS once (coalesced both ways), one
// thread per box decodes its row IN the staged copy (box constants computed once per box, not once per coordinate)
// and derives its class scores from it.  Same functions, same operation order: bit-identical to the two calls.
// (round 3) The copies are what the kernel is made of -- 490 MB at 608^2 batch 64 -- and a block that loads, computes and
// stores in turn keeps its loads in flight a third of the time (2.0 TB/s, 243-263 us, with LDS leaving room for ten waves per
// CU).  So the blocks are persistent and PIPELINED: a tile of 128 boxes is fetched into registers by 16-byte loads, all issued
// at once, while the tile before it is decoded in LDS and written out.  The score rows have an ODD pitch: a lane owns a box, and
// 24-float rows put every fourth lane on the same bank; the exponentials are kept there between the sum and the division.
constexpr int kDecBoxes = 128;

// HIST (round 4, yolo_decode_nms): the first radix histogram of the NMS selection (nms_hist_kernel pass 0: valid scores by their
// top 11 bits, per image) is taken HERE, from the scores the block holds in LDS, instead of in a pass of its own over the
// score array.  A thread counts runs of equal bins over its box's scores, adds them to a kDecHistBins-bin (512) block histogram (2 KiB: one
// more KiB and a CU holds four of these blocks instead of five) and the block flushes the non-empty bins -- two or three -- to the
// image's global histogram once per tile; a tile that straddles two images is flushed once per image.  512 bins: the scores
// made here are sigmoid x softmax <= 1.0 = bin 508 (a larger value -- impossible -- would go straight to the global histogram).
constexpr int kDecHistBins = 512;

template <int NV, int HIST>                                // NV: 16-byte loads per thread and tile: kDecBoxes * C / 4 / kDecBoxes
__global__ __launch_bounds__(kDecBoxes) void decode_scores_kernel(const float* __restrict__ out, float* __restrict__ rows,
                                                                  float* __restrict__ scores, int C, int ncls, int mode,
                                                                  int nbox, long long nboxes, GridDev g, unsigned vbits,
                                                                  unsigned* __restrict__ ghist) {
    extern __shared__ float4 sm4[];                      // kDecBoxes rows of C floats, then the score rows of SP floats
    float* sm = reinterpret_cast<float*>(sm4);
    float* so = sm + ((kDecBoxes * C + 3) & ~3);
    const int per = mode == 1 ? ncls : 1, SP = per | 1;
    unsigned* lh = reinterpret_cast<unsigned*>(so + kDecBoxes * SP);         // HIST: the block's histogram
    if (HIST) {
        for (int i = threadIdx.x; i < kDecHistBins; i += kDecBoxes) lh[i] = 0;
        __syncthreads();
    }
    // HIST: this thread's runs of equal bins over its box's scores -> the block histogram
    long long k0h = 0;                                     // (first box of the tile in flight, for hist_add's slow path)
    auto hist_add = [&](const float* e) {
        unsigned cur = 0xffffffffu, run = 0;
        for (int c = 0; c < per; ++c) {
            const unsigned u = __float_as_uint(e[c]);
            if (u >= vbits && u <= 0x7f800000u) {
                const unsigned bin = u >> 21;
                if (bin >= kDecHistBins) { atomicAdd(&ghist[(long long)((k0h + threadIdx.x) / nbox) * 2 * 2048 + bin], 1u); continue; }
                if (bin != cur) {
                    if (run) atomicAdd(&lh[cur], run);
                    cur = bin;
                    run = 0;
                }
                ++run;
            }
        }
        if (run) atomicAdd(&lh[cur], run);
    };
    auto hist_flush = [&](int img) {
        unsigned* gh = ghist + (long long)img * 2 * 2048;                  // (pass 0 section of the image's two histograms)
#pragma unroll
        for (int q = 0; q < kDecHistBins / kDecBoxes; ++q) {
            const int bin = threadIdx.x + q * kDecBoxes;
            const unsigned v = lh[bin];
            if (v) { atomicAdd(&gh[bin], v); lh[bin] = 0; }
        }
    };
    const long long ntiles = (nboxes + kDecBoxes - 1) / kDecBoxes;
    float4 v[NV];
    float vt = 0.f;                                        // (tile floats % 4: only a last, ragged tile has them)
    auto fetch = [&](long long tile) {
        const long long k0 = tile * kDecBoxes;
        const int n = (int)min((long long)kDecBoxes, nboxes - k0) * C, n4 = n >> 2;
        const float* src = out + k0 * C;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = (int)threadIdx.x + kDecBoxes * j;
            if (i < n4) v[j] = reinterpret_cast<const float4*>(src)[i];
        }
        if ((n4 << 2) + (int)threadIdx.x < n) vt = src[(n4 << 2) + threadIdx.x];
    };
    long long tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const long long k0 = tile * kDecBoxes;
        k0h = k0;
        const int nb = (int)min((long long)kDecBoxes, nboxes - k0), n = nb * C, n4 = n >> 2;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = (int)threadIdx.x + kDecBoxes * j;
            if (i < n4) sm4[i] = v[j];
        }
        if ((n4 << 2) + (int)threadIdx.x < n) sm[(n4 << 2) + threadIdx.x] = vt;
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);      // in flight until the top of the next trip
        __syncthreads();
        if ((int)threadIdx.x < nb) {
            float* p = sm + threadIdx.x * C;
            const int k = (int)((k0 + threadIdx.x) % nbox);
            float s, y, x, h, w, l, r, t, b;
            box_consts(g, k, s, y, x, h, w);
            decode_axis(p[2], p[4], s, x, (float)g.img_w, w, l, r);
            decode_axis(p[1], p[3], s, y, (float)g.img_h, h, t, b);
            const float obj = sigmoidf_ref(p[0]);
            p[0] = obj; p[1] = l; p[2] = t; p[3] = r; p[4] = b;
            if (mode == 1) {
                // eight classes at a time: their LDS reads, exponentials and divisions are independent instructions of one
                // wave (two or three waves share a SIMD here, so a chain of dependent LDS round trips is not hidden); the
                // maximum, the sum and every quotient are still taken in class order, value by value
                const float* cl = p + 6;
                float* e = so + threadIdx.x * SP;           // the exponentials are kept, not taken twice (same values)
                float m = -FLT_MAX, sum = 0.f, q[8];
                for (int c0 = 0; c0 < ncls; c0 += 8) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) q[i] = c0 + i < ncls ? cl[c0 + i] : -FLT_MAX;
#pragma unroll
                    for (int i = 0; i < 8; ++i) m = fmaxf(m, q[i]);
                }
                for (int c0 = 0; c0 < ncls; c0 += 8) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) q[i] = c0 + i < ncls ? cl[c0 + i] : 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) q[i] = expf(q[i] - m);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (c0 + i < ncls) { e[c0 + i] = q[i]; sum += q[i]; }
                }
                for (int c0 = 0; c0 < ncls; c0 += 8) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) q[i] = c0 + i < ncls ? e[c0 + i] : 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) q[i] = obj * (q[i] / sum);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (c0 + i < ncls) e[c0 + i] = q[i];
                }
            } else {
                so[threadIdx.x * SP] = obj;
            }
            if (HIST && (int)((k0 + threadIdx.x) / nbox) == (int)(k0 / nbox)) hist_add(so + threadIdx.x * SP);
        }
        __syncthreads();
        if (HIST) {
            const int b_first = (int)(k0 / nbox), b_last = (int)((k0 + nb - 1) / nbox);
            hist_flush(b_first);
            for (int img = b_first + 1; img <= b_last; ++img) {      // (a tile that straddles images: rare, block-uniform)
                __syncthreads();
                if ((int)threadIdx.x < nb && (int)((k0 + threadIdx.x) / nbox) == img) hist_add(so + threadIdx.x * SP);
                __syncthreads();
                hist_flush(img);
            }
        }
        {
            float4* dst = reinterpret_cast<float4*>(rows + k0 * C);
            for (int i = threadIdx.x; i < n4; i += kDecBoxes) dst[i] = sm4[i];
            if ((n4 << 2) + (int)threadIdx.x < n) rows[k0 * C + (n4 << 2) + threadIdx.x] = sm[(n4 << 2) + threadIdx.x];
            // flat element i = threadIdx.x + kDecBoxes j of the nb x per scores <-> (row, col), advanced without divisions
            const int er = kDecBoxes / per, ec = kDecBoxes % per;
            int row = (int)threadIdx.x / per, col = (int)threadIdx.x % per;
            for (int i = threadIdx.x; i < nb * per; i += kDecBoxes) {
                scores[k0 * per + i] = so[row * SP + col];
                row += er; col += ec;
                if (col >= per) { col -= per; ++row; }
            }
        }
        __syncthreads();                                   // the next trip overwrites the staged tile
    }
}

constexpr int kMaxDev = 64;

// ghist != nullptr: also take the NMS selection's first histogram (decode_scores_kernel<.., HIST = 1>)
static int launch_decode_scores(const float* out, float* rows, float* scores, int B, int C, const yolo_grid_desc* g, int mode,
                                hipStream_t st, unsigned vbits, unsigned* ghist) {
    if (!out || !rows || !scores || B <= 0 || C < 6 || (mode != 0 && mode != 1)) return YOLO_EINVAL;
    if (mode == 1 && C <= 6) return YOLO_EINVAL;
    if (C > 96) return YOLO_EUNSUPPORTED;                      // LDS staging: kDecBoxes boxes x (C + ncls) floats
    GridDev d; int nbox;
    int rc = make_grid(g, d, &nbox);
    if (rc) return rc;
    const long long nboxes = (long long)B * nbox;
    if ((C * kDecBoxes) % 4) return YOLO_EUNSUPPORTED;         // 16-byte copies: a tile's rows start on a 16-byte boundary
    const size_t lds = (size_t)(((kDecBoxes * C + 3) & ~3) + kDecBoxes * ((mode == 1 ? C - 6 : 1) | 1)) * sizeof(float) +
                       (ghist ? kDecHistBins * sizeof(unsigned) : 0);
    // per-DEVICE launch state (a process may switch the current device; hipFuncSetAttribute applies to the current one only):
    // CU count and "opted in to > 64 KiB of dynamic LDS", indexed by device id; relaxed atomics -- two host threads racing
    // here both write the same values
    static std::atomic<int> cu_of[kMaxDev];
    static std::atomic<int> opted_of[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return YOLO_EUNSUPPORTED;
    int cus = cu_of[dev].load(std::memory_order_relaxed);
    if (!cus) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cu_of[dev].store(cus = n, std::memory_order_relaxed);
    }
    long long per_cu = (long long)(160 * 1024 / lds);          // persistent blocks: as many as a CU's LDS holds
    if (per_cu > 8) per_cu = 8;
    const long long ntiles = (nboxes + kDecBoxes - 1) / kDecBoxes;
    const dim3 grid((unsigned)(ntiles < cus * per_cu ? ntiles : cus * per_cu));
    if (lds > 65536 && !opted_of[dev].load(std::memory_order_relaxed)) {     // (C + classes > 128: more than 64 KiB, once per device)
        if (hipFuncSetAttribute((const void*)decode_scores_kernel<24, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)decode_scores_kernel<24, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return YOLO_EUNSUPPORTED;                          // (a negative status like every other refusal, not a raw hipError_t)
        }
        opted_of[dev].store(1, std::memory_order_relaxed);
    }
#define YOLO_DEC_LAUNCH(NV, H)                                                                                          \
    YOLO_LAUNCH((decode_scores_kernel<NV, H>), grid, dim3(kDecBoxes), lds, st, out, rows, scores, C, C - 6, mode, nbox, \
                nboxes, d, vbits, ghist)
    if (C <= 32) { if (ghist) YOLO_DEC_LAUNCH(8, 1); else YOLO_DEC_LAUNCH(8, 0); }
    else { if (ghist) YOLO_DEC_LAUNCH(24, 1); else YOLO_DEC_LAUNCH(24, 0); }
#undef YOLO_DEC_LAUNCH
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

extern "C" int yolo_decode_scores(const float* out, float* rows, float* scores, int B, int C, const yolo_grid_desc* g,
                                  int mode, void* stream) {
    return launch_decode_scores(out, rows, scores, B, C, g, mode, (hipStream_t)stream, 0u, nullptr);
}

// ---- NMS ---------------------------------------------------------------------------------------
// One 1024-thread block per image.
//   1. exact top-k selection by (score desc, id asc): three histogram passes over the score bits
//      (11+11+10) find the k-th largest score T; candidates > T are collected, candidates == T are
//      appended in ascending id order until k is reached (ordered wavefront-ballot compaction).
//   2. bitonic sort of <=512 composite keys (score_bits<<32 | ~id) descending.
//   3. suppression bit-matrix over candidate pairs (same class && IoU > thr), then one wavefront
//      walks the candidates in order OR-ing rows of the matrix (lane w owns 64-bit word w).
// A candidate is VALID iff valid_thresh <= score <= +inf as floats: as bit patterns vbits <= u <= 0x7f800000 (negative scores and
// the padding sentinel have the sign bit set; NaN scores -- a NaN logit -- lie above +inf and are never candidates, as in the
// oracle's `scores >= valid_thresh`).
constexpr int NMS_MAXK = 512;
constexpr int NMS_THREADS = 1024;
// Chip-wide pre-selection (optional workspace): the single block per image above spends its time streaming the
// score array three times (546k candidates per image at 608x608, 64 blocks on 256 CUs).  With a workspace the
// same radix selection runs as grid-wide passes over all images -- 11-bit histogram, 11 more bits inside the
// selected bucket, then every valid candidate at or above the 22-bit threshold is appended to a per-image list
// (<= NMS_CAP entries; top-k plus one fine bucket) -- and the per-image block only sorts that list.  Exactly the
// same (score desc, id asc) order: the list is sorted by the same composite key.  A list that overflows (massive
// ties, e.g. the all-equal scores of a zero input) falls back to the single-block selection, decided on device.
constexpr int NMS_CAP = 4096;
constexpr int NMS_WS_PER_IMAGE = (2 * 2048 + 16) * 4 + NMS_CAP * 8;     // hist x2, sel[8] + cnt + pad, list

// (round 3) The grid-wide selection passes read sixteen scores per thread and trip -- four 16-byte loads in flight -- instead
// of one dword (4096 waves with one load each in flight took a memory latency per 256 scores: 73-78 us per pass over 140 MB).
// Element e of load q of the thread's batch is candidate base + 1024 q + e; slots past the end read as -0 (never valid).
__device__ __forceinline__ void load_score_batch(const float* __restrict__ sc, long long base, long long i1, bool vec,
                                                 unsigned (&u)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long i = base + 1024 * q;
        if (vec && i + 3 < i1) {
            const float4 v = *reinterpret_cast<const float4*>(sc + i);
            u[4 * q] = __float_as_uint(v.x); u[4 * q + 1] = __float_as_uint(v.y);
            u[4 * q + 2] = __float_as_uint(v.z); u[4 * q + 3] = __float_as_uint(v.w);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) u[4 * q + e] = i + e < i1 ? __float_as_uint(sc[i + e]) : 0x80000000u;
        }
    }
}

__global__ __launch_bounds__(256) void nms_hist_kernel(const float* __restrict__ scores, long long ncand,
                                                       unsigned vbits, int pass, unsigned* __restrict__ hist,
                                                       const unsigned* __restrict__ sel, long long per_block) {
    __shared__ unsigned h[2048];
    const int b = blockIdx.y;
    const unsigned* sl = sel + b * 16;
    if (pass == 1 && sl[4]) return;                       // fewer valid candidates than top-k: everything is taken
    for (int i = threadIdx.x; i < 2048; i += 256) h[i] = 0;
    __syncthreads();
    const float* sc = scores + (long long)b * ncand;
    const long long i0 = blockIdx.x * per_block, i1 = min(i0 + per_block, ncand);
    const unsigned q1 = pass == 1 ? sl[0] : 0;
    // a thread counts runs of equal bins privately: scores of neighbouring candidates often share their top bits
    // (every candidate of a random-weight net lands in two or three bins), and LDS atomics on one address serialise
    unsigned cur = 0xffffffffu, run = 0;
    const bool vec = (reinterpret_cast<unsigned long long>(sc) & 15ull) == 0;
    for (long long base = i0 + 4 * threadIdx.x; base < i1; base += 4096) {
        unsigned ub[16];
        load_score_batch(sc, base, i1, vec, ub);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const unsigned u = ub[e];
            if (u >= vbits && u <= 0x7f800000u) {
                unsigned bin = 0xffffffffu;
                if (pass == 0) bin = u >> 21;
                else if ((u >> 21) == q1) bin = (u >> 10) & 2047u;
                if (bin != cur) {
                    if (run) atomicAdd(&h[cur], run);
                    cur = bin;
                    run = 0;
                }
                if (bin != 0xffffffffu) ++run;
            }
        }
    }
    if (run) atomicAdd(&h[cur], run);
    __syncthreads();
    unsigned* gh = hist + ((long long)b * 2 + pass) * 2048;
    for (int i = threadIdx.x; i < 2048; i += 256)
        if (h[i]) atomicAdd(&gh[i], h[i]);
}

// one wavefront per image: the bucket in which the cumulative count from the top reaches `need`
__global__ __launch_bounds__(64) void nms_pick_kernel(const unsigned* __restrict__ hist, unsigned* __restrict__ sel,
                                                      int pass, int topk) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const unsigned* gh = hist + ((long long)b * 2 + pass) * 2048;
    unsigned* sl = sel + b * 16;
    if (pass == 1 && sl[4]) return;
    const unsigned need = pass == 0 ? (unsigned)topk : sl[2];
    unsigned hv[32], chunk = 0;
#pragma unroll
    for (int q = 0; q < 32; ++q) { hv[q] = gh[lane * 32 + q]; chunk += hv[q]; }
    unsigned above = 0, total = 0;                         // candidates in chunks above this lane's / in all chunks
    for (int l = 0; l < 64; ++l) {
        const unsigned c = __shfl(chunk, l, 64);
        if (l > lane) above += c;
        total += c;
    }
    if (total < need) {                                    // (only possible in pass 0)
        if (lane == 0) { sl[4] = 1; sl[5] = 0; }
        return;
    }
    if (above < need && above + chunk >= need) {           // exactly one lane
        unsigned cum = above;
        int q = 31;
        for (; q > 0; --q) {
            if (cum + hv[q] >= need) break;
            cum += hv[q];
        }
        const unsigned bucket = (unsigned)(lane * 32 + q);
        if (pass == 0) { sl[0] = bucket; sl[1] = cum; sl[2] = need - cum; sl[4] = 0; }
        else { sl[3] = bucket; sl[5] = (sl[0] << 21) | (bucket << 10); }
    }
}

__global__ __launch_bounds__(256) void nms_collect_kernel(const float* __restrict__ scores, long long ncand,
                                                          unsigned vbits, unsigned* __restrict__ sel,
                                                          unsigned long long* __restrict__ list, long long per_block) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    unsigned* sl = sel + b * 16;
    const unsigned thr = sl[4] ? vbits : max(vbits, sl[5]);
    unsigned* cnt = sl + 8;
    const float* sc = scores + (long long)b * ncand;
    unsigned long long* out = list + (long long)b * NMS_CAP;
    const long long i0 = blockIdx.x * per_block, i1 = min(i0 + per_block, ncand);
    const bool vec = (reinterpret_cast<unsigned long long>(sc) & 15ull) == 0;
    // (every wave of the block runs the same number of trips: the ballots below are wave-wide)
    for (long long base0 = i0; base0 < i1; base0 += 4096) {
        const long long base = base0 + 4 * threadIdx.x;
        unsigned ub[16];
        load_score_batch(sc, base, i1, vec, ub);
        unsigned mine = 0;                                 // bit e: element e of the batch is taken
#pragma unroll
        for (int e = 0; e < 16; ++e) mine |= (ub[e] >= thr && ub[e] <= 0x7f800000u) ? 1u << e : 0u;
        if (!__ballot(mine != 0)) continue;                // the common case: nothing at or above the threshold in 4096 scores
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bool take = (mine >> e) & 1u;
            const unsigned long long bal = __ballot(take);
            if (bal) {
                unsigned basepos = 0;
                const int leader = __ffsll((long long)bal) - 1;
                if (lane == leader) basepos = atomicAdd(cnt, (unsigned)__popcll(bal));
                basepos = __shfl(basepos, leader, 64);
                const unsigned pos = basepos + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
                const long long i = base + 1024 * (e >> 2) + (e & 3);
                if (take && pos < NMS_CAP)
                    out[pos] = ((unsigned long long)ub[e] << 32) | (unsigned)(0xffffffffu - (unsigned)i);
            }
        }
    }
}


__device__ __forceinline__ float box_iou(const float4 a, const float4 b) {
    const float iw = fmaxf(0.f, fminf(a.z, b.z) - fmaxf(a.x, b.x));
    const float ih = fmaxf(0.f, fminf(a.w, b.w) - fmaxf(a.y, b.y));
    const float inter = iw * ih;
    const float ua = (a.z - a.x) * (a.w - a.y) + (b.z - b.x) * (b.w - b.y) - inter;
    return ua > 0.f ? inter / ua : 0.f;
}

__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const float* __restrict__ rows,
                                                          const float* __restrict__ scores, int nbox, int C, int cpb,
                                                          float valid_thresh, float iou_thresh, int topk, int post_nms,
                                                          int* __restrict__ kept, float* __restrict__ kept_scores,
                                                          int* __restrict__ kept_count,
                                                          const unsigned long long* __restrict__ list,
                                                          const unsigned* __restrict__ sel) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned long long keys[NMS_CAP];
    __shared__ float4 cbox[NMS_MAXK];
    __shared__ unsigned ccls[NMS_MAXK];                  // class of candidate i (its id modulo the candidates per box), taken once
    __shared__ unsigned long long supp[NMS_MAXK][NMS_MAXK / 64];
    __shared__ unsigned sh_sel, sh_above, sh_cnt, sh_eqbase;
    __shared__ unsigned wave_tot[NMS_THREADS / 64];

    const int b = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const long long ncand = (long long)nbox * cpb;
    const float* sc = scores + (long long)b * ncand;
    const float* rw = rows + (long long)b * nbox * C;
    const unsigned vbits = __float_as_uint(fmaxf(valid_thresh, 0.f));
    unsigned n;
    int sortn;
    const unsigned nlist = list ? sel[b * 16 + 8] : 0xffffffffu;
    if (nlist <= (unsigned)NMS_CAP) {
    // ---- 1'. pre-selected list (chip-wide passes above): sort it, the first top-k keys are the selection ------
    for (int i = tid; i < NMS_CAP; i += NMS_THREADS) keys[i] = (unsigned)i < nlist ? list[(long long)b * NMS_CAP + i] : 0ull;
    n = min(nlist, (unsigned)topk);
    sortn = 64;
    while ((unsigned)sortn < nlist) sortn <<= 1;
    __syncthreads();
    } else {
    // ---- 1. k-th largest valid score via 3 histogram passes -------------------------------
    // scores are >= 0, so their uint bit patterns order like the floats.
    unsigned prefix = 0;     // bits fixed so far
    unsigned above = 0;      // candidates strictly above the current bucket
    unsigned need = (unsigned)topk;
    bool all_taken = false;  // fewer than topk valid candidates: take everything valid
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = shifts[pass], nb = 1 << widths[pass];
        for (int i = tid; i < 2048; i += NMS_THREADS) hist[i] = 0;
        __syncthreads();
        const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (shift + widths[pass]));
        for (long long i = tid; i < ncand; i += NMS_THREADS) {
            const unsigned u = __float_as_uint(sc[i]);
            if (u >= vbits && u <= 0x7f800000u && ((u & himask) == (prefix & himask)))
                atomicAdd(&hist[(u >> shift) & (nb - 1)], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned cum = 0; int sel = -1;
            for (int q = nb - 1; q >= 0; --q) {
                if (cum + hist[q] >= need) { sel = q; break; }
                cum += hist[q];
            }
            if (sel < 0) { sh_sel = 0xffffffffu; sh_above = cum; }
            else { sh_sel = (unsigned)sel; sh_above = cum; }
        }
        __syncthreads();
        if (sh_sel == 0xffffffffu) { all_taken = true; break; }
        prefix |= sh_sel << shift;
        above += sh_above;
        need -= sh_above;
        __syncthreads();
    }
    // now: candidates with bits > prefix are all selected (`above` of them); `need` more with bits == prefix
    // (lowest ids first).  all_taken: every valid candidate selected (count < topk).
    const unsigned T = all_taken ? vbits : prefix;
    if (tid == 0) { sh_cnt = 0; sh_eqbase = 0; }
    __syncthreads();
    const unsigned need_eq = all_taken ? 0xffffffffu : need;
    // strictly-above candidates (or all valid when all_taken): unordered append
    for (long long i = tid; i < ncand; i += NMS_THREADS) {
        const unsigned u = __float_as_uint(sc[i]);
        const bool valid = u >= vbits && u <= 0x7f800000u;
        const bool take = valid && (all_taken ? true : (u > T));
        if (take) {
            const unsigned pos = atomicAdd(&sh_cnt, 1u);
            if (pos < NMS_MAXK) keys[pos] = ((unsigned long long)u << 32) | (unsigned)(0xffffffffu - (unsigned)i);
        }
    }
    __syncthreads();
    // equal-to-T candidates in ascending id order (ordered compaction, NMS_THREADS ids per round)
    if (!all_taken) {
        for (long long base = 0; base < ncand; base += NMS_THREADS) {
            if (sh_eqbase >= need_eq) break;
            const long long i = base + tid;
            const bool eq = i < ncand && __float_as_uint(sc[i]) == T;
            const unsigned long long bal = __ballot(eq);
            const unsigned before = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wave_tot[wave] = __popcll(bal);
            __syncthreads();
            unsigned woff = 0, tot = 0;
            for (int w = 0; w < NMS_THREADS / 64; ++w) {
                if (w < wave) woff += wave_tot[w];
                tot += wave_tot[w];
            }
            const unsigned rank = sh_eqbase + woff + before;
            if (eq && rank < need_eq) {
                const unsigned pos = above + rank;
                if (pos < NMS_MAXK) keys[pos] = ((unsigned long long)T << 32) | (unsigned)(0xffffffffu - (unsigned)i);
            }
            __syncthreads();
            if (tid == 0) sh_eqbase += tot;
            __syncthreads();
        }
    }
    __syncthreads();
    n = all_taken ? min(sh_cnt, (unsigned)NMS_MAXK) : (unsigned)topk;
    if (!all_taken) n = min(n, above + min(sh_eqbase, need_eq));
    n = min(n, (unsigned)topk);
    for (int i = tid; i < NMS_MAXK; i += NMS_THREADS)
        if ((unsigned)i >= n) keys[i] = 0ull;
    sortn = NMS_MAXK;
    __syncthreads();
    }

    // ---- 2. bitonic sort, descending ---------------------------------------------------------
    for (int k = 2; k <= sortn; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < sortn; t += NMS_THREADS) {
                const int ixj = t ^ j;
                if (ixj > t) {
                    const unsigned long long a = keys[t], c = keys[ixj];
                    const bool desc = (t & k) == 0;
                    if (desc ? (a < c) : (a > c)) { keys[t] = c; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    // ---- 3. suppression matrix -----------------------------------------------------------------
    for (int i = tid; i < (int)n; i += NMS_THREADS) {
        const unsigned id = 0xffffffffu - (unsigned)(keys[i] & 0xffffffffull);
        const float* p = rw + (long long)(id / cpb) * C + 1;
        cbox[i] = make_float4(p[0], p[1], p[2], p[3]);
        ccls[i] = id % cpb;
    }
    __syncthreads();
    const int nwords = ((int)n + 63) / 64;
    for (int e = tid; e < (int)n * nwords; e += NMS_THREADS) {
        const int i = e / nwords, w = e - i * nwords;
        const unsigned ci = ccls[i];
        const float4 bi = cbox[i];
        unsigned long long m = 0;
        // (pairs j <= i are never looked at by the walk; the class test first: an integer compare against an IoU)
        const int q0 = max(0, i + 1 - w * 64), q1 = min(64, (int)n - w * 64);
        for (int q = q0; q < q1; ++q) {
            const int j = w * 64 + q;
            if (ccls[j] == ci && box_iou(bi, cbox[j]) > iou_thresh) m |= 1ull << q;
        }
        supp[i][w] = m;
    }
    __syncthreads();
    // ---- 4. greedy walk by one wavefront, 64 candidates at a time ------------------------------------
    // (round 4) Candidate by candidate the walk is a chain of dependent LDS round trips (broadcast the removed word, test,
    // fetch the row, OR: ~150 cycles x 400 candidates).  Here lane q holds candidate c*64+q's row for the CURRENT word, the
    // 64 decisions of a chunk are taken on registers (v_readlane of the rows, wave-uniform), and only then the rows of the
    // kept candidates are OR-ed into the later words -- independent LDS reads, one wait.  Same greedy order, same result.
    if (wave == 0) {
        unsigned long long removed = 0;     // lane w owns word w (nwords <= 8)
        int nk = 0;
        for (int c = 0; c < nwords && nk < post_nms; ++c) {
            unsigned long long rem = __shfl(removed, c, 64);                 // this chunk's word, final w.r.t. earlier chunks
            const int i = c * 64 + lane;
            const unsigned long long own = i < (int)n ? supp[i][c] : 0ull;      // candidate i's row inside its own chunk (bits j > i)
            const int cnt = min(64, (int)n - c * 64);
            unsigned long long keptm = 0;
            int room = post_nms - nk;
            for (int q = 0; q < cnt && room > 0; ++q) {
                if (!((rem >> q) & 1ull)) {
                    keptm |= 1ull << q;
                    --room;
                    const unsigned lo = __builtin_amdgcn_readlane((unsigned)own, q), hi = __builtin_amdgcn_readlane((unsigned)(own >> 32), q);
                    rem |= ((unsigned long long)hi << 32) | lo;
                }
            }
            // the kept candidates of the chunk: ids / scores in rank order, their rows into the later words
            if ((keptm >> lane) & 1ull) {
                const int rank = nk + __popcll(keptm & ((1ull << lane) - 1ull));
                const unsigned long long key = keys[i];
                kept[(long long)b * post_nms + rank] = (int)(0xffffffffu - (unsigned)(key & 0xffffffffull));
                kept_scores[(long long)b * post_nms + rank] = __uint_as_float((unsigned)(key >> 32));
            }
            nk += __popcll(keptm);
            unsigned long long add = 0;
            for (unsigned long long m = keptm; m; m &= m - 1) {
                const int q = __ffsll((long long)m) - 1;
                if (lane < nwords && lane > c) add |= supp[c * 64 + q][lane];
            }
            removed |= add;
        }
        for (int q = nk + lane; q < post_nms; q += 64) {
            kept[(long long)b * post_nms + q] = -1;
            kept_scores[(long long)b * post_nms + q] = 0.f;
        }
        if (lane == 0) kept_count[b] = nk;
    }
}

extern "C" long long yolo_nms_select_workspace_bytes(int B) {
    if (B <= 0) return YOLO_EINVAL;
    return (long long)B * NMS_WS_PER_IMAGE;
}

extern "C" long long yolo_nms_workspace_bytes(int B, int nbox, int ncls, int mode, int topk) {
    (void)topk;
    if (B <= 0 || nbox <= 0) return YOLO_EINVAL;
    return ((long long)B * nbox * (mode == 1 ? ncls : 1) * 4 + 15) / 16 * 16 + yolo_nms_select_workspace_bytes(B);
}

// hist0_taken: the selection workspace was zeroed and its pass-0 histograms filled by the caller (yolo_decode_nms)
static int nms_from_scores_impl(const float* rows, const float* scores, int B, int nbox, int C, int cand_per_box,
                                float valid_thresh, float iou_thresh, int topk, int post_nms, int* kept, float* kept_scores,
                                int* kept_count, void* select_workspace, hipStream_t st, bool hist0_taken) {
    if (!rows || !scores || !kept || !kept_scores || !kept_count) return YOLO_EINVAL;
    if (B <= 0 || nbox <= 0 || C < 5 || cand_per_box < 1 || post_nms < 1) return YOLO_EINVAL;
    if (topk < 1 || topk > NMS_MAXK) return YOLO_EUNSUPPORTED;
    const long long ncand = (long long)nbox * cand_per_box;
    if (ncand > 0x7fffffffLL) return YOLO_EUNSUPPORTED;
    const unsigned long long* list = nullptr;
    unsigned* sel = nullptr;
    if (select_workspace) {
        unsigned* hist = (unsigned*)select_workspace;
        sel = hist + (long long)B * 2 * 2048;
        list = (const unsigned long long*)(sel + (long long)B * 16);
        (void)hipGetLastError();
        if (!hist0_taken) (void)hipMemsetAsync(select_workspace, 0, (size_t)B * (2 * 2048 + 16) * 4, st);
        float vt = valid_thresh > 0.f ? valid_thresh : 0.f;
        unsigned vbits;
        memcpy(&vbits, &vt, 4);
        int G = 1024 / B;                                     // ~1024 blocks over the chip
        if (G < 1) G = 1;
        if (G > 64) G = 64;
        long long per = (ncand + G - 1) / G;
        per = (per + 255) / 256 * 256;
        G = (int)((ncand + per - 1) / per);
        if (!hist0_taken)
            YOLO_LAUNCH(nms_hist_kernel, dim3(G, B), dim3(256), 0, st, scores, ncand, vbits, 0, hist, (const unsigned*)sel, per);
        YOLO_LAUNCH(nms_pick_kernel, dim3(B), dim3(64), 0, st, (const unsigned*)hist, sel, 0, topk);
        YOLO_LAUNCH(nms_hist_kernel, dim3(G, B), dim3(256), 0, st, scores, ncand, vbits, 1, hist, (const unsigned*)sel, per);
        YOLO_LAUNCH(nms_pick_kernel, dim3(B), dim3(64), 0, st, (const unsigned*)hist, sel, 1, topk);
        YOLO_LAUNCH(nms_collect_kernel, dim3(G, B), dim3(256), 0, st, scores, ncand, vbits, sel,
                    (unsigned long long*)list, per);
    }
    YOLO_LAUNCH(nms_kernel, dim3(B), dim3(NMS_THREADS), 0, st, rows, scores, nbox, C, cand_per_box, valid_thresh,
                iou_thresh, topk, post_nms, kept, kept_scores, kept_count, list, (const unsigned*)sel);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

extern "C" int yolo_nms_from_scores(const float* rows, const float* scores, int B, int nbox, int C,
                                    int cand_per_box, float valid_thresh, float iou_thresh, int topk, int post_nms,
                                    int* kept, float* kept_scores, int* kept_count, void* select_workspace,
                                    void* stream) {
    return nms_from_scores_impl(rows, scores, B, nbox, C, cand_per_box, valid_thresh, iou_thresh, topk, post_nms, kept,
                                kept_scores, kept_count, select_workspace, (hipStream_t)stream, false);
}

// yolo_decode_scores + yolo_nms_from_scores as ONE entry (BASELINE configs[4]'s post-processing): the decode pass takes the
// selection's first histogram from the scores it holds in LDS, so the score array is streamed twice (second histogram,
// collect) instead of three times.  Same rows, scores and kept ids as the two calls (tests/test_gpu_detect.py).
// select_workspace: yolo_nms_select_workspace_bytes(B), required here.
extern "C" int yolo_decode_nms(const float* out, float* rows, float* scores, int B, int C, const yolo_grid_desc* g, int mode,
                               float valid_thresh, float iou_thresh, int topk, int post_nms, int* kept, float* kept_scores,
                               int* kept_count, void* select_workspace, void* stream) {
    if (!select_workspace || !kept || !kept_scores || !kept_count || !g) return YOLO_EINVAL;
    if (B <= 0 || C < 6 || post_nms < 1 || (mode != 0 && mode != 1)) return YOLO_EINVAL;
    if (topk < 1 || topk > NMS_MAXK) return YOLO_EUNSUPPORTED;
    GridDev d; int nbox;
    int rc = make_grid(g, d, &nbox);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
    (void)hipMemsetAsync(select_workspace, 0, (size_t)B * (2 * 2048 + 16) * 4, st);
    float vt = valid_thresh > 0.f ? valid_thresh : 0.f;
    unsigned vbits;
    memcpy(&vbits, &vt, 4);
    rc = launch_decode_scores(out, rows, scores, B, C, g, mode, st, vbits, (unsigned*)select_workspace);
    if (rc) return rc;
    return nms_from_scores_impl(rows, scores, B, nbox, C, mode == 1 ? C - 6 : 1, valid_thresh, iou_thresh, topk, post_nms, kept,
                                kept_scores, kept_count, select_workspace, st, true);
}

extern "C" int yolo_nms(const float* rows, int B, int nbox, int C, int mode, float valid_thresh, float iou_thresh,
                        int topk, int post_nms, int* kept, float* kept_scores, int* kept_count, void* workspace,
                        void* stream) {
    if (!workspace) return YOLO_EINVAL;
    if (mode != 0 && mode != 1) return YOLO_EINVAL;
    if (B <= 0 || nbox <= 0 || C < 6) return YOLO_EINVAL;
    int rc = yolo_nms_scores(rows, (float*)workspace, B, nbox, C, mode, stream);
    if (rc) return rc;
    const long long sbytes = (long long)B * nbox * (mode == 1 ? C - 6 : 1) * 4;
    return yolo_nms_from_scores(rows, (const float*)workspace, B, nbox, C, mode == 1 ? C - 6 : 1, valid_thresh,
                                iou_thresh, topk, post_nms, kept, kept_scores, kept_count,
                                (char*)workspace + (sbytes + 15) / 16 * 16, stream);
}

// ---- LPD plumbing (BASELINE config 1): LicencePlateDetectioin.predict_LP, LP_detection.py:147-162 --------
// out (1, C, h, w) float32 NCHW -> pred (C): arg-max of channel 0 over the h*w cells (first index among
// ties), then sigmoid(score), xyz * 1000, three angles (sigmoid - 0.5) * 2 * r_max * pi / 180.
__global__ __launch_bounds__(256) void predict_lp_kernel(const float* __restrict__ out, float* __restrict__ pred,
                                                         int* __restrict__ best_idx, int C, int hw, float r0, float r1,
                                                         float r2) {
    float bv = -FLT_MAX; int bi = 0x7fffffff;
    for (int k = threadIdx.x; k < hw; k += blockDim.x) {
        const float v = out[k];
        if (v > bv) { bv = v; bi = k; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        argmax_combine(bv, bi, ov, oi);
    }
    __shared__ float sv[4];
    __shared__ int si[4];
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) argmax_combine(bv, bi, sv[w], si[w]);
        if (bi == 0x7fffffff) bi = 0;          // all-NaN scores: index 0, as mxnet's argmax
        si[0] = bi;
        best_idx[0] = bi;
    }
    __syncthreads();
    const int k = si[0];
    const int c = threadIdx.x;
    if (c < C) {
        float v = out[(long long)c * hw + k];
        if (c == 0) v = sigmoidf_ref(v);
        else if (c < 4) v = v * 1000.f;
        else if (c < 7) {
            const float r = c == 4 ? r0 : (c == 5 ? r1 : r2);
            v = (sigmoidf_ref(v) - 0.5f) * 2.f * r;
            v = v * 3.14159274101257324f / 180.f;
        }
        pred[c] = v;
    }
}

extern "C" int yolo_predict_lp(const float* out, float* pred, int* best_idx, int C, int h, int w, float r_max0,
                               float r_max1, float r_max2, void* stream) {
    if (!out || !pred || !best_idx || C < 7 || C > 256 || h <= 0 || w <= 0) return YOLO_EINVAL;
    YOLO_LAUNCH(predict_lp_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out, pred, best_idx, C, h * w, r_max0,
                r_max1, r_max2);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ---- CarLPNet.predict_LP + LP_pose_activation (car_and_LP/YOLO.py:133-169) ----------------------------------
// out (B, HW, C) float32 NHWC as the LP branch writes it -> pred (B, 7): per image the cell with the highest
// sigmoid(score) (first index among ties -- the arg-max is taken over the SIGMOID, which saturates), then
// [sigmoid(score), xy*1000, z*1000, three angles (sigmoid - 0.5) * 2 * r_max * pi / 180].
__global__ __launch_bounds__(256) void predict_lp_nhwc_kernel(const float* __restrict__ out, float* __restrict__ pred,
                                                              int* __restrict__ best_idx, int C, int hw, float r0,
                                                              float r1, float r2) {
    const int b = blockIdx.x;
    const float* o = out + (long long)b * hw * C;
    float bv = -FLT_MAX; int bi = 0x7fffffff;
    for (int k = threadIdx.x; k < hw; k += blockDim.x) {
        const float v = sigmoidf_ref(o[(long long)k * C]);
        if (v > bv) { bv = v; bi = k; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        argmax_combine(bv, bi, ov, oi);
    }
    __shared__ float sv[4];
    __shared__ int si[4];
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) argmax_combine(bv, bi, sv[w], si[w]);
        if (bi == 0x7fffffff) bi = 0;          // all-NaN scores: index 0, as mxnet's argmax
        si[0] = bi;
        best_idx[b] = bi;
    }
    __syncthreads();
    const int k = si[0];
    const int c = threadIdx.x;
    if (c < 7) {
        float v = o[(long long)k * C + c];
        if (c == 0) v = sigmoidf_ref(v);
        else if (c < 4) v = v * 1000.f;
        else {
            const float r = c == 4 ? r0 : (c == 5 ? r1 : r2);
            v = (sigmoidf_ref(v) - 0.5f) * 2.f * r;
            v = v * 3.14159274101257324f / 180.f;
        }
        pred[b * 7 + c] = v;
    }
}

extern "C" int yolo_predict_lp_nhwc(const float* out, float* pred, int* best_idx, int B, int hw, int C, float r_max0,
                                    float r_max1, float r_max2, void* stream) {
    if (!out || !pred || !best_idx || B <= 0 || hw <= 0 || C < 7) return YOLO_EINVAL;
    YOLO_LAUNCH(predict_lp_nhwc_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, out, pred, best_idx, C, hw, r_max0,
                r_max1, r_max2);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

