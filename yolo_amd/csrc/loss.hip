// Target assignment and losses of the reference's training step (fp32, op-by-op like NDArray:
// compiled with -ffp-contract=off).
//   yolo_assign_targets  <- _find_best + the scatter of _loss_mask      car/YOLO.py:401-480
//   yolo_loss_fwd_bwd    <- _score_weight + _get_loss + backward        car/YOLO.py:482-498
//     gluon LogisticLoss(binary) / HuberLoss(rho=1) / SoftmaxCrossEntropyLoss(sparse_label=False),
//     each the mean over every non-batch axis of loss*sample_weight.
#include "common.h"
#include <float.h>

// record per (image, object): [valid, box index k = pixel*A + anchor, ty, tx, th, tw, rot, cls...]
#define REC_HEAD 7

struct AssignGrid {
    int nscale, A, img_h, img_w, nbox;
    int cum[5], step[4];
    float ah[4][8], aw[4][8];
};

__device__ __forceinline__ float iou_ltrb_yxhw(const float4 p, float ty, float tx, float th, float tw) {
    const float l2 = tx - tw / 2.f, t2 = ty - th / 2.f, r2 = tx + tw / 2.f, b2 = ty + th / 2.f;
    const float iw = fmaxf(fminf(r2, p.z) - fmaxf(l2, p.x), 0.f);
    const float ih = fmaxf(fminf(b2, p.w) - fmaxf(t2, p.y), 0.f);
    const float inter = iw * ih;
    const float pa = (p.z - p.x) * (p.w - p.y);
    const float ta = th * tw;
    return inter / (pa + ta - inter);
}

// one block per (image, object): arg-max IoU over all anchor boxes (first index among ties), then the
// inverse decode of car/YOLO.py:432-446.
__global__ __launch_bounds__(256) void assign_kernel(const float* __restrict__ labels,
                                                     const float4* __restrict__ anchors_ltrb,
                                                     float* __restrict__ rec, int lab_w, int ncls, AssignGrid g) {
    const int bo = blockIdx.x;
    const float* L = labels + (long long)bo * lab_w;
    float* R = rec + (long long)bo * (REC_HEAD + ncls);
    if (L[0] < 0.f) {                                   // "no object" row (car/YOLO.py:468)
        if (threadIdx.x == 0) R[0] = 0.f;
        return;
    }
    const float Ly = L[1], Lx = L[2], Lh = L[3], Lw = L[4];
    float bv = -FLT_MAX;
    int bi = 0x7fffffff;
    for (int k = threadIdx.x; k < g.nbox; k += blockDim.x) {
        const float v = iou_ltrb_yxhw(anchors_ltrb[k], Ly, Lx, Lh, Lw);
        if (v > bv) { bv = v; bi = k; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    __shared__ float sv[4];
    __shared__ int si[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        const int k = bi == 0x7fffffff ? 0 : bi;          // NaN label / all-NaN IoU: box 0, as mxnet's argmax
        const int px = k / g.A, anc = k - px * g.A;
        int layer = 0;
        for (int q = 1; q < g.nscale; ++q)
            if (k >= g.cum[q]) layer = q;
        const float4 b = anchors_ltrb[k];
        const float step = (float)g.step[layer];
        float sty = (Ly - (b.w + b.y) / 2.f) * (float)g.img_h / step + 0.5f;
        sty = fminf(fmaxf(sty, 0.0001f), 0.9999f);
        float stx = (Lx - (b.z + b.x) / 2.f) * (float)g.img_w / step + 0.5f;
        stx = fminf(fmaxf(stx, 0.0001f), 0.9999f);
        R[0] = 1.f;
        R[1] = (float)k;
        R[2] = -logf(1.f / sty - 1.f);
        R[3] = -logf(1.f / stx - 1.f);
        R[4] = logf(Lh / g.ah[layer][anc]);
        R[5] = logf(Lw / g.aw[layer][anc]);
        R[6] = L[5];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < ncls; c += blockDim.x) R[REC_HEAD + c] = L[6 + c];
}

extern "C" int yolo_assign_targets(const float* labels, const float* anchors_ltrb, float* records, int B, int nobj,
                                   int ncls, const yolo_grid_desc* gd, void* stream) {
    if (!labels || !anchors_ltrb || !records || !gd || B <= 0 || nobj <= 0 || ncls < 0) return YOLO_EINVAL;
    if (gd->nscale < 1 || gd->nscale > 4 || gd->A < 1 || gd->A > 8) return YOLO_EINVAL;
    AssignGrid g;
    g.nscale = gd->nscale; g.A = gd->A; g.img_h = gd->img_h; g.img_w = gd->img_w;
    int cum = 0;
    for (int i = 0; i < 4; ++i) {
        g.cum[i] = cum;
        g.step[i] = i < gd->nscale ? gd->step[i] : 1;
        if (i < gd->nscale) {
            cum += gd->gh[i] * gd->gw[i] * gd->A;
            for (int a = 0; a < gd->A; ++a) {
                g.ah[i][a] = gd->anchors_hw[(i * gd->A + a) * 2];
                g.aw[i][a] = gd->anchors_hw[(i * gd->A + a) * 2 + 1];
            }
        }
    }
    g.cum[4] = cum;
    g.nbox = cum;
    YOLO_LAUNCH(assign_kernel, dim3(B * nobj), dim3(256), 0, (hipStream_t)stream, labels,
                (const float4*)anchors_ltrb, records, 6 + ncls, ncls, g);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ------------------------------------------------------------------------------------------------
// losses + gradient w.r.t. the logits
// ------------------------------------------------------------------------------------------------
struct LossCfg {
    float s_score, s_g1, s_g2, s_g3, s_cls;      // spec `scale` of the score / three Huber groups / class terms
    float pos_w, neg_w;                          // positive_weight / negative_weight
    int nh, g1, g2;                              // Huber channels 1..nh in three groups of g1, g2, nh-g1-g2
};

// logits (B, nbox, C): [score | Huber group 1 | group 2 | group 3 | cls...];  dlogits same shape;  losses (5, B)
// accumulated atomically (caller zero-fills).  One thread per box.  record per (image, object):
// [valid, box index, Huber targets (nh), cls (C-1-nh)].
//   car head (car/YOLO.py:491-498):          nh = 5: yx (2), hw (2), rot (1)
//   LP branch (LP_detection.py:354-360):     nh = 6: xy (2), z (1), r (3)
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ logits, const float* __restrict__ rec,
                                                   float* __restrict__ dlogits, float* __restrict__ losses, int B,
                                                   int nbox, int C, int nobj, LossCfg cfg) {
    const int b = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int nh = cfg.nh;
    const int ncls = C - 1 - nh;
    const int rhead = 2 + nh;
    const int recw = rhead + ncls;
    float l_s = 0.f, l_1 = 0.f, l_2 = 0.f, l_3 = 0.f, l_c = 0.f;
    if (k < nbox) {
        const float* p = logits + ((long long)b * nbox + k) * C;
        float* d = dlogits + ((long long)b * nbox + k) * C;
        // the LAST valid record that maps to this box wins (the reference's scatter loop overwrites)
        const float* R = nullptr;
        for (int o = 0; o < nobj; ++o) {
            const float* r = rec + ((long long)b * nobj + o) * recw;
            if (r[0] > 0.f && (int)r[1] == k) R = r;
        }
        const float mask = R ? 1.f : 0.f;
        const float inv_n = 1.f / (float)nbox;
        // score: LogisticLoss(binary), weight = where(mask>0, pos, neg) * scale
        {
            const float x = p[0], y = mask;
            const float w = (R ? cfg.pos_w : cfg.neg_w) * cfg.s_score;
            const float l = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
            l_s = l * w * inv_n;
            d[0] = (1.f / (1.f + expf(-x)) - y) * w * inv_n;
        }
        // Huber groups: HuberLoss(rho=1), weight = mask * scale, mean over (N,A,k) with k the group width
        {
            const int g3 = nh - cfg.g1 - cfg.g2;
            const float w1 = mask * cfg.s_g1 * inv_n / (float)cfg.g1;
            const float w2 = mask * cfg.s_g2 * inv_n / (float)cfg.g2;
            const float w3 = mask * cfg.s_g3 * inv_n / (float)(g3 > 0 ? g3 : 1);
            for (int j = 0; j < nh; ++j) {
                const float y = R ? R[2 + j] : 0.f;
                const float df = p[1 + j] - y, ad = fabsf(df);
                const float l = ad > 1.f ? ad - 0.5f : 0.5f * ad * ad;
                const float gr = ad > 1.f ? (df > 0.f ? 1.f : -1.f) : df;
                const int grp = j < cfg.g1 ? 0 : (j < cfg.g1 + cfg.g2 ? 1 : 2);
                const float w = grp == 0 ? w1 : (grp == 1 ? w2 : w3);
                if (grp == 0) l_1 += l * w; else if (grp == 1) l_2 += l * w; else l_3 += l * w;
                d[1 + j] = gr * w;
            }
        }
        // class: soft-label softmax cross-entropy, weight = mask * scale, mean over (N,A,1)
        {
            const float w = mask * cfg.s_cls * inv_n;
            const float* pc = p + 1 + nh;
            float* dc = d + 1 + nh;
            if (R) {
                float m = -FLT_MAX;
                for (int c = 0; c < ncls; ++c) m = fmaxf(m, pc[c]);
                float se = 0.f, sy = 0.f;
                for (int c = 0; c < ncls; ++c) { se += expf(pc[c] - m); sy += R[rhead + c]; }
                const float lse = m + logf(se);
                float l = 0.f;
                for (int c = 0; c < ncls; ++c) {
                    const float y = R[rhead + c];
                    l -= y * (pc[c] - lse);
                    dc[c] = (expf(pc[c] - lse) * sy - y) * w;
                }
                l_c = l * w;
            } else {
                for (int c = 0; c < ncls; ++c) dc[c] = 0.f;
            }
        }
    }
    // block reduction of the five partial losses -> one atomic per block per loss
    __shared__ float red[5][4];
    float v[5] = {l_s, l_1, l_2, l_3, l_c};
#pragma unroll
    for (int q = 0; q < 5; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off, 64);
        if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = v[q];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const float s_ = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        atomicAdd(&losses[threadIdx.x * B + b], s_);
    }
}

static int loss_launch(const float* logits, const float* records, float* dlogits, float* losses, int B, int nbox, int C,
                       int nobj, const float* scales5, float pos_w, float neg_w, int nh, int g1, int g2, void* stream) {
    if (!logits || !records || !dlogits || !losses || !scales5 || B <= 0 || nbox <= 0 || nobj <= 0) return YOLO_EINVAL;
    if (nh < 2 || g1 < 1 || g2 < 1 || g1 + g2 > nh || C < 1 + nh) return YOLO_EINVAL;
    LossCfg cfg{scales5[0], scales5[1], scales5[2], scales5[3], scales5[4], pos_w, neg_w, nh, g1, g2};
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
    (void)hipMemsetAsync(losses, 0, sizeof(float) * 5 * B, st);
    YOLO_LAUNCH(loss_kernel, dim3((nbox + 255) / 256, B), dim3(256), 0, st, logits, records, dlogits, losses, B, nbox, C,
                nobj, cfg);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

extern "C" int yolo_loss_fwd_bwd(const float* logits, const float* records, float* dlogits, float* losses, int B,
                                 int nbox, int C, int nobj, const float* scales5, float pos_w, float neg_w,
                                 void* stream) {
    if (C < 6) return YOLO_EINVAL;
    return loss_launch(logits, records, dlogits, losses, B, nbox, C, nobj, scales5, pos_w, neg_w, 5, 2, 2, stream);
}

// ---- CarLPNet: licence-plate targets and losses (LP_detection.py:258-360) -----------------------------------------
// record per (image, object): [valid, cell index h_f*w_+w_f, tX, tY, tZ, tr1, tr2, tr3, one-hot class (ncls)]
__global__ void assign_lp_kernel(const float* __restrict__ labels, float* __restrict__ rec, int total, int lab_w, int ncls,
                                 int fh, int fw, float step, float rm0, float rm1, float rm2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float* L = labels + (long long)i * lab_w;
    float* R = rec + (long long)i * (8 + ncls);
    if (L[0] < 0.f) { R[0] = 0.f; return; }
    int hf = (int)(L[8] / step), wf = (int)(L[7] / step);          // int(): truncation, as the reference
    hf = min(max(hf, 0), fh - 1);
    wf = min(max(wf, 0), fw - 1);
    R[0] = 1.f;
    R[1] = (float)(hf * fw + wf);
    R[2] = L[1] / 1000.f; R[3] = L[2] / 1000.f; R[4] = L[3] / 1000.f;
    const float rm[3] = {rm0, rm1, rm2};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float sg = L[4 + q] / rm[q] / 2.f + 0.5f;
        R[5 + q] = -logf(1.f / sg - 1.f);                            // nd_inv_sigmoid, yolo_gluon.py:365
    }
    const int cls = (int)L[lab_w - 1];
    for (int c = 0; c < ncls; ++c) R[8 + c] = c == cls ? 1.f : 0.f;
}

extern "C" int yolo_assign_targets_lp(const float* labels, float* records, int B, int nobj, int lab_w, int ncls, int img_h,
                                      int img_w, int step, float r_max0_deg, float r_max1_deg, float r_max2_deg,
                                      void* stream) {
    if (!labels || !records || B <= 0 || nobj <= 0 || lab_w < 10 || ncls < 1 || step <= 0) return YOLO_EINVAL;
    const int total = B * nobj;
    const float k = 3.14159265358979323846f / 180.f;
    YOLO_LAUNCH(assign_lp_kernel, dim3((total + 63) / 64), dim3(64), 0, (hipStream_t)stream, labels, records, total, lab_w,
                ncls, img_h / step, img_w / step, (float)step, r_max0_deg * k, r_max1_deg * k, r_max2_deg * k);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

extern "C" int yolo_loss_lp_fwd_bwd(const float* logits, const float* records, float* dlogits, float* losses, int B,
                                    int ncell, int C, int nobj, const float* scales5, float pos_w, float neg_w,
                                    void* stream) {
    if (C < 8) return YOLO_EINVAL;
    return loss_launch(logits, records, dlogits, losses, B, ncell, C, nobj, scales5, pos_w, neg_w, 6, 2, 1, stream);
}
