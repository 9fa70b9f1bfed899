/* yolo_amd.h -- C ABI of libyolo_amd.so: the MI355X (gfx950) YOLOv3 hot path.
 *
 * The reference (n8886919/YOLO) has no FFI/plugin interface: its seam is the Python object
 * protocol between the task drivers and MXNet operators.  Each entry
 * point below names the reference call site / MXNet operator it replaces.  All pointers are
 * DEVICE pointers unless a name ends in _host.  The caller owns every buffer; the library
 * allocates no persistent device memory.  Every function is asynchronous on `stream`
 * (a hipStream_t passed as void*), returns 0 on success, <0 on invalid argument /
 * unsupported shape, >0 = hipError_t from the launch.  Nothing throws across the ABI.
 *
 * Layout: activations are NHWC ("pixel-major, channel-contiguous") in HBM, dtype
 * YOLO_F32, YOLO_BF16, YOLO_F16 or a split type (YOLO_BF16X3 / YOLO_F16X3: two planes per pixel, below); images enter as NCHW float32 exactly as the reference feeds them
 * (car/YOLO.py:381, yolo_gluon.py:354) and head logits leave as (B, sum HW, A, C) float32
 * exactly as CarNet returns them (car/utils.py:95, basic_yolo.py:102-103).
 */
#ifndef YOLO_AMD_H
#define YOLO_AMD_H
#ifdef __cplusplus
extern "C" {
#endif

#define YOLO_F32 0
#define YOLO_BF16 1
#define YOLO_F16 2     /* IEEE half: the reference's own reduced precision (use_fp16 -> net.cast('float16'), car/YOLO.py:98-100; executor fp16 flag
                          yolo_gluon.py:204-214).  Inference entry points only (pack, fold, nchw<->nhwc, conv_fwd, stem, res_block); the
                          training entries take YOLO_F32 | YOLO_BF16 */

/* SPLIT bf16 ("bf16x3", round 6): the path on which the north-star tolerance (decoded boxes within 1e-3 of the fp32 reference) and the
 * bf16 MFMA rate meet.  A value v is stored as TWO bf16 numbers, hi = bf16(v) and lo = bf16(v - hi) (16 significant bits), and every
 * product w * x of a convolution is taken as w_hi x_hi + w_hi x_lo + w_lo x_hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
 * (three MFMAs per product; the exact-fp32 MFMA of YOLO_F32 is sixteen times slower than one).  Storage of an (N,H,W,C) activation:
 * per pixel a hi PLANE of Cp = round_up(C, 32) values (C real, the rest zero: whole 32-channel K-chunks), then -- `lo offset`
 * elements further (dense: Cp) -- the lo plane: dense pixel stride 2 * Cp elements of 2 bytes.  The kernels never write the pad
 * channels and read them as operands of zero weights: THE CALLER ZEROES A BUFFER WITH C % 32 != 0 ONCE (a NaN there would poison the
 * sums).  Inference entry points only: yolo_packed_weight_bytes / yolo_pack_conv_weights (image [w_hi | w_hi | w_lo] over
 * 3 * Cp / 32 K-chunks; Cin % 8 == 0), yolo_conv_fwd (pipelined kernels; Cout % 8 == 0 unless out_f32; no stats / tail),
 * yolo_stem_conv_fwd.  Everything else returns YOLO_EUNSUPPORTED / YOLO_EINVAL for it. */
#define YOLO_BF16X3 3
/* The same scheme on IEEE-half pairs (22 significant bits; the dropped term is 2^-22 of a product): decoded boxes indistinguishable from
 * the fp32 path's (6e-5 on the D53 random-BN nets, where YOLO_BF16X3 measures 3e-4) at the same three MFMAs per product -- for data inside
 * half's range (|v| < 65504; the lo planes live in the subnormals, which v_mfma_f32_32x32x16_f16 honours: tools/probes/f16_denorm_probe.py).
 * Same entry points, layout and restrictions as YOLO_BF16X3. */
#define YOLO_F16X3 4

#define YOLO_OK 0
#define YOLO_EINVAL (-1)
#define YOLO_EUNSUPPORTED (-2)

/* ABI revision = the layout of every struct and the argument list of every entry below.  A caller compiled against another
 * revision must not call anything else: the library reads the WHOLE yolo_conv_desc on every call (revision 2 appended the
 * tail_* fields, revision 3 added YOLO_F16, revision 4 YOLO_BF16X3 and the *_lo_offset fields), so a shorter struct from an older header would be read past its end.
 * yolo_amd/lib.py:load() refuses a library whose yolo_version() differs from the YOLO_ABI_VERSION it was written against. */
#define YOLO_ABI_VERSION 4
int yolo_version(void);

/* ---- parameter preparation ------------------------------------------------------------- */

/* Bytes of the packed weight image for a conv (Cout, Cin, ksize) in `dtype`. */
long long yolo_packed_weight_bytes(int Cout, int Cin, int ksize, int dtype);

/* OIHW float32 weights (gluon Conv2D layout, basic_yolo.py:20-26,98) -> the K-chunked,
 * LDS-swizzled image the implicit-GEMM kernel streams: [chunk][tap][Cout_pad][64 B]. */
int yolo_pack_conv_weights(const float* w_oihw, void* packed, int Cout, int Cin, int ksize,
                           int dtype, void* stream);

/* Inference-mode BatchNorm folding (gluon BatchNorm eps=1e-5):
 * scale = gamma/sqrt(var+eps), bias = beta - mean*scale, both padded with zeros to
 * yolo_padded_channels(C) floats.  gamma==NULL: scale=1, bias=beta (plain conv bias,
 * YOLOOutput basic_yolo.py:98; beta may be NULL -> 0). */
int yolo_padded_channels(int C);
int yolo_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var,
                 float eps, float* scale, float* bias, int C, void* stream);

/* ---- image plumbing -------------------------------------------------------------------- */

/* (N,C,H,W) float32 -> (N,H,W,Cpad) dtype, channels C..Cpad-1 zero.  Replaces the implicit
 * layout the reference hands to its first Convolution (car/YOLO.py:381). */
int yolo_nchw_to_nhwc(const float* x, void* y, int N, int C, int H, int W, int Cpad, int dtype,
                      void* stream);
/* (N,H,W,C) uint8 -> (N,C,H,W) float32 / 255: cv_img_2_ndarray, yolo_gluon.py:335-357. */
int yolo_image_u8_to_nchw(const unsigned char* img, float* y, int N, int H, int W, int C,
                          void* stream);
/* NHWC dtype -> NCHW float32 (debug / parity taps). */
int yolo_nhwc_to_nchw(const void* x, float* y, int N, int C, int H, int W, int dtype, void* stream);

/* ---- convolution ----------------------------------------------------------------------- */

/* Fused Conv(k in {1,3}, stride in {1,2}, pad k/2, no bias) + folded BN + LeakyReLU(slope)
 * [+ residual add after the activation].  Replaces gluoncv _conv2d (Convolution + BatchNorm
 * + LeakyReLU, call sites basic_yolo.py:20,24,26,118,121), DarknetBasicBlockV3's elementwise
 * add, and YOLOOutput's Conv2D+bias+transpose+reshape (basic_yolo.py:98-103) when out_f32=1. */
typedef struct yolo_conv_desc {
    const void* x;         /* (N,H,W,Cin) dtype                                               */
    const void* w_packed;  /* yolo_pack_conv_weights image                                    */
    const float* scale;    /* [padded Cout] folded BN scale                                   */
    const float* bias;     /* [padded Cout] folded BN bias / conv bias; scale and bias both NULL =
                              identity epilogue: y = conv(x) (+ residual), slope ignored       */
    const void* residual;  /* (N,Ho,Wo,Cout) dtype or NULL; added after the activation        */
    void* y;               /* (N,Ho,Wo,Cout) dtype, or float32 when out_f32                   */
    int N, H, W, Cin, Cout;
    int ksize, stride;
    int dtype;             /* YOLO_F32 | YOLO_BF16 | YOLO_F16 | YOLO_BF16X3 | YOLO_F16X3: activations and weights */
    int out_f32;           /* 1: y is float32 (head logits)                                   */
    float slope;           /* LeakyReLU negative slope in [0, 1]; 1.0f = linear               */
    long long y_batch_stride; /* elements between images in y; 0 = dense Ho*Wo*Cout           */
    long long y_pixel_stride; /* elements between pixels in y; 0 = dense Cout                 */
    int algo;              /* 0 = library heuristic; 1 = generic kernel; >= 2 = a specific pipelined
                              tile variant (csrc/conv_pipe.hip), YOLO_EUNSUPPORTED if not eligible    */
    long long x_pixel_stride; /* elements between pixels in x; 0 = dense Cin.  > Cin: x is a channel slice of a wider
                              NHWC buffer (the route half of a concat buffer, car/utils.py:93); images stay
                              H*W*x_pixel_stride apart                                        */
    int upsample2x;        /* 1: every output pixel is stored to the 2x2 patch (2oy..2oy+1, 2ox..2ox+1) of a
                              (N,2Ho,2Wo,*) map -- gluoncv _upsample(x, stride=2) (car/utils.py:92) fused into the
                              producing convolution; strides then refer to that map (y_batch_stride 0 =
                              4*Ho*Wo*y_pixel_stride); no residual, no out_f32                */
    /* Training step: Gluon BatchNorm's batch statistics taken in the convolution's epilogue instead of
     * in a pass of their own over the tensor.  stats != NULL: the kernel also writes yolo_conv_stats_rows() partial rows
     * [row][2][yolo_padded_channels(Cout)] float32 of per-channel sums over its pixel tiles (no atomics), which
     * yolo_bn_train_fwd_partials / yolo_bn_train_bwd_partials reduce in double.  stats_mode 1: sum(y), sum(y^2) of the
     * output y (the BatchNorm that follows this convolution in the forward pass).  stats_mode 2 (a data gradient,
     * y = d(loss)/d(z) of the layer BEHIND it): sum(da), sum(da * xhat) with da = y * lrelu'(gamma*xhat + beta),
     * xhat = (stats_y - mean) * invstd, stats_y = that layer's raw convolution output (same shape as y, dense).
     * bf16, pipelined kernels only (YOLO_EUNSUPPORTED otherwise: run the separate reduction). */
    void* stats;
    int stats_mode;
    const void* stats_y;
    const float* stats_mean;
    const float* stats_invstd;
    const float* stats_gamma;
    const float* stats_beta;
    float stats_slope;
    /* A 1x1 convolution fused BEHIND this one (the "tail"): tail_y = act(conv1x1(y)) -- the next DarknetBasicBlockV3's first
     * _conv2d (basic_yolo.py:26), a YOLODetectionBlockV3's 1x1 after its 3x3 (basic_yolo.py:118), or YOLOOutput after the tip
     * (basic_yolo.py:98-105, tail_out_f32 = 1) -- computed by the same kernel from the output tile it has just stored: no second
     * launch, no HBM read of y; y is still written.  tail_w_packed: yolo_pack_conv_weights image of the (tail_cout, Cout, 1, 1)
     * weights; tail_scale / tail_bias padded like scale / bias; tail_y (N,Ho,Wo,tail_cout) dtype or float32; strides as for y.
     * Bit-identical to the separate 1x1 launch.  Needs: bf16, ksize 3, Cout <= 256 and a multiple of 32, tail_cout <= 128, no
     * out_f32 / upsample2x / stats on this convolution, and an 8-wave pipelined variant whose tile holds every channel of a pixel
     * (256-cout tiles: algo 2, 6; stride 2: 10, 16, 18; Cout <= 128 also the 128-cout tiles 7; 9, 17; algo 0 picks one) -- YOLO_EUNSUPPORTED otherwise (run the two launches). */
    const void* tail_w_packed;
    const float* tail_scale;
    const float* tail_bias;
    void* tail_y;
    int tail_cout;
    int tail_out_f32;
    float tail_slope;
    long long tail_y_batch_stride;
    long long tail_y_pixel_stride;
    /* split types only (YOLO_BF16X3 / YOLO_F16X3; ignored otherwise): elements between a pixel's hi plane and its lo plane in x and in y; 0 = dense
     * (round_up(Cin, 32), round_up(Cout, 32)).  A channel slice of a wider split buffer of Ctot channels (the halves of a concat
     * buffer, car/utils.py:93) has pixel stride 2 * Ctot and lo offset Ctot.  Dense strides of a split tensor count both padded
     * planes (pixel stride 2 * round_up(C, 32)); the residual is dense; y_lo_offset is not used when out_f32. */
    long long x_lo_offset;
    long long y_lo_offset;
} yolo_conv_desc;

int yolo_conv_fwd(const yolo_conv_desc* d, void* stream);
/* Partial rows yolo_conv_fwd would write into d->stats (> 0), or YOLO_EUNSUPPORTED if the kernel it would launch for `d`
 * has no statistics epilogue.  Host-only, no launch; d->stats only has to be non-NULL. */
int yolo_conv_stats_rows(const yolo_conv_desc* d);
/* Name of the kernel instantiation yolo_conv_fwd would launch for `d` (as rocprofv3 prints it);
 * used by bench.py to attribute measured time to the dominant kernel.  Host-only, no launch. */
int yolo_conv_kernel_name(const yolo_conv_desc* d, char* buf, int len);

/* The network's first _conv2d (basic_yolo.py:20) fused with the image layout change: Conv3x3 s1 p1 over
 * the (N,3,H,W) float32 NCHW image exactly as the reference feeds it (car/YOLO.py:381) + folded BN +
 * LeakyReLU -> (N,H,W,Cout) bf16 NHWC.  w_oihw: (Cout,3,3,3) float32 (unpacked); Cout % 4 == 0, <= 64;
 * dtype YOLO_BF16 / YOLO_F16 (YOLO_F32 callers use yolo_nchw_to_nhwc + yolo_conv_fwd: EUNSUPPORTED here), or YOLO_BF16X3 / YOLO_F16X3: a direct
 * fp32 convolution on the vector pipe (K = 27: nothing for the matrix pipe to win) whose output is stored split, dense
 * (N,H,W,[hi Cout | lo Cout]); Cout % 8 == 0 there. */
int yolo_stem_conv_fwd(const float* x_nchw, const float* w_oihw, const float* scale, const float* bias,
                       void* y, int N, int H, int W, int Cin, int Cout, int dtype, float slope,
                       void* stream);

/* The same with BatchNorm's batch sums of the stored output taken in the kernel (training step; see yolo_conv_desc.stats):
 * partials = yolo_stem_stats_rows() rows of [2][Cout] float32 (pass cout_pad = Cout to yolo_bn_train_fwd_partials).
 * Cout % 8 == 0 and 64 % (Cout / 8) == 0 (YOLO_EUNSUPPORTED otherwise). */
int yolo_stem_stats_rows(int N, int H, int W, int Cout);
int yolo_stem_conv_fwd_stats(const float* x_nchw, const float* w_oihw, const float* scale, const float* bias,
                             void* y, int N, int H, int W, int Cin, int Cout, int dtype, float slope,
                             float* partials, void* stream);

/* One DarknetBasicBlockV3 (basic_yolo.py:26; gluoncv darknet.py: x + conv3x3(C)(conv1x1(C/2)(x)), each conv with
 * folded BN + LeakyReLU, no activation after the add) as ONE inference kernel for the first stages: x, y (N,H,W,C)
 * bf16 NHWC; w1_packed / w2_packed = yolo_pack_conv_weights images of the (C/2,C,1,1) and (C,C/2,3,3) convs; scale /
 * bias = yolo_fold_bn of each layer.  The C/2-channel map between the two convs stays in LDS and x is read once.  Same
 * rounding points as yolo_conv_fwd run twice (mid rounded to bf16 once; residual added in fp32 before the output's
 * one rounding).  YOLO_BF16 and C in {64, 128} only (YOLO_EUNSUPPORTED otherwise: run the two layers separately). */
int yolo_res_block_fwd(const void* x, const void* w1_packed, const float* scale1, const float* bias1,
                       const void* w2_packed, const float* scale2, const float* bias2, void* y, int N, int H, int W,
                       int C, int dtype, float slope, void* stream);

/* The first TWO layers fused for inference: the stem above and the first stage's down-sampling _conv2d (basic_yolo.py:24,
 * Conv3x3 s2 p1, C1 -> C2) + folded BN + LeakyReLU -> y (N,(H-1)/2+1,(W-1)/2+1,C2) bf16 NHWC; the full-resolution
 * C1-channel map between them stays in LDS.  w1_oihw (C1,3,3,3) float32; w2_packed = yolo_pack_conv_weights image of
 * the (C2,C1,3,3) conv.  Bit-identical to yolo_stem_conv_fwd followed by yolo_conv_fwd.  C1 == 32, C2 == 64 and
 * YOLO_BF16 only (YOLO_EUNSUPPORTED otherwise: run the two layers separately). */
int yolo_stem_down_fwd(const float* x_nchw, const float* w1_oihw, const float* scale1, const float* bias1,
                       const void* w2_packed, const float* scale2, const float* bias2, void* y, int N, int H, int W,
                       int C1, int C2, int dtype, float slope, void* stream);

/* Synthetic-target compositing of RenderCar.render (car/render_car.py:135-137): out = clip((bg / 255) * (1 - mask) +
 * fg * mask, 0, 1) over n float32 elements (n % 4 == 0; (B,3,H,W) tensors: bg 0..255, fg and mask 0..1). */
int yolo_composite(const float* bg, const float* fg, const float* mask, float* out, long long n, void* stream);
/* The blend of LPGenerator.add (yolo_modules/licence_plate_render/__init__.py:163-164), which pastes the projected plate
 * onto images that are ALREADY 0..1: out = clip(bg * (1 - mask) + fg * mask, 0, 1). */
int yolo_composite_unit(const float* bg, const float* fg, const float* mask, float* out, long long n, void* stream);

/* 2x nearest up-sample of `up` (N,H/2,W/2,C1) + channel concat with `route` (N,H,W,C2) ->
 * (N,H,W,C1+C2), up-sampled channels first: gluoncv _upsample + F.concat, car/utils.py:92-93. */
int yolo_upsample2x_concat(const void* up, const void* route, void* y, int N, int H, int W,
                           int C1, int C2, int dtype, void* stream);

/* ---- detection post-processing --------------------------------------------------------- */

/* Anchor-grid description shared by decode / assignment (car/YOLO.py:112-155):
 * nscale scales fine->coarse; scale i has grid (gh[i], gw[i]), stride step[i] px and A anchors
 * anchors_hw[(i*A+a)*2+{0,1}] = (h, w) as fractions of the image. */
typedef struct yolo_grid_desc {
    int nscale, A;
    int img_h, img_w;
    int gh[4], gw[4], step[4];
    float anchors_hw[4 * 8 * 2];
} yolo_grid_desc;

/* out (B, N, A, C) float32 logits, channel order [obj, ty, tx, th, tw, rot, cls...]
 * (slice_point [1,3,5,6,C], car/v1/spec.yaml:6) -> rows (B, N*A, C) float32
 * [sigmoid(obj), l, t, r, b, rot, cls...]: _yxhw_to_ltrb + the concat in predict,
 * car/YOLO.py:552-579. */
int yolo_decode(const float* out, float* rows, int B, int C, const yolo_grid_desc* g, void* stream);

/* Per-image arg-max of sigmoid(obj) over N*A boxes (lowest index among ties, mxnet argmax),
 * and the reference's output row [score, y, x, h, w, rot, cls...]: car/YOLO.py:581-597.
 * `out` are the raw logits (B,N,A,C); pred (B,C) float32; best_idx (B) int32. */
int yolo_predict_top1(const float* out, float* pred, int* best_idx, int B, int C,
                      const yolo_grid_desc* g, void* stream);

/* LicencePlateDetectioin.predict_LP (licence_plate/LP_detection.py:147-162; BASELINE config 1 plumbing):
 * out (1,C,h,w) float32 NCHW -> pred (C) [sigmoid(score), xyz*1000, 3 angles (sigmoid-.5)*2*r_max*pi/180, rest
 * raw] of the cell with the highest channel-0 value (first among ties); best_idx (1) int32. */
int yolo_predict_lp(const float* out, float* pred, int* best_idx, int C, int h, int w, float r_max0,
                    float r_max1, float r_max2, void* stream);

/* CarLPNet.predict_LP + LP_pose_activation (car_and_LP/YOLO.py:133-169): out (B, h*w, C) float32 NHWC (the LP branch
 * output, C = LP_slice_point[-1] >= 7, channel order [score, x, y, z, r1, r2, r3, class...]) -> pred (B, 7): per image
 * the cell with the highest sigmoid(score) (first among ties): [sigmoid(score), xyz * 1000, three angles
 * (sigmoid - 0.5) * 2 * r_max * pi / 180]; best_idx (B) the chosen cells. */
int yolo_predict_lp_nhwc(const float* out, float* pred, int* best_idx, int B, int hw, int C, float r_max0,
                         float r_max1, float r_max2, void* stream);

/* get_iou(predict, target, mode=2), yolo_gluon.py:127-168: boxes (n,4) ltrb vs one target
 * [c,y,x,h,w] (5 floats, device) -> iou (n). */
int yolo_iou_ltrb_vs_yxhw(const float* boxes, const float* target, float* iou, int n, void* stream);
/* get_iou(predict, target, mode=1) -- the reference's DEFAULT mode (yolo_gluon.py:127): target [c,l,t,r,b]; keeps the
 * reference's target_area = target[3] * target[4] (yolo_gluon.py:166; = r * b in this mode), so a caller that omits
 * `mode` gets the reference's numbers, quirk included. */
int yolo_iou_ltrb_vs_cltrb(const float* boxes, const float* target, float* iou, int n, void* stream);

/* Greedy per-class NMS over decoded rows (not in the reference -- SURVEY.md S1; semantics =
 * SURVEY App. A.8).  mode 0: score = sigmoid(obj), class-agnostic; mode 1: candidates are
 * (box, class) pairs scored sigmoid(obj)*softmax(cls)_c, suppression within a class.
 * kept (B, post_nms) int32 candidate ids in score order padded with -1; kept_scores same
 * shape; kept_count (B).  workspace: yolo_nms_workspace_bytes() = the score array + the selection workspace. */
long long yolo_nms_workspace_bytes(int B, int nbox, int ncls, int mode, int topk);
/* The two halves of yolo_nms, exposed so the score array can be inspected / injected:
 * scores (B, nbox) [mode 0] or (B, nbox*ncls) [mode 1]; candidate id = box*cand_per_box + class.
 * select_workspace (yolo_nms_select_workspace_bytes(B), may be NULL): with it the top-k pre-selection runs as
 * chip-wide passes over all images instead of one block per image (same result; ~6x faster at 608x608). */
long long yolo_nms_select_workspace_bytes(int B);
int yolo_nms_scores(const float* rows, float* scores, int B, int nbox, int C, int mode, void* stream);
/* yolo_decode + yolo_nms_scores in one pass over the logits (the logits are read once; bit-identical results):
 * out (B,N,A,C) -> rows (B,nbox,C) and scores as above. */
int yolo_decode_scores(const float* out, float* rows, float* scores, int B, int C, const yolo_grid_desc* g, int mode,
                       void* stream);
int yolo_nms_from_scores(const float* rows, const float* scores, int B, int nbox, int C,
                         int cand_per_box, float valid_thresh, float iou_thresh, int topk,
                         int post_nms, int* kept, float* kept_scores, int* kept_count,
                         void* select_workspace, void* stream);
int yolo_nms(const float* rows, int B, int nbox, int C, int mode, float valid_thresh,
             float iou_thresh, int topk, int post_nms, int* kept, float* kept_scores,
             int* kept_count, void* workspace, void* stream);
/* yolo_decode_scores + yolo_nms_from_scores as ONE call -- the post-processing of BASELINE configs[4] (608x608 inference incl.
 * anchor decode + NMS; no counterpart in the reference): the decode pass takes the selection's first
 * radix histogram from the scores it holds in LDS, so the score array is streamed twice instead of three times.  Identical
 * rows, scores and kept ids to the two calls.  select_workspace: yolo_nms_select_workspace_bytes(B), REQUIRED here. */
int yolo_decode_nms(const float* out, float* rows, float* scores, int B, int C, const yolo_grid_desc* g, int mode,
                    float valid_thresh, float iou_thresh, int topk, int post_nms, int* kept, float* kept_scores,
                    int* kept_count, void* select_workspace, void* stream);

/* ---- training step (fp32 parity path) ------------------------------------------------------- */

/* Packed weight image of the data-gradient convolution of a forward conv (Cout_f, Cin_f, k): dx =
 * conv_stride1(dy [dilated 2x for stride 2], W'), W'[ci][co][a][b] = W[co][ci][k-1-a][k-1-b]; run it with
 * yolo_conv_fwd (scale 1, bias 0, slope 1, residual = gradient to accumulate into).  Bytes:
 * yolo_packed_weight_bytes(Cin_f, Cout_f, k, dtype).  Replaces the autograd backward of Convolution
 * (sum(losses).backward(), car/YOLO.py:394). */
int yolo_pack_conv_weights_dgrad(const float* w_oihw, void* packed, int Cout_f, int Cin_f, int ksize,
                                 int dtype, void* stream);

/* Sub-pixel form of the data gradient of a 3x3 stride-2 pad-1 convolution (Cout_f, Cin_f) -- the down-sampling convs of
 * every stage, basic_yolo.py:24 -- for even input sizes: instead of zero-dilating dy (4x the arithmetic), the four
 * sub-pixel phases of dx are four output-channel blocks of one 2x2-window convolution over dy:
 *   dx[n, 2m+a, 2n+b, c] = sum_{ty,tx in {0,1}} sum_k W'[(2a+b) Cin_f + c][k][ty][tx] dy[n, m+ty, n+tx, k]   (zero beyond)
 * with W'[..a..][k][ty] = W[k][c][1] for (a,ty) = (0,0), W[k][c][2] for (1,0), W[k][c][0] for (1,1), 0 for (0,1)
 * (likewise b/tx).  yolo_pack_conv_weights_dgrad_s2 writes that image (yolo_packed_weight_bytes(4 Cin_f, Cout_f, 2,
 * dtype) bytes; batch records: Cout = 4 Cin_f, Cin = Cout_f, ksize = 2, dgrad = 2).  yolo_conv_dgrad_s2 runs it:
 * d->x = dy (N,H,W,Cin = Cout_f), d->Cout = 4 Cin_f, d->y = dx (N,2H,2W,Cin_f) dense, d->residual = a gradient to
 * accumulate onto (same shape as dx) or NULL, d->scale / d->bias over 4 Cin_f (padded) channels, ksize / stride /
 * strides ignored; bf16 only.  YOLO_EUNSUPPORTED (Cin_f % 8, Cout_f % 32, halo does not fit): use yolo_dilate2x +
 * yolo_conv_fwd instead.  d->algo: 0 = heuristic, or one of 2 / 6 / 10 / 4 (256 / 192 / 128 px x 256, 128 x 128). */
int yolo_pack_conv_weights_dgrad_s2(const float* w_oihw, void* packed, int Cout_f, int Cin_f, int dtype, void* stream);
int yolo_conv_dgrad_s2(const yolo_conv_desc* d, void* stream);

/* Every conv of a network packed in ONE launch (the training step re-packs all images after each update).
 * items_device: device array of n_items records { const float* w_oihw; void* packed; int Cout, Cin, ksize, dgrad; }
 * (32 bytes each; dgrad = 1: the data-gradient image of the FORWARD conv (Cin, Cout swapped as in
 * yolo_pack_conv_weights_dgrad's arguments)); first_block_device: n_items + 1 prefix sums of
 * yolo_pack_batch_blocks(Cout, Cin, ksize, dtype) over the items; total_blocks = the last prefix. */
typedef struct yolo_pack_item { const float* w_oihw; void* packed; int Cout, Cin, ksize, dgrad; } yolo_pack_item;
long long yolo_pack_batch_blocks(int Cout, int Cin, int ksize, int dtype);
int yolo_pack_conv_weights_batch(const void* items_device, const long long* first_block_device, int n_items,
                                 long long total_blocks, int dtype, void* stream);
/* The forward and the data-gradient image of every listed conv from ONE read of its weights (YOLO_BF16; Cout and Cin
 * multiples of 32; 1x1 and 3x3): items_device = n_items records { const float* w_oihw; void* packed_fwd; void*
 * packed_dgrad; int Cout, Cin, ksize, reserved; } (40 bytes), first_block_device = prefix sums of yolo_pack_pair_blocks()
 * (YOLO_EUNSUPPORTED for a conv this entry does not take).  Bit-identical to yolo_pack_conv_weights +
 * yolo_pack_conv_weights_dgrad except that the padding rows of the images are not written: zero-fill the buffers once. */
typedef struct yolo_pack_pair { const float* w_oihw; void* packed_fwd; void* packed_dgrad; int Cout, Cin, ksize, reserved; } yolo_pack_pair;
long long yolo_pack_pair_blocks(int Cout, int Cin, int ksize);
int yolo_pack_conv_weights_pairs(const void* items_device, const long long* first_block_device, int n_items,
                                 long long total_blocks, void* stream);

/* Gluon BatchNorm in training mode fused with LeakyReLU (+ residual add):
 * batch mean / biased variance over (N,H,W) of the NHWC conv output y (npix x C, dtype; C % 8 == 0),
 * z = lrelu(gamma*(y-mean)*invstd + beta) [+ residual]; running stats r = momentum*r + (1-momentum)*batch
 * (running_var takes the biased variance).  Statistics are accumulated in double.  workspace: 2*C doubles (shared by the
 * backward), ZERO-FILLED by the caller before its first use; every call leaves it zeroed again. */
int yolo_bn_train_fwd(const void* y, const float* gamma, const float* beta, const void* residual,
                      void* z, float* mean, float* invstd, float* running_mean, float* running_var,
                      double* workspace, long long npix, int C, float eps, float momentum, float slope,
                      int dtype, void* stream);
/* Backward of the above w.r.t. y, gamma, beta given dz (the residual branch receives dz unchanged). */
int yolo_bn_train_bwd(const void* dz, const void* y, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, void* dy, float* dgamma, float* dbeta,
                      double* workspace, long long npix, int C, float slope, int dtype, void* stream);

/* The same two calls as TWO launches each: the per-layer finalize / parameter-gradient launches are folded into the apply
 * pass.  `workspace` (2*C doubles) must be ZERO on entry and is left dirty; `zero_next` (a different buffer of
 * zero_next_count doubles -- the next call may have more channels --, or NULL) is zeroed for the caller's next BatchNorm
 * call: callers alternate two workspaces.  Same arithmetic as yolo_bn_train_fwd / _bwd.  yolo_bn_train_fwd_pp does NOT
 * work in place: z == y is refused with YOLO_EINVAL (every block re-reads y at pixel 0, the pivot of its shifted sums,
 * while another block writes z there); yolo_bn_train_fwd may run in place. */
int yolo_bn_train_fwd_pp(const void* y, const float* gamma, const float* beta, const void* residual,
                         void* z, float* mean, float* invstd, float* running_mean, float* running_var,
                         double* workspace, double* zero_next, int zero_next_count, long long npix, int C, float eps,
                         float momentum, float slope, int dtype, void* stream);
int yolo_bn_train_bwd_pp(const void* dz, const void* y, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, void* dy, float* dgamma, float* dbeta,
                         double* workspace, double* zero_next, int zero_next_count, long long npix, int C, float slope,
                         int dtype, void* stream);
/* The same two calls with the reduction pass over the tensor(s) replaced by the partial rows a convolution's statistics
 * epilogue wrote (yolo_conv_desc.stats: stats_mode 1 for the forward call, 2 for the backward call; rows =
 * yolo_conv_stats_rows(), cout_pad = yolo_padded_channels(C)): one small kernel sums the rows in double, then the apply
 * pass runs as in the _pp calls.  YOLO_BF16 only. */
int yolo_bn_train_fwd_partials(const float* partials, int rows, int cout_pad, const void* y, const float* gamma,
                               const float* beta, const void* residual, void* z, float* mean, float* invstd,
                               float* running_mean, float* running_var, double* workspace, double* zero_next,
                               int zero_next_count, long long npix, int C, float eps, float momentum, float slope,
                               int dtype, void* stream);
int yolo_bn_train_bwd_partials(const float* partials, int rows, int cout_pad, const void* dz, const void* y,
                               const float* mean, const float* invstd, const float* gamma, const float* beta,
                               void* dy, float* dgamma, float* dbeta, double* workspace, double* zero_next,
                               int zero_next_count, long long npix, int C, float slope, int dtype, void* stream);


/* Weight gradient of Conv(k, stride, pad k/2): dw (Cout,Cin,k,k) float32 += sum over pixels of
 * dy (N,Ho,Wo,[dy_pixel_stride]) x (N,H,W,Cin), NHWC `dtype`.  The caller zero-fills dw once per step.
 * YOLO_F32: MFMA 32x32x2 f32.  YOLO_BF16: MFMA 32x32x16 bf16 fed by transposing LDS reads
 * (ds_read_b64_tr_b16), fp32 accumulation; needs `workspace` of yolo_conv_wgrad_workspace_bytes(), ZERO-FILLED by
 * the caller before its first use; every call leaves it zeroed again.  bf16: x must be smaller than 4 GiB (32-bit
 * buffer offsets; YOLO_EUNSUPPORTED otherwise). */
long long yolo_conv_wgrad_workspace_bytes(int Cin, int Cout, int ksize, int dtype);
int yolo_conv_wgrad(const void* dy, const void* x, float* dw_oihw, int N, int H, int W, int Cin, int Cout,
                    int ksize, int stride, long long dy_pixel_stride, int dtype, void* workspace,
                    void* stream);
/* The same with an explicit kernel choice (tuning and tests): algo 0 = the library's choice (what yolo_conv_wgrad
 * runs); 1 = the register-staged kernels (per-tap / column-strip / row-group); 2 / 3 = the pipelined row-walk kernel of
 * the 3x3 stride-1 layers with Cin and Cout multiples of 64 (LDS-DMA ring, x fragments of the three kernel rows held
 * in registers), one 16-column walker / four 4-column walkers per block; 4 = 3 with 8-wave blocks whose two halves walk
 * different row slices and are summed through LDS before the atomics; 5 / 6 = the LDS-DMA GEMM of the 1x1 layers (Cin, Cout
 * multiples of 128) with a 128 x 128 / 256 cout x 128 cin tile; YOLO_EUNSUPPORTED outside a kernel's domain. */
int yolo_conv_wgrad_algo(const void* dy, const void* x, float* dw_oihw, int N, int H, int W, int Cin, int Cout,
                         int ksize, int stride, long long dy_pixel_stride, int dtype, void* workspace, int algo,
                         void* stream);
/* db[c] += sum over npix rows of dy (row stride pixel_stride, 0 = C): YOLOOutput's bias gradient. */
int yolo_bias_grad(const void* dy, float* db, long long npix, int C, long long pixel_stride, int dtype,
                   void* stream);
/* float32 (B, rows, [src strides]) C values per row -> dense (B*rows, Cpad) of `dtype`, zero padded. */
int yolo_gather_rows(const float* src, void* dst, int B, long long rows_per_batch, int C, int Cpad,
                     long long src_batch_stride, long long src_row_stride, int dtype, void* stream);
/* D (N,H,W,C): D[n,2y,2x,:] = dy[n,y,x,:] (dy is (N,Ho,Wo,C)), zeros elsewhere: turns the stride-2 data
 * gradient into a stride-1 convolution.  C % 8 == 0. */
int yolo_dilate2x(const void* dy, void* d, int N, int H, int W, int Ho, int Wo, int C, int dtype,
                  void* stream);
/* Backward of yolo_upsample2x_concat: dcat (N,H,W,C1+C2) -> dup (N,H/2,W/2,C1), droute (N,H,W,C2);
 * accumulate_* != 0 adds into the destination. */
int yolo_upsample2x_concat_bwd(const void* dcat, void* dup, void* droute, int N, int H, int W, int C1,
                               int C2, int accumulate_up, int accumulate_route, int dtype, void* stream);
int yolo_add(const void* a, const void* b, void* y, long long n, int dtype, void* stream);

/* _find_best + the scatter of _loss_mask (car/YOLO.py:401-480): labels (B,nobj,6+ncls)
 * [cls,y,x,h,w,rot,dist...] (cls < 0 = no object), anchors_ltrb (nbox,4) = _get_default_ltrb
 * (car/YOLO.py:209-240) -> records (B,nobj,7+ncls) [valid, box index, ty,tx,th,tw, rot, cls...]. */
int yolo_assign_targets(const float* labels, const float* anchors_ltrb, float* records, int B, int nobj,
                        int ncls, const yolo_grid_desc* g, void* stream);
/* _score_weight + _get_loss + the backward of sum(losses) (car/YOLO.py:482-498, :394): logits
 * (B,nbox,C) -> dlogits (B,nbox,C), losses (5,B) [score, box_yx, box_hw, rotate, class].
 * scales5_host: 5 floats on the HOST (spec `scale`; rotate is 0 unless car_rotate). */
int yolo_loss_fwd_bwd(const float* logits, const float* records, float* dlogits, float* losses, int B,
                      int nbox, int C, int nobj, const float* scales5_host, float pos_w, float neg_w,
                      void* stream);

/* CarLPNet's licence-plate branch (licence_plate/LP_detection.py:258-360, car_and_LP/YOLO.py:262-300).
 * yolo_assign_targets_lp = _find_best_LP + the scatter of _loss_mask_LP: labels (B, nobj, lab_w >= 10) rows
 * [flag (<0: none), X, Y, Z (mm), r1, r2, r3 (rad), x_px, y_px, ..., type] -> records (B, nobj, 8 + ncls)
 * [valid, cell = clip(int(y_px/step))*w_ + clip(int(x_px/step)), XYZ/1000, inv_sigmoid(r/r_max/2 + 0.5) x3,
 * one-hot type].  yolo_loss_lp_fwd_bwd = _score_weight_LP + _get_loss_LP + backward on the LP output
 * (B, ncell, C) [score, xy(2), z(1), r(3), class(C-7)]: losses (5, B) [LP_score, LP_xy, LP_z, LP_r, LP_class]. */
int yolo_assign_targets_lp(const float* labels, float* records, int B, int nobj, int lab_w, int ncls, int img_h,
                           int img_w, int step, float r_max0_deg, float r_max1_deg, float r_max2_deg, void* stream);
int yolo_loss_lp_fwd_bwd(const float* logits, const float* records, float* dlogits, float* losses, int B,
                         int ncell, int C, int nobj, const float* scales5, float pos_w, float neg_w, void* stream);
/* mxnet Adam, t = 1-based update count, rescale = 1/global batch
 * (trainer.step(batch_size), car/YOLO.py:396). */
int yolo_adam_step(float* w, const float* grad, float* m, float* v, long long n, int t, float lr,
                   float beta1, float beta2, float eps, float rescale, void* stream);
/* The same update with rescale = 1 / *global_batch_dev read ON THE DEVICE: the batch_size of trainer.step(batch_size)
 * (car/YOLO.py:396) as the SUM of the ranks' shard sizes, which the caller's gradient all-reduce delivers in a slot
 * of the last bucket (yolo_amd/train.py) -- uneven shards (split_render_data, yolo_gluon.py:100-124) need no
 * collective of their own and no host synchronisation. */
int yolo_adam_step_dev(float* w, const float* grad, float* m, float* v, long long n, int t, float lr,
                       float beta1, float beta2, float eps, const float* global_batch_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif
