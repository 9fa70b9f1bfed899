// Kernel argument block and host helpers shared by the generic (conv_igemm.hip) and pipelined
// (conv_pipe.hip) implicit-GEMM convolutions.
#pragma once
#include "common.h"

// Exact unsigned division by a launch-constant divisor (n < 2^31): q = (n * mul) >> sh.  The per-tile set-up
// does ~20 divisions per thread (pixel -> strip row / image / column); at ~30 instructions each they were a
// measurable share of short-K tiles.
struct FastDiv {
    unsigned mul, sh;
};
static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    f.sh = 31 + s;
    f.mul = (unsigned)(((1ull << f.sh) + d - 1) / d);
    return f;
}
__device__ __forceinline__ int fdiv(int n, FastDiv f) {
    return (int)(((unsigned long long)(unsigned)n * f.mul) >> f.sh);
}

struct ConvArgs {
    const char* x;
    const char* wp;
    const float* scale;
    const float* bias;
    const char* res;
    char* y;
    int N, H, W, Cin, Ho, Wo, Cout, Cout_pad;
    int TWt, nstrips, tiles_per_strip, PW, total_i;
    int row_swz, TP;            // conv_pipe 3x3 stride 1: halo pitch TWt + 4 and a row-relative unit swizzle (see launch_pipe); TP = PW - 4 * row_swz (stride 2: (PW - TWt) / 4)
    int halo_strict;        // conv_pipe.hip launch_pipe: leave one slot of the staged halo unused (the 2x2-window 4-wave tile)
    int tile_px;            // conv_igemm.hip: strip pixels a tile covers (= its 128 unless the shape needs row-limited tiles); 0 elsewhere
    int nchunks, tiles_c;
    int buf32;              // conv_epilogue.h: 1 = y, the residual (and stats_y, tail_y) extents fit 31-bit byte offsets: the epilogue's
                            // loads / stores are unconditional buffer accesses (out-of-range offset = no access)
    int lab;                // lab build only (YOLO_EPI_AB): epilogue ablation bits -- 1 drop the stores, 2 drop the residual loads, 4 skip the epilogue, 8 no scale / bias loads, 16 no LDS transpose; K-loop probes (wrong results): 32 every second barrier dropped, 64 no weight DMAs, 128 no input DMAs, 256 3x3 input fragments from consecutive slots (no bank conflicts)
    int vblocks;            // conv_pipe.hip: number of (pixel tile, cout tile) units = the grid size unless the blocks are persistent
    int out_f32;
    int x_ps;       // elements between input pixels (>= Cin: x may be a channel slice of a wider NHWC buffer)
    int up2;        // 1: every output pixel is stored to its 2x2 patch of the (N,2Ho,2Wo) map (nearest 2x up-sampling)
    int d2s;        // 1: Cout = 4 sub-pixel phases x Cout/4 channels, stored depth-to-space into (N,2Ho,2Wo,Cout/4)
    float slope;
    long long y_bs, y_ps;
    long long r_bs, r_ps;   // residual strides (elements); differ from y's when y is a channel slice of a wider buffer
    FastDiv d_PW, d_H1, d_TWt, d_Ho, d_HoWo, d_tc, d_tps;   // divisors PW, H+1, TWt, Ho, Ho*Wo, tiles_c, tiles_per_strip
    // BatchNorm batch statistics folded into the epilogue (training step; conv_epilogue.h): per (pixel tile, pixel wave)
    // partial column sums [rows][2][Cout_pad] fp32, summed in double by bn_stats_finish_kernel (train.hip)
    float* stats;           // nullptr: none
    int stats_mode;         // 1: sum(y), sum(y^2) of the output; 2: sum(da), sum(da * xhat) of the BatchNorm BEHIND the output
    const char* s_y;        // mode 2: that layer's raw convolution output (same shape as the output here), dense
    const float* s_mean; const float* s_invstd; const float* s_gamma; const float* s_beta;
    float s_slope;
    // A 1x1 convolution fused BEHIND this one (yolo_conv_desc.tail_*; conv_pipe.hip, kernel flag 3): computed by the same block
    // from the output tile it has just stored.  t_wp == nullptr: none.
    const char* t_wp;       // packed 1x1 weights (Cin = this conv's Cout)
    const float* t_scale; const float* t_bias;
    char* t_y;
    int t_cout, t_out_f32;
    float t_slope;
    long long t_y_bs, t_y_ps;
    // split types (YOLO_BF16X3, common.h): the K loop walks 3 * x3_n chunks [x_hi | x_lo | x_hi] of a pixel whose lo plane sits
    // x_lo bytes behind its hi plane: source byte offset of chunk c = c * 64 + (c >= x3_n ? x3_adj1 : 0) + (c >= 2 x3_n ? x3_adj2 : 0)
    int x3_n, x3_adj1, x3_adj2;
    long long y_lo, r_lo;   // ELEMENT offset of the lo plane in a pixel of y / of the residual
};

static inline void conv_args_fastdiv(ConvArgs& a) {
    a.d_PW = make_fastdiv(a.PW); a.d_H1 = make_fastdiv(a.H + 1); a.d_TWt = make_fastdiv(a.TWt);
    a.d_Ho = make_fastdiv(a.Ho); a.d_HoWo = make_fastdiv((unsigned)a.Ho * a.Wo); a.d_tc = make_fastdiv(a.tiles_c);
    a.d_tps = make_fastdiv(a.tiles_per_strip);
}

// `name` != nullptr: write the kernel instantiation that WOULD run (rocprofv3's demangled name) and
// do not launch.
struct NameOut { char* buf; int len; int* stats_rows; };     // stats_rows: receives the partial-row count (-1: the kernel has no statistics epilogue)

// Worst-case number of LDS slots of the zero-padded input halo tile for a BP-pixel output tile on
// strips of width d (stride S; KS = 3: 3x3, pad 1; KS = 2: 2x2 window anchored at the output pixel, zero row / column
// after the last).
static inline int conv_halo_slots(int BP, int d, int Ho, int H, int S, long long total_rows, int KS = 3, int pitch_extra = 0) {
    long long nro = (BP - 1 + d - 1) / d + 1;
    if (nro > total_rows) nro = total_rows;
    const long long nb = (nro - 1 + Ho - 1) / Ho;
    const int extra = H + 1 - Ho * S;
    const long long NR = (long long)S * (nro - 1) + KS + nb * (extra > 0 ? extra : 0);
    const int PW = (d - 1) * S + KS + pitch_extra;
    return (int)(NR * PW);
}

// conv_pipe.hip: pipelined variants (algo >= 2); YOLO_EUNSUPPORTED if the shape is not eligible.
int conv_pipe_dispatch(ConvArgs& a, int ks, int stride, int dtype, int algo, hipStream_t st, const NameOut* nm);
// conv_sk.hip: 1x1 with the K range split over the waves of one block (algo 30-35); YOLO_EUNSUPPORTED if not eligible.
int conv_sk_dispatch(ConvArgs& a, int ks, int stride, int dtype, int algo, hipStream_t st, const NameOut* nm);
// conv_stream.hip: weights-stationary streaming kernel for the small-channel layers (algo 13 / 14).
int conv_stream_dispatch(const ConvArgs& a, int ks, int stride, int dtype, int algo, hipStream_t st, const NameOut* nm);
