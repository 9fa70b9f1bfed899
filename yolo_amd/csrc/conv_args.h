// Kernel argument block and host helpers shared by the generic (conv_igemm.hip) and pipelined
// (conv_pipe.hip) implicit-GEMM convolutions.
#pragma once
#include "common.h"

struct ConvArgs {
    const char* x;
    const char* wp;
    const float* scale;
    const float* bias;
    const char* res;
    char* y;
    int N, H, W, Cin, Ho, Wo, Cout, Cout_pad;
    int TWt, nstrips, tiles_per_strip, PW, total_i;
    int nchunks, tiles_c;
    int out_f32;
    float slope;
    long long y_bs, y_ps;
    int dbg;            // experiment knobs (YOLO_DBG env, 0 in production): see conv_pipe.hip
};

// `name` != nullptr: write the kernel instantiation that WOULD run (rocprofv3's demangled name) and
// do not launch.
struct NameOut { char* buf; int len; };

// Worst-case number of LDS slots of the zero-padded input halo tile for a BP-pixel output tile on
// strips of width d (stride S, 3x3, pad 1).
static inline int conv_halo_slots(int BP, int d, int Ho, int H, int S, long long total_rows) {
    long long nro = (BP - 1 + d - 1) / d + 1;
    if (nro > total_rows) nro = total_rows;
    const long long nb = (nro - 1 + Ho - 1) / Ho;
    const int extra = H + 1 - Ho * S;
    const long long NR = (long long)S * (nro - 1) + 3 + nb * (extra > 0 ? extra : 0);
    const int PW = (d - 1) * S + 3;
    return (int)(NR * PW);
}

// conv_pipe.hip: pipelined variants (algo >= 2); YOLO_EUNSUPPORTED if the shape is not eligible.
int conv_pipe_dispatch(ConvArgs& a, int ks, int stride, int dtype, int algo, hipStream_t st, const NameOut* nm);
