// bf16 weight gradient of the 3x3 stride-1 layers (Cin, Cout multiples of 64) as a PIPELINED row walk for gfx950:
//   dW[tap][co][ci] += sum over pixels of dy[p][co] * x[p @ tap][ci]          (Convolution backward w.r.t. weights,
//   car/YOLO.py:393 `sum(losses).backward()` through every _conv2d of basic_yolo.py:20-26,118-121).
// MFMA 32x32x16 bf16 with K = output pixels; both operands are [pixel][channel] in HBM, so the fragments come from
// the transposing LDS read ds_read_b64_tr_b16 (tools/probes/tr_b16_probe.hip).
//
// What the older kernels of train.hip leave on the table (wgrad_rows_kernel: ~600 TFLOP/s on the 26x26 / 13x13 maps,
// the per-tap wgrad_bf16_kernel: less on the 52x52 ones): they stage through registers behind two __syncthreads per
// chunk, and they read NINE shifted x fragments from LDS for every dy fragment (ten fragment reads per nine MFMAs;
// ds_read_b64 needs ~4 waves per SIMD to reach its rate, these kernels hold two).  Here:
//   * the images are one stack of padded rows R = n*(H+1) + y + 1 (row 0 mod H+1 is the zero row two images share,
//     as in the forward kernels' padded-strip scheme); a WALKER owns SC columns and walks down a range of stacked
//     rows.  A K-step (16 pixels) is the current row of 16/SC walkers that share the columns and differ in the row
//     range, so narrow maps lose nothing to K padding with SC = 4 (52 = 13 x 4, 26 -> 28; 16 / SC = 1 walker of 16
//     columns for the wide or 13-wide maps);
//   * the x fragments of the three kernel rows are kept in REGISTERS across steps: output row R needs x rows R-1,
//     R, R+1, of which only R+1 is new -- four fragment reads (1 dy + 3 kw shifts of the new x row) per nine MFMAs;
//   * a block (4 waves = 2 x 2 over 64 cout x 64 cin, 9 x 32 x 32 accumulators per wave) runs phases of three
//     steps; the rows of a phase (3 new x rows with their column halo, 3 dy rows: 16 KiB) arrive by LDS-DMA
//     (global_load_lds_dwordx4) in a ring of RD slots, RD - 1 phases ahead, behind ONE counted s_waitcnt vmcnt +
//     s_barrier per phase; padding, image edges and range ends are DMA'd from a zero page (no predication in the
//     loop).  Lane-linear DMA images: the bank-conflict swizzle (64-byte halves of a pixel's 128 bytes exchanged
//     on odd 256-byte lines) is applied to the per-lane SOURCE address and again by the reader;
//   * blocks that share the rows (the cout x cin tiles of one walker set) are consecutive on one XCD.
// Partial sums of the blocks are added atomically straight into the OIHW fp32 gradient (each wave transposes its
// accumulators through LDS first: 256 contiguous bytes per atomic instruction, no workspace, no finishing pass).
#include "common.h"
#include "conv_args.h"
#include <type_traits>
#include <stdlib.h>

namespace {
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __attribute__((aligned(64))) unsigned int wgw_zero_page[16];

struct WalkArgs {
    const char* dy;
    const char* x;
    const char* zero;              // 64 zero bytes (padding, image edges, range ends are DMA'd from here)
    float* dwt;
    int H, W, Cin, Cout;
    unsigned x_rowb, dy_rowb;      // bytes per image row of x / dy
    unsigned x_pixb, dy_pixb;      // bytes per pixel
    int tiles_ci, ntiles, ncolseg;
    int L;                         // stacked rows per walker (a multiple of 3)
    int NR;                        // N * (H + 1) stacked rows
    int nphase;                    // 1 (fragment pre-load) + L / 3
    int dbg;                       // timing ablations (YOLO_WW_DBG): 1 = no epilogue atomics, 2 = DMAs from the zero page only
    FastDiv d_h1, d_tiles, d_colseg;
};

__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst_uniform) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ int xcd_order(int bid, int nblk) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ uint2 tr_read(const char* p) {
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p));
}
__device__ __forceinline__ void mma(f32x16& c, const uint4& a, const uint4& b) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
}  // namespace

// SC: columns per walker (16 / SC walkers per block); RD: ring depth; KH: K-halves per block -- KH = 2: 8 waves, waves 4-7
// run the same tile over the NEXT row slice in their own ring and hand their accumulators to waves 0-3 through LDS before
// the atomics (half the atomic volume at the same two waves per SIMD)
template <int SC, int RD, int KH = 1>
__global__ __launch_bounds__(256 * KH, 2 / KH) void wgrad_walk_kernel(WalkArgs a) {
    constexpr int NW = 16 / SC;
    constexpr int XPX = SC + 2;                         // x pixels per walker row (column halo)
    constexpr int WXB = XPX * 128;                      // bytes of a walker's x row (64 channels)
    constexpr int XROWB = (NW * WXB + 511) / 512 * 512;  // (whole 512-byte line pairs: the swizzle does not depend on the row)
    constexpr int SLOT_X = 3 * XROWB;
    constexpr int SLOT_XP = (SLOT_X + 1023) / 1024 * 1024;       // (whole DMAs: a DMA is x or dy, never both)
    constexpr int DYROWB = 16 * 128;
    constexpr int SLOT = 16384;
    constexpr int ND = SLOT / 4096;                     // DMAs per wave per phase
    static_assert(SLOT_XP + 3 * DYROWB <= SLOT, "slot layout");
    static_assert(XROWB % 512 == 0 && SLOT_XP % 512 == 0 && DYROWB % 512 == 0 && SLOT % 512 == 0, "the swizzle follows 256-byte lines");
    __shared__ __attribute__((aligned(1024))) char smem_all[KH * RD * SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave8 >> 2, wave = wave8 & 3;
    char* smem = smem_all + half * RD * SLOT;           // this half's ring
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)smem;
    const int wm = wave & 1, wn = wave >> 1;

    const int lid = xcd_order(blockIdx.x, gridDim.x);
    const int rest = fdiv(lid, a.d_tiles);
    const int tile = lid - rest * a.ntiles;
    const int bslice = fdiv(rest, a.d_colseg);
    const int cseg = rest - bslice * a.ncolseg;
    const int slice = bslice * KH + half;
    const int tco = tile / a.tiles_ci, tci = tile - tco * a.tiles_ci;
    const int co0 = tco * 64, ci0 = tci * 64;
    const int c0 = cseg * SC;
    const int L = a.L, NR = a.NR, H1 = a.H + 1;
    const int Ra0 = slice * NW * L;                     // first stacked row of walker 0

    // ---- per-lane DMA descriptors.  DMA k of this wave fills slot bytes [(k*4 + wave)*1024 + lane*16, +16) -------
    //   code = row-in-phase | walker << 2 ; co = byte offset of the 16-byte unit inside its image row (~0: padding column)
    unsigned d_co[ND];
    int d_r0[ND];                                       // stacked row of the unit in slot-phase 0
    int d_rl[ND];                                       // rows >= this are not loaded (dy: the walker's range end)
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        const int o = (k * 4 + wave) * 1024 + lane * 16;
        const int par = (o >> 8) & 1;
        const int u = ((o >> 4) & 7) ^ (par << 2);      // the channel unit this LDS unit must hold
        int i, w, col;
        unsigned co = 0xffffffffu;
        if (o < SLOT_XP) {
            i = o / XROWB;
            const int rem = o - i * XROWB;
            w = rem / WXB;
            const int px = (rem - w * WXB) >> 7;
            col = c0 - 1 + px;
            if (o < SLOT_X && rem < NW * WXB && col >= 0 && col < a.W) co = (unsigned)col * a.x_pixb + (unsigned)(ci0 * 2 + u * 16);
            d_r0[k] = Ra0 + w * L + i - 2;              // x rows of slot-phase q: s + 1 + i, s = Ra + 3 (q - 1)
            d_rl[k] = NR;
        } else {
            const int o2 = o - SLOT_XP;
            i = o2 / DYROWB;
            const int kk = (o2 - i * DYROWB) >> 7;
            w = kk / SC;
            col = c0 + kk - w * SC;
            if (i < 3 && col < a.W) co = (unsigned)col * a.dy_pixb + (unsigned)(co0 * 2 + u * 16);
            d_r0[k] = Ra0 + w * L + i - 3;              // dy rows of slot-phase q: s + i
            d_rl[k] = min(Ra0 + (w + 1) * L, NR);
        }
        d_co[k] = co;
    }
    const uint32_t wave_lds = lds0 + wave * 1024;
    auto issue = [&](int k, int qs) {                   // DMA k of slot-phase qs into ring slot qs % RD
        const bool isx = (k * 4 + wave) * 1024 < SLOT_XP;                   // (wave-uniform)
        const int R = d_r0[k] + 3 * qs;
        const int n = fdiv(R, a.d_h1);
        const int y = R - n * H1 - 1;
        const bool ok = (unsigned)R < (unsigned)d_rl[k] && y >= 0 && d_co[k] != 0xffffffffu;
        const unsigned rowb = isx ? a.x_rowb : a.dy_rowb;
        const char* base = isx ? a.x : a.dy;
        const char* src = base + ((size_t)(unsigned)(R - n - 1) * rowb + d_co[k]);
        src = (ok && !(a.dbg & 2)) ? src : a.zero;
        dma16(src, wave_lds + (qs % RD) * SLOT + k * 4096);
    };

    // ---- prologue: slot-phases 0 .. RD-2 ---------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < RD - 1; ++q)
#pragma unroll
        for (int k = 0; k < ND; ++k) issue(k, q);

    // ---- per-lane fragment addresses (bytes inside a slot; row i of the phase adds i * XROWB / i * DYROWB) -----
    const int g = lane >> 4, j16 = lane & 15;
    int xa[3][2], da[2];
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) {
        const int kk = (g >> 1) * 8 + hl * 4 + (j16 >> 2);          // K index = pixel of the K-step
        const int w = kk / SC, pc = kk - w * SC;
        const int sub = (g & 1) * 2 + ((j16 & 3) >> 1), in8 = (j16 & 1) * 8;
        {
            const int pb = SLOT_XP + kk * 128;
            da[hl] = pb + (((wm * 4 + sub) ^ (((pb >> 8) & 1) << 2)) << 4) + in8;
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int pb = (w * XPX + pc + kw) * 128;
            xa[kw][hl] = pb + (((wn * 4 + sub) ^ (((pb >> 8) & 1) << 2)) << 4) + in8;
        }
    }

    f32x16 acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;
    uint4 X[3][3];                                      // [x row mod 3][kw]
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) X[s_][kw] = make_uint4(0, 0, 0, 0);

    auto xfrag = [&](const char* sl, int i, int kw) {
        const uint2 lo = tr_read(sl + i * XROWB + xa[kw][0]), hi = tr_read(sl + i * XROWB + xa[kw][1]);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    };
    auto dfrag = [&](const char* sl, int i) {
        const uint2 lo = tr_read(sl + i * DYROWB + da[0]), hi = tr_read(sl + i * DYROWB + da[1]);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    };

    wait_vm<(RD - 2) * ND>();
    __builtin_amdgcn_s_barrier();

    // ---- phase 0: the two x rows above the walkers' first output row go into the fragment registers -------------
    {
#pragma unroll
        for (int k = 0; k < ND; ++k) issue(k, RD - 1);
#pragma unroll
        for (int i = 1; i < 3; ++i)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) X[(i + 1) % 3][kw] = xfrag(smem, i, kw);
        wait_vm<(RD - 2) * ND>();
        __builtin_amdgcn_s_barrier();
    }

    // ---- main loop ---------------------------------------------------------------------------------------------
    const int nphase = a.nphase;
    int ring = 1 % RD;
    for (int q = 1; q < nphase; ++q) {
        const char* sl = smem + ring * SLOT;
        uint4 A = dfrag(sl, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            uint4 Xn[3];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) Xn[kw] = xfrag(sl, i, kw);
            uint4 An = A;
            if (i < 2) An = dfrag(sl, i + 1);
            // kernel rows 0 and 1 from the registers (x rows R-1, R), row 2 from the row just read (R+1)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) mma(acc[kw], A, X[(i + 2) % 3][kw]);
            __builtin_amdgcn_sched_barrier(0);
            issue(i, q + RD - 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) mma(acc[3 + kw], A, X[i % 3][kw]);
            if (i == 1) {
                __builtin_amdgcn_sched_barrier(0);
                issue(3, q + RD - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                X[(i + 1) % 3][kw] = Xn[kw];
                mma(acc[6 + kw], A, Xn[kw]);
            }
            A = An;
        }
        static_assert(ND == 4, "issue points above");
        wait_vm<(RD - 2) * ND>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        ring = (ring + 1 == RD) ? 0 : ring + 1;
    }
    wait_vm<0>();                                       // (the tail's dead DMAs must land before the LDS is re-used / released)

    // ---- epilogue: straight into the OIHW gradient.  A lane holds 16 cout x 1 cin x 9 taps; dw[co][ci][kh][kw] wants the
    //      (cin, tap) pairs of one cout contiguous (288 floats per wave), so every wave transposes its tile through its
    //      own 9 KiB of the (now idle) ring in four passes of 8 cout rows and adds 256 contiguous bytes per instruction.
    __builtin_amdgcn_s_barrier();                       // every wave is done reading the ring
    if (a.dbg & 1) return;
    if constexpr (KH == 2) {
        // waves 4-7 park their accumulators in LDS (lane-linear, 256 B per register; taps 0-4 then 5-8: 80 / 64 KiB), waves
        // 0-3 add them to their own and go on alone
        float* park = (float*)smem_all + wave * (5 * 16 * 64);
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const int t0 = part * 5, t1 = part ? 9 : 5;
            if (half == 1) {
#pragma unroll
                for (int tp = t0; tp < t1; ++tp)
#pragma unroll
                    for (int r = 0; r < 16; ++r) park[((tp - t0) * 16 + r) * 64 + lane] = acc[tp][r];
            }
            __syncthreads();
            if (half == 0) {
#pragma unroll
                for (int tp = t0; tp < t1; ++tp)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tp][r] += park[((tp - t0) * 16 + r) * 64 + lane];
            }
            __syncthreads();
        }
        if (half == 1) return;
    }
    const int l31 = lane & 31, h = lane >> 5;
    float* scr = (float*)(smem_all + wave * 9216);
    float* drow = a.dwt + ((long long)(co0 + wm * 32) * a.Cin + (ci0 + wn * 32)) * 9;
    const long long co_pitch = (long long)a.Cin * 9;
    const int rot = (a.dbg & 4) ? 0 : (int)((unsigned)(slice * 7 + cseg * 11) % 36u);
    // (round 5) The walk over the scratch -- element idx = tt * 64 + lane of [8 rows][288], tt = rot, rot + 1, ... mod 36 -- carries
    // its row / remainder / gradient offset along instead of dividing per atomic (the rotation makes tt a run-time value: the
    // compiler spent ~14 vector instructions per atomic on idx / 288, the row's cout and a 64-bit multiply by the cout pitch; the
    // epilogue is ~20 % of the kernel and issues nothing else).  Row r of pass p is cout 8 p + r: one pitch per carry.
    const int idx0 = rot * 64 + lane;
    const int row0 = idx0 / 288;
    const int rem0 = idx0 - row0 * 288;
    const int pitch = (int)co_pitch;                    // (elements; the host bounds Cout * Cin * 9 below 2^31)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) scr[((j + 4 * h) * 32 + l31) * 9 + tp] = acc[tp][4 * p + j];
        int idx = idx0, row = row0, rem = rem0;
        int off = (8 * p + row0) * pitch + rem0;        // element offset of (cout 8 p + row, rem) from drow
#pragma unroll
        for (int t = 0; t < 36; ++t) {
            // (the K-slices of a tile finish together: each starts its pass at another row, so that they do not queue on
            //  the same addresses of the memory-side atomic units)
            atomicAdd(drow + off, scr[idx]);
            idx += 64; rem += 64; off += 64;
            if (rem >= 288) { rem -= 288; row += 1; off += pitch - 288; }
            if (row == 8) { row = 0; idx -= 2304; off -= 8 * pitch; }
        }
    }
}

template <int SC, int RD, int KH = 1>
static int wgrad_walk_launch(const void* dy, const void* x, float* dwt, int N, int H, int W, int Cin, int Cout, long long ps,
                             int target_blocks, hipStream_t st) {
    constexpr int NW = 16 / SC * KH;                    // walkers (row ranges) per block
    target_blocks /= KH;
    WalkArgs a;
    a.dy = (const char*)dy; a.x = (const char*)x; a.dwt = dwt;
    void* zp = nullptr;
    if (hipGetSymbolAddress(&zp, HIP_SYMBOL(wgw_zero_page)) != hipSuccess || !zp) return YOLO_EINVAL;
    a.zero = (const char*)zp;
    a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.x_pixb = (unsigned)Cin * 2; a.dy_pixb = (unsigned)ps * 2;
    a.x_rowb = (unsigned)W * a.x_pixb; a.dy_rowb = (unsigned)W * a.dy_pixb;
    a.tiles_ci = Cin / 64;
    a.ntiles = a.tiles_ci * (Cout / 64);
    a.ncolseg = (W + SC - 1) / SC;
    a.NR = N * (H + 1);
    const long long per_slice = (long long)a.ntiles * a.ncolseg;
    long long slices = target_blocks / per_slice;            // rounded DOWN: a few blocks over one resident round cost a second one
    if (slices < 1) slices = 1;
    // rows per walker: a multiple of 3, at least 6 phases of work per block
    long long Lw = (a.NR + slices * NW - 1) / (slices * NW);
    Lw = (Lw + 2) / 3 * 3;
    if (Lw < 18) Lw = 18;
    slices = (a.NR + Lw * NW - 1) / (Lw * NW);               // (never more than before: Lw was rounded up)
    a.L = (int)Lw;
    a.nphase = 1 + a.L / 3;
    static const int dbg = (int)YOLO_LAB_ENV("YOLO_WW_DBG", 0);
    a.dbg = dbg;
    const long long grid = per_slice * slices;
    if (grid > 0x7fffffffLL) return YOLO_EUNSUPPORTED;
    a.d_h1 = make_fastdiv((unsigned)(H + 1));
    a.d_tiles = make_fastdiv((unsigned)a.ntiles);
    a.d_colseg = make_fastdiv((unsigned)a.ncolseg);
    YOLO_LAUNCH((wgrad_walk_kernel<SC, RD, KH>), dim3((unsigned)grid), dim3(256 * KH), 0, st, a);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// 1x1 weight gradient: dW[co][ci] += sum over pixels of dy[p][co] * x[p][ci] -- a plain GEMM with K = pixels, both operands
// K-strided.  Same machinery as the row walk (LDS-DMA ring, counted waits, one barrier per phase, transposing fragment
// reads, zero page past the slice end) without its register re-use: a 1x1 moves 2-3x the bytes per MFMA.  Block = 4 waves
// (2 x 2), wave tile (MI*32) cout x (NI*32) cin, phase = 32 pixels.  LDS rows are one pixel of the tile (256 or 512
// bytes): the 64-byte chunk c of pixel p sits at chunk c ^ (p & 3), so the four pixel rows of a transposing read fall on
// four different bank quarters.  [cout][cin] IS the OIHW layout of a 1x1: the accumulators are added straight into the
// gradient.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct GemmArgs {
    const char* dy;
    const char* x;
    const char* zero;
    float* dw;
    int Cin, Cout;
    unsigned x_pixb, dy_pixb;
    int tiles_ci, ntiles;
    long long P;                   // pixels
    int Ls;                        // pixels per K-slice (a multiple of 32)
    int nphase;                    // Ls / 32
    FastDiv d_tiles;
};
}  // namespace

template <int MI, int NI, int RD>
__global__ __launch_bounds__(256, (MI * NI > 4) ? 2 : 3) void wgrad_gemm_kernel(GemmArgs a) {
    constexpr int KP = 32;
    constexpr int BM = 2 * MI * 32, BN = 2 * NI * 32;
    constexpr int PBD = BM * 2, PBX = BN * 2;           // LDS row bytes (one pixel of the tile)
    constexpr int SLOT_D = KP * PBD, SLOT = KP * (PBD + PBX);
    constexpr int ND = SLOT / 4096;                     // DMAs per wave per phase
    static_assert(SLOT % 4096 == 0 && SLOT_D % 1024 == 0 && PBD >= 256 && PBX >= 256, "slot layout");
    __shared__ __attribute__((aligned(1024))) char smem[RD * SLOT];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int lid = xcd_order(blockIdx.x, gridDim.x);
    const int slice = fdiv(lid, a.d_tiles);
    const int tile = lid - slice * a.ntiles;
    const int tco = tile / a.tiles_ci, tci = tile - tco * a.tiles_ci;
    const int co0 = tco * BM, ci0 = tci * BN;
    const long long p0 = (long long)slice * a.Ls;
    const long long p1 = min(p0 + a.Ls, a.P);

    // ---- per-lane DMA sources (advance by one phase = 32 pixels per issue) --------------------------------------------
    const char* d_src[ND];
    int d_left[ND];                                     // phases for which DMA k still loads real pixels (then: the zero page)
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        const int o = (k * 4 + wave) * 1024 + lane * 16;
        const bool isd = (k * 4 + wave) * 1024 < SLOT_D;           // (wave-uniform: a DMA is dy or x, never both)
        const int oo = isd ? o : o - SLOT_D;
        const int pb = isd ? PBD : PBX;
        const int row = oo / pb, unit = (oo - row * pb) >> 4;
        const int chunk = (unit >> 2) ^ (row & 3);
        const unsigned pixb = isd ? a.dy_pixb : a.x_pixb;
        const long long left = p1 - p0 - row;
        d_left[k] = left > 0 ? (int)((left + KP - 1) / KP) : 0;
        d_src[k] = (isd ? a.dy + (size_t)co0 * 2 : a.x + (size_t)ci0 * 2) + (size_t)(p0 + row) * pixb + chunk * 64 + (unit & 3) * 16;
    }
    const uint32_t wave_lds = lds0 + wave * 1024;
    auto issue = [&](int k, int ring) {
        const bool isd = (k * 4 + wave) * 1024 < SLOT_D;
        const char* src = d_left[k] > 0 ? d_src[k] : a.zero;
        dma16(src, wave_lds + ring * SLOT + k * 4096);
        d_src[k] += (isd ? a.dy_pixb : a.x_pixb) * KP;
        d_left[k] -= 1;
    };
#pragma unroll
    for (int q = 0; q < RD - 1; ++q)
#pragma unroll
        for (int k = 0; k < ND; ++k) issue(k, q);

    // ---- per-lane fragment offsets: pixel row k of a K-step = (g >> 1) * 8 + hl * 4 + (j16 >> 2), so k & 3 = j16 >> 2 ------
    const int g = lane >> 4, j16 = lane & 15;
    const int sw = j16 >> 2;
    const int within = (g & 1) * 32 + (j16 & 3) * 8;
    int ao[MI], bo[NI], krow[2];
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) krow[hl] = (g >> 1) * 8 + hl * 4 + (j16 >> 2);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) ao[mi] = (((wm * MI + mi) ^ sw) << 6) + within;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bo[ni] = SLOT_D + (((wn * NI + ni) ^ sw) << 6) + within;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    wait_vm<(RD - 2) * ND>();
    __builtin_amdgcn_s_barrier();
    const int nphase = a.nphase;
    int ring = 0;
    for (int q = 0; q < nphase; ++q) {
        const char* sl = smem + ring * SLOT;
        const int wr = ring == 0 ? RD - 1 : ring - 1;   // the slot the previous phase read
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 af[MI], bf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const uint2 lo = tr_read(sl + (ks * 16 + krow[0]) * PBD + ao[mi]), hi = tr_read(sl + (ks * 16 + krow[1]) * PBD + ao[mi]);
                af[mi] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const uint2 lo = tr_read(sl + (ks * 16 + krow[0]) * PBX + bo[ni]), hi = tr_read(sl + (ks * 16 + krow[1]) * PBX + bo[ni]);
                bf[ni] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    mma(acc[mi][ni], af[mi], bf[ni]);
                    // the DMAs of the phase RD - 1 ahead, spread between the MFMAs
                    constexpr int NM = 2 * MI * NI;
                    const int m = (ks * MI + mi) * NI + ni;
#pragma unroll
                    for (int k = 0; k < ND; ++k)
                        if (m == (k * NM) / ND) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue(k, wr);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
        }
        wait_vm<(RD - 2) * ND>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        ring = (ring + 1 == RD) ? 0 : ring + 1;
    }
    wait_vm<0>();
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int ci = ci0 + (wn * NI + ni) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                atomicAdd(a.dw + (long long)co * a.Cin + ci, acc[mi][ni][r]);
            }
        }
}

template <int MI, int NI>
static int wgrad_gemm_launch(const void* dy, const void* x, float* dw, long long P, int Cin, int Cout, long long ps, hipStream_t st) {
    constexpr int BM = 2 * MI * 32, BN = 2 * NI * 32, RD = 3;    // (3 slots: 48 / 72 KiB per block -> 3 / 2 blocks per CU)
    constexpr int SLOT = 32 * (BM + BN) * 2;
    GemmArgs a;
    a.dy = (const char*)dy; a.x = (const char*)x; a.dw = dw;
    void* zp = nullptr;
    if (hipGetSymbolAddress(&zp, HIP_SYMBOL(wgw_zero_page)) != hipSuccess || !zp) return YOLO_EINVAL;
    a.zero = (const char*)zp;
    a.Cin = Cin; a.Cout = Cout; a.x_pixb = (unsigned)Cin * 2; a.dy_pixb = (unsigned)ps * 2;
    a.tiles_ci = Cin / BN;
    a.ntiles = a.tiles_ci * (Cout / BM);
    a.P = P;
    // K-slices: ONE block per CU's worth (256).  Every block ends with its 64 KiB tile of atomics onto a gradient of a
    // few MiB: at 768 blocks (what fits) the 11-GFLOP layers took 46-50 us, at 256 they take 29-32, the 45-GFLOP ones 67-72
    // either way (YOLO_WG_SLOTS: the knob of that sweep)
    static const int slots_env = (int)YOLO_LAB_ENV("YOLO_WG_SLOTS", 0);
    const int resident = slots_env ? slots_env : 256;
    long long slices = resident / a.ntiles;
    if (slices < 1) slices = 1;
    long long Ls = (P + slices - 1) / slices;
    Ls = (Ls + 31) / 32 * 32;
    if (Ls < 256) Ls = 256;                                      // at least 8 phases per block
    slices = (P + Ls - 1) / Ls;
    a.Ls = (int)Ls;
    a.nphase = (int)(Ls / 32);
    a.d_tiles = make_fastdiv((unsigned)a.ntiles);
    const long long grid = (long long)a.ntiles * slices;
    if (grid > 0x7fffffffLL) return YOLO_EUNSUPPORTED;
    YOLO_LAUNCH((wgrad_gemm_kernel<MI, NI, RD>), dim3((unsigned)grid), dim3(256), 0, st, a);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
}

// variant: 0 = by the channel counts; 1 = 128 x 128 tile; 2 = 256 cout x 128 cin.  EUNSUPPORTED outside the domain.
int wgrad_gemm_dispatch(const void* dy, const void* x, float* dw, long long P, int Cin, int Cout, long long ps, int variant,
                        hipStream_t st) {
    if ((Cin % 128) || (Cout % 128) || (ps % 8) || P < 64) return YOLO_EUNSUPPORTED;
    if (P * Cin * 2 >= 0x7fffff00LL * 2 || P * ps * 2 >= 0x7fffff00LL * 2) return YOLO_EUNSUPPORTED;
    // measured (bs 64) against the register-staged per-tap kernel: 45-GFLOP head layers 96-125 -> 64-72 us, 11-GFLOP backbone
    // layers 46-49 -> 29-32 us; the 256 x 128 tile loses everywhere (one block per SIMD pair) -- so: the 128 x 128 tile
    if (variant == 0) variant = 1;
    if (variant == 2 && (Cout % 256)) return YOLO_EUNSUPPORTED;
    if (variant == 1) return wgrad_gemm_launch<2, 2>(dy, x, dw, P, Cin, Cout, ps, st);
    if (variant == 2) return wgrad_gemm_launch<4, 2>(dy, x, dw, P, Cin, Cout, ps, st);
    return YOLO_EUNSUPPORTED;
}

// variant: 0 = the default; 1 = one 16-column walker; 2 = four 4-column walkers; 3 = 2, with 8-wave blocks of two K-halves.  EUNSUPPORTED when the
// shape is outside the kernel's domain (the caller falls back on the kernels of train.hip).
int wgrad_walk_dispatch(const void* dy, const void* x, float* dwt, int N, int H, int W, int Cin, int Cout, long long ps,
                        int variant, hipStream_t st) {
    if ((Cin % 64) || (Cout % 64) || (ps % 8) || W < 4 || H < 2) return YOLO_EUNSUPPORTED;
    if ((long long)N * H * W * Cin * 2 >= 0xffffff00LL || (long long)N * H * W * ps * 2 >= 0xffffff00LL) return YOLO_EUNSUPPORTED;
    if ((long long)N * (H + 1) >= 0x3fffffffLL) return YOLO_EUNSUPPORTED;
    // four 4-column walkers win on the wider maps of the 416 / 608 families (19 ... 152): no K padding beyond W % 4, and
    // 40 instead of 34 staged pixels per K-step row cost less than the 16-column walker's padding
    if (variant == 0) {
        // (maps of at most 16 columns: one 16-column walker covers a row with the same K padding as four narrow ones and a
        //  quarter of the blocks -- 13x13, 1024 -> 2048: 439 -> 390 us, 512 -> 1024: a tie)
        static const int sc = (int)YOLO_LAB_ENV("YOLO_WW_SC", 0);     // (A/B knob)
        variant = sc ? (sc == 16 ? 1 : 2) : (W <= 16 ? 1 : 2);
    }
    // (ablation knobs, read once: YOLO_WW_TARGET = blocks per launch, YOLO_WW_RD = ring depth, YOLO_WW_DBG see WalkArgs)
    static const int target = (int)YOLO_LAB_ENV("YOLO_WW_TARGET", 512);    // two blocks per CU
    static const int rd = (int)YOLO_LAB_ENV("YOLO_WW_RD", 4);
#define WW_CASE(SC_, RD_) if (variant == (SC_ == 16 ? 1 : 2) && rd == RD_) return wgrad_walk_launch<SC_, RD_>(dy, x, dwt, N, H, W, Cin, Cout, ps, target, st);
    WW_CASE(16, 3) WW_CASE(16, 4) WW_CASE(16, 5) WW_CASE(4, 3) WW_CASE(4, 4) WW_CASE(4, 5)
#undef WW_CASE
    if (variant == 3) return wgrad_walk_launch<4, 4, 2>(dy, x, dwt, N, H, W, Cin, Cout, ps, target, st);   // two K-halves per block
    return YOLO_EUNSUPPORTED;
}
