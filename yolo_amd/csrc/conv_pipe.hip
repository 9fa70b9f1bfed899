// Pipelined implicit-GEMM convolution for gfx950: the fast path for 3x3 (stride 1 and 2), 1x1 and the 2x2-window
// form of the stride-2 data gradient (yolo_conv_dgrad_s2), for layers whose K-chunks are whole
// (Cin*sizeof(T) % 64 == 0).  Same math, data layout, packed-weight image and epilogue as conv_igemm.hip (the
// generic path); what differs is the memory pipeline:
//
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR
//     staging, no ds_write pass).  The LDS image is lane-linear, so the bank-conflict XOR swizzle
//     is applied to the per-lane SOURCE address (weights: at pack time) and again on the read.
//   * weights stream through a ring of 3-4 slots (4 wherever the extra slot does not cost a co-resident block; the
//     1x1 variants may ask for more), one ring slot per "phase" (one tap of a 3x3 / 2x2, or KC K-chunks of a 1x1);
//     the zero-padded input halo tile is double-buffered per K-chunk and read by all taps.  Loads run ring-depth - 1
//     phases ahead, interleaved between the MFMAs of a phase; the only wait is a COUNTED s_waitcnt vmcnt(N)
//     (N = loads of the phases still allowed in flight) in front of ONE raw s_barrier per phase, so DMA stays in
//     flight across barriers.  Zero padding / image edges are DMA'd from a zero page: no predication anywhere in
//     the main loop.
//   * XCD-aware block order: the blocks that share an input tile (different cout tiles) are
//     consecutive on one XCD so its L2 serves the re-reads.
#ifndef YOLO_PIPE_PART
#define YOLO_PIPE_PART 0
#endif
#include "common.h"
#include "conv_args.h"
#include "stamp.h"
#include "conv_epilogue.h"
#include <stdio.h>
#include <type_traits>
#include <atomic>

// The file is compiled FOUR times (csrc/Makefile), kernel kind x element type, so that the ~270 instantiations build in parallel:
//   YOLO_PIPE_PART 0: 3x3 kernels, bf16 + conv_pipe_dispatch     1: 2x2-window and 1x1 kernels, bf16 (conv_pipe_dispatch_b)
//                  2: 3x3 kernels, f16 and f32 (.._dispatch_c)   3: 1x1 kernels, f16 and f32 (conv_pipe_dispatch_d)
//                  4: 3x3 stride-2 kernels, bf16 (conv_pipe_dispatch_e)
//                  5: 3x3 stride-1 kernels, bf16, the second half of the tile variants (conv_pipe_dispatch_f)
//                  6: 3x3 kernels, split bf16 (YOLO_BF16X3; conv_pipe_dispatch_g)      7: 1x1 kernels, split bf16 (conv_pipe_dispatch_h)
//                  8 / 9: the same for split f16 (YOLO_F16X3; conv_pipe_dispatch_i / _j)
#define YOLO_PIPE_3X3 (YOLO_PIPE_PART == 0 || YOLO_PIPE_PART == 2 || YOLO_PIPE_PART == 4 || YOLO_PIPE_PART == 5 || YOLO_PIPE_PART == 6 || YOLO_PIPE_PART == 8)
// (which stride-1 3x3 variants a unit holds: bf16 is split over units 0 and 5)
#define YOLO_PIPE_S1A (YOLO_PIPE_PART == 0 || YOLO_PIPE_PART == 2 || YOLO_PIPE_PART == 6 || YOLO_PIPE_PART == 8)
#define YOLO_PIPE_S1B (YOLO_PIPE_PART == 5 || YOLO_PIPE_PART == 2 || YOLO_PIPE_PART == 6 || YOLO_PIPE_PART == 8)
namespace { __device__ __attribute__((aligned(64))) unsigned int yolo_zero_page[16]; }

typedef __attribute__((address_space(3))) char lds_char;

// One 16-byte-per-lane LDS-DMA: LDS[lds_dst + lane*16 .. +16) = *gsrc (per-lane source).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst_uniform) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst_uniform)
                 : "memory");
}
// The same without saving / restoring m0 (nothing the compiler emits for gfx950 in these kernels reads m0 -- LDS instructions
// have not needed it since GFX9 -- and the generic form's save + restore is two of the four scalar instructions of every DMA):
// every K loop of this file since round 5 (the lean 1x1 loops since round 3); the fused tail keeps the saving form.
__device__ __forceinline__ void glds16_m0(const void* gsrc, uint32_t lds_dst_uniform) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The kernel's argument block read afresh from the kernarg segment (ConvArgs is the kernel's only argument: offset 0).  The
// optimiser cannot tell two such reads apart from two different structs, so the fields a region uses are s_load'ed where the
// region starts instead of at the top of the kernel -- and do not stay in SGPRs across the regions that do not use them.  The
// persistent tile loop keeps every tile-set-up scalar (seven division constants, the dimensions, the strides, six pointers)
// live across the epilogue otherwise: the ISA of the round-4 epilogue spent 70 of its 330 vector-pipe instructions per
// 32-pixel slab on v_readlane / v_writelane SGPR spill traffic (buffer descriptors rebuilt from spilled halves before every
// load and store, each followed by the s_nop its hazard needs).
__device__ __forceinline__ const ConvArgs& kargs_fresh() {
    const __attribute__((address_space(4))) ConvArgs* p = (const __attribute__((address_space(4))) ConvArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const ConvArgs*)p;
}

template <typename T> struct FragP;
template <> struct FragP<bf16_t> {
    static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                    0, 0);
    }
};
template <> struct FragP<f16_t> {
    static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct FragP<bf16x3_t> : FragP<bf16_t> {};
template <> struct FragP<f16x3_t> : FragP<f16_t> {};
template <> struct FragP<float> {
    static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

// bijective XCD remap (8 XCDs, block b runs on XCD b % 8): logical ids are contiguous per XCD
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// KS=3 (stride 1): phase = one tap; X halo tile of XSLOTS slots, double-buffered per K-chunk.
// KS=1           : phase = one K-chunk; X tile = BP slots, 3-deep ring like the weights.
// The LDS-DMAs of a phase are NOT issued in a burst after the barrier (that stalls both waves of a
// SIMD in the VMEM issue queue at the same moment and idles the matrix pipe: measured -25..-35 %):
// they are interleaved between the MFMAs, and the second half of the waves (which shares SIMDs with
// the first half) issues them at shifted positions, so a wave stuck in a DMA issue is covered by its
// SIMD partner's MFMAs.
template <typename T, int KS, int WAVES_P, int WAVES_C, int MI, int NI, int XSLOTS, int S = 1, int RD = 0, int KC = 1, int LEAN = 0, int STATS = 0>
__global__ __launch_bounds__(WAVES_P* WAVES_C * 64) void conv_pipe_kernel(ConvArgs a0) {
    constexpr int PT = 1;
    constexpr int NW = WAVES_P * WAVES_C;
    constexpr int NT = NW * 64;
    constexpr int BP = WAVES_P * NI * 32;
    constexpr int BC = WAVES_C * MI * 32;
    constexpr int NTAP = KS * KS;
    constexpr int PPC = NTAP / PT;                       // phases per K-chunk
    static_assert(NTAP % PT == 0 && (KS != 1 || PT == 1), "phase shape");
    // KC (1x1 only): K chunks of 32 bf16 channels per phase.  The small-map 1x1 layers spend ~600 cycles per
    // one-chunk phase on the fixed chain barrier -> LDS reads -> 4 MFMAs -> wait; two chunks per phase halve the
    // number of barriers for the same bytes in flight.
    static_assert(KC == 1 || KS == 1, "several K chunks per phase: 1x1 only");
    constexpr int W_STAGE1 = PT * BC * 64;               // bytes of one chunk's weight slab
    constexpr int W_STAGE = KC * W_STAGE1;               // bytes per weight ring slot
    constexpr int WL1 = W_STAGE1 / (NT * 16);
    constexpr int WL = KC * WL1;                         // LDS-DMAs per thread per phase (weights)
    static_assert(W_STAGE1 % (NT * 16) == 0, "weight stage must be whole DMAs");
    constexpr int X_STAGE1 = XSLOTS * 64;
    constexpr int X_STAGE = KC * X_STAGE1;
    constexpr int XL1 = X_STAGE1 / (NT * 16);
    constexpr int XL = KC * XL1;                         // LDS-DMAs per thread per X tile
    static_assert(X_STAGE1 % (NT * 16) == 0, "input stage must be whole DMAs");
    static_assert(S != 2 || XSLOTS % 8 == 0, "stride 2: slots are permuted in groups of 8");
    // 1x1: depth of the X and W rings (loads run R1-1 phases ahead): 4 where that keeps the blocks per CU, else 3
#ifndef YOLO_RING1
#define YOLO_RING1 4
#endif
    constexpr int EPI1_ = WAVES_P * WAVES_C * YOLO_EPI_WAVE_BYTES_MI(MI);
    constexpr int L13_ = 3 * KC * (XSLOTS * 64 + WAVES_C * MI * 32 * 64) > EPI1_ ? 3 * KC * (XSLOTS * 64 + WAVES_C * MI * 32 * 64) : EPI1_;
    constexpr int L1N_ = YOLO_RING1 * KC * (XSLOTS * 64 + WAVES_C * MI * 32 * 64) > EPI1_ ? YOLO_RING1 * KC * (XSLOTS * 64 + WAVES_C * MI * 32 * 64) : EPI1_;
    // RD != 0 (1x1 only): an explicit ring depth.  The small-map 1x1 layers are bound by the latency of their loads (a
    // 64 x 128 tile has 12 KB per phase: 32 phases of K = 1024 took ~1000 cycles each with 3 phases in flight), so
    // some variants trade co-resident blocks for a deeper ring.
    constexpr int R1 = RD ? RD : ((L1N_ <= 163840 && 163840 / L1N_ == 163840 / L13_) ? YOLO_RING1 : 3);
    static_assert(RD == 0 || KS == 1, "explicit ring depth: 1x1 only");
    constexpr int XBUFS = (KS != 1) ? 2 : R1;
    static_assert(KS != 1 || XSLOTS == BP, "1x1: one slot per pixel");
    constexpr int PAD = KS / 2 - (KS == 2 ? 1 : 0);          // 3x3: 1; 2x2 (sub-pixel data gradient) and 1x1: 0
    constexpr int W_OFF = XBUFS * X_STAGE;
    // 3x3: the X DMAs of the next chunk are spread over the phases of this chunk (phase q issues
    // j = q, q+PPC, ...)

    // 3x3 weight ring: 4 slots (loads run 3 phases ahead: K loop -4..-9 % cycles vs 3 slots, measured with the phase
    // stamps; 5 slots are slower again) wherever the extra slot does not cost a co-resident block per CU
#ifndef YOLO_WRING
#define YOLO_WRING 4
#endif
    constexpr int EPI_BYTES_ = WAVES_P * WAVES_C * YOLO_EPI_WAVE_BYTES_MI(MI);
    constexpr int LDS3 = XBUFS * X_STAGE + 3 * W_STAGE > EPI_BYTES_ ? XBUFS * X_STAGE + 3 * W_STAGE : EPI_BYTES_;
    constexpr int LDSN = XBUFS * X_STAGE + YOLO_WRING * W_STAGE > EPI_BYTES_ ? XBUFS * X_STAGE + YOLO_WRING * W_STAGE : EPI_BYTES_;
    constexpr int WR = (KS != 1) ? ((LDSN <= 163840 && 163840 / LDSN == 163840 / LDS3) ? YOLO_WRING : 3) : R1;
    constexpr int PIPE_BYTES = XBUFS * X_STAGE + WR * W_STAGE;
    constexpr int EPI_BYTES = NW * YOLO_EPI_WAVE_BYTES_MI(MI);
    __shared__ __attribute__((aligned(16))) char smem[PIPE_BYTES > EPI_BYTES ? PIPE_BYTES : EPI_BYTES];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)smem;

    STAMP(0); STAMP_ID();
    // a.vblocks tiles over gridDim.x blocks: a resident grid walks them (launch_pipe: one round of blocks, persistent), or one
    // block per tile when the launch has no more tiles than that
    for (int vb = blockIdx.x; vb < a0.vblocks; vb += gridDim.x) {
    const ConvArgs& a = kargs_fresh();                 // (tile set-up and K loop: their own reads of the argument block)
    const char* const ax = a.x;                        // (read HERE: a conditional read inside the K loop is not hoisted out of it)
    // The per-thread constants are derived from a LAUNDERED thread id inside the tile loop: as loop invariants the compiler
    // hoists them -- and every address built from them -- out of the loop and keeps them in registers across the K loop
    // (+40-50 VGPRs: the 4-wave 192 x 128 tile lost its second wave per SIMD, the 8-wave 256 x 256 tile spilled: -10 % on the
    // whole pass, measured).
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wave_p = wave % WAVES_P, wave_c = wave / WAVES_P;
    const int lid = xcd_remap(vb, a.vblocks);
    const int tile_p = fdiv(lid, a.d_tc);
    const int tile_c = lid - tile_p * a.tiles_c;
    const int strip = fdiv(tile_p, a.d_tps);
    const int i0 = (tile_p - strip * a.tiles_per_strip) * BP;
    const int co0 = tile_c * BC;
    const int H = a.H, W = a.W, Ho = a.Ho, TWt = a.TWt, PW = a.PW;
    const int row_bytes = a.x_ps * (int)sizeof(T);          // pitch of an input pixel

    int Rin_lo = 0, HS = XSLOTS, x0 = 0;
    if constexpr (KS != 1) {
        const int i_last = min(i0 + BP, a.total_i) - 1;
        const int r_first = fdiv(i0, a.d_TWt), r_last = fdiv(i_last, a.d_TWt);
        const int n_f = fdiv(r_first, a.d_Ho), n_l = fdiv(r_last, a.d_Ho);
        Rin_lo = n_f * (H + 1) + (r_first - n_f * Ho) * S;
        const int Rin_hi = n_l * (H + 1) + (r_last - n_l * Ho) * S + KS - 1;
        HS = (Rin_hi - Rin_lo + 1) * PW;
        x0 = strip * TWt * S - PAD;
    }

    const char* wsrc = a.wp + (long long)co0 * 64 + tid * 16;
    const long long wplane = (long long)a.Cout_pad * 64;
    static_assert((BC * 4) % NT == 0 || NT % (BC * 4) == 0, "tap index must be uniform per DMA");

    unsigned xo[XL1];
    const int nchunks = a.nchunks;
    // split types: byte offset of K-chunk c inside a pixel -- the passes [x_hi | x_lo | x_hi] over the pixel's two planes (conv_args.h)
    const int x3n = IsSplit<T>::value ? a.x3_n : 0, x3a1 = IsSplit<T>::value ? a.x3_adj1 : 0, x3a2 = IsSplit<T>::value ? a.x3_adj2 : 0;
    auto chunk_off = [&](int cc) -> long long {
        if constexpr (IsSplit<T>::value) return (long long)(cc * 64 + (cc >= x3n ? x3a1 : 0) + (cc >= 2 * x3n ? x3a2 : 0));
        else return (long long)cc * 64;
    };
    const int nphase = nchunks * PPC / KC;
    const uint32_t wave_lds = lds0 + wave * 1024;        // this wave's 1 KiB lane-linear window per DMA

    // weights of global phase gp (clamped: the tail re-loads the last plane into a dead ring slot so
    // every phase issues the same number of DMAs and the counted waits stay exact)
    // weight DMA j (of WL) of global phase gp; the tail is clamped: it re-loads the last plane into a
    // dead ring slot so every phase issues the same number of DMAs and the counted waits stay exact
    auto issue_w1 = [&](int gp, int j) {
#ifdef YOLO_LAB
        if ((a.lab & 64) && gp >= WR) return;           // (lab probe, WRONG results: no weight DMAs after the first ring: what the L2 -> LDS weight stream costs)
#endif
        const int kc = j / WL1, jj = j - kc * WL1;
        const int g = min(gp, nphase - 1) * KC + kc;
        glds16_m0(wsrc + (long long)g * wplane + (long long)jj * NT * 16,
               wave_lds + W_OFF + (gp % WR) * W_STAGE + kc * W_STAGE1 + jj * NT * 16);
    };
    auto issue_w = [&](int gp) {
#pragma unroll
        for (int j = 0; j < WL; ++j) issue_w1(gp, j);
    };
    // input DMA j of chunk c into X buffer `buf`
    // W1 (wave tiles of 128 couts: ONE wave per SIMD): the address of the zero page is taken once -- re-materialised per
    // DMA it is an s_load + s_waitcnt lgkmcnt(0), which also drains the wave's LDS reads and nothing else runs on the SIMD
    constexpr bool W1 = KS == 3 && MI == 4;
    const char* zero_page = (const char*)yolo_zero_page;
    if constexpr (W1) asm volatile("" : "+s"(zero_page));
    auto issue_x = [&](int j, int c, int buf) {
#ifdef YOLO_LAB
        if ((a.lab & 128) && c >= 2) return;            // (lab probe, WRONG results: no input DMAs after the first two chunks)
#endif
        const int kc = j / XL1, jj = j - kc * XL1;
        const int cc = (KS != 1) ? min(c, nchunks - 1) : min(c, nphase - 1) * KC + kc;      // (1x1: c counts phases)
        const char* src = (xo[jj] != 0xffffffffu) ? ax + ((long long)xo[jj] + chunk_off(cc)) : zero_page;
        glds16_m0(src, wave_lds + buf * X_STAGE + kc * X_STAGE1 + jj * NT * 16);
    };
    // Input DMAs per phase of a 3x3: one (phase q issues j = q) -- or, for a tile with more input DMAs per thread than phases (round 6: the
    // 4-wave stride-2 tile of the split type), XPP per phase in the FIRST phases, so that the last two phases of a chunk issue none:
    // the counted wait at the end of a phase leaves that phase's and the previous phase's DMAs in flight, and the next chunk's first
    // phase reads the buffer they fill.  (The first cut of that tile issued j = q, q + PPC -- input DMAs in phase 8 -- and computed
    // on a stale unit now and then: found by tests/test_gpu_split.py::test_split_d53_logits_vs_fp32_oracle[measure].)
    constexpr int XPP = (KS != 1 && XL > PPC) ? (XL + PPC - 3) / (PPC - 2) : 1;
    static_assert(KS == 1 || XL <= PPC || (KS == 3 && S == 2 && XPP * (PPC - 2) >= XL), "more input DMAs than phases: stride-2 tiles only");
    static_assert(BC * 4 >= NT, "one weight DMA covers rows of a single tap plane");

    // ---- prologue: the first weight DMAs go out before the (division-heavy) input address set-up, and the
    //      MFMA operand bases are computed while the first DMAs are in flight -------------------------------
    issue_w(0);
    // ---- per-thread DMA sources: byte offset of the slot's 16-byte unit for chunk 0 (32-bit: the host
    //      checks the activation tensor is < 4 GiB), or ~0 for a zero-padding slot (DMA'd from the zero page)
#pragma unroll
    for (int j = 0; j < XL1; ++j) {
        const int u = tid + j * NT;
        const int pos = u >> 2, part = u & 3;           // LDS position of the unit; `slot` = the halo slot it holds
        const int slot = (S == 2) ? ((pos & ~7) | ((pos & 3) << 1) | ((pos >> 2) & 1)) : pos;
        bool valid;
        long long off;
        int swz = (S == 2 ? pos >> 3 : pos >> 2);
        if constexpr (KS != 1) {
            const int rr = fdiv(slot, a.d_PW), cc = slot - rr * PW;
            if constexpr (S == 1) swz -= rr * a.row_swz;         // (row-relative swizzle: launch_pipe)
            else swz -= (rr >> 1) * a.TP;                        // (stride 2: TP = (PW - TWt) / 4 per OUTPUT row, or 0)
            const int Rr = Rin_lo + rr;
            const int n = fdiv(Rr, a.d_H1);
            const int yy = Rr - n * (H + 1) - PAD;
            const int xx = x0 + cc;
            valid = slot < HS && yy >= 0 && yy < H && n < a.N && xx >= 0 && xx < W && cc < (TWt - 1) * S + KS;      // (cc beyond: pitch padding)
            off = ((long long)(n * H + yy) * W + xx) * row_bytes;
        } else {
            const int i = i0 + slot;
            valid = i < a.total_i;
            off = (long long)i * row_bytes;
        }
        const int lp = (part ^ (swz & 3)) * 16;
        xo[j] = valid ? (unsigned)(off + lp) : 0xffffffffu;
    }
    if constexpr (KS != 1) {
#pragma unroll
        for (int j = 0; j < XL; ++j) issue_x(j, 0, 0);
#pragma unroll
        for (int g = 1; g < WR - 1; ++g) issue_w(g);
    } else {
#pragma unroll
        for (int j = 0; j < XL; ++j) issue_x(j, 0, 0);
#pragma unroll
        for (int g = 1; g < R1 - 1; ++g) {
#pragma unroll
            for (int j = 0; j < XL; ++j) issue_x(j, g, g);
            issue_w(g);
        }
    }
    // ---- per-lane MFMA operand bases ------------------------------------------------------------
    // RSW (3x3 stride 1): t00 = slot00 - 4 * row_swz * (halo row) -- the index the unit swizzle is taken from; a tap moves it by
    // dy * TP + dx where the slot moves by dy * PW + dx
    // RS2 (3x3 stride 2): t00 = the lane's halo row; the swizzle subtracts TP = (PW - TWt) / 4 per output row (launch_pipe)
    constexpr bool RSW = KS != 1 && S == 1;
    constexpr bool RS2 = KS == 3 && S == 2;
    const int TP = a.TP;
    int slot00[NI], t00[(RSW || RS2) ? NI : 1];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int i = i0 + (wave_p * NI + ni) * 32 + l31;
        const int ii = (i < a.total_i) ? i : i0;
        if constexpr (KS != 1) {
            const int r = fdiv(ii, a.d_TWt);
            const int tx = ii - r * TWt;
            const int n = fdiv(r, a.d_Ho);
            const int hrow = n * (H + 1) + (r - n * Ho) * S - Rin_lo;
            slot00[ni] = hrow * PW + tx * S;
            if constexpr (RSW) t00[ni] = hrow * TP + tx;
            if constexpr (RS2) t00[ni] = hrow;
        } else {
            slot00[ni] = ii - i0;
        }
    }
    const int aoff0 = (wave_c * MI * 32 + l31) * 64 + ((h ^ ((l31 >> 2) & 3)) << 4);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    if constexpr (KS != 1) wait_vmcnt<(WR - 2) * WL>();
    else wait_vmcnt<(R1 - 2) * (WL + XL)>();
    __builtin_amdgcn_s_barrier();
    STAMP(1);

    constexpr int NM = 2 * KC * MI * NI;                // MFMA "steps" per phase (one 16-byte operand pair each)
    auto phase = [&](auto shift_c, int c, int q, int gp) {
        constexpr int SHIFT = decltype(shift_c)::value;
        const int nx = (KS != 1) ? (XPP == 1 ? (q < XL ? 1 : 0) : (XL - q * XPP < 0 ? 0 : (XL - q * XPP < XPP ? XL - q * XPP : XPP))) : XL;      // input DMAs of this phase
        const int nd = nx + WL;
        const int stride = (NM - SHIFT) / (nd > 0 ? nd : 1) > 0 ? (NM - SHIFT) / (nd > 0 ? nd : 1) : 1;
        auto issue_item = [&](int k) {
            __builtin_amdgcn_sched_barrier(0);
            if (k < nx) {
                if constexpr (KS != 1) issue_x(XPP == 1 ? q : q * XPP + k, c + 1, (c + 1) & 1);
                else issue_x(k, gp + R1 - 1, (gp + R1 - 1) % R1);
            } else {
                issue_w1(gp + WR - 1, k - nx);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        const char* Wl = smem + W_OFF + (gp % WR) * W_STAGE;
        const char* Xl = smem + ((KS != 1) ? (c & 1) : (gp % R1)) * X_STAGE;
        int bx[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            int slot = (KS != 1) ? slot00[ni] + (q / KS) * PW + (q % KS) : slot00[ni];
#ifdef YOLO_LAB
            if (KS == 3 && (a.lab & 256)) slot = (wave_p * NI + ni) * 32 + l31 + q;      // (lab probe, WRONG results: consecutive slots = conflict-free fragment reads)
#endif
            if constexpr (S == 2) {
                // stride 2: the 32 pixels of a fragment are every other slot.  Inside each group of 8 slots the even ones
                // sit in the first 256 bytes and the odd ones in the second, so that 16 lanes still cover four whole
                // 256-byte lines (with the slots in pixel order they would touch half the banks: two-way conflicts on
                // every input fragment read, which bound these kernels); the unit swizzle follows the group index
                const int pos = (slot & ~7) | ((slot & 1) << 2) | ((slot >> 1) & 3);
                bx[ni] = pos * 64 + ((h ^ (((slot >> 3) - ((t00[ni] + q / KS) >> 1) * TP) & 3)) << 4);
            } else if constexpr (RSW) {
                int t = t00[ni] + (q / KS) * TP + (q % KS);
#ifdef YOLO_LAB
                if (a.lab & 256) t = slot;
#endif
                bx[ni] = slot * 64 + ((h ^ ((t >> 2) & 3)) << 4);
            } else {
                bx[ni] = slot * 64 + ((h ^ ((slot >> 2) & 3)) << 4);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2 * KC; ++ks) {
            const int kofw = (ks >> 1) * W_STAGE1, kofx = (ks >> 1) * X_STAGE1;     // sub-chunk of this phase (KC > 1)
            uint4 af[MI], bf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[mi] = *(const uint4*)(Wl + kofw + mi * 2048 + (aoff0 ^ ((ks & 1) * 32)));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bf[ni] = *(const uint4*)(Xl + kofx + (bx[ni] ^ ((ks & 1) * 32)));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    FragP<T>::mma(af[mi], bf[ni], acc[mi][ni]);
                    const int m = (ks * MI + mi) * NI + ni;
#pragma unroll
                    for (int k = 0; k < XL + WL; ++k)
                        if (k < nd && m == min(SHIFT + k * stride, NM - 1)) issue_item(k);
                }
        }
        // phase gp+1's data must have landed: everything except what this phase issued for gp+2 (the
        // input DMAs are issued before the weight DMAs, so leaving WL outstanding covers them too)
        if constexpr (KS != 1) {
            // (round 3, found by tools/fuzz_dgrad.py) 2x2 window with as many input DMAs per tile as phases per chunk (XL == PPC:
            // the 4-wave 128 x 128 and the 8-wave 256 x 256 tiles): the LAST phase of a chunk issues an input DMA that the very
            // next phase reads -- with a weight ring of four slots the counted wait above still lets it fly (it is older than
            // this phase's weight DMAs only), and the tile computed on a stale unit whenever its halo reached into the last
            // quarter of the staged slots.  There the wait keeps this phase's weight DMAs in flight and nothing else.
            if (KS == 2 && XL == PPC && q == PPC - 1) wait_vmcnt<((WR - 2) * WL < WL ? (WR - 2) * WL : WL)>();
            else wait_vmcnt<(WR - 2) * WL>();
        } else {
            wait_vmcnt<(R1 - 2) * (WL + XL)>();
        }
#ifdef YOLO_LAB
        // (lab, YOLO_EPI_AB bit 32: every second barrier dropped -- WRONG results, a timing probe of what a barrier costs;
        //  bit 64: the counted waits of those phases dropped as well)
        if (!((a.lab & 32) && (q & 1)))
#endif
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);      // keep the next phase's address math out of this phase (VGPR pressure)
    };
    if constexpr (KS == 1 && LEAN) {
        // ---- lean 1x1 K loop.  With one or two waves per SIMD (the small maps: 340 blocks of a 64 x 128 tile at 13x13)
        //      a wave issues an instruction every ~4 cycles and nothing fills the gaps, so the ~75 instructions of a
        //      generic phase -- ring-slot modulo, 64-bit source address multiply-adds, zero-page selects, m0 save / restore
        //      per DMA -- cost more than its 4 MFMAs (128 cycles).  Here the loop is unrolled by the ring depth (every
        //      LDS offset is an immediate), DMA sources are pointers advanced by one add per phase, and m0 is written
        //      once per DMA. ------------------------------------------------------------------------------------------
        static_assert(PPC == 1, "1x1");
        const char* wq[WL];                                  // source of weight DMA j for the NEXT phase to issue
        const char* xq[XL];
        unsigned xinc[XL1];                                  // 64 * KC for a pixel row, 0 for the zero page
        {
            const int g0 = (R1 - 1) * KC;                    // the prologue issued phases 0 .. R1-2
#pragma unroll
            for (int j = 0; j < WL; ++j) {
                const int kc = j / WL1, jj = j - kc * WL1;
                wq[j] = wsrc + (long long)(g0 + kc) * wplane + (long long)jj * NT * 16;
            }
#pragma unroll
            for (int j = 0; j < XL; ++j) {
                const int kc = j / XL1, jj = j - kc * XL1;
                const bool ok = xo[jj] != 0xffffffffu;
                xq[j] = ok ? ax + ((long long)xo[jj] + chunk_off(g0 + kc)) : (const char*)yolo_zero_page;
                if (kc == 0) xinc[jj] = ok ? 64u * KC : 0u;
            }
        }
        const long long winc = (long long)KC * wplane;
        auto lean_phase = [&](auto slot_c, int gp) {
            constexpr int U = decltype(slot_c)::value;       // ring slot read in this phase; (U + R1 - 1) % R1 is written
            constexpr int UW = (U + R1 - 1) % R1;
            const char* Wl = smem + W_OFF + U * W_STAGE;
            const char* Xl = smem + U * X_STAGE;
            int bx[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bx[ni] = slot00[ni] * 64 + ((h ^ ((slot00[ni] >> 2) & 3)) << 4);
            constexpr int ND = XL + WL;
            constexpr int STRIDE = (NM - 1) / ND > 0 ? (NM - 1) / ND : 1;
#pragma unroll
            for (int ks = 0; ks < 2 * KC; ++ks) {
                const int kofw = (ks >> 1) * W_STAGE1, kofx = (ks >> 1) * X_STAGE1;
                uint4 af[MI], bf[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) af[mi] = *(const uint4*)(Wl + kofw + mi * 2048 + (aoff0 ^ ((ks & 1) * 32)));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bf[ni] = *(const uint4*)(Xl + kofx + (bx[ni] ^ ((ks & 1) * 32)));
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        FragP<T>::mma(af[mi], bf[ni], acc[mi][ni]);
                        const int m = (ks * MI + mi) * NI + ni;
#pragma unroll
                        for (int k = 0; k < ND; ++k)
                            if (m == (1 + k * STRIDE < NM - 1 ? 1 + k * STRIDE : NM - 1)) {
                                __builtin_amdgcn_sched_barrier(0);
                                if (k < XL) {
                                    const int kc = k / XL1, jj = k - kc * XL1;
                                    glds16_m0(xq[k], wave_lds + UW * X_STAGE + kc * X_STAGE1 + jj * NT * 16);
                                } else {
                                    const int j = k - XL, kc = j / WL1, jj = j - kc * WL1;
                                    glds16_m0(wq[j], wave_lds + W_OFF + UW * W_STAGE + kc * W_STAGE1 + jj * NT * 16);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                    }
            }
            if (gp + R1 < nphase) {                          // (tail: the pointers stay on the last phase -> dead reloads)
#pragma unroll
                for (int j = 0; j < WL; ++j) wq[j] += winc;
#pragma unroll
                for (int j = 0; j < XL; ++j) xq[j] += xinc[j % XL1];
                if constexpr (IsSplit<T>::value) {
                    // the pointers now address the phase that starts at chunk g: entering the lo pass / the second hi pass they
                    // jump by the plane adjustment (the host checks that a phase of KC chunks never straddles two passes)
                    const int g = (gp + R1) * KC;
                    if (g == x3n || g == 2 * x3n) {
                        const int adj = g == x3n ? x3a1 : x3a2;
#pragma unroll
                        for (int j = 0; j < XL; ++j) xq[j] += xinc[j % XL1] ? adj : 0;
                    }
                }
            }
            wait_vmcnt<(R1 - 2) * (WL + XL)>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int gp = 0; gp < nphase; gp += R1) {
            if constexpr (R1 > 0) { lean_phase(std::integral_constant<int, 0>{}, gp); }
            if constexpr (R1 > 1) { if (gp + 1 < nphase) lean_phase(std::integral_constant<int, 1 % R1>{}, gp + 1); }
            if constexpr (R1 > 2) { if (gp + 2 < nphase) lean_phase(std::integral_constant<int, 2 % R1>{}, gp + 2); }
            if constexpr (R1 > 3) { if (gp + 3 < nphase) lean_phase(std::integral_constant<int, 3 % R1>{}, gp + 3); }
            static_assert(R1 <= 4, "lean loop: ring depth <= 4");
        }
    } else if constexpr (W1) {
        // ---- one wave per SIMD (256 accumulator registers per wave): nothing covers a wave's LDS latency but the wave itself,
        //      so the fragments are double-buffered in registers.  A phase is two K-steps of MI*NI MFMAs; the reads of step
        //      t + 1 are issued before the MFMAs of step t.  The next phase's first fragments come from the next ring slot,
        //      hence the counted wait + barrier sit in the MIDDLE of a phase (after its first step, by which time every DMA
        //      of the phase has been issued) and the second step's MFMAs run over the first reads of the next phase. --------
        static_assert(PPC == 9 && KC == 1 && S == 1, "W1: 3x3 stride 1");
        uint4 fa[2][MI], fb[2][NI];
        auto read_frags = [&](auto buf_c, int c, int q, int gp, int ks) {
            constexpr int B = decltype(buf_c)::value;
            const char* Wl = smem + W_OFF + (gp % WR) * W_STAGE;
            const char* Xl = smem + (c & 1) * X_STAGE;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fa[B][mi] = *(const uint4*)(Wl + mi * 2048 + (aoff0 ^ (ks * 32)));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int slot = slot00[ni] + (q / KS) * PW + (q % KS);
                const int t = t00[ni] + (q / KS) * TP + (q % KS);
                const int bxn = slot * 64 + ((h ^ ((t >> 2) & 3)) << 4);
                fb[B][ni] = *(const uint4*)(Xl + (bxn ^ (ks * 32)));
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        read_frags(std::integral_constant<int, 0>{}, 0, 0, 0, 0);
        for (int c = 0; c < nchunks; ++c) {
#pragma unroll
            for (int q = 0; q < PPC; ++q) {
                const int gp = c * PPC + q;
                const int nx = q < XL ? 1 : 0;
                constexpr int NMM = MI * NI;
                // ---- step 0: this phase's second fragments are read first; all DMAs of the phase go out between the MFMAs
                read_frags(std::integral_constant<int, 1>{}, c, q, gp, 1);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        FragP<T>::mma(fa[0][mi], fb[0][ni], acc[mi][ni]);
                        const int m = mi * NI + ni;
                        constexpr int STRIDE = (NMM - 2) / (WL + 1) > 0 ? (NMM - 2) / (WL + 1) : 1;
#pragma unroll
                        for (int k = 0; k < WL + 1; ++k)
                            if (m == min(1 + k * STRIDE, NMM - 1)) {
                                if (k == 0) {
                                    if (nx) {
                                        __builtin_amdgcn_sched_barrier(0);
                                        issue_x(q, c + 1, (c + 1) & 1);
                                        __builtin_amdgcn_sched_barrier(0);
                                    }
                                } else {
                                    __builtin_amdgcn_sched_barrier(0);
                                    issue_w1(gp + WR - 1, k - 1);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                    }
                // ---- middle: phase gp+1's data has landed (this phase's DMAs and the previous phase's weight DMAs may still
                //      be in flight), everyone's reads of the slots about to be overwritten are complete
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (q < XL) wait_vmcnt<2 * WL + 1>(); else wait_vmcnt<2 * WL>();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- step 1 over the first reads of the next phase
                if (q + 1 < PPC) read_frags(std::integral_constant<int, 0>{}, c, q + 1, gp + 1, 0);
                else read_frags(std::integral_constant<int, 0>{}, c + 1, 0, gp + 1, 0);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) FragP<T>::mma(fa[1][mi], fb[1][ni], acc[mi][ni]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
    for (int c = 0; c < nchunks / KC; ++c) {
#pragma unroll
        for (int q = 0; q < PPC; ++q) {
            const int gp = c * PPC + q;
            phase(std::integral_constant<int, 1>{}, c, q, gp);
        }
    }
    }
    STAMP(2);
    // ---- epilogue (conv_epilogue.h): every wave transposes its slab through its own LDS scratch ------
    // (round 5) The output offsets -- a dozen divisions per lane -- and the epilogue's reads of the argument block are done
    // BEFORE the K loop's drain: the phase stamps (tools/stamp_probe.py) showed ~1.5 k cycles of drain + barrier (the tail's dead
    // DMAs making their round trip) followed by ~1 k cycles of this set-up, one after the other, with nothing else to issue.
    {
    const ConvArgs& a = kargs_fresh();                 // (the epilogue's own reads: nothing of the set-up stays live for it)
    const int Ho = a.Ho, Wo = a.Wo, TWt = a.TWt;
    long long yoff[NI];                  // output element offset of each lane's pixels (-1: none)
    long long roff[NI];                  // residual element offset (dense tensor; differs when y is strided)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int i = i0 + (wave_p * NI + ni) * 32 + l31;
        int n, pix;
        if constexpr (KS != 1) {
            const int r = fdiv(min(i, a.total_i - 1), a.d_TWt);
            const int tx = min(i, a.total_i - 1) - r * TWt;
            n = fdiv(r, a.d_Ho);
            pix = (r - n * Ho) * Wo + strip * TWt + tx;
            // sub-pixel output: pixel (oy, ox) owns the 2x2 patch at (2 oy, 2 ox) of the (2Ho, 2Wo) map (y_ps = Cout / 4)
            if (a.d2s || a.up2) pix = (r - n * Ho) * 4 * Wo + 2 * (strip * TWt + tx);
        } else {
            n = fdiv(min(i, a.total_i - 1), a.d_HoWo);
            pix = min(i, a.total_i - 1) - n * (Ho * Wo);
            if (a.up2) {                                  // (1x1: TWt == Wo)
                const int oy = fdiv(pix, a.d_TWt);
                pix = oy * 4 * Wo + 2 * (pix - oy * Wo);
            }
        }
        yoff[ni] = (i < a.total_i) ? (long long)n * a.y_bs + (long long)pix * a.y_ps : -1;
        roff[ni] = (long long)n * a.r_bs + (long long)pix * a.r_ps;
    }
    wait_vmcnt<0>();                     // the tail's dead DMAs must land before the block's LDS is released
    __builtin_amdgcn_s_barrier();        // all waves are done reading the pipeline's LDS
    STAMP(3);
    // (STATS: one partial row of BatchNorm sums per (pixel tile, pixel wave))
    constexpr int ESTATS = STATS == 3 ? 0 : STATS;       // (3 = the fused tail 1x1 below, not a statistics mode)
    float* srow = ESTATS ? a.stats + ((size_t)tile_p * WAVES_P + wave_p) * 2 * a.Cout_pad : nullptr;
    conv_epilogue<T, MI, NI, ESTATS>(acc, yoff, smem + wave * YOLO_EPI_WAVE_BYTES_MI(MI), a, co0 + wave_c * MI * 32, lane, roff, srow);
    STAMP(4);
    if constexpr (STATS == 3) {
        // ---- TAIL (round 4): the 1x1 convolution that follows this one in the network -- the next residual block's first conv
        //      (basic_yolo.py:26, darknet.py DarknetBasicBlockV3), a detection block's 1x1 after its 3x3, YOLOOutput after the
        //      tip (basic_yolo.py:98-105) -- computed HERE from the output tile this block has just stored: z = act(W1 . y) over
        //      the tile's pixels, K = this conv's Cout (one 256-cout tile holds every channel of a pixel).  The tile is read back
        //      through L2 by LDS-DMA, chunk by chunk, exactly as a 1x1 kernel would stage it (same operands, same K order:
        //      bit-identical to the separate launch), but without a second kernel, its boundary, or the HBM read of y. ------
        static_assert(KS == 3 && WAVES_C == 4 && sizeof(T) == 2, "tail: 3x3, bf16, tiles of 4 cout waves (the tail's 128 couts = 4 x 32)");
        constexpr int BPX = (BP + NT / 4 - 1) / (NT / 4) * (NT / 4);        // X slots per phase (whole DMAs; the surplus reads zeros)
        constexpr int X2_STAGE = BPX * 64, W2_STAGE = 128 * 64, S2 = X2_STAGE + W2_STAGE;
        // ring depth: as many K chunks in flight as the block's LDS holds (<= 8 = every chunk of a 256-channel tile): the tail
        // is a chain of L2 round trips -- six to eight MFMAs per wave and chunk -- so its time is the latency it cannot overlap
        constexpr int LDS_ALL = PIPE_BYTES > EPI_BYTES ? PIPE_BYTES : EPI_BYTES;
        constexpr int R2 = LDS_ALL / S2 > 8 ? 8 : LDS_ALL / S2;
        constexpr int XL2 = X2_STAGE / (NT * 16), WL2 = W2_STAGE / (NT * 16), ND2 = XL2 + WL2;
        static_assert(W2_STAGE % (NT * 16) == 0 && R2 >= 3, "tail: LDS");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this wave's output stores have reached L2 ...
        __syncthreads();                                                    // ... and everyone's; the epilogue's scratch is free
        unsigned xs2[XL2];
#pragma unroll
        for (int j = 0; j < XL2; ++j) {
            const int u = tid + j * NT;
            const int slot = u >> 2, part = u & 3;
            const int i = i0 + slot;
            const bool valid = slot < BP && i < a.total_i;
            const int ic = min(i, a.total_i - 1);
            const int r = fdiv(ic, a.d_TWt);
            const int tx = ic - r * TWt;
            const int n = fdiv(r, a.d_Ho);
            const int pix = (r - n * Ho) * Wo + strip * TWt + tx;
            const long long off = ((long long)n * a.y_bs + (long long)pix * a.y_ps) * 2;
            xs2[j] = valid ? (unsigned)(off + ((part ^ ((slot >> 2) & 3)) * 16)) : 0xffffffffu;
        }
        const char* const ay = a.y;
        const char* const atwp = a.t_wp;
        const int nph2 = a.Cout >> 5;                                       // K chunks of 32 channels (the host checks Cout % 32 == 0)
        const long long wplane2 = (long long)round_up(a.t_cout, YOLO_COUT_PAD) * 64;
        const char* zp2 = (const char*)yolo_zero_page;
        auto issue2 = [&](int c) {
            const int cc = min(c, nph2 - 1);                                // (tail phases re-load the last chunk into a dead slot)
            const uint32_t base = wave_lds + (c % R2) * S2;
#pragma unroll
            for (int j = 0; j < XL2; ++j)
                glds16(xs2[j] != 0xffffffffu ? ay + ((size_t)xs2[j] + (size_t)cc * 64) : zp2, base + j * NT * 16);
#pragma unroll
            for (int j = 0; j < WL2; ++j)
                glds16(atwp + (long long)cc * wplane2 + (long long)(tid + j * NT) * 16, base + X2_STAGE + j * NT * 16);
        };
        f32x16 acc2[1][NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[0][ni][r] = 0.f;
#pragma unroll
        for (int c = 0; c < R2 - 1; ++c) issue2(c);
        const int aoff2 = (wave_c * 32 + l31) * 64 + ((h ^ ((l31 >> 2) & 3)) << 4);
        int bx2[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int slot = (wave_p * NI + ni) * 32 + l31;
            bx2[ni] = slot * 64 + ((h ^ ((slot >> 2) & 3)) << 4);
        }
        wait_vmcnt<(R2 - 2) * ND2>();
        __builtin_amdgcn_s_barrier();
        for (int c = 0; c < nph2; ++c) {
            issue2(c + R2 - 1);
            const char* X2l = smem + (c % R2) * S2;
            const char* W2l = X2l + X2_STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 af = *(const uint4*)(W2l + (aoff2 ^ (ks * 32)));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const uint4 bf = *(const uint4*)(X2l + (bx2[ni] ^ (ks * 32)));
                    FragP<T>::mma(af, bf, acc2[0][ni]);
                }
            }
            wait_vmcnt<(R2 - 2) * ND2>();
            __builtin_amdgcn_s_barrier();
        }
        wait_vmcnt<0>();
        __syncthreads();                                                    // everyone is done with the ring: it becomes the scratch
        long long yoff2[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int i = i0 + (wave_p * NI + ni) * 32 + l31;
            const int ic = min(i, a.total_i - 1);
            const int r = fdiv(ic, a.d_TWt);
            const int tx = ic - r * TWt;
            const int n = fdiv(r, a.d_Ho);
            const int pix = (r - n * Ho) * Wo + strip * TWt + tx;
            yoff2[ni] = (i < a.total_i) ? (long long)n * a.t_y_bs + (long long)pix * a.t_y_ps : -1;
        }
        ConvArgs b = a;
        b.scale = a.t_scale; b.bias = a.t_bias; b.res = nullptr; b.y = a.t_y; b.Cout = a.t_cout; b.out_f32 = a.t_out_f32;
        b.up2 = 0; b.d2s = 0; b.slope = a.t_slope; b.y_bs = a.t_y_bs; b.y_ps = a.t_y_ps; b.stats = nullptr;
        conv_epilogue<T, 1, NI, 0>(acc2, yoff2, smem + wave * YOLO_EPI_WAVE_BYTES_MI(1), b, wave_c * 32, lane, nullptr, nullptr);
    }
#if defined(YOLO_STAMP_ON)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(5);
#endif
    }
    if (vb + (int)gridDim.x < a0.vblocks) __syncthreads();     // the next tile's DMAs overwrite the epilogue's scratch
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// PERSISTENT blocks (round 4): the grid is ONE resident round -- blocks per CU of the instantiation that is launched x CUs -- and a
// block walks tiles vb = blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8: a block's tiles stay on its XCD).  No
// workgroup dispatch between the tiles of a CU: +1.8 % on the 608x608 bs 64 pass (1444 tiles per CU slot over a launch), +0.2 % at
// 416x416 bs 32 (about one round anyway), same-box A/B.  The resident size is cached per DEVICE and per instantiation; a failed
// query is not cached.  Lab build: YOLO_PIPE_PERSIST=0 = one block per tile (A/B knob), n > 0 = a grid of n blocks.
template <auto kernel>
static long long pipe_resident_grid(int threads, long long grid) {
    static const int persist = (int)YOLO_LAB_ENV("YOLO_PIPE_PERSIST", -1);
    if (persist == 0) return grid;
    if (persist > 0) return grid > persist ? persist : grid;
    constexpr int kMaxDev = 16;
    static std::atomic<int> resident[kMaxDev];      // (zero-initialised; a race writes the same value twice)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return grid; }
    int res = (dev >= 0 && dev < kMaxDev) ? resident[dev].load(std::memory_order_relaxed) : 0;
    if (!res) {
        int n = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kernel, threads, 0) != hipSuccess || n < 1 ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) {
            (void)hipGetLastError();
            return grid;                                // (this launch: one block per tile; the next one asks again)
        }
        res = n * (cus & ~7);
        if (dev >= 0 && dev < kMaxDev) resident[dev].store(res, std::memory_order_relaxed);
    }
    return grid > res ? res : grid;
}

template <typename T, int KS, int WAVES_P, int WAVES_C, int MI, int NI, int XSLOTS, int S = 1, int RD = 0, int KC = 1, int LEAN = 0>
static int launch_pipe(ConvArgs& a, hipStream_t st, const NameOut* name) {
    constexpr int BP = WAVES_P * NI * 32, BC = WAVES_C * MI * 32;
    if (KC > 1 && (a.nchunks % KC || a.nchunks < 2 * KC)) return YOLO_EUNSUPPORTED;
    if (IsSplit<T>::value && KC > 1 && (a.x3_n % KC)) return YOLO_EUNSUPPORTED;      // (a phase must not straddle two K passes)
    if (LEAN && a.nchunks / KC < 4) return YOLO_EUNSUPPORTED;       // the lean loop starts after a full ring of phases
    a.row_swz = 0;
    if (KS != 1) {
        // 3x3 stride 1: halo rows of pitch TWt + 4 instead of TWt + 2 where that fits the buffer.  A 32-pixel input fragment
        // crosses strip rows, and ds_read_b128 is served in 16-lane groups ({0-3, 12-15, 20-27}, ...) that are conflict-free when
        // their 16 (slot % 4, 16-byte unit) pairs differ: with pitch = TWt (mod 4) and the unit swizzle ((slot >> 2) - halo row) & 3
        // that pair is the pixel's index in the strip (row * TWt + column) mod 16 for every tap -- 16 different values in every
        // group.  With pitch TWt + 2 and swizzle (slot >> 2) & 3 each row crossing shifts it by 2: SQ_LDS_BANK_CONFLICT was 36 % of
        // the LDS-active cycles of these kernels; a lab probe with conflict-free (wrong) addresses ran them 2-5 % faster
        // (tools/ab_bank.sh).
        int best = -1, best_hs = 1 << 30;
        static const int rsw_off = (int)YOLO_LAB_ENV("YOLO_NO_ROW_SWZ", 0);       // (lab A/B: the round-4 layout)
        // The 2x2 window (stride-2 data gradient) likewise with TWt + 4 for TWt + 1.  Stride 2: a fragment's lanes are same-parity
        // slots 2 apart (the 8-slot permutation keeps them on whole 256-byte lines) and an output row is 2 halo rows: pitch = TWt
        // (mod 4) -- 0-3 slots more than 2 TWt + 1 -- and the swizzle ((slot >> 3) - (halo row >> 1) * (PW - TWt) / 4) & 3.
        int best_pad = 0;
        for (int pass = rsw_off ? 1 : 0; pass < 2 && best < 0; ++pass) {
            for (int d = 1; d <= a.Wo; ++d) {
                if (a.Wo % d) continue;
                const int pad = pass ? 0 : (S == 1 ? 4 - (KS - 1) : (4 - ((d + 1) & 3)) & 3);
                const int hs = conv_halo_slots(BP, d, a.Ho, a.H, S, (long long)a.N * a.Ho, KS, pad);
                if (hs <= XSLOTS - (a.halo_strict ? 1 : 0) && hs <= best_hs) { best = d; best_hs = hs; best_pad = pad; }
            }
            if (best >= 0) a.row_swz = pass ? 0 : 1;
        }
        if (best < 0) return YOLO_EUNSUPPORTED;
        a.TWt = best;
        a.PW = (best - 1) * S + KS + best_pad;
    } else {
        a.TWt = a.Wo;
        a.PW = a.Wo;
    }
    a.TP = S == 1 ? a.PW - 4 * a.row_swz : (a.row_swz ? (a.PW - a.TWt) / 4 : 0);
    a.nstrips = a.Wo / a.TWt;
    const long long tot = (long long)a.N * a.Ho * a.TWt;
    if (tot > 0x7fffffffLL) return YOLO_EUNSUPPORTED;
    a.total_i = (int)tot;
    a.tiles_per_strip = (a.total_i + BP - 1) / BP;
    a.tiles_c = (a.Cout + BC - 1) / BC;
    long long grid = (long long)a.nstrips * a.tiles_per_strip * a.tiles_c;
    if (grid > 0x7fffffffLL) return YOLO_EUNSUPPORTED;
    a.vblocks = (int)grid;
    conv_args_fastdiv(a);
    // BatchNorm statistics in the epilogue: bf16, the transposed store path (conv_epilogue.h), no sub-pixel / up-sampled
    // stores; the data-gradient sums (mode 2) only on stride-1 kernels
    constexpr bool kStats = IsBf16<T>::value && KS != 2;
    const bool stats_ok = kStats && !a.out_f32 && !a.d2s && !a.up2 && (a.Cout % 8) == 0 && (a.y_ps % 8) == 0 && (a.y_bs % 8) == 0 &&
                          (a.stats_mode == 1 || (a.stats_mode == 2 && S == 1 && a.y_ps == a.Cout));
    if (a.stats && !stats_ok) return YOLO_EUNSUPPORTED;
    // fused tail 1x1 (kernel flag 3): bf16 3x3 tiles that hold all the channels of a pixel (tiles_c == 1: Cout <= the tile's 128 or
    // 256 couts) and have four cout waves (each takes 32 of the tail's <= 128 couts)
    constexpr bool kTail = sizeof(T) == 2 && !IsSplit<T>::value && KS == 3 && WAVES_C == 4 && RD == 0 && KC == 1 && LEAN == 0;       // (128- or 256-cout tiles of 8 waves)
    if (a.t_wp) {
        if (!kTail || a.stats || a.out_f32 || a.up2 || a.d2s || a.tiles_c != 1 || (a.Cout % 32) || a.t_cout < 1 || a.t_cout > 128)
            return YOLO_EUNSUPPORTED;
        if ((long long)a.N * a.y_bs * 2 >= 0xffffff00LL) return YOLO_EUNSUPPORTED;         // 32-bit DMA source offsets into y
    }
    if (name) {
        snprintf(name->buf, name->len, "void conv_pipe_kernel<%s, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d>(ConvArgs)",
                 Elem<T>::name, KS, WAVES_P, WAVES_C, MI, NI, XSLOTS, S, RD, KC, LEAN,
                 a.t_wp ? 3 : (a.stats && stats_ok) ? a.stats_mode : 0);
        if (name->stats_rows) *name->stats_rows = (a.stats && stats_ok) ? a.nstrips * a.tiles_per_strip * WAVES_P : -1;
        return YOLO_OK;
    }
#define YOLO_PIPE_LAUNCH(ST_)                                                                                      \
    do {                                                                                                           \
        constexpr auto kern = conv_pipe_kernel<T, KS, WAVES_P, WAVES_C, MI, NI, XSLOTS, S, RD, KC, LEAN, ST_>;     \
        YOLO_LAUNCH(kern, dim3((unsigned)pipe_resident_grid<kern>(WAVES_P * WAVES_C * 64, grid)),                  \
                    dim3(WAVES_P * WAVES_C * 64), 0, st, a);                                                       \
        YOLO_LAUNCH_CHECK();                                                                                       \
        return YOLO_OK;                                                                                            \
    } while (0)
    if constexpr (kTail) {
        if (a.t_wp) YOLO_PIPE_LAUNCH(3);
    }
    if constexpr (kStats) {
        if (a.stats && a.stats_mode == 1) YOLO_PIPE_LAUNCH(1);
        if constexpr (S == 1) {
            if (a.stats && a.stats_mode == 2) YOLO_PIPE_LAUNCH(2);
        }
    }
    YOLO_PIPE_LAUNCH(0);
#undef YOLO_PIPE_LAUNCH
}

// algo ids (yolo_conv_desc.algo): 1 = generic (conv_igemm.hip); pipelined variants:
//   2: 8 waves, 256 px x 256 cout (wave tile 128x64)   3: 8 waves, 256 px x 128 cout (wave tile 64x64)
//   4: 4 waves, 128 px x 128 cout (wave tile 64x64)    5: 8 waves, 128 px x 256 cout (wave tile 128x32)
//   6: 8 waves, 192 px x 256 cout (wave tile 96x64)    7: 8 waves, 192 px x 128 cout (wave tile 96x32)
//   8: 4 waves, 192 px x 128 cout (wave tile 96x64)    -- 192-pixel tiles exist to cut tile quantisation
//   9 / 10: 3x3 stride 2, 8 waves, 128 px x 128 / 256 cout; 17 / 16: the same with a smaller halo buffer and a
//           4-slot weight ring
//   11: 4 waves, 64 px x 128 cout; 12 (1x1 only): 4 waves, 64 px x 256 cout -- small-M layers (13x13 maps at
//       batch 32 have 5408 pixels: more, smaller tiles fill the chip)
//   26: 3x3 stride 1, bf16: 4 waves, 256 px x 256 cout, wave tile 128x128 -- ONE wave per SIMD (256 accumulator registers),
//       a third fewer LDS fragment reads per MFMA than algo 2, fragments double-buffered in registers (the W1 loop).
//       Measured at batch 64: 1328 vs 1311 TFLOP/s on 19x19 1024->2048, 1263 vs 1276 on 38x38 512->1024, 1120 vs 1172 on
//       76x76 256->512; a 192 px x 256 cout sibling (wave tile 128x96) ties algo 6 at batch 32 (1167 vs 1168, 1100 vs 1125)
//       and was removed: the long-K layers are not bound by LDS traffic or by the loop structure (power / clock)
//   (also tried: algo 4 with a 2-slot weight ring = 48 KB of LDS, three blocks per CU -- 4-10 % faster than algo 4 on the short-K
//    layers, never faster than the 192-pixel tiles there, which fit neither three blocks of LDS nor of registers; removed)
int conv_pipe_dispatch_b(ConvArgs& a, int ks, int dtype, int algo, hipStream_t st, const NameOut* nm);                 // (unit 1)
int conv_pipe_dispatch_c(ConvArgs& a, int ks, int stride, int dtype, int algo, hipStream_t st, const NameOut* nm);     // (unit 2)
int conv_pipe_dispatch_d(ConvArgs& a, int ks, int dtype, int algo, hipStream_t st, const NameOut* nm);                 // (unit 3)
int conv_pipe_dispatch_e(ConvArgs& a, int algo, hipStream_t st, const NameOut* nm);                                    // (unit 4)
int conv_pipe_dispatch_f(ConvArgs& a, int algo, hipStream_t st, const NameOut* nm);                                    // (unit 5)
int conv_pipe_dispatch_g(ConvArgs& a, int ks, int stride, int algo, hipStream_t st, const NameOut* nm);                // (unit 6)
int conv_pipe_dispatch_h(ConvArgs& a, int ks, int algo, hipStream_t st, const NameOut* nm);                            // (unit 7)
int conv_pipe_dispatch_i(ConvArgs& a, int ks, int stride, int algo, hipStream_t st, const NameOut* nm);                // (unit 8)
int conv_pipe_dispatch_j(ConvArgs& a, int ks, int algo, hipStream_t st, const NameOut* nm);                            // (unit 9)

template <typename T>
static int pipe_dispatch_t(ConvArgs& a, int ks, int stride, int algo, hipStream_t st, const NameOut* nm) {
#if YOLO_PIPE_PART == 0
    if (ks == 3 && stride == 2) return conv_pipe_dispatch_e(a, algo, st, nm);      // (bf16: a unit of its own)
#endif
#if YOLO_PIPE_PART == 2 || YOLO_PIPE_PART == 4 || YOLO_PIPE_PART == 6 || YOLO_PIPE_PART == 8
    if (ks == 3 && stride == 2) {
        // stride 2: the input footprint is ~4x the output tile, so tiles are 128 output pixels
        switch (algo) {
            case 9: return launch_pipe<T, 3, 2, 4, 1, 2, 896, 2>(a, st, nm);     // 8 waves, 128 px x 128 cout
            case 10: return launch_pipe<T, 3, 2, 4, 2, 2, 896, 2>(a, st, nm);    // 8 waves, 128 px x 256 cout
            // same tiles with a 640-slot halo buffer (enough for every strip shape of the 416/608 families): leaves room
            // for the 4th weight-ring slot
            case 16: return launch_pipe<T, 3, 2, 4, 2, 2, 640, 2>(a, st, nm);
            case 17: return launch_pipe<T, 3, 2, 4, 1, 2, 640, 2>(a, st, nm);
            case 18: return launch_pipe<T, 3, 2, 4, 2, 2, 768, 2>(a, st, nm);    // (768 slots: tiles that cross image boundaries)
            case 42:                                                              // (split type: a 64-cout tile, see algo 40; ten input DMAs per thread and chunk: two in phase 0)
                if constexpr (IsSplit<T>::value) return launch_pipe<T, 3, 4, 1, 2, 1, 640, 2>(a, st, nm);   // 4 waves, 128 px x 64 cout
                break;
        }
        return YOLO_EUNSUPPORTED;
    }
#endif
#if YOLO_PIPE_PART == 4
    return YOLO_EUNSUPPORTED;
#elif YOLO_PIPE_3X3
    if (ks == 3) {
        switch (algo) {
#if YOLO_PIPE_S1A
            case 2: return launch_pipe<T, 3, 2, 4, 2, 4, 512>(a, st, nm);
            case 3: return launch_pipe<T, 3, 4, 2, 2, 2, 512>(a, st, nm);
            case 4: return launch_pipe<T, 3, 2, 2, 2, 2, 256>(a, st, nm);
            case 6: return launch_pipe<T, 3, 2, 4, 2, 3, 384>(a, st, nm);
            case 7: return launch_pipe<T, 3, 2, 4, 1, 3, 384>(a, st, nm);
            case 8: return launch_pipe<T, 3, 2, 2, 2, 3, 384>(a, st, nm);      // (384 slots = 80 KB with the 4-slot ring: still two blocks per CU; room for the padded pitch on 13-wide strips)
#endif
#if YOLO_PIPE_PART == 0
            default: return conv_pipe_dispatch_f(a, algo, st, nm);
#endif
#if YOLO_PIPE_S1B
            case 5: return launch_pipe<T, 3, 1, 8, 1, 4, 256>(a, st, nm);
            case 11: return launch_pipe<T, 3, 2, 2, 2, 1, 192>(a, st, nm);
            // (round 5) pixel-heavy tiles: the same wave tiles as algo 6 / 2 (96 x 64, 128 x 64) arranged 4 x 2 instead of 2 x 4.  A 3x3
            // re-reads its weights for every tap and its input once per K chunk, so per output the L2 -> LDS stream of a 384 x 128
            // tile is ~40 % smaller than that of the 192 x 256 one (X 30 KB + W 9 x 8 KB against 24.5 + 9 x 16 per chunk) -- and a lab
            // probe that drops the weight DMAs (wrong results, timing only: tools/ab_kloop.sh) runs these kernels 8-12 % faster.
            case 27: return launch_pipe<T, 3, 4, 2, 2, 3, 768>(a, st, nm);     // 8 waves, 384 px x 128 cout (wave tile 96x64)
            case 28: return launch_pipe<T, 3, 4, 2, 2, 4, 1024>(a, st, nm);    // 8 waves, 512 px x 128 cout (wave tile 128x64)
            case 41:                                                            // (split type: 64-cout tiles, see algo 40)
                if constexpr (IsSplit<T>::value) return launch_pipe<T, 3, 4, 1, 2, 2, 512>(a, st, nm);      // 4 waves, 256 px x 64 cout
                break;
            case 26:                                                            // 4 waves, 256 px x 256 cout (wave tile 128x128)
                // (not for the split type: beside the split epilogue's second plane the 256-register accumulator spills)
                if constexpr (sizeof(T) == 2 && !IsSplit<T>::value) return launch_pipe<T, 3, 2, 2, 4, 4, 512>(a, st, nm);
                break;
#endif
        }
        return YOLO_EUNSUPPORTED;
    }
#if YOLO_PIPE_PART == 5
    return YOLO_EUNSUPPORTED;
#elif YOLO_PIPE_PART == 0
    return conv_pipe_dispatch_b(a, ks, Elem<T>::dtype, algo, st, nm);
#elif YOLO_PIPE_PART == 6
    return conv_pipe_dispatch_h(a, ks, algo, st, nm);
#elif YOLO_PIPE_PART == 8
    return conv_pipe_dispatch_j(a, ks, algo, st, nm);
#else
    return conv_pipe_dispatch_d(a, ks, Elem<T>::dtype, algo, st, nm);
#endif
#else
    if (ks == 2) {
        // 2x2 window (yolo_conv_dgrad_s2, bf16 only): four phases per K chunk
        if constexpr (IsBf16<T>::value) {
            switch (algo) {
                case 2: return launch_pipe<T, 2, 2, 4, 2, 4, 512>(a, st, nm);    // 256 px x 256 cout
                case 6: return launch_pipe<T, 2, 2, 4, 2, 3, 384>(a, st, nm);    // 192 px x 256 cout
                case 10: return launch_pipe<T, 2, 2, 4, 2, 2, 384>(a, st, nm);   // 128 px x 256 cout
                case 4:                                                           // 128 px x 128 cout, 4 waves (see the wait at the end of `phase`)
                    return launch_pipe<T, 2, 2, 2, 2, 2, 256>(a, st, nm);
            }
        }
        return YOLO_EUNSUPPORTED;
    }
    {
        switch (algo) {
            // 1x1: the lean K loop wherever the K extent allows it (>= 4 phases), the generic loop otherwise
#define YOLO_PIPE1(...)                                                                        \
    {                                                                                          \
        ConvArgs b = a;                                                                        \
        const int rc = launch_pipe<T, 1, __VA_ARGS__, 1, 0, 1, 1>(b, st, nm);                  \
        if (rc != YOLO_EUNSUPPORTED) { a = b; return rc; }                                     \
        return launch_pipe<T, 1, __VA_ARGS__>(a, st, nm);                                      \
    }
            case 2: YOLO_PIPE1(2, 4, 2, 4, 256)
            case 3: YOLO_PIPE1(4, 2, 2, 2, 256)
            case 4: YOLO_PIPE1(2, 2, 2, 2, 128)
            case 5: YOLO_PIPE1(1, 8, 1, 4, 128)
            case 8: YOLO_PIPE1(2, 2, 2, 3, 192)
            case 11: YOLO_PIPE1(2, 2, 2, 1, 64)
            case 12: YOLO_PIPE1(1, 4, 2, 2, 64)
            // (round 6, split type only) 64-cout tiles for the first stages' 32- / 64-channel layers: every other variant stages >= 128
            // weight rows per phase, and on the split path -- where these layers are not covered by the fused stem / residual-block /
            // streaming kernels of the 2-byte paths -- a 1x1 64 -> 32 at 208x208 moved twice its input bytes in zero weight rows
            // (567 us against an HBM floor of ~120).  41 / 42: the 3x3 stride-1 / stride-2 siblings.
            case 40:
                if constexpr (IsSplit<T>::value) YOLO_PIPE1(4, 1, 2, 2, 256)        // 4 waves, 256 px x 64 cout
                break;
#undef YOLO_PIPE1
            // deep-ring variants (one block per CU, 5-7 phases of loads in flight; generic loop)
            case 19: return launch_pipe<T, 1, 1, 4, 2, 2, 64, 1, 7>(a, st, nm);      // 64 px x 256 cout
            case 20: return launch_pipe<T, 1, 2, 2, 2, 1, 64, 1, 8>(a, st, nm);      // 64 px x 128 cout
            case 21: return launch_pipe<T, 1, 2, 2, 2, 2, 128, 1, 6>(a, st, nm);     // 128 px x 128 cout
            // 3-slot rings (round 3): LESS LDS per block so that one more block fits a CU -- the large-map 1x1 layers are
            // HBM-bound and a block runs load -> compute -> store in sequence; a third (fourth) co-resident block overlaps them
            case 36: return launch_pipe<T, 1, 2, 2, 2, 2, 128, 1, 3, 1, 1>(a, st, nm);     // 128 px x 128 cout, 48 KB: 3 blocks per CU
            case 37: return launch_pipe<T, 1, 2, 2, 2, 1, 64, 1, 3, 1, 1>(a, st, nm);      // 64 px x 128 cout, 39 KB: 4 blocks per CU
            case 38: return launch_pipe<T, 1, 1, 8, 1, 4, 128, 1, 3, 1, 1>(a, st, nm);     // 128 px x 256 cout (8 waves), 78 KB: 2 blocks per CU
            case 39: return launch_pipe<T, 1, 2, 2, 2, 3, 192, 1, 3, 1, 1>(a, st, nm);     // 192 px x 128 cout, 60 KB: 2 blocks per CU
            // two K chunks per phase (half the barriers), lean loop
            case 22: return launch_pipe<T, 1, 2, 2, 2, 1, 64, 1, 0, 2, 1>(a, st, nm);   // 64 px x 128 cout
            case 23: return launch_pipe<T, 1, 2, 2, 2, 2, 128, 1, 0, 2, 1>(a, st, nm);  // 128 px x 128 cout
            case 24: return launch_pipe<T, 1, 2, 2, 2, 3, 192, 1, 0, 2, 1>(a, st, nm);  // 192 px x 128 cout
            case 25: return launch_pipe<T, 1, 1, 4, 2, 2, 64, 1, 0, 2, 1>(a, st, nm);   // 64 px x 256 cout
        }
    }
    return YOLO_EUNSUPPORTED;
#endif
}

#if YOLO_PIPE_PART == 5
int conv_pipe_dispatch_f(ConvArgs& a, int algo, hipStream_t st, const NameOut* nm) { return pipe_dispatch_t<bf16_t>(a, 3, 1, algo, st, nm); }
#elif YOLO_PIPE_PART == 4
int conv_pipe_dispatch_e(ConvArgs& a, int algo, hipStream_t st, const NameOut* nm) { return pipe_dispatch_t<bf16_t>(a, 3, 2, algo, st, nm); }
#elif YOLO_PIPE_PART == 1
int conv_pipe_dispatch_b(ConvArgs& a, int ks, int dtype, int algo, hipStream_t st, const NameOut* nm) {
    return pipe_dispatch_t<bf16_t>(a, ks, 1, algo, st, nm);
}
#elif YOLO_PIPE_PART == 3
int conv_pipe_dispatch_d(ConvArgs& a, int ks, int dtype, int algo, hipStream_t st, const NameOut* nm) {
    if (dtype == YOLO_F16) return pipe_dispatch_t<f16_t>(a, ks, 1, algo, st, nm);
    return pipe_dispatch_t<float>(a, ks, 1, algo, st, nm);
}
#elif YOLO_PIPE_PART == 6
int conv_pipe_dispatch_g(ConvArgs& a, int ks, int stride, int algo, hipStream_t st, const NameOut* nm) { return pipe_dispatch_t<bf16x3_t>(a, ks, stride, algo, st, nm); }
#elif YOLO_PIPE_PART == 7
int conv_pipe_dispatch_h(ConvArgs& a, int ks, int algo, hipStream_t st, const NameOut* nm) { return pipe_dispatch_t<bf16x3_t>(a, ks, 1, algo, st, nm); }
#elif YOLO_PIPE_PART == 8
int conv_pipe_dispatch_i(ConvArgs& a, int ks, int stride, int algo, hipStream_t st, const NameOut* nm) { return pipe_dispatch_t<f16x3_t>(a, ks, stride, algo, st, nm); }
#elif YOLO_PIPE_PART == 9
int conv_pipe_dispatch_j(ConvArgs& a, int ks, int algo, hipStream_t st, const NameOut* nm) { return pipe_dispatch_t<f16x3_t>(a, ks, 1, algo, st, nm); }
#elif YOLO_PIPE_PART == 2
int conv_pipe_dispatch_c(ConvArgs& a, int ks, int stride, int dtype, int algo, hipStream_t st, const NameOut* nm) {
    if (dtype == YOLO_F16) return pipe_dispatch_t<f16_t>(a, ks, stride, algo, st, nm);
    return pipe_dispatch_t<float>(a, ks, stride, algo, st, nm);
}
#else
int conv_pipe_dispatch(ConvArgs& a, int ks, int stride, int dtype, int algo, hipStream_t st, const NameOut* nm) {
    if ((ks != 1 && ks != 2 && ks != 3) || (stride != 1 && !(ks == 3 && stride == 2))) return YOLO_EUNSUPPORTED;
    if (a.d2s && ks != 2) return YOLO_EUNSUPPORTED;
    if (!dtype_split(dtype) && (a.Cin * elem_size(dtype)) % 64) return YOLO_EUNSUPPORTED;      // (split planes are padded to whole chunks by the caller)
    if (ks == 1 && a.nchunks < 2) return YOLO_EUNSUPPORTED;     // a 1x1 needs >= 2 phases; a 3x3 has 9 per chunk
    if ((long long)a.N * a.H * a.W * a.x_ps * elem_size(dtype) >= 0xffffff00LL) return YOLO_EUNSUPPORTED;
    if (dtype == YOLO_BF16) return pipe_dispatch_t<bf16_t>(a, ks, stride, algo, st, nm);
    if (dtype == YOLO_BF16X3) return ks == 2 ? YOLO_EUNSUPPORTED : conv_pipe_dispatch_g(a, ks, stride, algo, st, nm);
    if (dtype == YOLO_F16X3) return ks == 2 ? YOLO_EUNSUPPORTED : conv_pipe_dispatch_i(a, ks, stride, algo, st, nm);
    return conv_pipe_dispatch_c(a, ks, stride, dtype, algo, st, nm);
}
#endif
