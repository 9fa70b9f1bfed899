// Shared epilogue of the implicit-GEMM convolutions: folded BN (scale, bias), LeakyReLU, residual
// add, one rounding to the activation dtype, NHWC store.
//
// The MFMA accumulator holds, per lane, 4 consecutive couts of ONE pixel (D rows = cout), which
// as a direct store is an 8-byte write at a pixel-strided address and makes every residual load a
// dependent round trip.  Instead each wave transposes its 32-pixel x (MI*32)-cout slab through a
// private LDS scratch (fp32, row stride padded by 16 B: conflict-free b128 writes and reads) so
// that afterwards a lane owns 16 contiguous output bytes of a pixel row:
//   * scale/bias are loaded once per wave tile (the lane's couts do not change across rows),
//   * all residual loads of a slab are issued back-to-back (one latency, not 16),
//   * loads and stores are 16 B per lane, LPR lanes covering one contiguous cout run of a pixel.
// Rounding order is unchanged: fp32 (acc*scale+bias) -> LeakyReLU -> + residual (fp32) -> round.
#pragma once
#include "conv_args.h"
#include <type_traits>
#include "stamp.h"

// bytes of LDS scratch one wave needs (max over MI in {1,2}): 32 rows x (64*4+16) + 2 x 32 x 8 (output offsets) + 2 x 32 x 8
// (residual offsets, when the residual's strides differ from the output's)
#define YOLO_EPI_WAVE_BYTES 9728
// (wave tiles of more than 64 couts: the scratch row grows with MI)
#define YOLO_EPI_WAVE_BYTES_MI(MI) ((MI) <= 2 ? YOLO_EPI_WAVE_BYTES : 32 * ((MI) * 128 + 16) + 4 * 32 * 8)

// (wave_lds_fence(): common.h)
// STATS (bf16, transposed path only; the host checks): BatchNorm batch statistics of the training step taken here instead
// of in a pass of their own over the tensor.  After the transpose a lane owns 8 channels of one pixel row, so the column
// sums are plain per-lane accumulations over the lane's rows; the 64 / LPR lanes that share the channels are combined
// through the wave's scratch once per tile and the wave writes ONE partial row (no atomics: the atomic version of this
// fusion doubled the step time).  1: sum(v), sum(v^2) of the output; 2: the backward sums of the BatchNorm whose output
// gradient this kernel produces (a data gradient): da = v * lrelu'(gamma*xhat + beta), sum(da), sum(da * xhat).
template <typename T, int MI, int NI, int STATS = 0>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[MI][NI], const long long (&yoff)[NI], char* wsm,
                                              const ConvArgs& a, int co_w, int lane, const long long* roff = nullptr,
                                              float* srow = nullptr) {
    constexpr int WN = MI * 32;                 // couts of the wave tile
    constexpr int RS = WN * 4 + 16;             // fp32 row stride in the scratch (bytes)
    constexpr int ES = (int)sizeof(T);
    constexpr int CPL = 16 / ES;                // couts per lane after the transpose
    constexpr int LPR = WN / CPL;               // lanes per pixel row
    constexpr int RPP = 64 / LPR;               // rows per pass
    constexpr int NPASS = 32 / RPP;
    static_assert(32 * RS + 4 * 32 * 8 <= YOLO_EPI_WAVE_BYTES_MI(MI), "scratch size");
#ifdef YOLO_LAB
    if (a.lab & 4) return;                      // (lab: YOLO_EPI_AB ablation bits, conv_args.h)
#endif
    const int l31 = lane & 31, h = lane >> 5;
    const float slope = a.slope;
    // scale == bias == nullptr: identity epilogue (the training step's raw convolutions and data gradients: BN and the
    // activation are separate passes there) -- skips two loads and three VALU operations per output
    const bool ident = a.scale == nullptr;

    if (a.out_f32 && a.res == nullptr && (a.Cout % 2) == 0 && (a.y_ps % 2) == 0 && (a.y_bs % 2) == 0 &&
        ((size_t)a.y % 8) == 0) {
        // ---- fp32 head logits (YOLOOutput, Cout = A*C = 90: rows of 360 bytes, 8-byte aligned) ---------------
        // same LDS transpose, then 8-byte stores: a wave writes 2 pixel rows x (WN couts x 4 B) contiguous bytes per
        // pass instead of 4-byte stores scattered over 32 rows (the per-element path below)
        constexpr int LPR2 = WN / 2, RPP2 = 64 / LPR2, NPASS2 = 32 / RPP2;
        const int col2 = lane % LPR2, rowa = lane / LPR2;
        const int co2 = co_w + col2 * 2;
        const bool ok2 = co2 < a.Cout;
        const float sc0 = ident ? 1.f : a.scale[co2], sc1 = ident ? 1.f : a.scale[co2 + 1];
        const float bi0 = ident ? 0.f : a.bias[co2], bi1 = ident ? 0.f : a.bias[co2 + 1];
        long long* ytab2 = (long long*)(wsm + 32 * RS);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            if (h == 0) ytab2[l31] = yoff[ni];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
                    *(f32x4*)(wsm + l31 * RS + (mi * 32 + 8 * g + 4 * h) * 4) = v;
                }
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < NPASS2; ++k) {
                const int row = rowa + k * RPP2;
                const long long yo2 = ok2 ? ytab2[row] : -1;
                const float2 t2 = *(const float2*)(wsm + row * RS + col2 * 8);
                float2 o2;
                o2.x = ident ? t2.x : leaky(t2.x * sc0 + bi0, slope);
                o2.y = ident ? t2.y : leaky(t2.y * sc1 + bi1, slope);
                if (yo2 >= 0) *(float2*)(a.y + (yo2 + co2) * 4) = o2;
            }
        }
        return;
    }
    if (a.out_f32 || (a.Cout % CPL) != 0 || ((a.y_ps * ES) % 16) != 0 || ((a.y_bs * ES) % 16) != 0) {
        // ---- generic path: arbitrary Cout / strides / fp32 logits (small head-output layers) -----
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            if (yoff[ni] < 0) continue;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = co_w + mi * 32 + 8 * g + 4 * h;
                    if (co >= a.Cout) continue;
                    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
                    if (!ident) { sc = *(const f32x4*)(a.scale + co); bi = *(const f32x4*)(a.bias + co); }
                    const long long o = yoff[ni] + co;
                    const long long ro = (a.res && roff) ? roff[ni] + co : o;       // (dense residual, strided y)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (co + e >= a.Cout) continue;
                        float t = acc[mi][ni][4 * g + e];
                        if (!ident) t = leaky(t * sc[e] + bi[e], slope);
                        if (a.out_f32) {
                            ((float*)a.y)[o + e] = t;
                        } else if constexpr (ES == 2) {
                            if (a.res) t += Elem<T>::lo(((const uint16_t*)a.res)[ro + e]);
                            if constexpr (IsSplit<T>::value) {
                                if (a.res) t += Elem<T>::lo(((const uint16_t*)a.res)[ro + e + a.r_lo]);
                            }
                            const uint16_t b16 = (uint16_t)(Elem<T>::pack2(t, 0.f) & 0xffffu);
                            ((uint16_t*)a.y)[o + e] = b16;
                            if (a.up2) {
                                ((uint16_t*)a.y)[o + e + a.y_ps] = b16;
                                ((uint16_t*)a.y)[o + e + 2LL * a.Wo * a.y_ps] = b16;
                                ((uint16_t*)a.y)[o + e + (2LL * a.Wo + 1) * a.y_ps] = b16;
                            }
                            if constexpr (IsSplit<T>::value) {          // the lo plane: what the hi value's rounding left
                                const uint16_t l16 = (uint16_t)(Elem<T>::pack2(t - Elem<T>::lo(b16), 0.f) & 0xffffu);
                                const long long ol = o + e + a.y_lo;
                                ((uint16_t*)a.y)[ol] = l16;
                                if (a.up2) {
                                    ((uint16_t*)a.y)[ol + a.y_ps] = l16;
                                    ((uint16_t*)a.y)[ol + 2LL * a.Wo * a.y_ps] = l16;
                                    ((uint16_t*)a.y)[ol + (2LL * a.Wo + 1) * a.y_ps] = l16;
                                }
                            }
                        } else {
                            if (a.res) t += ((const float*)a.res)[ro + e];
                            ((float*)a.y)[o + e] = t;
                            if (a.up2) {
                                ((float*)a.y)[o + e + a.y_ps] = t;
                                ((float*)a.y)[o + e + 2LL * a.Wo * a.y_ps] = t;
                                ((float*)a.y)[o + e + (2LL * a.Wo + 1) * a.y_ps] = t;
                            }
                        }
                    }
                }
            }
        }
        return;
    }

    // ---- transposed path ---------------------------------------------------------------------------
    const int col = lane % LPR;
    const int row0 = lane / LPR;
    const int co = co_w + col * CPL;
    const bool co_ok = co < a.Cout;
    // element offset of this lane's channel run inside its pixel; sub-pixel mode (a.d2s, the stride-2 data gradient):
    // channel block ph = co / C of 4 x C is pixel (ph >> 1, ph & 1) of the 2x2 output patch (C % CPL == 0 is checked
    // by the host)
    long long cofs = co;
    if (a.d2s) {
        const int C4 = a.Cout >> 2, ph = co / C4;
        cofs = (long long)((ph >> 1) * 2 * a.Wo + (ph & 1)) * C4 + (co - ph * C4);
    }
    float sc[CPL], bi[CPL];
#pragma unroll
    for (int q = 0; q < CPL / 4; ++q) {
        f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
#ifdef YOLO_LAB
        if (!ident && !(a.lab & 8)) {
#else
        if (!ident) {
#endif
            s4 = *(const f32x4*)(a.scale + co + 4 * q);     // arrays are padded to the cout tile
            b4 = *(const f32x4*)(a.bias + co + 4 * q);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[4 * q + e] = s4[e]; bi[4 * q + e] = b4[e]; }
    }
    long long* ytab = (long long*)(wsm + 32 * RS);
    long long* rtab = ytab + 64;
    const bool has_res = a.res != nullptr;
    // the residual is a dense (N,Ho,Wo,Cout) tensor; when y is a channel slice of a wider buffer its offsets differ
    const bool res_sep = has_res && roff != nullptr && (a.r_ps != a.y_ps || a.r_bs != a.y_bs);
    float ssum[CPL], qsum[CPL], smu[CPL], sis[CPL], sga[CPL], sbe[CPL];
    if constexpr (STATS != 0) {
        static_assert(IsBf16<T>::value, "statistics epilogue: bf16 only");
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
            ssum[e] = qsum[e] = 0.f;
            if (STATS == 2) {
                smu[e] = co_ok ? a.s_mean[co + e] : 0.f; sis[e] = co_ok ? a.s_invstd[co + e] : 0.f;
                sga[e] = co_ok ? a.s_gamma[co + e] : 0.f; sbe[e] = co_ok ? a.s_beta[co + e] : 0.f;
            }
        }
    }
    if (a.buf32) {
        // ---- (round 4) every load and store of this path is UNCONDITIONAL: a buffer access whose byte offset is -1 for a lane
        //      that has nothing to read or write (out of range: loads return zeros, stores are dropped).  With the accesses
        //      inside `if (offset >= 0)` branches the compiler cannot count the memory operations in flight, so its
        //      s_waitcnt for the residual it needs became vmcnt(0): EVERY pass waited for the stores of the pass before it
        //      and for the next slab's prefetch (seen in the ISA; the epilogue ran at 2.5x its instruction-issue bound).
        //      The host sets buf32 when the tensors' extents fit 31-bit byte offsets; the branching form below remains for
        //      larger ones. ---------------------------------------------------------------------------------------------
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(has_res ? a.res : a.y), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)(STATS == 2 ? a.s_y : a.y), 0, 0x7fffffff, 0x00020000);
        int yb[2][NPASS];                       // byte offset of the lane's 16-byte piece in y (-1: none)
        u32x4_t rb[2][NPASS], sb[2][NPASS];
        u32x4_t rl[IsSplit<T>::value ? 2 : 1][IsSplit<T>::value ? NPASS : 1];      // split types: the residual's lo plane
        const int ylo_b = IsSplit<T>::value ? (int)(a.y_lo * ES) : 0, rlo_b = IsSplit<T>::value ? (int)(a.r_lo * ES) : 0;
        // (round 5) The uniform decisions -- identity epilogue, residual, 2x2 up-sampled stores -- are taken ONCE around the slab loop:
        // the common combinations are compiled as specialisations (the flag a compile-time constant), the rest runs the generic
        // form with the flags read at run time (FLAG < 0).  Inside the loop the branches cut every pass into basic blocks: the
        // results of the scale / bias block were copied into the registers the residual block expects (4 v_mov_b64 per pass), the
        // up-sampling test cost a compare and three selects per pass, and the LDS addresses were recomputed per pass.  The
        // epilogue is bound by the vector pipe (two waves per SIMD, ~260 VALU instructions each per 32-pixel slab: stamp_probe.py).
        const char* const rbase = wsm + row0 * RS + col * CPL * 4;       // this lane's piece of row row0; pass k, quad q: + k * RPP * RS + q * 16
        char* const wbase = wsm + l31 * RS + 16 * h;                       // this lane's row of the slab; (mi, g): + (mi * 32 + 8 * g) * 4
        auto run = [&](auto id_c, auto rs_c, auto up_c) __attribute__((always_inline)) {
        constexpr int ID_ = decltype(id_c)::value, RS_ = decltype(rs_c)::value, UP_ = decltype(up_c)::value;
        const bool f_ident = ID_ < 0 ? ident : (bool)ID_, f_res = RS_ < 0 ? has_res : (bool)RS_, f_up2 = UP_ < 0 ? (bool)a.up2 : (bool)UP_;
        auto prefetch_b = [&](int ni) __attribute__((always_inline)) {
            if (h == 0) ytab[(ni & 1) * 32 + l31] = yoff[ni];
            if (res_sep && h == 1) rtab[(ni & 1) * 32 + l31] = roff[ni];
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < NPASS; ++k) {
                const long long y_ = ytab[(ni & 1) * 32 + row0 + k * RPP];
                const bool ok = co_ok && y_ >= 0;
                yb[ni & 1][k] = ok ? (int)((y_ + cofs) * ES) : -1;
#ifdef YOLO_LAB
                if (a.lab & 1) yb[ni & 1][k] = -1;
#endif
                if (f_res) {
                    const long long r_ = res_sep ? rtab[(ni & 1) * 32 + row0 + k * RPP] : y_;
#ifdef YOLO_LAB
                    rb[ni & 1][k] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (ok && !(a.lab & 2)) ? (int)((r_ + cofs) * ES) : -1, 0, 0);
#else
                    rb[ni & 1][k] = __builtin_amdgcn_raw_buffer_load_b128(rrs, ok ? (int)((r_ + cofs) * ES) : -1, 0, 0);
#endif
                    if constexpr (IsSplit<T>::value)
                        rl[ni & 1][k] = __builtin_amdgcn_raw_buffer_load_b128(rrs, ok ? (int)((r_ + cofs) * ES) + rlo_b : -1, 0, 0);
                }
                if constexpr (STATS == 2) sb[ni & 1][k] = __builtin_amdgcn_raw_buffer_load_b128(srs, yb[ni & 1][k], 0, 0);
            }
        };
        const int up_a = (int)(a.y_ps * ES), up_b = (int)(2LL * a.Wo * a.y_ps * ES);      // (up2: the other pixels of the 2x2 patch)
        STAMP(8);
        prefetch_b(0);
        STAMP(9);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            if (ni + 1 < NI) prefetch_b(ni + 1);
#ifdef YOLO_LAB
            if (!(a.lab & 16))
#endif
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
                    *(f32x4*)(wbase + (mi * 32 + 8 * g) * 4) = v;
                }
            wave_lds_fence();
            if (ni < 3) STAMP(10 + 2 * ni);
#pragma unroll
            for (int k = 0; k < NPASS; ++k) {
                float v[CPL];
#pragma unroll
                for (int q = 0; q < CPL / 4; ++q) {
#ifdef YOLO_LAB
                    f32x4 t4 = {acc[0][ni][4 * q], acc[0][ni][4 * q + 1], acc[0][ni][4 * q + 2], acc[0][ni][4 * q + 3]};      // (ablation 16: no transpose, wrong values)
                    if (!(a.lab & 16)) t4 = *(const f32x4*)(rbase + k * RPP * RS + q * 16);
#else
                    const f32x4 t4 = *(const f32x4*)(rbase + k * RPP * RS + q * 16);
#endif
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * q + e] = t4[e];
                }
                if (!f_ident) {
#pragma unroll
                    for (int e = 0; e < CPL; ++e) {
                        const float t = v[e] * sc[e] + bi[e];
                        v[e] = leaky(t, slope);
                    }
                }
                const int ob = yb[ni & 1][k];
                u32x4_t ov;
                if constexpr (ES == 2) {
                    if (f_res) {
                        const uint32_t w[4] = {rb[ni & 1][k].x, rb[ni & 1][k].y, rb[ni & 1][k].z, rb[ni & 1][k].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v[2 * q] += Elem<T>::lo(w[q]);
                            v[2 * q + 1] += Elem<T>::hi(w[q]);
                        }
                        if constexpr (IsSplit<T>::value) {
                            const uint32_t wl[4] = {rl[ni & 1][k].x, rl[ni & 1][k].y, rl[ni & 1][k].z, rl[ni & 1][k].w};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                v[2 * q] += Elem<T>::lo(wl[q]);
                                v[2 * q + 1] += Elem<T>::hi(wl[q]);
                            }
                        }
                    }
                    ov.x = Elem<T>::pack2(v[0], v[1]); ov.y = Elem<T>::pack2(v[2], v[3]);
                    ov.z = Elem<T>::pack2(v[4], v[5]); ov.w = Elem<T>::pack2(v[6], v[7]);
                    if constexpr (IsSplit<T>::value) {
                        // the lo plane = what the hi values' rounding left, rounded once more: 16 significant bits per stored value
                        const uint32_t hw[4] = {ov.x, ov.y, ov.z, ov.w};
                        u32x4_t ol;
                        ol.x = Elem<T>::pack2(v[0] - Elem<T>::lo(hw[0]), v[1] - Elem<T>::hi(hw[0]));
                        ol.y = Elem<T>::pack2(v[2] - Elem<T>::lo(hw[1]), v[3] - Elem<T>::hi(hw[1]));
                        ol.z = Elem<T>::pack2(v[4] - Elem<T>::lo(hw[2]), v[5] - Elem<T>::hi(hw[2]));
                        ol.w = Elem<T>::pack2(v[6] - Elem<T>::lo(hw[3]), v[7] - Elem<T>::hi(hw[3]));
                        __builtin_amdgcn_raw_buffer_store_b128(ol, yrs, ob >= 0 ? ob + ylo_b : -1, 0, 0);
                        if (f_up2) {
                            __builtin_amdgcn_raw_buffer_store_b128(ol, yrs, ob >= 0 ? ob + ylo_b + up_a : -1, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(ol, yrs, ob >= 0 ? ob + ylo_b + up_b : -1, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(ol, yrs, ob >= 0 ? ob + ylo_b + up_b + up_a : -1, 0, 0);
                        }
                    }
                    if constexpr (STATS != 0) {
                        const uint32_t ow[4] = {ov.x, ov.y, ov.z, ov.w};
                        float vr[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) vr[e] = ((e & 1) ? Elem<T>::hi(ow[e >> 1]) : Elem<T>::lo(ow[e >> 1]));
                        if constexpr (STATS == 1) {
                            // (lanes past the end of the tensor compute a copy of the tile's first pixel: masked like the store)
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float m = ob >= 0 ? vr[e] : 0.f;
                                ssum[e] += m; qsum[e] += m * m;
                            }
                        } else {
                            const uint32_t yw[4] = {sb[ni & 1][k].x, sb[ni & 1][k].y, sb[ni & 1][k].z, sb[ni & 1][k].w};
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float yv = ((e & 1) ? Elem<T>::hi(yw[e >> 1]) : Elem<T>::lo(yw[e >> 1]));
                                const float xh = (yv - smu[e]) * sis[e];
                                const float da = (ob >= 0) ? vr[e] * ((sga[e] * xh + sbe[e]) > 0.f ? 1.f : a.s_slope) : 0.f;
                                ssum[e] += da; qsum[e] += da * xh;
                            }
                        }
                    }
                } else {
                    if (f_res) {
                        v[0] += __uint_as_float(rb[ni & 1][k].x); v[1] += __uint_as_float(rb[ni & 1][k].y);
                        v[2] += __uint_as_float(rb[ni & 1][k].z); v[3] += __uint_as_float(rb[ni & 1][k].w);
                    }
                    ov.x = __float_as_uint(v[0]); ov.y = __float_as_uint(v[1]); ov.z = __float_as_uint(v[2]); ov.w = __float_as_uint(v[3]);
                }
                __builtin_amdgcn_raw_buffer_store_b128(ov, yrs, ob, 0, 0);
                if (f_up2) {                            // the other three pixels of the 2x2 patch (row pitch 2*Wo pixels)
                    __builtin_amdgcn_raw_buffer_store_b128(ov, yrs, ob >= 0 ? ob + up_a : -1, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(ov, yrs, ob >= 0 ? ob + up_b : -1, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(ov, yrs, ob >= 0 ? ob + up_b + up_a : -1, 0, 0);
                }
            }
            if (ni < 3) STAMP(11 + 2 * ni);
        }
        };
        typedef std::integral_constant<int, 0> F0;
        typedef std::integral_constant<int, 1> F1;
        typedef std::integral_constant<int, -1> FR;
        if (a.up2) run(FR{}, FR{}, FR{});                                   // (two transition convolutions per net)
        else if (!ident) { if (has_res) run(F0{}, F1{}, F0{}); else run(F0{}, F0{}, F0{}); }
        else { if (has_res) run(F1{}, F1{}, F0{}); else run(F1{}, F0{}, F0{}); }
    } else {
    // The residual loads of slab ni+1 are issued BEFORE slab ni is processed (two register sets), so their latency
    // hides under the previous slab's transpose/arithmetic/stores -- the phase stamps showed one exposed memory
    // round trip per 32-pixel slab doubling the epilogue time.  (Prefetching all slabs at once spills on the
    // 8-wave variants.)  Output offsets go through a small LDS table: the transpose changes which pixel a lane owns.
    long long yo[2][NPASS];
    uint4 rv[2][NPASS];
    uint4 rvl[IsSplit<T>::value ? 2 : 1][IsSplit<T>::value ? NPASS : 1];      // split types: the residual's lo plane
    uint4 sv[2][NPASS];                     // STATS == 2: the forward raw outputs under this lane's gradients
    auto prefetch = [&](int ni) {
        if (h == 0) ytab[(ni & 1) * 32 + l31] = yoff[ni];
        if (res_sep && h == 1) rtab[(ni & 1) * 32 + l31] = roff[ni];
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < NPASS; ++k) yo[ni & 1][k] = co_ok ? ytab[(ni & 1) * 32 + row0 + k * RPP] : -1;
        if (has_res) {
#pragma unroll
            for (int k = 0; k < NPASS; ++k) {
                rv[ni & 1][k] = make_uint4(0, 0, 0, 0);
                const long long ro = res_sep ? rtab[(ni & 1) * 32 + row0 + k * RPP] : yo[ni & 1][k];
                if (yo[ni & 1][k] >= 0) rv[ni & 1][k] = *(const uint4*)(a.res + (ro + cofs) * ES);
                if constexpr (IsSplit<T>::value) {
                    rvl[ni & 1][k] = make_uint4(0, 0, 0, 0);
                    if (yo[ni & 1][k] >= 0) rvl[ni & 1][k] = *(const uint4*)(a.res + (ro + cofs + a.r_lo) * ES);
                }
            }
        }
        if constexpr (STATS == 2) {
#pragma unroll
            for (int k = 0; k < NPASS; ++k) {
                sv[ni & 1][k] = make_uint4(0, 0, 0, 0);
                if (yo[ni & 1][k] >= 0) sv[ni & 1][k] = *(const uint4*)(a.s_y + (yo[ni & 1][k] + cofs) * ES);
            }
        }
    };
    prefetch(0);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        if (ni + 1 < NI) prefetch(ni + 1);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
                *(f32x4*)(wsm + l31 * RS + (mi * 32 + 8 * g + 4 * h) * 4) = v;
            }
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
            float v[CPL];
#pragma unroll
            for (int q = 0; q < CPL / 4; ++q) {
                const f32x4 t4 = *(const f32x4*)(wsm + (row0 + k * RPP) * RS + (col * CPL + 4 * q) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * q + e] = t4[e];
            }
            if (!ident) {
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    const float t = v[e] * sc[e] + bi[e];
                    v[e] = leaky(t, slope);
                }
            }
            uint4 ov;
            if constexpr (ES == 2) {
                if (has_res) {
                    const uint32_t w[4] = {rv[ni & 1][k].x, rv[ni & 1][k].y, rv[ni & 1][k].z, rv[ni & 1][k].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[2 * q] += Elem<T>::lo(w[q]);
                        v[2 * q + 1] += Elem<T>::hi(w[q]);
                    }
                    if constexpr (IsSplit<T>::value) {
                        const uint32_t wl[4] = {rvl[ni & 1][k].x, rvl[ni & 1][k].y, rvl[ni & 1][k].z, rvl[ni & 1][k].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v[2 * q] += Elem<T>::lo(wl[q]);
                            v[2 * q + 1] += Elem<T>::hi(wl[q]);
                        }
                    }
                }
                ov = make_uint4(Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3]), Elem<T>::pack2(v[4], v[5]),
                                Elem<T>::pack2(v[6], v[7]));
                if constexpr (IsSplit<T>::value) {
                    if (yo[ni & 1][k] >= 0) {
                        const uint32_t hw[4] = {ov.x, ov.y, ov.z, ov.w};
                        const uint4 ol = make_uint4(Elem<T>::pack2(v[0] - Elem<T>::lo(hw[0]), v[1] - Elem<T>::hi(hw[0])),
                                                    Elem<T>::pack2(v[2] - Elem<T>::lo(hw[1]), v[3] - Elem<T>::hi(hw[1])),
                                                    Elem<T>::pack2(v[4] - Elem<T>::lo(hw[2]), v[5] - Elem<T>::hi(hw[2])),
                                                    Elem<T>::pack2(v[6] - Elem<T>::lo(hw[3]), v[7] - Elem<T>::hi(hw[3])));
                        char* dl = a.y + (yo[ni & 1][k] + cofs + a.y_lo) * ES;
                        *(uint4*)dl = ol;
                        if (a.up2) {
                            *(uint4*)(dl + a.y_ps * ES) = ol;
                            *(uint4*)(dl + 2LL * a.Wo * a.y_ps * ES) = ol;
                            *(uint4*)(dl + (2LL * a.Wo + 1) * a.y_ps * ES) = ol;
                        }
                    }
                }
                if constexpr (STATS != 0) {
                    // the statistics of the STORED (bf16-rounded) values: what the separate reduction pass would read
                    const uint32_t ow[4] = {ov.x, ov.y, ov.z, ov.w};
                    float vr[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) vr[e] = ((e & 1) ? Elem<T>::hi(ow[e >> 1]) : Elem<T>::lo(ow[e >> 1]));
                    if constexpr (STATS == 1) {
                        // (lanes past the end of the tensor compute a copy of the tile's first pixel: masked like the store)
                        if (yo[ni & 1][k] >= 0) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) { ssum[e] += vr[e]; qsum[e] += vr[e] * vr[e]; }
                        }
                    } else {
                        const uint32_t yw[4] = {sv[ni & 1][k].x, sv[ni & 1][k].y, sv[ni & 1][k].z, sv[ni & 1][k].w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float yv = ((e & 1) ? Elem<T>::hi(yw[e >> 1]) : Elem<T>::lo(yw[e >> 1]));
                            const float xh = (yv - smu[e]) * sis[e];
                            const float da = (yo[ni & 1][k] >= 0) ? vr[e] * ((sga[e] * xh + sbe[e]) > 0.f ? 1.f : a.s_slope) : 0.f;
                            ssum[e] += da; qsum[e] += da * xh;
                        }
                    }
                }
            } else {
                if (has_res) {
                    v[0] += __uint_as_float(rv[ni & 1][k].x); v[1] += __uint_as_float(rv[ni & 1][k].y);
                    v[2] += __uint_as_float(rv[ni & 1][k].z); v[3] += __uint_as_float(rv[ni & 1][k].w);
                }
                ov = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                                __float_as_uint(v[3]));
            }
            if (yo[ni & 1][k] >= 0) {
                char* dst = a.y + (yo[ni & 1][k] + cofs) * ES;
                *(uint4*)dst = ov;
                if (a.up2) {                            // the other three pixels of the 2x2 patch (row pitch 2*Wo pixels)
                    *(uint4*)(dst + a.y_ps * ES) = ov;
                    *(uint4*)(dst + 2LL * a.Wo * a.y_ps * ES) = ov;
                    *(uint4*)(dst + (2LL * a.Wo + 1) * a.y_ps * ES) = ov;
                }
            }
        }
    }
    }
    if constexpr (STATS != 0) {
        // combine the RPP lanes that share a channel run: every lane parks its 16 sums in the scratch (64 B per lane), then
        // lane o (and o + 64) of the LPR * 16 outputs adds its column and writes it into the wave's partial row
        float* sc4 = (float*)wsm;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc4[lane * 16 + e] = ssum[e]; sc4[lane * 16 + 8 + e] = qsum[e]; }
        wave_lds_fence();
#pragma unroll
        for (int o = lane; o < LPR * 16; o += 64) {
            const int cx = o >> 4, val = o & 15;
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < RPP; ++r) t += sc4[(r * LPR + cx) * 16 + val];
            const int c = co_w + cx * CPL + (val & 7);
            if (c < a.Cout_pad) srow[(val >> 3) * a.Cout_pad + c] = t;
        }
    }
}
