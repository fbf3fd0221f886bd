// Epilogue shared by the split-fp16 convolution kernels (gl_conv_h3.hip: tap-gather form, gl_conv_halo.hip: halo form):
//   v = act(acc * scale[c] + shift[c])  [-> PixelNorm over the channels of a position]  -> split layout / fp32 / planar store.
// C tile (16 x 16): column (position) = lane & 15, row (channel) = 4 * (lane >> 4) + reg; a wave owns TC x TP tiles, the workgroup
// WC x WP waves.  o4[j]: output position index of the lane's column in position tile j (-1: none).
#pragma once
#include "gl_conv.h"
#include <type_traits>

namespace gl_h3 {

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

// returns whether a split store of this lane had to clamp to the fp16 range.
// pool_dj: position tile j ^ pool_dj holds the row below / above tile j's (fused tap + pool only; 0 when the caller never fuses one)
template <int WC, int WP, int TC, int TP>
__device__ __forceinline__ bool epilogue(const GlGatherConv &p, v4f (&acc)[TC][TP], int c0, int wc, int wp_, int lane, const int (&o4)[TP], char *smem,
                                         int pool_dj = 0)
{
    const int frow = lane & 15, fk = lane >> 4;
    const float relu_floor = p.act == 1 ? 0.0f : -__builtin_inff();
    const float neg_slope = p.act == 2 ? 0.2f : 1.0f;
    bool saturated = false;
    // fused PixelNorm: all channels of a position sit in this workgroup (host-checked: cols <= HTC).  Pass 1 activates in place, rounds every
    // value to what the split layout holds (hi + lo) and sums the squares per position in ONE canonical order -- per 16-channel tile: the
    // lane's 4 channels as an fmaf chain, then (g0 + g1) + (g2 + g3) over the 4 lane groups; then a balanced binary tree over the tiles --
    // which pixelnorm_split_kernel (gl_pggan.hip) reproduces on stored values: fused or not, and whatever the tile shape, the stored
    // activations are bit-identical (a pass of another size may pick another tile).  Pass 2 below stores v * inv.
    // The fused epilogues (PixelNorm, LPIPS tap + pool, toRGB) exist only in the tiles whose waves hold 4 x 4 accumulator tiles.  In the 8 x 4 forms
    // (<2,4,8,4>, <1,8,8,4>) they spill hundreds of registers: measured, PixelNorm fused into the 256-channel tile made PGGAN-256 2 % SLOWER than
    // the separate pass, the relu3_3 tap fused there made VGG16 2 % slower -- and code the K loop never runs can still cost it an accumulator
    // (tools/check_loop_spills.py).  The launcher never asks those tiles for a fused epilogue (gl_conv_h3_tile_channels returns 0 for them).
    constexpr bool kFusable = TC <= 4;
    const bool pixnorm = kFusable && p.pixnorm_act > 0.0f;
    const bool tap = kFusable && p.tap_V != nullptr;
    float pinv[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) pinv[j] = 1.0f;
    if constexpr (kFusable) if (pixnorm || tap) {
        float tss[TC][TP];
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int ch = c0 + wc * 16 * TC + i * 16 + 4 * fk;
            const bool real = ch < p.cols;
            const int chm = ch < p.cmod ? ch : ch % p.cmod;     // ch and cmod are multiples of 4: (ch + r) % cmod = chm + r
            float sc[4], sh[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc[r] = real ? p.scale[chm + r] : 0.0f; sh[r] = real ? p.shift[chm + r] : 0.0f; }
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                float s1 = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = fmaf(acc[i][j][r], sc[r], sh[r]);
                    t = fmaxf(fmaxf(t, t * neg_slope), relu_floor);
                    const float c = fminf(fmaxf(t, -65504.0f), 65504.0f);
                    saturated |= (c != t) && (o4[j] >= 0);
                    const _Float16 h = (_Float16)c;
                    const float vq = __fadd_rn((float)h, (float)(_Float16)__fsub_rn(c, (float)h));
                    acc[i][j][r] = vq;
                    s1 = fmaf(vq, vq, s1);
                }
                s1 += __shfl_xor(s1, 16, 64);
                s1 += __shfl_xor(s1, 32, 64);
                tss[i][j] = s1;
            }
        }
        // the (up to 16) tile sums of a position are combined as a balanced binary tree over the tile index, absent tiles counting 0
        float ss[TP];
        constexpr int NT = WC * TC;                                      // tiles of 16 channels in the workgroup tile (4, 8 or 16)
        static_assert(NT == 4 || NT == 8 || NT == 16, "fused PixelNorm: 4, 8 or 16 channel tiles");
        if constexpr (WC > 1) {
            float *xs = reinterpret_cast<float *>(smem + 8192);          // [WC * TC][WP][TP][16]
            if (fk == 0) {
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j) xs[(((wc * TC + i) * WP + wp_) * TP + j) * 16 + frow] = tss[i][j];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                float t[NT];
#pragma unroll
                for (int gt = 0; gt < NT; ++gt) t[gt] = xs[((gt * WP + wp_) * TP + j) * 16 + frow];
#pragma unroll
                for (int w = 1; w < NT; w *= 2)
#pragma unroll
                    for (int k = 0; k < NT; k += 2 * w) t[k] = __fadd_rn(t[k], t[k + w]);
                ss[j] = t[0];
            }
        } else {
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                float t[NT];
#pragma unroll
                for (int gt = 0; gt < NT; ++gt) t[gt] = tss[gt][j];
#pragma unroll
                for (int w = 1; w < NT; w *= 2)
#pragma unroll
                    for (int k = 0; k < NT; k += 2 * w) t[k] = __fadd_rn(t[k], t[k + w]);
                ss[j] = t[0];
            }
        }
        if (tap) {
            // ---- LPIPS tap (+ 2 x 2 max-pool) from the rounded activations in acc; nothing else is stored
#pragma unroll
            for (int j = 0; j < TP; ++j) pinv[j] = __fdiv_rn(p.tap_scale, __fadd_rn(__fsqrt_rn(ss[j]), p.tap_eps));
            const int HW = p.Ho * p.Wo, Wp = p.Wo >> 1;
            int img[TP], pin[TP];
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int o = o4[j] >= 0 ? o4[j] : 0;
                img[j] = o / HW;
                pin[j] = o - img[j] * HW;
            }
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                const int ch = c0 + wc * 16 * TC + i * 16 + 4 * fk;
                if (ch >= p.cols) continue;
                float cf[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) cf[r] = p.tap_coef[ch + r];
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    if (o4[j] < 0) continue;
                    v4h oh, ol;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t = gl_tap_value(acc[i][j][r], pinv[j], cf[r]);
                        oh[r] = (_Float16)t;
                        ol[r] = (_Float16)__fsub_rn(t, (float)oh[r]);
                    }
                    const int64_t k = p.tap_off + (int64_t)pin[j] * p.cols + ch;
                    if (p.tap_fmt) {
                        *reinterpret_cast<v4h *>(gl_vrow_elem(p.tap_V, p.tap_ldv, p.tap_row0 + img[j], k)) = oh;
                    } else {
                        char *dst = p.tap_V + (p.tap_row0 + img[j]) * p.tap_ldv + (k >> 5) * 128 + (k & 31) * 2;
                        *reinterpret_cast<v4h *>(dst) = oh;
                        *reinterpret_cast<v4h *>(dst + 64) = ol;
                    }
                }
                if (p.tap_pool) {
                    auto pool = [&](auto dj_c) {
                        constexpr int DJ = decltype(dj_c)::value;
#pragma unroll
                        for (int j = 0; j < TP; ++j) {
                            if ((j & DJ) != 0) continue;                  // the lower row of a pair of tiles is handled with the upper one
                            float m[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                m[r] = fmaxf(acc[i][j][r], acc[i][(j ^ DJ) % TP][r]);
                                m[r] = fmaxf(m[r], __shfl_xor(m[r], 1, 64));   // the column next to it
                            }
                            if ((frow & 1) != 0 || o4[j] < 0) continue;
                            const int y = pin[j] / p.Wo, x = pin[j] - y * p.Wo;
                            char *dst = p.tap_pool + ((int64_t)img[j] * (HW >> 2) + (y >> 1) * Wp + (x >> 1)) * p.cols * 4 + (ch >> 5) * 128 + (ch & 31) * 2;
                            v4h hi, lo;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                hi[r] = (_Float16)m[r];
                                lo[r] = (_Float16)__fsub_rn(m[r], (float)hi[r]);
                            }
                            *reinterpret_cast<v4h *>(dst) = hi;
                            *reinterpret_cast<v4h *>(dst + 64) = lo;
                        }
                    };
                    if (pool_dj == 1) pool(std::integral_constant<int, 1>());
                    else pool(std::integral_constant<int, 2>());
                }
            }
            return saturated;
        }
        const float A = p.pixnorm_act;
#pragma unroll
        for (int j = 0; j < TP; ++j) pinv[j] = __fdiv_rn(A, __fsqrt_rn(__fadd_rn(__fdiv_rn(ss[j], (float)p.cols), __fmul_rn(__fmul_rn(1e-8f, A), A))));
        if (p.rgb_out) {
            // ---- toRGB on the normalised values (what the split store would hold: hi + lo), nothing else is stored
            float res[TP][4];
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                float t[TC][4];
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    const int ch = c0 + wc * 16 * TC + i * 16 + 4 * fk;
#pragma unroll
                    for (int c = 0; c < 4; ++c) t[i][c] = 0.0f;
                    if (ch < p.cols) {
                        float vv[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = __fmul_rn(acc[i][j][r], pinv[j]);
                            const float cl = fminf(fmaxf(v, -65504.0f), 65504.0f);
                            saturated |= (cl != v) && (o4[j] >= 0);
                            const _Float16 hi = (_Float16)cl;
                            const _Float16 lo = (_Float16)fmaf(acc[i][j][r], pinv[j], -(float)hi);
                            vv[r] = __fadd_rn((float)hi, (float)lo);
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (c < p.rgb_n) {
                                const float4 w = *reinterpret_cast<const float4 *>(p.rgb_w + (int64_t)c * p.cols + ch);
                                t[i][c] = fmaf(w.w, vv[3], fmaf(w.z, vv[2], fmaf(w.y, vv[1], fmaf(w.x, vv[0], 0.0f))));
                            }
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int w = 1; w < TC; w *= 2)
#pragma unroll
                        for (int k = 0; k < TC; k += 2 * w) t[k][c] = __fadd_rn(t[k][c], t[k + w][c]);
                    res[j][c] = t[0][c];
                }
            }
            if constexpr (WC > 1) {
                // the other channel half of the position sits in the partner wave: the top level of the tree over the tiles
                static_assert(WC == 2, "fused toRGB: one or two waves along the channels");
                float *xr = reinterpret_cast<float *>(smem + 8192 + 16384);          // [WP][TP][4][64]
                __syncthreads();
                if (wc == 1) {
#pragma unroll
                    for (int j = 0; j < TP; ++j)
#pragma unroll
                        for (int c = 0; c < 4; ++c) xr[((wp_ * TP + j) * 4 + c) * 64 + lane] = res[j][c];
                }
                __syncthreads();
                if (wc == 1) return saturated;
#pragma unroll
                for (int j = 0; j < TP; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) res[j][c] = __fadd_rn(res[j][c], xr[((wp_ * TP + j) * 4 + c) * 64 + lane]);
            }
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                float out4[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float sacc = res[j][c];
                    sacc = __fadd_rn(sacc, __shfl_xor(sacc, 16, 64));
                    sacc = __fadd_rn(sacc, __shfl_xor(sacc, 32, 64));
                    out4[c] = c < p.rgb_n ? fmaf(sacc, p.rgb_inv_act, p.rgb_b[c]) : 0.0f;
                }
                if (fk == 0 && o4[j] >= 0) *reinterpret_cast<float4 *>(p.rgb_out + (int64_t)o4[j] * 4) = make_float4(out4[0], out4[1], out4[2], out4[3]);
            }
            return saturated;
        }
    }
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const int ch = c0 + wc * 16 * TC + i * 16 + 4 * fk;     // first of this lane's 4 consecutive channels
        if (ch >= p.cols) continue;                             // cols is a multiple of 4 (host-checked)
        float sc[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (!pixnorm) {
            const int chm = ch < p.cmod ? ch : ch % p.cmod;     // ch and cmod are multiples of 4: (ch + r) % cmod = chm + r
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc[r] = p.scale[chm + r]; sh[r] = p.shift[chm + r]; }
        }
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int o = o4[j];
            if (o < 0) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (pixnorm) {
                    v[r] = __fmul_rn(acc[i][j][r], pinv[j]);
                } else {
                    float t = fmaf(acc[i][j][r], sc[r], sh[r]);
                    v[r] = fmaxf(fmaxf(t, t * neg_slope), relu_floor);
                }
            }
            if (p.out_mode == 2) {
                // split layout: [o][cols/32][hi 32 | lo 32] halves
                char *dst = reinterpret_cast<char *>(p.out) + (int64_t)o * p.cols * 4 + (ch >> 5) * 128 + (ch & 31) * 2;
                v4h hi, lo;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float c = fminf(fmaxf(v[r], -65504.0f), 65504.0f);
                    saturated |= (c != v[r]);
                    hi[r] = (_Float16)c;
                    // after the fused PixelNorm the low half is the exact residual of the product (one fma), as pixelnorm_split_kernel computes it
                    lo[r] = pixnorm ? (_Float16)fmaf(acc[i][j][r], pinv[j], -(float)hi[r]) : (_Float16)__fsub_rn(c, (float)hi[r]);
                }
                *reinterpret_cast<v4h *>(dst) = hi;
                *reinterpret_cast<v4h *>(dst + 64) = lo;
            } else if (p.planar) {
#pragma unroll
                for (int r = 0; r < 4; ++r) p.out[(int64_t)(ch + r) * p.ld_planar + o] = v[r];
            } else {
                *reinterpret_cast<float4 *>(p.out + (int64_t)o * p.cols + ch) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
    return saturated;
}

}  // namespace gl_h3
