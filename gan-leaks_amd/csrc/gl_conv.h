// Tap-gather implicit GEMM on the fp32 matrix cores: one kernel template for every dense
// convolution shape on the path (ConvTranspose2d k4 s1 p0 on a 1x1 input, the four sub-pixel
// phases of ConvTranspose2d k4 s2 p1, and -- later rounds -- 3x3 p1 convolutions).
#pragma once
#include "gl_common.h"

struct GlGatherConv {
    // input activations, NHWC fp32: [n_img][H >> up][W >> up][Cin]
    const float *in;
    unsigned in_bytes;          // size of the stored input tensor in bytes (filled by gl_launch_gather_conv; < 3 GiB)
    int64_t positions;          // n_img * H * W   (GEMM M: one row per base-grid position)
    int H, W, Cin;              // H, W: the convolution's input grid
    int up;                     // 1: `in` is stored at (H/2) x (W/2) and read through nearest-neighbour 2x upsampling
    // packed weights: [phases][cols_pad][K] fp32, K = ntaps * Cin contiguous, ordered by gl_conv_k_index
    const float *wpack;
    int cols;                   // real GEMM columns (output channels per position)
    int cols_pad;               // multiple of 128 (wide tile) or of 64 (narrow tile)
    int ntaps;
    // per phase: taps packed 2 bits each (value+1) -> input offset (dy,dx) in {-1,0,1}
    uint32_t tap_dy[4], tap_dx[4];
    // output, NHWC fp32: [n_img][Ho][Wo][cols]; position (y,x) of phase p writes (y*omul+oy[p], x*omul+ox[p])
    float *out;
    int Ho, Wo, omul;
    int oy[4], ox[4];
    // planar != 0: write column-major out[c * ld_planar + position] instead (single phase, identity position map)
    int planar;
    int64_t ld_planar;
    // gather_conv_h3 only: 0 = fp32 output (row-major or planar as above), 2 = split-fp16 layout [pos][cols/32][hi 32 | lo 32]
    int out_mode;
    int *sat_flag;              // filled by the launcher: counter of workgroups that clamped a split store
    // epilogue: v = act(acc * scale[c % cmod] + shift[c % cmod]) (+ residual[same index as out], row-major only);
    // act 0 none, 1 ReLU, 2 LeakyReLU(0.2), 3 tanh, 4 sigmoid
    const float *scale, *shift;
    int cmod;
    int act;
    const float *residual;
    const float *zero;          // >= 16 B of zeros (source of out-of-image taps)
    // gather_conv_h3 only, optional: the NEXT layer fused into this one's epilogue when that layer is a per-position GEMM over all of this
    // layer's 64 or 128 output channels with 48 columns (the generator's ConvTranspose2d(C -> 3, k4 s2 p1) in scatter form): the activations never
    // leave the registers, only tail_out[col * tail_ld + output position] = tail_scale * sum_c act[c] * w[c][col] is written.
    // tail_w: gl_pack_tail_weights_host layout.  `out` is ignored then.
    const void *tail_w;
    float *tail_out;
    int64_t tail_ld;
    float tail_scale;
    // gather_conv_h3 only, optional (out_mode 2, all `cols` channels inside one tile -- gl_conv_h3_tile_channels() says how many that is):
    // PGGAN's PixelNorm (gan_models/pggan/model_torch.py:25-31) applied to the activated outputs before they are stored,
    // v -> v / sqrt(mean_c v^2 / A^2 + 1e-8) with A = pixnorm_act the factor the stored activations carry; 0 = off
    float pixnorm_act;
    // gather_conv_h3 only, optional (same condition on the tile, see gl_conv_h3_tap_fusable): an LPIPS tap (attack_models/lpips.py:110-118,
    // normalize_tensor + the lin layer's weights) and the 2 x 2 max-pool that follows it, taken from the activated outputs in the epilogue:
    //   V[img][tap_off + pin * cols + c] = v / (|v|_channels + eps) * tap_coef[c]   (pin = position inside the image; tap_fmt 1: fp16 row
    //   of tap_ldv bytes, 0: split row), tap_pool[img][y/2][x/2][c] = max of the 2 x 2 window in the split layout.  `out` is NOT written.
    // The sum of squares over the channels uses the epilogue's canonical order (gl_conv_h3_epi.h), which lpips_tap_*_split_kernel repeat.
    // gather_conv_h3 only, with pixnorm_act > 0 and cols <= 128 (so that every tile shape a pass may get holds all channels): PGGAN's toRGB, the
    // 1 x 1 convolution to <= 4 image channels that follows the last block (gan_models/pggan/model_torch.py:60-62,84-88), taken from the normalised
    // activations while they are in registers: rgb_out[o * 4 + c] = rgb_b[c] + rgb_inv_act * sum_ch rgb_w[c * cols + ch] * v[ch]   (v = the values the
    // split store would hold; the sum in the epilogue's canonical order: fp32 fma chain over the 4 channels of a lane, balanced tree over the
    // 16-channel tiles, then (g0 + g1) + (g2 + g3) over the lane groups).  `out` is NOT written: the 2.1 GB of a 128-image pass at 256 x 256
    // never reach HBM.
    const float *rgb_w, *rgb_b;
    float *rgb_out;             // nullptr = off
    int rgb_n;
    float rgb_inv_act;
    char *tap_V;                // nullptr = off; base of the row buffer
    const float *tap_coef;
    int64_t tap_ldv, tap_off;   // tap_ldv: bytes per row, or (fp16 rows only) -(cells per row) for the K-blocked layout, see gl_vrow_elem
    int64_t tap_row0;           // buffer row of the pass's image 0
    int tap_fmt;
    char *tap_pool;
    float tap_scale, tap_eps;   // V = v * (tap_scale / (sqrt(sum v^2) + tap_eps)) * coef
};

// fp16 search rows (LPIPS feature rows for the nearest-neighbour search) come in two layouts, chosen by the row length alone so that writers and
// the search agree without a flag in the ABI (gl_vrow_blocked):
//   row-major   [row][K] halves                                   -- rows shorter than 2 MiB (images up to 64 x 64 ... 80 x 80)
//   K-blocked   [row / 256][K / 64][row % 256][64 halves]          -- longer rows (96 x 96: 2.3 MiB, 128 x 128: 4 MiB, 256 x 256: 16 MiB per row)
// The search kernel reads the same 128-byte K slice of the 512 rows of a tile together.  With 16 MiB rows those are 512 different 2 MiB pages
// at every slice: measured on the persistent kernel at 256 x 256 (round 3, tools/pmc_tlb_search.sh) 44 % of the UTCL1 translations miss and the
// UTCL2 is busy 72 % of the time, against 0.013 % / 3 % at 64 x 64, and the kernel runs at 0.49 of the fp16 peak per busy cluster instead of
// 0.58.  K-blocked, a tile's slice is ONE contiguous 32 KiB piece per operand.  A buffer of n rows holds ceil(n / 256) * 256 rows.
static inline bool gl_vrow_blocked(int64_t K1_halves) { return K1_halves * 2 >= (2ll << 20); }
static inline int64_t gl_vrow_capacity(int64_t n, int64_t K1_halves) { return gl_vrow_blocked(K1_halves) ? (n + 255) / 256 * 256 : n; }

#if defined(__HIPCC__)
// address of half k of fp16 search row `row`: ldv >= 0: row-major with ldv bytes per row; ldv < 0: K-blocked with -ldv cells of 64 halves per row
__device__ __forceinline__ char *gl_vrow_elem(char *V, int64_t ldv, int64_t row, int64_t k)
{
    if (ldv >= 0) return V + row * ldv + k * 2;
    return V + ((row >> 8) * (-ldv) + (k >> 6)) * 32768 + (row & 255) * 128 + (k & 63) * 2;
}

// (v * inv) * coef with both products rounded to fp32 and MATERIALISED: without the empty asm hipcc is free to fold a product into the
// conversion or subtraction that follows it (v_fma_mix*, fma contraction), differently in different kernels, and the fused and the
// stand-alone tap would stop agreeing bit for bit.
__device__ __forceinline__ float gl_tap_value(float v, float inv, float coef)
{
    float x = v * inv;
    asm volatile("" : "+v"(x));
    float t = x * coef;
    asm volatile("" : "+v"(t));
    return t;
}
#endif

// host: rows [48][channels] (channels = 64 or 128) of fp32 tail weights (row = GEMM column) -> the operand image the fused epilogue reads
// (12 KiB per 64 channels): [channel half][column tile 3][k step 2][hi | lo][row 16][k group 4][8 halves], the 8 halves of k group g holding
// channels half * 64 + (2 * step + e / 4) * 16 + 4 * g + e % 4  (the order in which a lane of the epilogue holds its activated outputs)
void gl_pack_tail_weights_host(const float *w48xC, int channels, float scale, void *out);

// position of (tap, ci) inside a packed weight row: channel chunks of 32 outermost, taps inside a chunk
static inline int64_t gl_conv_k_index(int tap, int ci, int ntaps) { return ((int64_t)(ci / 32) * ntaps + tap) * 32 + (ci % 32); }

int gl_launch_gather_conv(gl_ctx *ctx, const GlGatherConv &p, int phases);

// split-fp16 variant (gl_conv_h3.hip): `in` and `wpack` are in the split layout (see that file); same parameter block
int gl_launch_gather_conv_h3(gl_ctx *ctx, const GlGatherConv &p, int phases);
// output channels one workgroup tile of gl_launch_gather_conv_h3 will cover for this problem as far as the fused epilogues (PixelNorm, tap,
// toRGB) are concerned: they need cols <= that; 0 for the tiles that have none
int gl_conv_h3_tile_channels(const GlGatherConv &p, int phases);
// whether gl_launch_gather_conv_h3 can take the tap_* fields for this problem (all channels in one tile, the 2 x 2 window inside one wave)
bool gl_conv_h3_tap_fusable(const GlGatherConv &p, int phases);
// halo form for narrow 3 x 3 layers at high resolution (gl_conv_halo.hip); same results bit for bit as gl_launch_gather_conv_h3's own kernel
bool gl_conv_halo_applies(const GlGatherConv &p, int phases);
int gl_launch_conv_halo_h3(gl_ctx *ctx, const GlGatherConv &p);
// fp32 rows [n][d] -> split layout [n][dpad/32][128 B] of (value * scale)
int gl_launch_split_rows(gl_ctx *ctx, const float *in, int64_t n, int d, int dpad, float scale, void *out);
// host: packed fp32 weight rows [rows][K] -> split layout of (value * scale); `out` holds rows * K * 4 bytes
void gl_split_weights_host(const float *w, size_t rows, size_t K, float scale, void *out);

// second half of the generator tail: col2im of P[pos][(ky*4+kx)*3+co] (the 48-column scatter-form GEMM
// of ConvTranspose2d(Cin -> 3, k4 s2 p1)) + bias + tanh (+ 8-bit quantisation).
// P: column-major [48][ldp] over positions (n, y, x); out_f32 / out_u8: NCHW [n][3][2H][2W], either may be NULL.
int gl_launch_col2im_rgb_tanh(gl_ctx *ctx, const float *P, int64_t ldp, int64_t n_img, int H, int W, const float *bias, float *out_f32,
                              uint8_t *out_u8);
