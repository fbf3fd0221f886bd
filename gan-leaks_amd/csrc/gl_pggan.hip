// Progressive-GAN generator behind the C ABI (gan_models/pggan/model_torch.py:49-88):
//   initial : PixelNorm -> ConvTranspose2d(z, C, 4,1,0)+bias -> LeakyReLU(0.2) -> WSConv3x3 -> LeakyReLU -> PixelNorm
//   step s  : nearest x2 -> [WSConv3x3 -> LeakyReLU -> PixelNorm] x 2          (ConvBlock, :33-47)
//   output  : tanh(alpha * rgb[steps](out) + (1 - alpha) * rgb[steps-1](upscaled))   (fade_in, :71-72)
//             steps == 0: rgb[0](out) without tanh (:78-79)
// WSConv2d (:8-22) scales its INPUT by sqrt(2/(C_in k^2)); the factor is folded into the packed weights.
// Every convolution runs in gather_conv_kernel (fp32 MFMA): 3x3 = 9 taps, the nearest-neighbour upsampling
// is fused into the gather (GlGatherConv::up), toRGB = 1 tap with 3 (padded to 64) columns.
#include "gl_conv.h"
#include <cstdlib>
#include <cmath>
#include <vector>

namespace {

const double kFactors[9] = {1, 1, 1, 1, 1.0 / 2, 1.0 / 4, 1.0 / 8, 1.0 / 16, 1.0 / 32};   // model_torch.py:6
constexpr int kBlocks = 8;

// rows: x / sqrt(mean(x^2) + 1e-8), written zero-padded to dpad.  One wave per row.
__global__ void __launch_bounds__(256) pixelnorm_rows_kernel(const float *__restrict__ in, int64_t n, int d, int dpad, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= n) return;
    const float *src = in + row * d;
    float ss = 0.0f;
    for (int c = lane; c < d; c += 64) { const float t = src[c]; ss = fmaf(t, t, ss); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = 1.0f / sqrtf(ss / (float)d + 1e-8f);
    for (int c = lane; c < dpad; c += 64) out[row * dpad + c] = c < d ? src[c] * inv : 0.0f;
}

// split-layout flavours (gl_conv_h3.hip): values are stored as halves of (value * kPgAct)
constexpr float kPgAct = 16.0f;

__global__ void __launch_bounds__(256) pixelnorm_rows_split_kernel(const float *__restrict__ in, int64_t n, int d, int dpad, char *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= n) return;
    const float *src = in + row * d;
    float ss = 0.0f;
    for (int c = lane; c < d; c += 64) { const float t = src[c]; ss = fmaf(t, t, ss); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = kPgAct / sqrtf(ss / (float)d + 1e-8f);
    char *dst = out + row * dpad * 4;
    for (int c = lane; c < dpad; c += 64) {
        const float v = c < d ? src[c] * inv : 0.0f;
        const _Float16 hi = (_Float16)v;
        char *q = dst + (c >> 5) * 128 + (c & 31) * 2;
        *reinterpret_cast<_Float16 *>(q) = hi;
        *reinterpret_cast<_Float16 *>(q + 64) = (_Float16)(v - (float)hi);
    }
}

// in place on the split layout: stored v = A x  ->  A x / sqrt(mean x^2 + eps) = A v / sqrt(mean v^2 + eps A^2); C % 16 == 0.
// One wave per position.  The sum of squares follows the canonical order of the fused form in gather_conv_h3_kernel's epilogue (per 16-channel
// tile: 4 channels per lane as an fmaf chain, then (g0 + g1) + (g2 + g3); a balanced binary tree over the 16 tiles of every 256 channels;
// the 256-channel groups in ascending order) and every product / sum is individually rounded, so a layer gives the same bits whether
// its PixelNorm ran fused or here.
__global__ void __launch_bounds__(256) pixelnorm_split_kernel(char *__restrict__ x, int64_t positions, int C)
{
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t pos = wave; pos < positions; pos += nwaves) {
        char *p = x + pos * C * 4;
        float ss = 0.0f;
        for (int c = 4 * lane; c - 4 * lane < C; c += 256) {   // 16 tiles (256 channels) per sweep: lane = tile * 4 + group
            float s1 = 0.0f;
            if (c < C) {
                const char *q = p + (c >> 5) * 128 + (c & 31) * 2;
                const h4 hi = *reinterpret_cast<const h4 *>(q), lo = *reinterpret_cast<const h4 *>(q + 64);
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float v = __fadd_rn((float)hi[r], (float)lo[r]); s1 = fmaf(v, v, s1); }
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) s1 = __fadd_rn(s1, __shfl_xor(s1, o, 64));
            ss = c == 4 * lane ? s1 : __fadd_rn(ss, s1);
        }
        const float inv = __fdiv_rn(kPgAct, __fsqrt_rn(__fadd_rn(__fdiv_rn(ss, (float)C), __fmul_rn(__fmul_rn(1e-8f, kPgAct), kPgAct))));
        for (int c = 4 * lane; c < C; c += 256) {
            char *q = p + (c >> 5) * 128 + (c & 31) * 2;
            const h4 hi = *reinterpret_cast<const h4 *>(q), lo = *reinterpret_cast<const h4 *>(q + 64);
            h4 nh, nl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = __fadd_rn((float)hi[r], (float)lo[r]);
                nh[r] = (_Float16)__fmul_rn(v, inv);
                nl[r] = (_Float16)fmaf(v, inv, -(float)nh[r]);          // the exact residual of the product, spelled out (the fused form does the same)
            }
            *reinterpret_cast<h4 *>(q) = nh;
            *reinterpret_cast<h4 *>(q + 64) = nl;
        }
    }
}

// NHWC, in place: one wave per position
__global__ void __launch_bounds__(256) pixelnorm_nhwc_kernel(float *__restrict__ x, int64_t positions, int C)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t pos = wave; pos < positions; pos += nwaves) {
        float *p = x + pos * C;
        float ss = 0.0f;
        for (int c = lane; c < C; c += 64) { const float t = p[c]; ss = fmaf(t, t, ss); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float inv = 1.0f / sqrtf(ss / (float)C + 1e-8f);
        for (int c = lane; c < C; c += 64) p[c] *= inv;
    }
}

__device__ __forceinline__ uint32_t quantize_half(float x)
{
    // gan_models/pggan/train.py:238: gen(...) * 0.5 + 0.5, then ToPILImage mul(255).byte()
    float t = __fmul_rn(__fadd_rn(__fmul_rn(x, 0.5f), 0.5f), 255.0f);
    t = fminf(fmaxf(truncf(t), 0.0f), 255.0f);
    return (uint32_t)(int)t;
}

// a: [n][R][R][ld] (rgb of the last block, ld >= nc floats per pixel); b: [n][R/2][R/2][ld] (rgb of its input, read through
// nearest x2) or NULL.  out NCHW.  use_tanh = 0 for steps == 0.
__global__ void __launch_bounds__(256) pggan_output_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t n, int R, int nc, int ld, float alpha,
                                                           int use_tanh, float *__restrict__ out_f32, uint8_t *__restrict__ out_u8)
{
    const int64_t total = n * nc * R * R;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % R);
        int64_t r = i / R;
        const int y = (int)(r % R);
        r /= R;
        const int c = (int)(r % nc);
        const int64_t im = r / nc;
        float v = a[((im * R + y) * R + x) * ld + c];
        if (use_tanh) {
            if (b) {
                const int Rh = R / 2;
                const float u = b[((im * Rh + (y >> 1)) * Rh + (x >> 1)) * ld + c];
                v = __fadd_rn(__fmul_rn(alpha, v), __fmul_rn(1.0f - alpha, u));
            }
            v = tanhf(v);
        }
        if (out_f32) out_f32[i] = v;
        if (out_u8) out_u8[i] = (uint8_t)quantize_half(v);
    }
}

int pg_upload(gl_ctx *ctx, float **dev, const std::vector<float> &host)
{
    if (*dev) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(*dev); *dev = nullptr; }
    GL_HIP(hipMalloc((void **)dev, host.size() * sizeof(float)));
    GL_HIP(hipMemcpyAsync(*dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    GL_HIP(hipStreamSynchronize(ctx->stream));
    return GL_OK;
}

int pad_to(int v, int m) { return (int)gl_ceil_div(v, m) * m; }
int cols_pad_of(int cols) { return cols % 128 == 0 ? cols : pad_to(cols, 64); }

// WSConv weight [Cout][Cin][k][k] (k = 1 or 3) -> packed [cols_pad][k*k*Cin] with the input scale folded in
std::vector<float> pack_ws(const float *w, int cout, int cin, int k)
{
    const int taps = k * k, K = taps * cin, cp = cols_pad_of(cout);
    const double scale = std::sqrt(2.0 / ((double)cin * k * k));
    std::vector<float> pk((size_t)cp * K, 0.0f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < taps; ++t) pk[(size_t)co * K + gl_conv_k_index(t, ci, taps)] = (float)((double)w[((size_t)co * cin + ci) * taps + t] * scale);
    return pk;
}

}  // namespace

struct gl_pggan;
namespace {
int pg_make_h3(gl_pggan *g, void *h3_slot, const std::vector<float> &pk, size_t rows, size_t K, const float *bias, int nbias, bool rgb);
}

struct gl_pggan {
    gl_ctx *ctx;
    int z_dim, z_pad, C, nc;
    int cin[kBlocks], cout[kBlocks];
    float *w_init, *b_init, *w_i3, *b_i3;
    float *w_blk[kBlocks][2], *b_blk[kBlocks][2];
    float *w_rgb[kBlocks + 1], *b_rgb[kBlocks + 1];
    bool have_init, have_blk[kBlocks], have_rgb[kBlocks + 1];
    float *ones;
    // split-fp16 path (precision 1): per convolution the weights in the split layout (* 2^wexp) and the folded epilogue constants
    int precision;
    struct H3 { float *w, *scale, *shift; } h_init, h_i3, h_blk[kBlocks][2], h_rgb[kBlocks + 1];
    int64_t chunk, ws_imgs;
    size_t ws_act_elems, ws_rgb_elems;      // floats allocated per activation / rgb buffer
    float *ws_z, *ws_buf[3], *ws_rgb[2];
};

namespace {

int rgb_cin(const gl_pggan *g, int j) { return j == 0 ? g->C : g->cout[j - 1]; }

// split-fp16 copy of a packed convolution: rows padded to a multiple of 128, values * 2^wexp (max |w| 2^wexp in [2^12, 2^13)),
// epilogue constants: activations are stored * kPgAct on both sides, so scale = 2^-wexp, shift = bias * kPgAct;
// toRGB writes true fp32 values: scale = 2^-wexp / kPgAct, shift = bias (4 columns: the 4th is padding).
int pg_make_h3(gl_pggan *g, void *h3_slot, const std::vector<float> &pk, size_t rows, size_t K, const float *bias, int nbias, bool rgb)
{
    gl_pggan::H3 *h = reinterpret_cast<gl_pggan::H3 *>(h3_slot);
    const size_t rows128 = (size_t)gl_ceil_div((int64_t)rows, 128) * 128;
    float mx = 0.0f;
    for (float v : pk) mx = std::fmax(mx, std::fabs(v));
    int e = mx > 0.0f ? (int)std::floor(std::log2(8191.0f / mx)) : 0;
    e = e > 30 ? 30 : (e < -30 ? -30 : e);
    std::vector<float> padded(rows128 * K, 0.0f), split(rows128 * K);
    std::copy(pk.begin(), pk.begin() + rows * K, padded.begin());
    gl_split_weights_host(padded.data(), rows128, K, std::ldexp(1.0f, e), split.data());
    const int nvec = rgb ? 4 : nbias;
    std::vector<float> sc(nvec, rgb ? std::ldexp(1.0f / kPgAct, -e) : std::ldexp(1.0f, -e)), sh(nvec, 0.0f);
    for (int i = 0; i < nbias; ++i) sh[i] = rgb ? bias[i] : bias[i] * kPgAct;
    int rc = pg_upload(g->ctx, &h->w, split);
    if (rc == GL_OK) rc = pg_upload(g->ctx, &h->scale, sc);
    if (rc == GL_OK) rc = pg_upload(g->ctx, &h->shift, sh);
    return rc;
}

// pixnorm: the layer is followed by PixelNorm.  In split-fp16 mode it is applied in the convolution's epilogue whenever one tile holds all
// output channels (everything up to 256 channels); *pixnorm_done says whether it was.
// rgb_tail >= 0: this is the last convolution before toRGB layer `rgb_tail`; when its PixelNorm rides in the epilogue and it has <= 128 channels
// (every tile shape then holds all of them, whatever the pass size) toRGB is taken there too and written to rgb_out [pos][4]; *rgb_done says so.
int pg_conv(gl_pggan *g, const float *in, int64_t m, int H, int W, int up, int Cin, const float *w, const float *bias, int cols, int ntaps, int act,
            float *out, const gl_pggan::H3 *h3 = nullptr, bool rgb = false, bool *pixnorm_done = nullptr, int rgb_tail = -1, float *rgb_out = nullptr,
            bool *rgb_done = nullptr)
{
    if (pixnorm_done) *pixnorm_done = false;
    if (rgb_done) *rgb_done = false;
    GlGatherConv p = {};
    p.in = in; p.positions = m * H * W; p.H = H; p.W = W; p.Cin = Cin; p.up = up;
    p.wpack = w; p.cols = cols; p.cols_pad = cols_pad_of(cols); p.ntaps = ntaps;
    if (ntaps == 1) { p.tap_dy[0] = 1; p.tap_dx[0] = 1; }
    else {
        uint32_t dy = 0, dx = 0;
        for (int t = 0; t < 9; ++t) { dy |= (uint32_t)(t / 3) << (2 * t); dx |= (uint32_t)(t % 3) << (2 * t); }
        p.tap_dy[0] = dy; p.tap_dx[0] = dx;
    }
    p.out = out; p.Ho = H; p.Wo = W; p.omul = 1;
    p.scale = g->ones; p.shift = bias; p.cmod = cols; p.act = act; p.zero = g->ctx->zero_page;
    if (g->precision == 1 && h3) {
        p.wpack = h3->w; p.scale = h3->scale; p.shift = h3->shift;
        if (rgb) { p.cols = 4; p.cmod = 4; p.out_mode = 0; }      // 3 real + 1 padding column, fp32 [pos][4]
        else p.out_mode = 2;
        p.cols_pad = (int)gl_ceil_div(p.cols, 128) * 128;
        const int fuse = gl_tuning_int("GL_PIXNORM_FUSE", 1);      // 0: always the separate kernel (debugging)
        if (fuse && pixnorm_done && !rgb && p.cols <= gl_conv_h3_tile_channels(p, 1)) {
            p.pixnorm_act = kPgAct;
            *pixnorm_done = true;
            const int fuse_rgb = gl_tuning_int("GL_RGB_FUSE", 1);
            if (fuse_rgb && rgb_tail >= 0 && rgb_out && rgb_done && p.cols <= 128 && g->nc <= 4) {
                p.rgb_w = g->w_rgb[rgb_tail]; p.rgb_b = g->b_rgb[rgb_tail]; p.rgb_out = rgb_out; p.rgb_n = g->nc; p.rgb_inv_act = 1.0f / kPgAct;
                *rgb_done = true;
            }
        }
        return gl_launch_gather_conv_h3(g->ctx, p, 1);
    }
    return gl_launch_gather_conv(g->ctx, p, 1);
}

int pg_pixelnorm(gl_pggan *g, float *x, int64_t positions, int C)
{
    int64_t blocks = gl_ceil_div(positions, 4);
    if (blocks > 8192) blocks = 8192;
    if (g->precision == 1)
        hipLaunchKernelGGL(pixelnorm_split_kernel, dim3((unsigned)blocks), dim3(256), 0, g->ctx->stream, reinterpret_cast<char *>(x), positions, C);
    else
        hipLaunchKernelGGL(pixelnorm_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, g->ctx->stream, x, positions, C);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

}  // namespace

extern "C" {

int gl_pggan_create(gl_ctx *ctx, int z_dim, int in_channels, int img_channels, gl_pggan **out)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && out, "gl_pggan_create: NULL argument");
    GL_REQUIRE(z_dim > 0 && z_dim <= 4096 && in_channels >= 32 && in_channels % 32 == 0 && img_channels > 0 && img_channels <= 4,
               "gl_pggan_create: unsupported sizes z_dim=%d in_channels=%d (multiple of 32) img_channels=%d", z_dim, in_channels, img_channels);
    gl_pggan *g = new gl_pggan();
    g->ctx = ctx; g->z_dim = z_dim; g->z_pad = pad_to(z_dim, 32); g->C = in_channels; g->nc = img_channels;
    for (int i = 0; i < kBlocks; ++i) {
        g->cin[i] = (int)(in_channels * kFactors[i]);
        g->cout[i] = (int)(in_channels * kFactors[i + 1]);
        g->w_blk[i][0] = g->w_blk[i][1] = g->b_blk[i][0] = g->b_blk[i][1] = nullptr;
        g->have_blk[i] = false;
    }
    for (int j = 0; j <= kBlocks; ++j) { g->w_rgb[j] = g->b_rgb[j] = nullptr; g->have_rgb[j] = false; }
    g->w_init = g->b_init = g->w_i3 = g->b_i3 = nullptr;
    g->have_init = false;
    g->ones = nullptr;
    g->precision = 1;
    g->h_init = g->h_i3 = gl_pggan::H3{nullptr, nullptr, nullptr};
    for (int i = 0; i < kBlocks; ++i) g->h_blk[i][0] = g->h_blk[i][1] = gl_pggan::H3{nullptr, nullptr, nullptr};
    for (int j = 0; j <= kBlocks; ++j) g->h_rgb[j] = gl_pggan::H3{nullptr, nullptr, nullptr};
    g->chunk = 0; g->ws_imgs = 0; g->ws_act_elems = 0; g->ws_rgb_elems = 0;
    g->ws_z = nullptr; g->ws_buf[0] = g->ws_buf[1] = g->ws_buf[2] = nullptr; g->ws_rgb[0] = g->ws_rgb[1] = nullptr;
    std::vector<float> one(16 * (size_t)in_channels, 1.0f);
    int rc = pg_upload(ctx, &g->ones, one);
    if (rc != GL_OK) { delete g; return rc; }
    *out = g;
    return GL_OK;
}

int gl_pggan_destroy(gl_pggan *g)
{
    gl_make_current(g ? g->ctx : nullptr);
    if (!g) return GL_OK;
    (void)hipStreamSynchronize(g->ctx->stream);
    (void)hipFree(g->w_init); (void)hipFree(g->b_init); (void)hipFree(g->w_i3); (void)hipFree(g->b_i3); (void)hipFree(g->ones);
    for (int i = 0; i < kBlocks; ++i)
        for (int k = 0; k < 2; ++k) { (void)hipFree(g->w_blk[i][k]); (void)hipFree(g->b_blk[i][k]); }
    for (int j = 0; j <= kBlocks; ++j) { (void)hipFree(g->w_rgb[j]); (void)hipFree(g->b_rgb[j]); }
    (void)hipFree(g->ws_z);
    for (int k = 0; k < 3; ++k) (void)hipFree(g->ws_buf[k]);
    for (int k = 0; k < 2; ++k) (void)hipFree(g->ws_rgb[k]);
    auto free_h3 = [](gl_pggan::H3 &h) { (void)hipFree(h.w); (void)hipFree(h.scale); (void)hipFree(h.shift); };
    free_h3(g->h_init); free_h3(g->h_i3);
    for (int i = 0; i < kBlocks; ++i) { free_h3(g->h_blk[i][0]); free_h3(g->h_blk[i][1]); }
    for (int j = 0; j <= kBlocks; ++j) free_h3(g->h_rgb[j]);
    delete g;
    return GL_OK;
}

int gl_pggan_set_precision(gl_pggan *g, int mode)
{
    gl_make_current(g ? g->ctx : nullptr);
    GL_REQUIRE(g && (mode == 0 || mode == 1), "gl_pggan_set_precision: mode must be 0 or 1");
    g->precision = mode;
    return GL_OK;
}

int gl_pggan_set_chunk(gl_pggan *g, int64_t images_per_pass)
{
    gl_make_current(g ? g->ctx : nullptr);
    GL_REQUIRE(g && images_per_pass >= 0, "gl_pggan_set_chunk: bad argument");
    g->chunk = images_per_pass;
    return GL_OK;
}

/* initial.1.{weight [z][C][4][4], bias [C]} and initial.3.{conv.weight [C][C][3][3], bias [C]} */
int gl_pggan_set_initial(gl_pggan *g, const float *convt_w, const float *convt_b, const float *ws_w, const float *ws_b)
{
    gl_make_current(g ? g->ctx : nullptr);
    GL_REQUIRE(g && convt_w && convt_b && ws_w && ws_b, "gl_pggan_set_initial: NULL argument");
    const int C = g->C, K = g->z_pad;
    std::vector<float> pk((size_t)16 * C * K, 0.0f);   // GEMM [16*C][z_pad]: column (ky*4+kx)*C + co -> NHWC 4x4xC
    for (int ci = 0; ci < g->z_dim; ++ci)
        for (int co = 0; co < C; ++co)
            for (int t = 0; t < 16; ++t) pk[((size_t)t * C + co) * K + ci] = convt_w[((size_t)ci * C + co) * 16 + t];
    int rc = pg_upload(g->ctx, &g->w_init, pk);
    if (rc == GL_OK) rc = pg_upload(g->ctx, &g->b_init, std::vector<float>(convt_b, convt_b + C));
    const std::vector<float> pk3 = pack_ws(ws_w, C, C, 3);
    if (rc == GL_OK) rc = pg_upload(g->ctx, &g->w_i3, pk3);
    if (rc == GL_OK) rc = pg_upload(g->ctx, &g->b_i3, std::vector<float>(ws_b, ws_b + C));
    if (rc == GL_OK) rc = pg_make_h3(g, &g->h_init, pk, (size_t)16 * C, (size_t)K, convt_b, C, false);
    if (rc == GL_OK) rc = pg_make_h3(g, &g->h_i3, pk3, (size_t)cols_pad_of(C), (size_t)9 * C, ws_b, C, false);
    if (rc != GL_OK) return rc;
    g->have_init = true;
    return GL_OK;
}

/* prog_blocks.{block}.conv{1,2}.{conv.weight, bias} */
int gl_pggan_set_block(gl_pggan *g, int block, const float *conv1_w, const float *conv1_b, const float *conv2_w, const float *conv2_b)
{
    gl_make_current(g ? g->ctx : nullptr);
    GL_REQUIRE(g && block >= 0 && block < kBlocks && conv1_w && conv1_b && conv2_w && conv2_b, "gl_pggan_set_block: bad argument");
    const int ci = g->cin[block], co = g->cout[block];
    GL_REQUIRE(co >= 1, "gl_pggan_set_block: block %d has no channels at in_channels=%d", block, g->C);
    if (ci % 32 != 0 || co % 32 != 0) { g->have_blk[block] = false; return GL_OK; }   // too narrow for the 32-channel K slices: forward() refuses this depth
    const std::vector<float> p1 = pack_ws(conv1_w, co, ci, 3), p2 = pack_ws(conv2_w, co, co, 3);
    int rc = pg_upload(g->ctx, &g->w_blk[block][0], p1);
    if (rc == GL_OK) rc = pg_upload(g->ctx, &g->b_blk[block][0], std::vector<float>(conv1_b, conv1_b + co));
    if (rc == GL_OK) rc = pg_upload(g->ctx, &g->w_blk[block][1], p2);
    if (rc == GL_OK) rc = pg_upload(g->ctx, &g->b_blk[block][1], std::vector<float>(conv2_b, conv2_b + co));
    if (rc == GL_OK) rc = pg_make_h3(g, &g->h_blk[block][0], p1, (size_t)cols_pad_of(co), (size_t)9 * ci, conv1_b, co, false);
    if (rc == GL_OK) rc = pg_make_h3(g, &g->h_blk[block][1], p2, (size_t)cols_pad_of(co), (size_t)9 * co, conv2_b, co, false);
    if (rc != GL_OK) return rc;
    g->have_blk[block] = true;
    return GL_OK;
}

/* rgb_layers.{j}.{conv.weight [nc][C_j][1][1], bias [nc]}; j = 0 is initial_rgb */
int gl_pggan_set_rgb(gl_pggan *g, int j, const float *w, const float *b)
{
    gl_make_current(g ? g->ctx : nullptr);
    GL_REQUIRE(g && j >= 0 && j <= kBlocks && w && b, "gl_pggan_set_rgb: bad argument");
    const int ci = rgb_cin(g, j);
    GL_REQUIRE(ci >= 1, "gl_pggan_set_rgb: layer %d has no input channels", j);
    if (ci % 32 != 0) { g->have_rgb[j] = false; return GL_OK; }
    const std::vector<float> pr = pack_ws(w, g->nc, ci, 1);
    int rc = pg_upload(g->ctx, &g->w_rgb[j], pr);
    if (rc == GL_OK) rc = pg_upload(g->ctx, &g->b_rgb[j], std::vector<float>(b, b + g->nc));
    if (rc == GL_OK) rc = pg_make_h3(g, &g->h_rgb[j], pr, (size_t)cols_pad_of(g->nc), (size_t)ci, b, g->nc, true);
    if (rc != GL_OK) return rc;
    g->have_rgb[j] = true;
    return GL_OK;
}

/* z_dev [n][z_dim] -> [n][nc][R][R], R = 4 * 2^steps.  out_f32_dev = Generator.forward(x, steps, alpha);
 * out_u8_dev = the bytes pggan/train.py:238-246 writes (x*0.5+0.5, mul(255).byte()).  Either may be NULL. */
int gl_pggan_forward(gl_pggan *g, const float *z_dev, int64_t n, int steps, float alpha, float *out_f32_dev, uint8_t *out_u8_dev)
{
    gl_make_current(g ? g->ctx : nullptr);
    GL_REQUIRE(g && n >= 0 && steps >= 0 && steps <= kBlocks, "gl_pggan_forward: bad argument (steps in [0,8])");
    if (!g->have_init) { gl_set_error("gl_pggan_forward: initial block not loaded"); return GL_ERR_STATE; }
    for (int s = 0; s < steps; ++s) {
        GL_REQUIRE(g->cin[s] % 32 == 0 && g->cout[s] % 32 == 0, "gl_pggan_forward: block %d has %d -> %d channels; the K slices need multiples of 32", s,
                   g->cin[s], g->cout[s]);
        if (!g->have_blk[s]) { gl_set_error("gl_pggan_forward: prog_blocks.%d not loaded", s); return GL_ERR_STATE; }
    }
    if (!g->have_rgb[steps] || (steps > 0 && !g->have_rgb[steps - 1])) { gl_set_error("gl_pggan_forward: rgb_layers.%d/%d not loaded", steps, steps - 1); return GL_ERR_STATE; }
    if (n == 0) return GL_OK;
    GL_REQUIRE(z_dev && (out_f32_dev || out_u8_dev), "gl_pggan_forward: NULL z or no output requested");
    gl_ctx *ctx = g->ctx;
    const int R = 4 << steps, nc = g->nc, C = g->C;

    // workspace: three activation buffers sized for the largest layer of this depth
    size_t act = (size_t)16 * C;
    {
        int hw = 16;
        for (int s = 0; s < steps; ++s) {
            hw *= 4;
            const size_t a = (size_t)hw * (g->cout[s] > 0 ? g->cout[s] : 1);
            if (a > act) act = a;
        }
    }
    // default: ~2.25 GiB per buffer, a multiple of 32 images: every heavy layer's grid is then a whole number of rounds of the 256 CUs
    // (a 32 x 32 x 512 layer has 8 workgroups per image); PGGAN-256: 128 images per pass, 375 TFLOP/s against 344 at the former 45
    int64_t want = g->chunk > 0 ? g->chunk : (int64_t)((2304ull << 20) / (act * 4));
    if (g->chunk <= 0 && want > 32) want -= want % 32;
    const int64_t cap = (int64_t)(0xB0000000ull / (act * 4));                           // 32-bit buffer descriptors: one activation tensor < 3 GiB
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    if (want > n) want = n;
    const size_t rgb_elems = (size_t)R * R * 4;      // up to 4 floats per pixel (split path pads 3 -> 4)
    // `want` images per pass for THIS depth; the buffers only ever grow (ws_imgs latent rows, ws_act_elems / ws_rgb_elems floats per buffer)
    const size_t need_act = (size_t)want * act, need_rgb = (size_t)want * rgb_elems;
    if (want > g->ws_imgs || need_act > g->ws_act_elems || need_rgb > g->ws_rgb_elems) {
        GL_HIP(hipStreamSynchronize(ctx->stream));
        const int64_t z_rows = want > g->ws_imgs ? want : g->ws_imgs;
        const size_t act_alloc = need_act > g->ws_act_elems ? need_act : g->ws_act_elems;
        const size_t rgb_alloc = need_rgb > g->ws_rgb_elems ? need_rgb : g->ws_rgb_elems;
        (void)hipFree(g->ws_z);
        for (int k = 0; k < 3; ++k) { (void)hipFree(g->ws_buf[k]); g->ws_buf[k] = nullptr; }
        for (int k = 0; k < 2; ++k) { (void)hipFree(g->ws_rgb[k]); g->ws_rgb[k] = nullptr; }
        g->ws_z = nullptr;
        g->ws_imgs = 0; g->ws_act_elems = 0; g->ws_rgb_elems = 0;
        GL_HIP(gl_device_alloc(g->ctx, (void **)&g->ws_z, (size_t)z_rows * g->z_pad * 4));
        for (int k = 0; k < 3; ++k) GL_HIP(gl_device_alloc(g->ctx, (void **)&g->ws_buf[k], act_alloc * 4));
        for (int k = 0; k < 2; ++k) GL_HIP(gl_device_alloc(g->ctx, (void **)&g->ws_rgb[k], rgb_alloc * 4 + 64));
        g->ws_imgs = z_rows;
        g->ws_act_elems = act_alloc;
        g->ws_rgb_elems = rgb_alloc;
    }
    const int64_t img_elems = (int64_t)nc * R * R;
    int rc;

    for (int64_t i0 = 0; i0 < n; i0 += want) {
        const int64_t m = (n - i0 < want) ? n - i0 : want;
        const bool h3 = g->precision == 1;
        if (h3)
            hipLaunchKernelGGL(pixelnorm_rows_split_kernel, dim3((unsigned)gl_ceil_div(m, 4)), dim3(256), 0, ctx->stream, z_dev + i0 * g->z_dim, m, g->z_dim,
                               g->z_pad, reinterpret_cast<char *>(g->ws_z));
        else
            hipLaunchKernelGGL(pixelnorm_rows_kernel, dim3((unsigned)gl_ceil_div(m, 4)), dim3(256), 0, ctx->stream, z_dev + i0 * g->z_dim, m, g->z_dim, g->z_pad,
                               g->ws_z);
        GL_LAUNCH_CHECK();
        // ConvTranspose2d(z, C, 4, 1, 0) + bias + LeakyReLU : one GEMM to NHWC 4x4xC
        {
            GlGatherConv p = {};
            p.in = g->ws_z; p.positions = m; p.H = 1; p.W = 1; p.Cin = g->z_pad;
            p.wpack = g->w_init; p.cols = 16 * C; p.cols_pad = cols_pad_of(16 * C); p.ntaps = 1; p.tap_dy[0] = 1; p.tap_dx[0] = 1;
            p.out = g->ws_buf[0]; p.Ho = 1; p.Wo = 1; p.omul = 1;
            p.scale = g->ones; p.shift = g->b_init; p.cmod = C; p.act = 2; p.zero = ctx->zero_page;
            if (h3) {
                p.wpack = g->h_init.w; p.scale = g->h_init.scale; p.shift = g->h_init.shift; p.out_mode = 2;
                p.cols_pad = (int)gl_ceil_div(p.cols, 128) * 128;
                rc = gl_launch_gather_conv_h3(ctx, p, 1);
            } else {
                rc = gl_launch_gather_conv(ctx, p, 1);
            }
            if (rc != GL_OK) return rc;
        }
        bool normed = false;
        rc = pg_conv(g, g->ws_buf[0], m, 4, 4, 0, C, g->w_i3, g->b_i3, C, 9, 2, g->ws_buf[1], &g->h_i3, false, &normed);
        if (rc != GL_OK) return rc;
        if (!normed) rc = pg_pixelnorm(g, g->ws_buf[1], m * 16, C);
        if (rc != GL_OK) return rc;
        int cur = 1, prev = 1, hw = 4;
        bool rgb_done = false;
        for (int s = 0; s < steps; ++s) {
            hw *= 2;
            const int b1 = (cur + 1) % 3, b2 = (cur + 2) % 3;
            rc = pg_conv(g, g->ws_buf[cur], m, hw, hw, 1, g->cin[s], g->w_blk[s][0], g->b_blk[s][0], g->cout[s], 9, 2, g->ws_buf[b1], &g->h_blk[s][0], false, &normed);
            if (rc == GL_OK && !normed) rc = pg_pixelnorm(g, g->ws_buf[b1], m * hw * hw, g->cout[s]);
            if (rc == GL_OK) rc = pg_conv(g, g->ws_buf[b1], m, hw, hw, 0, g->cout[s], g->w_blk[s][1], g->b_blk[s][1], g->cout[s], 9, 2, g->ws_buf[b2], &g->h_blk[s][1], false, &normed,
                                          s + 1 == steps ? steps : -1, g->ws_rgb[0], &rgb_done);
            if (rc == GL_OK && !normed) rc = pg_pixelnorm(g, g->ws_buf[b2], m * hw * hw, g->cout[s]);
            if (rc != GL_OK) return rc;
            prev = cur;      // input of this block (low resolution): `upscaled` of the reference is its x2 view
            cur = b2;
        }
        // toRGB (unless the last convolution's epilogue has already written it)
        if (!rgb_done) {
            rc = pg_conv(g, g->ws_buf[cur], m, hw, hw, 0, rgb_cin(g, steps), g->w_rgb[steps], g->b_rgb[steps], nc, 1, 0, g->ws_rgb[0], &g->h_rgb[steps], true);
            if (rc != GL_OK) return rc;
        }
        const float *brgb = nullptr;
        if (steps > 0 && alpha != 1.0f) {
            // rgb[steps-1](upscaled): the 1x1 convolution commutes with nearest upsampling, so it runs at half resolution.
            // With alpha == 1 the term is multiplied by exactly 0 (finite values), so it is skipped.
            rc = pg_conv(g, g->ws_buf[prev], m, hw / 2, hw / 2, 0, rgb_cin(g, steps - 1), g->w_rgb[steps - 1], g->b_rgb[steps - 1], nc, 1, 0, g->ws_rgb[1], &g->h_rgb[steps - 1], true);
            if (rc != GL_OK) return rc;
            brgb = g->ws_rgb[1];
        }
        {
            const int64_t tot = m * img_elems;
            int64_t blocks = gl_ceil_div(tot, 256);
            if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL(pggan_output_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, g->ws_rgb[0], brgb, m, R, nc, g->precision == 1 ? 4 : nc, alpha, steps > 0 ? 1 : 0,
                               out_f32_dev ? out_f32_dev + i0 * img_elems : nullptr, out_u8_dev ? out_u8_dev + i0 * img_elems : nullptr);
            GL_LAUNCH_CHECK();
        }
    }
    return GL_OK;
}

}  // extern "C"
