// DCGAN / WGAN-GP generator behind the C ABI: weight repacking (host) + the layer schedule.
//   gan_models/dcgan/model_torch.py:75-96  Generator(z_dim, channels_img, features_g)
//   gan_models/wgangp/model.py:37-58       identical graph and state_dict keys
// Layer l = 0..3: ConvTranspose2d(bias=False) -> BatchNorm2d (eval: running stats) -> ReLU
// Layer 4:        ConvTranspose2d(+bias) -> tanh           (-> 8-bit code of the generate branch)
// Activations are NHWC fp32 and stay resident in HBM for a whole pass of `chunk` images.
#include "gl_conv.h"
#include <cmath>
#include <vector>

struct gl_dcgan {
    gl_ctx *ctx;
    int z_dim, z_pad, nc, fg;
    int cin[5], cout[5];
    float *wpack[5];           // device, packed
    float *scale[4], *shift[4];
    float *bias_out;
    bool have_w[5], have_bn[4], have_bias;
    int64_t chunk, ws_chunk;   // requested / allocated images per pass
    float *ws_z, *ws_a[4];     // z padded; outputs of layers 0..3
    float *ws_p;               // scatter-form output of layer 4: [img][H*W][16 taps * nc]
    float *ident_scale, *ident_shift;   // epilogue constants (1, 0) for the layer-4 GEMM
    // optional self-attention on the output of layer 2 (VAEGAN: gan_models/vaegan/ops.py:86-120)
    bool have_att;
    float *att_wq, *att_bq, *att_wk, *att_bk, *att_wv, *att_bv, att_gamma;
    float *ws_att;
    // split-fp16 path (gl_conv_h3.hip): weights in the split layout scaled by 2^wexp, epilogue constants folded for it
    int precision;             // 0 = fp32 MFMA (exact fp32 products), 1 = split-fp16 (three fp16 MFMAs per product, ~22-bit operands)
    float *wsplit[5];
    int wexp[5];
    std::vector<float> h_scale[4], h_shift[4];
    float *scale_h3[5], *shift_h3[5];
    bool h3_dirty;
};

// activations of the split path are stored multiplied by this power of two (keeps small values out of the fp16 subnormals)
static const float kActScale = 16.0f;

namespace {

__global__ void pad_rows_kernel(const float *__restrict__ in, int64_t n, int d, int dpad, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * dpad) return;
    const int64_t r = i / dpad;
    const int c = (int)(i - r * dpad);
    out[i] = c < d ? in[r * d + c] : 0.0f;
}

// sub-pixel decomposition of ConvTranspose2d(k=4, s=2, p=1): output row 2y+py receives
//   py=0: ky=1 from input row y, ky=3 from y-1;   py=1: ky=0 from y+1, ky=2 from y.
const int kKy[2][2] = {{1, 3}, {0, 2}};
const int kDy[2][2] = {{0, -1}, {1, 0}};

int upload(gl_ctx *ctx, float **dev, const std::vector<float> &host)
{
    if (!*dev) GL_HIP(hipMalloc((void **)dev, host.size() * sizeof(float)));
    GL_HIP(hipMemcpyAsync(*dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    GL_HIP(hipStreamSynchronize(ctx->stream));
    return GL_OK;
}


// Self-attention (SAGAN style) over the T = 256 positions of a 16 x 16 map, per image
// (gan_models/vaegan/ops.py:101-120):
//   q = Wq x + bq, k = Wk x + bk  (C/8 channels), energy_ij = q_i . k_j, att = softmax_j(energy),
//   out_i = sum_j att_ij (Wv x_j + bv) = Wv (sum_j att_ij x_j) + bv   (rows of att sum to 1),  y = gamma * out + x
// One workgroup per image, one thread per position i.  Everything indexed by j or by a weight is wave-uniform
// (scalar loads / LDS broadcast); per thread: q_i (C/8 regs), the running sum over j (C regs).
template <int C>
__global__ void __launch_bounds__(256) self_attention_kernel(const float *__restrict__ x, float *__restrict__ y, const float *__restrict__ wq,
                                                             const float *__restrict__ bq, const float *__restrict__ wk, const float *__restrict__ bk,
                                                             const float *__restrict__ wv, const float *__restrict__ bv, float gamma)
{
    constexpr int DK = C / 8, T = 256;
    __shared__ float ks[T][DK];
    const int i = threadIdx.x;
    const float *ximg = x + (int64_t)blockIdx.x * T * C;
    const float *xi = ximg + (int64_t)i * C;
    float q[DK], kk[DK];
#pragma unroll
    for (int d = 0; d < DK; ++d) { q[d] = bq[d]; kk[d] = bk[d]; }
    for (int c = 0; c < C; c += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(xi + c);
#pragma unroll
        for (int d = 0; d < DK; ++d) {
            q[d] = fmaf(wq[d * C + c + 0], v.x, q[d]); q[d] = fmaf(wq[d * C + c + 1], v.y, q[d]);
            q[d] = fmaf(wq[d * C + c + 2], v.z, q[d]); q[d] = fmaf(wq[d * C + c + 3], v.w, q[d]);
            kk[d] = fmaf(wk[d * C + c + 0], v.x, kk[d]); kk[d] = fmaf(wk[d * C + c + 1], v.y, kk[d]);
            kk[d] = fmaf(wk[d * C + c + 2], v.z, kk[d]); kk[d] = fmaf(wk[d * C + c + 3], v.w, kk[d]);
        }
    }
#pragma unroll
    for (int d = 0; d < DK; ++d) ks[i][d] = kk[d];
    __syncthreads();
    float m = -__builtin_inff();
    for (int j = 0; j < T; ++j) {
        float e = 0.0f;
#pragma unroll
        for (int d = 0; d < DK; ++d) e = fmaf(q[d], ks[j][d], e);
        m = fmaxf(m, e);
    }
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0f;
    float ssum = 0.0f;
    for (int j = 0; j < T; ++j) {
        float e = 0.0f;
#pragma unroll
        for (int d = 0; d < DK; ++d) e = fmaf(q[d], ks[j][d], e);
        const float pj = __expf(e - m);
        ssum += pj;
        const float *xj = ximg + (int64_t)j * C;       // wave-uniform address
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = fmaf(pj, xj[c], acc[c]);
    }
    const float inv = 1.0f / ssum;
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] *= inv;
    float *yi = y + ((int64_t)blockIdx.x * T + i) * C;
    for (int co = 0; co < C; ++co) {
        float o = bv[co];
#pragma unroll
        for (int c = 0; c < C; ++c) o = fmaf(wv[co * C + c], acc[c], o);
        yi[co] = fmaf(gamma, o, xi[co]);
    }
}

int ensure_workspace(gl_dcgan *g, int64_t n)
{
    int64_t want = g->chunk > 0 ? g->chunk : 4096;
    if (n < want) want = n;
    if (want <= g->ws_chunk) return GL_OK;
    GL_HIP(hipStreamSynchronize(g->ctx->stream));
    (void)hipFree(g->ws_z);
    g->ws_z = nullptr;
    for (int l = 0; l < 4; ++l) { (void)hipFree(g->ws_a[l]); g->ws_a[l] = nullptr; }
    (void)hipFree(g->ws_p);
    g->ws_p = nullptr;
    (void)hipFree(g->ws_att);
    g->ws_att = nullptr;
    g->ws_chunk = 0;
    GL_HIP(hipMalloc((void **)&g->ws_z, (size_t)want * g->z_pad * 4));
    int hw = 16;
    for (int l = 0; l < 4; ++l) {
        GL_HIP(hipMalloc((void **)&g->ws_a[l], (size_t)want * hw * g->cout[l] * 4));
        hw *= 4;
    }
    GL_HIP(hipMalloc((void **)&g->ws_p, (size_t)want * 32 * 32 * 16 * g->nc * 4));
    GL_HIP(hipMalloc((void **)&g->ws_att, (size_t)want * 256 * g->cout[2] * 4));
    g->ws_chunk = want;
    return GL_OK;
}

}  // namespace

extern "C" {

int gl_dcgan_create(gl_ctx *ctx, int z_dim, int channels_img, int features_g, gl_dcgan **out)
{
    GL_REQUIRE(ctx && out, "gl_dcgan_create: NULL argument");
    GL_REQUIRE(z_dim > 0 && z_dim <= 4096, "gl_dcgan_create: z_dim=%d unsupported", z_dim);
    GL_REQUIRE(channels_img == 3, "gl_dcgan_create: channels_img=%d unsupported (the RGB tail kernel writes 3 channels)", channels_img);
    GL_REQUIRE(features_g > 0 && features_g % 16 == 0, "gl_dcgan_create: features_g=%d must be a multiple of 16 (K slices of 32 channels)", features_g);
    gl_dcgan *g = new gl_dcgan();
    g->ctx = ctx;
    g->z_dim = z_dim;
    g->z_pad = (int)gl_ceil_div(z_dim, 32) * 32;
    g->nc = channels_img;
    g->fg = features_g;
    const int ch[6] = {z_dim, features_g * 16, features_g * 8, features_g * 4, features_g * 2, channels_img};
    for (int l = 0; l < 5; ++l) { g->cin[l] = ch[l]; g->cout[l] = ch[l + 1]; g->wpack[l] = nullptr; g->have_w[l] = false; }
    for (int l = 0; l < 4; ++l) { g->scale[l] = g->shift[l] = nullptr; g->have_bn[l] = false; g->ws_a[l] = nullptr; }
    g->bias_out = nullptr;
    g->have_bias = false;
    g->chunk = 0;
    g->ws_chunk = 0;
    g->ws_z = nullptr;
    g->ws_p = nullptr;
    g->precision = 1;
    g->h3_dirty = true;
    for (int l = 0; l < 5; ++l) { g->wsplit[l] = nullptr; g->wexp[l] = 0; g->scale_h3[l] = g->shift_h3[l] = nullptr; }
    g->have_att = false;
    g->att_wq = g->att_bq = g->att_wk = g->att_bk = g->att_wv = g->att_bv = nullptr;
    g->att_gamma = 0.0f;
    g->ws_att = nullptr;
    g->ident_scale = g->ident_shift = nullptr;
    {
        std::vector<float> one(16 * channels_img, 1.0f), zero(16 * channels_img, 0.0f);
        int rc = upload(ctx, &g->ident_scale, one);
        if (rc == GL_OK) rc = upload(ctx, &g->ident_shift, zero);
        if (rc != GL_OK) { delete g; return rc; }
    }
    *out = g;
    return GL_OK;
}

int gl_dcgan_destroy(gl_dcgan *g)
{
    if (!g) return GL_OK;
    (void)hipStreamSynchronize(g->ctx->stream);
    for (int l = 0; l < 5; ++l) (void)hipFree(g->wpack[l]);
    for (int l = 0; l < 4; ++l) { (void)hipFree(g->scale[l]); (void)hipFree(g->shift[l]); (void)hipFree(g->ws_a[l]); }
    (void)hipFree(g->bias_out);
    (void)hipFree(g->ws_z);
    (void)hipFree(g->ws_p);
    (void)hipFree(g->ws_att);
    for (int l = 0; l < 5; ++l) { (void)hipFree(g->wsplit[l]); (void)hipFree(g->scale_h3[l]); (void)hipFree(g->shift_h3[l]); }
    (void)hipFree(g->att_wq); (void)hipFree(g->att_bq); (void)hipFree(g->att_wk); (void)hipFree(g->att_bk); (void)hipFree(g->att_wv); (void)hipFree(g->att_bv);
    (void)hipFree(g->ident_scale);
    (void)hipFree(g->ident_shift);
    delete g;
    return GL_OK;
}

int gl_dcgan_set_chunk(gl_dcgan *g, int64_t images_per_pass)
{
    GL_REQUIRE(g && images_per_pass >= 0, "gl_dcgan_set_chunk: bad argument");
    g->chunk = images_per_pass;
    return GL_OK;
}

int gl_dcgan_set_conv_weight(gl_dcgan *g, int layer, const float *w)
{
    GL_REQUIRE(g && w && layer >= 0 && layer < 5, "gl_dcgan_set_conv_weight: bad argument");
    const int ci_n = g->cin[layer], co_n = g->cout[layer];
    auto W = [&](int ci, int co, int ky, int kx) { return w[(((int64_t)ci * co_n + co) * 4 + ky) * 4 + kx]; };
    std::vector<float> pk;
    if (layer == 0) {
        // 1x1 input: a plain GEMM [n][z_pad] x [16*C1][z_pad]^T, column = (ky*4+kx)*C1 + co  (NHWC 4x4xC1)
        const int K = g->z_pad, cols = 16 * co_n, cols_pad = (int)gl_ceil_div(cols, 128) * 128;
        pk.assign((size_t)cols_pad * K, 0.0f);
        for (int ky = 0; ky < 4; ++ky)
            for (int kx = 0; kx < 4; ++kx)
                for (int co = 0; co < co_n; ++co)
                    for (int ci = 0; ci < ci_n; ++ci) pk[((size_t)(ky * 4 + kx) * co_n + co) * K + ci] = W(ci, co, ky, kx);
    } else if (layer < 4) {
        // four phases x four taps; [phase][co_pad][tap*Cin + ci]
        const int K = 4 * ci_n, cols_pad = (int)gl_ceil_div(co_n, 128) * 128;
        pk.assign((size_t)4 * cols_pad * K, 0.0f);
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px)
                for (int ty = 0; ty < 2; ++ty)
                    for (int tx = 0; tx < 2; ++tx) {
                        const int ky = kKy[py][ty], kx = kKy[px][tx], tap = ty * 2 + tx, phase = py * 2 + px;
                        for (int co = 0; co < co_n; ++co) {
                            float *dst = &pk[((size_t)phase * cols_pad + co) * K];
                            for (int ci = 0; ci < ci_n; ++ci) dst[gl_conv_k_index(tap, ci, 4)] = W(ci, co, ky, kx);
                        }
                    }
    } else {
        // scatter form: one GEMM column per (ky, kx, co); [64][Cin], column = (ky*4+kx)*nc + co
        const int K = ci_n, cols_pad = (int)gl_ceil_div(16 * co_n, 64) * 64;
        pk.assign((size_t)cols_pad * K, 0.0f);
        for (int ky = 0; ky < 4; ++ky)
            for (int kx = 0; kx < 4; ++kx)
                for (int co = 0; co < co_n; ++co)
                    for (int ci = 0; ci < ci_n; ++ci) pk[((size_t)(ky * 4 + kx) * co_n + co) * K + ci] = W(ci, co, ky, kx);
    }
    int rc = upload(g->ctx, &g->wpack[layer], pk);
    if (rc != GL_OK) return rc;
    {
        // split-fp16 copy: rows padded to 128 columns, values scaled by 2^wexp so that max |w| * 2^wexp is in [2^12, 2^13)
        const size_t Kl = layer == 0 ? (size_t)g->z_pad : (layer < 4 ? (size_t)4 * ci_n : (size_t)ci_n);
        const size_t rows_real = pk.size() / Kl / (layer > 0 && layer < 4 ? 4 : 1);     // cols_pad of the fp32 pack, per phase
        const size_t phases = layer > 0 && layer < 4 ? 4 : 1;
        const size_t rows128 = (size_t)gl_ceil_div((int64_t)rows_real, 128) * 128;
        float mx = 0.0f;
        for (float v : pk) mx = std::fmax(mx, std::fabs(v));
        int e = mx > 0.0f ? (int)std::floor(std::log2(8191.0f / mx)) : 0;
        if (e > 30) e = 30;
        if (e < -30) e = -30;
        g->wexp[layer] = e;
        std::vector<float> padded(phases * rows128 * Kl, 0.0f);
        for (size_t ph = 0; ph < phases; ++ph)
            for (size_t r = 0; r < rows_real; ++r)
                std::copy(pk.begin() + (ph * rows_real + r) * Kl, pk.begin() + (ph * rows_real + r + 1) * Kl, padded.begin() + (ph * rows128 + r) * Kl);
        std::vector<float> split(padded.size());
        gl_split_weights_host(padded.data(), phases * rows128, Kl, std::ldexp(1.0f, e), split.data());
        rc = upload(g->ctx, &g->wsplit[layer], split);
        if (rc != GL_OK) return rc;
        g->h3_dirty = true;
    }
    g->have_w[layer] = true;
    return GL_OK;
}

int gl_dcgan_set_bn(gl_dcgan *g, int layer, const float *gamma, const float *beta, const float *mean, const float *var, float eps)
{
    GL_REQUIRE(g && gamma && beta && mean && var && layer >= 0 && layer < 4, "gl_dcgan_set_bn: bad argument");
    const int c = g->cout[layer];
    std::vector<float> sc(c), sh(c);
    for (int i = 0; i < c; ++i) {
        // y = (x - mean) / sqrt(var + eps) * gamma + beta  (nn.BatchNorm2d, eval)  ==  x * sc + sh
        const double s = (double)gamma[i] / std::sqrt((double)var[i] + (double)eps);
        sc[i] = (float)s;
        sh[i] = (float)((double)beta[i] - (double)mean[i] * s);
    }
    int rc = upload(g->ctx, &g->scale[layer], sc);
    if (rc != GL_OK) return rc;
    rc = upload(g->ctx, &g->shift[layer], sh);
    if (rc != GL_OK) return rc;
    g->h_scale[layer] = sc;
    g->h_shift[layer] = sh;
    g->h3_dirty = true;
    g->have_bn[layer] = true;
    return GL_OK;
}

int gl_dcgan_set_out_bias(gl_dcgan *g, const float *bias)
{
    GL_REQUIRE(g && bias, "gl_dcgan_set_out_bias: bad argument");
    std::vector<float> b(bias, bias + g->nc);
    int rc = upload(g->ctx, &g->bias_out, b);
    if (rc != GL_OK) return rc;
    g->have_bias = true;
    return GL_OK;
}

/* epilogue of layer 0..3 set directly: y = conv(x) * scale[c] + shift[c], then ReLU.  Used when the caller folds more than
 * BatchNorm into it (VAEGAN: 1/sigma of the spectral norm and the ConvTranspose bias, gan_models/vaegan/train.py:112-135). */
int gl_dcgan_set_affine(gl_dcgan *g, int layer, const float *scale, const float *shift)
{
    GL_REQUIRE(g && scale && shift && layer >= 0 && layer < 4, "gl_dcgan_set_affine: bad argument");
    const int c = g->cout[layer];
    int rc = upload(g->ctx, &g->scale[layer], std::vector<float>(scale, scale + c));
    if (rc == GL_OK) rc = upload(g->ctx, &g->shift[layer], std::vector<float>(shift, shift + c));
    if (rc != GL_OK) return rc;
    g->h_scale[layer].assign(scale, scale + c);
    g->h_shift[layer].assign(shift, shift + c);
    g->h3_dirty = true;
    g->have_bn[layer] = true;
    return GL_OK;
}

/* SelfAttention(in_dim = C2) on the output of layer 2 (16 x 16 x C2, C2 = features_g * 4): query/key conv weights [C2/8][C2],
 * value conv weight [C2][C2] (1x1 convs), biases, gamma.  gan_models/vaegan/ops.py:86-120. */
int gl_dcgan_set_attention(gl_dcgan *g, const float *wq, const float *bq, const float *wk, const float *bk, const float *wv, const float *bv, float gamma)
{
    GL_REQUIRE(g && wq && bq && wk && bk && wv && bv, "gl_dcgan_set_attention: NULL argument");
    const int C = g->cout[2], DK = C / 8;
    GL_REQUIRE(C == 64 || C == 128, "gl_dcgan_set_attention: attention width %d unsupported (64 or 128)", C);
    int rc = upload(g->ctx, &g->att_wq, std::vector<float>(wq, wq + (size_t)DK * C));
    if (rc == GL_OK) rc = upload(g->ctx, &g->att_bq, std::vector<float>(bq, bq + DK));
    if (rc == GL_OK) rc = upload(g->ctx, &g->att_wk, std::vector<float>(wk, wk + (size_t)DK * C));
    if (rc == GL_OK) rc = upload(g->ctx, &g->att_bk, std::vector<float>(bk, bk + DK));
    if (rc == GL_OK) rc = upload(g->ctx, &g->att_wv, std::vector<float>(wv, wv + (size_t)C * C));
    if (rc == GL_OK) rc = upload(g->ctx, &g->att_bv, std::vector<float>(bv, bv + C));
    if (rc != GL_OK) return rc;
    g->att_gamma = gamma;
    g->have_att = true;
    return GL_OK;
}

/* 0 = fp32 MFMA (every product exact in fp32), 1 = split-fp16 (default): operands carried as hi + lo halves (~22 bits), three fp16
 * MFMAs per product, fp32 accumulation.  Both meet the 1e-4 parity bound; 1 is ~2-3x faster. */
int gl_dcgan_set_precision(gl_dcgan *g, int mode)
{
    GL_REQUIRE(g && (mode == 0 || mode == 1), "gl_dcgan_set_precision: mode must be 0 or 1");
    g->precision = mode;
    return GL_OK;
}

static int dcgan_prepare_h3(gl_dcgan *g)
{
    if (!g->h3_dirty) return GL_OK;
    for (int l = 0; l < 4; ++l) {
        // stored activation = true value * kActScale; acc = 2^wexp * kActScale * conv  =>  scale' = s / 2^wexp, shift' = t * kActScale
        std::vector<float> sc(g->h_scale[l].size()), sh(g->h_shift[l].size());
        for (size_t i = 0; i < sc.size(); ++i) { sc[i] = std::ldexp(g->h_scale[l][i], -g->wexp[l]); sh[i] = g->h_shift[l][i] * kActScale; }
        int rc = upload(g->ctx, &g->scale_h3[l], sc);
        if (rc == GL_OK) rc = upload(g->ctx, &g->shift_h3[l], sh);
        if (rc != GL_OK) return rc;
    }
    std::vector<float> sc(16 * g->nc, std::ldexp(1.0f / kActScale, -g->wexp[4])), sh(16 * g->nc, 0.0f);
    int rc = upload(g->ctx, &g->scale_h3[4], sc);
    if (rc == GL_OK) rc = upload(g->ctx, &g->shift_h3[4], sh);
    if (rc != GL_OK) return rc;
    g->h3_dirty = false;
    return GL_OK;
}

int gl_dcgan_forward(gl_dcgan *g, const float *z_dev, int64_t n, float *out_f32_dev, uint8_t *out_u8_dev)
{
    GL_REQUIRE(g && n >= 0, "gl_dcgan_forward: bad argument");
    for (int l = 0; l < 5; ++l)
        if (!g->have_w[l] || (l < 4 && !g->have_bn[l])) {
            gl_set_error("gl_dcgan_forward: weights of layer %d not loaded", l);
            return GL_ERR_STATE;
        }
    if (!g->have_bias) { gl_set_error("gl_dcgan_forward: gen.4.bias not loaded"); return GL_ERR_STATE; }
    if (n == 0) return GL_OK;
    GL_REQUIRE(z_dev && (out_f32_dev || out_u8_dev), "gl_dcgan_forward: NULL z or no output requested");
    gl_ctx *ctx = g->ctx;
    int rc = ensure_workspace(g, n);
    if (rc != GL_OK) return rc;
    const int64_t img_elems = (int64_t)g->nc * 64 * 64;
    // split-fp16 path unless the caller asked for fp32 products or the (fp32) attention block sits in the stack
    const bool h3 = g->precision == 1 && !g->have_att;
    if (h3) {
        rc = dcgan_prepare_h3(g);
        if (rc != GL_OK) return rc;
    }
    auto launch = [&](GlGatherConv &p, int layer, int phases) {
        if (!h3) return gl_launch_gather_conv(ctx, p, phases);
        p.wpack = g->wsplit[layer];
        p.cols_pad = (int)gl_ceil_div(p.cols, 128) * 128;
        p.scale = g->scale_h3[layer];
        p.shift = g->shift_h3[layer];
        p.out_mode = layer < 4 ? 2 : 0;
        return gl_launch_gather_conv_h3(ctx, p, phases);
    };

    // balanced passes: ceil(n / passes) images each instead of full passes plus a small ragged one
    const int64_t passes = gl_ceil_div(n, g->ws_chunk);
    const int64_t per_pass = gl_ceil_div(n, passes);
    for (int64_t i0 = 0; i0 < n; i0 += per_pass) {
        const int64_t m = (n - i0 < per_pass) ? n - i0 : per_pass;
        if (h3) {
            rc = gl_launch_split_rows(ctx, z_dev + i0 * g->z_dim, m, g->z_dim, g->z_pad, kActScale, g->ws_z);
            if (rc != GL_OK) return rc;
        } else {
            const int64_t tot = m * g->z_pad;
            hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)gl_ceil_div(tot, 256)), dim3(256), 0, ctx->stream, z_dev + i0 * g->z_dim, m, g->z_dim,
                               g->z_pad, g->ws_z);
            GL_LAUNCH_CHECK();
        }
        // layer 0: ConvT k4 s1 p0 on 1x1 -> 4x4xC1 (one GEMM), BN + ReLU
        {
            GlGatherConv p = {};
            p.in = g->ws_z; p.positions = m; p.H = 1; p.W = 1; p.Cin = g->z_pad;
            p.wpack = g->wpack[0]; p.cols = 16 * g->cout[0]; p.cols_pad = (int)gl_ceil_div(p.cols, 128) * 128; p.ntaps = 1;
            p.tap_dy[0] = 1; p.tap_dx[0] = 1;   // offset 0
            p.out = g->ws_a[0]; p.Ho = 1; p.Wo = 1; p.omul = 1; p.oy[0] = 0; p.ox[0] = 0;
            p.scale = g->scale[0]; p.shift = g->shift[0]; p.cmod = g->cout[0]; p.act = 1; p.zero = ctx->zero_page;
            rc = launch(p, 0, 1);
            if (rc != GL_OK) return rc;
        }
        // layers 1..3: ConvT k4 s2 p1 as four 2x2-tap sub-pixel convolutions, BN + ReLU
        int hw = 4;
        for (int l = 1; l < 4; ++l) {
            GlGatherConv p = {};
            p.in = (l == 3 && g->have_att) ? g->ws_att : g->ws_a[l - 1];
            p.positions = m * hw * hw; p.H = hw; p.W = hw; p.Cin = g->cin[l];
            p.wpack = g->wpack[l]; p.cols = g->cout[l]; p.cols_pad = (int)gl_ceil_div(p.cols, 128) * 128; p.ntaps = 4;
            for (int py = 0; py < 2; ++py)
                for (int px = 0; px < 2; ++px) {
                    uint32_t dy = 0, dx = 0;
                    for (int ty = 0; ty < 2; ++ty)
                        for (int tx = 0; tx < 2; ++tx) {
                            const int tap = ty * 2 + tx;
                            dy |= (uint32_t)(kDy[py][ty] + 1) << (2 * tap);
                            dx |= (uint32_t)(kDy[px][tx] + 1) << (2 * tap);
                        }
                    p.tap_dy[py * 2 + px] = dy; p.tap_dx[py * 2 + px] = dx;
                    p.oy[py * 2 + px] = py; p.ox[py * 2 + px] = px;
                }
            p.out = g->ws_a[l]; p.Ho = 2 * hw; p.Wo = 2 * hw; p.omul = 2;
            p.scale = g->scale[l]; p.shift = g->shift[l]; p.cmod = g->cout[l]; p.act = 1; p.zero = ctx->zero_page;
            rc = launch(p, l, 4);
            if (rc != GL_OK) return rc;
            hw *= 2;
            if (l == 2 && g->have_att) {
                GL_REQUIRE(hw == 16, "gl_dcgan_forward: attention expects a 16 x 16 map");
                if (g->cout[2] == 128)
                    hipLaunchKernelGGL(self_attention_kernel<128>, dim3((unsigned)m), dim3(256), 0, ctx->stream, g->ws_a[2], g->ws_att, g->att_wq, g->att_bq,
                                       g->att_wk, g->att_bk, g->att_wv, g->att_bv, g->att_gamma);
                else
                    hipLaunchKernelGGL(self_attention_kernel<64>, dim3((unsigned)m), dim3(256), 0, ctx->stream, g->ws_a[2], g->ws_att, g->att_wq, g->att_bq,
                                       g->att_wk, g->att_bk, g->att_wv, g->att_bv, g->att_gamma);
                GL_LAUNCH_CHECK();
            }
        }
        // layer 4: ConvT k4 s2 p1 -> 3 channels: scatter-form GEMM (48 columns) on the matrix cores,
        // then col2im + bias + tanh (+ quantise)
        {
            GlGatherConv p = {};
            p.in = g->ws_a[3]; p.positions = m * hw * hw; p.H = hw; p.W = hw; p.Cin = g->cin[4];
            p.wpack = g->wpack[4]; p.cols = 16 * g->nc; p.cols_pad = (int)gl_ceil_div(p.cols, 64) * 64; p.ntaps = 1;
            p.tap_dy[0] = 1; p.tap_dx[0] = 1;
            p.out = g->ws_p; p.Ho = hw; p.Wo = hw; p.omul = 1; p.oy[0] = 0; p.ox[0] = 0; p.planar = 1; p.ld_planar = m * hw * hw;
            p.scale = g->ident_scale; p.shift = g->ident_shift; p.cmod = p.cols; p.act = 0; p.zero = ctx->zero_page;
            rc = launch(p, 4, 1);
            if (rc != GL_OK) return rc;
        }
        rc = gl_launch_col2im_rgb_tanh(ctx, g->ws_p, m * hw * hw, m, hw, hw, g->bias_out, out_f32_dev ? out_f32_dev + i0 * img_elems : nullptr,
                                       out_u8_dev ? out_u8_dev + i0 * img_elems : nullptr);
        if (rc != GL_OK) return rc;
    }
    return GL_OK;
}

}  // extern "C"
