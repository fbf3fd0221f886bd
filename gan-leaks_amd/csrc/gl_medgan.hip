// medGAN tabular generator + autoencoder decoder behind the C ABI (gan_models/medgan/model.py:13-73, use at
// gan_models/medgan/train.py:306-312):
//   out1 = z    + ReLU(BatchNorm1d(Linear(z)))          gen_block1, BN eps 1e-3, eval (running statistics)
//   out2 = out1 + tanh(BatchNorm1d(Linear(out1)))       gen_block2 (genDim = 128)
//   decoded = sigmoid(Linear(out2))  (binary=True)  or  ReLU(Linear(out2));  rows >= 0.5 -> 1 else 0
// Each layer is one gather_conv launch (1 tap, H = W = 1: a plain fp32-MFMA GEMM) with the BatchNorm folded into the
// epilogue scale/shift and the residual added there.
#include "gl_conv.h"
#include <cmath>
#include <vector>

struct gl_medgan {
    gl_ctx *ctx;
    int z_dim, hidden, F, binary;
    float *w[3], *scale[3], *shift[3];
    bool have_gen, have_dec;
    int64_t ws_rows;
    float *ws_a, *ws_b;
};

namespace {

int mg_upload(gl_ctx *ctx, float **dev, const std::vector<float> &host)
{
    if (*dev) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(*dev); *dev = nullptr; }
    GL_HIP(hipMalloc((void **)dev, host.size() * sizeof(float)));
    GL_HIP(hipMemcpyAsync(*dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    GL_HIP(hipStreamSynchronize(ctx->stream));
    return GL_OK;
}

int mg_cols_pad(int cols) { return cols % 128 == 0 ? cols : (int)gl_ceil_div(cols, 64) * 64; }

// nn.Linear weight [out][in] -> packed [cols_pad][in]  (in % 32 == 0)
std::vector<float> pack_linear(const float *w, int out, int in)
{
    std::vector<float> pk((size_t)mg_cols_pad(out) * in, 0.0f);
    for (int o = 0; o < out; ++o)
        for (int i = 0; i < in; ++i) pk[(size_t)o * in + i] = w[(size_t)o * in + i];
    return pk;
}

int mg_layer(gl_medgan *g, int l, const float *in, int64_t n, int cin, int cols, int act, const float *residual, float *out)
{
    GlGatherConv p = {};
    p.in = in; p.positions = n; p.H = 1; p.W = 1; p.Cin = cin;
    p.wpack = g->w[l]; p.cols = cols; p.cols_pad = mg_cols_pad(cols); p.ntaps = 1; p.tap_dy[0] = 1; p.tap_dx[0] = 1;
    p.out = out; p.Ho = 1; p.Wo = 1; p.omul = 1;
    p.scale = g->scale[l]; p.shift = g->shift[l]; p.cmod = cols; p.act = act; p.residual = residual; p.zero = g->ctx->zero_page;
    return gl_launch_gather_conv(g->ctx, p, 1);
}

__global__ void threshold_kernel(const float *__restrict__ x, int64_t count, float *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) out[i] = x[i] >= 0.5f ? 1.0f : 0.0f;
}

}  // namespace

extern "C" {

int gl_medgan_create(gl_ctx *ctx, int z_dim, int hidden_size, int input_size, int binary, gl_medgan **out)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && out, "gl_medgan_create: NULL argument");
    // the reference's residual adds force z_dim == hidden_size == genDim == 128 (model.py:49,66,71)
    GL_REQUIRE(z_dim == 128 && hidden_size == 128, "gl_medgan_create: the residual generator needs z_dim == hidden_size == 128 (model.py:49-71), got %d / %d", z_dim,
               hidden_size);
    GL_REQUIRE(input_size >= 0 && input_size <= (1 << 20), "gl_medgan_create: bad input_size");
    gl_medgan *g = new gl_medgan();
    g->ctx = ctx; g->z_dim = z_dim; g->hidden = hidden_size; g->F = input_size; g->binary = binary != 0;
    for (int l = 0; l < 3; ++l) g->w[l] = g->scale[l] = g->shift[l] = nullptr;
    g->have_gen = g->have_dec = false;
    g->ws_rows = 0; g->ws_a = g->ws_b = nullptr;
    *out = g;
    return GL_OK;
}

int gl_medgan_destroy(gl_medgan *g)
{
    gl_make_current(g ? g->ctx : nullptr);
    if (!g) return GL_OK;
    (void)hipStreamSynchronize(g->ctx->stream);
    for (int l = 0; l < 3; ++l) { (void)hipFree(g->w[l]); (void)hipFree(g->scale[l]); (void)hipFree(g->shift[l]); }
    (void)hipFree(g->ws_a); (void)hipFree(g->ws_b);
    delete g;
    return GL_OK;
}

/* block = 0: gen_block1 (Linear z->hidden, BN, ReLU); block = 1: gen_block2 (Linear hidden->128, BN, tanh).
 * lin_w [out][in], lin_b [out]; BN vectors [out]; eps = 1e-3 in the reference (model.py:52,57) */
int gl_medgan_set_gen_block(gl_medgan *g, int block, const float *lin_w, const float *lin_b, const float *gamma, const float *beta, const float *mean,
                            const float *var, float eps)
{
    gl_make_current(g ? g->ctx : nullptr);
    GL_REQUIRE(g && (block == 0 || block == 1) && lin_w && lin_b && gamma && beta && mean && var, "gl_medgan_set_gen_block: bad argument");
    const int in = block == 0 ? g->z_dim : g->hidden, out = 128;
    std::vector<float> sc(out), sh(out);
    for (int i = 0; i < out; ++i) {
        const double s = (double)gamma[i] / std::sqrt((double)var[i] + (double)eps);
        sc[i] = (float)s;
        sh[i] = (float)(((double)lin_b[i] - (double)mean[i]) * s + (double)beta[i]);   // BN(Wx + b) = (Wx) s + (b - mean) s + beta
    }
    int rc = mg_upload(g->ctx, &g->w[block], pack_linear(lin_w, out, in));
    if (rc == GL_OK) rc = mg_upload(g->ctx, &g->scale[block], sc);
    if (rc == GL_OK) rc = mg_upload(g->ctx, &g->shift[block], sh);
    if (rc != GL_OK) return rc;
    g->have_gen = g->w[0] && g->w[1];
    return GL_OK;
}

/* Autoencoder.decoder.0.{weight [F][hidden], bias [F]} (model.py:28) */
int gl_medgan_set_decoder(gl_medgan *g, const float *w, const float *b)
{
    gl_make_current(g ? g->ctx : nullptr);
    GL_REQUIRE(g && w && b && g->F > 0, "gl_medgan_set_decoder: bad argument (input_size must be > 0)");
    std::vector<float> one(g->F, 1.0f);
    int rc = mg_upload(g->ctx, &g->w[2], pack_linear(w, g->F, g->hidden));
    if (rc == GL_OK) rc = mg_upload(g->ctx, &g->scale[2], one);
    if (rc == GL_OK) rc = mg_upload(g->ctx, &g->shift[2], std::vector<float>(b, b + g->F));
    if (rc != GL_OK) return rc;
    g->have_dec = true;
    return GL_OK;
}

/* Generator.forward: z_dev [n][128] -> hidden_out_dev [n][128] */
int gl_medgan_generate(gl_medgan *g, const float *z_dev, int64_t n, float *hidden_out_dev)
{
    gl_make_current(g ? g->ctx : nullptr);
    GL_REQUIRE(g && n >= 0, "gl_medgan_generate: bad argument");
    if (!g->have_gen) { gl_set_error("gl_medgan_generate: generator blocks not loaded"); return GL_ERR_STATE; }
    if (n == 0) return GL_OK;
    GL_REQUIRE(z_dev && hidden_out_dev, "gl_medgan_generate: NULL device pointer");
    if (n > g->ws_rows) {
        GL_HIP(hipStreamSynchronize(g->ctx->stream));
        (void)hipFree(g->ws_a);
        g->ws_a = nullptr;
        GL_HIP(hipMalloc((void **)&g->ws_a, (size_t)n * 128 * 4));
        g->ws_rows = n;
    }
    int rc = mg_layer(g, 0, z_dev, n, g->z_dim, 128, 1, z_dev, g->ws_a);                 // out1 = z + ReLU(BN(Linear z))
    if (rc == GL_OK) rc = mg_layer(g, 1, g->ws_a, n, g->hidden, 128, 3, g->ws_a, hidden_out_dev);   // out2 = out1 + tanh(BN(Linear out1))
    return rc;
}

/* Autoencoder.decode: hidden_dev [n][128] -> decoded_dev [n][F] (sigmoid or ReLU); binary_dev (optional) = decoded >= 0.5 */
int gl_medgan_decode(gl_medgan *g, const float *hidden_dev, int64_t n, float *decoded_dev, float *binary_dev)
{
    gl_make_current(g ? g->ctx : nullptr);
    GL_REQUIRE(g && n >= 0, "gl_medgan_decode: bad argument");
    if (!g->have_dec) { gl_set_error("gl_medgan_decode: decoder not loaded"); return GL_ERR_STATE; }
    if (n == 0) return GL_OK;
    GL_REQUIRE(hidden_dev && decoded_dev, "gl_medgan_decode: NULL device pointer");
    int rc = mg_layer(g, 2, hidden_dev, n, g->hidden, g->F, g->binary ? 4 : 1, nullptr, decoded_dev);
    if (rc != GL_OK) return rc;
    if (binary_dev) {
        const int64_t count = n * g->F;
        int64_t blocks = gl_ceil_div(count, 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(threshold_kernel, dim3((unsigned)blocks), dim3(256), 0, g->ctx->stream, decoded_dev, count, binary_dev);
        GL_LAUNCH_CHECK();
    }
    return GL_OK;
}

}  // extern "C"
