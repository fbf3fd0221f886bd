// DCGAN / WGAN-GP generator behind the C ABI: weight repacking (host) + the layer schedule.
//   gan_models/dcgan/model_torch.py:75-96  Generator(z_dim, channels_img, features_g)
//   gan_models/wgangp/model.py:37-58       identical graph and state_dict keys
// Layer l = 0..3: ConvTranspose2d(bias=False) -> BatchNorm2d (eval: running stats) -> ReLU
// Layer 4:        ConvTranspose2d(+bias) -> tanh           (-> 8-bit code of the generate branch)
// Activations are NHWC fp32 and stay resident in HBM for a whole pass of `chunk` images.
#include "gl_conv.h"
#include <cmath>
#include <cstdlib>
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
    float *att_w, *att_wsplit, *att_bias, *att_ones, *att_scale_h3, att_gamma;   // [q | k | v] 1x1 convolutions as one GEMM
    int att_cols, att_cols_pad, att_wexp;
    float *ws_att, *ws_qkv;
    // split-fp16 path (gl_conv_h3.hip): weights in the split layout scaled by 2^wexp, epilogue constants folded for it
    // optional spectral normalisation of layers 0..3 (VAEGAN: gan_models/vaegan/ops.py:23-75): w_bar as [C_in][C_out * 16], the power-iteration
    // state u [C_in], v [C_out * 16] and gamma / sqrt(var + eps) per output channel stay on the device; every forward advances u, v and
    // rewrites the epilogue scale as bn_scale / sigma
    bool have_sn[4];
    float *sn_w[4], *sn_u[4], *sn_v[4], *sn_wv[4], *sn_bns[4];
    int sn_iters;
    bool sn_hold;              // next forward(s) reuse the current sigma (a re-run of the same call)
    int precision;             // 0 = fp32 MFMA (exact fp32 products), 1 = split-fp16 (three fp16 MFMAs per product, ~22-bit operands)
    float *wsplit[5];
    bool fuse_tail;            // default on (gl_dcgan_set_fuse_tail)
    void *tail_w;              // layer 4 packed for the epilogue of layer 3 (gl_pack_tail_weights_host), when layer 3 has 64 or 128 channels
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


// Self-attention (SAGAN style) over the T = 256 positions of a 16 x 16 map, per image (gan_models/vaegan/ops.py:101-120):
//   q = Wq x + bq, k = Wk x + bk  (C/8 channels), v = Wv x + bv;  energy_ij = q_i . k_j, att = softmax_j(energy),
//   out_i = sum_j att_ij v_j,  y = gamma * out + x
// The three 1x1 convolutions run as ONE gather_conv GEMM with 2 C/8 + C columns (fp32 rows [position][q | k | v]); this
// kernel is the attention proper on the fp32 matrix cores (v_mfma_f32_32x32x2f32: exact fp32 products).  One workgroup
// of 8 waves per image, v (T x C) and kT (C/8 x T) staged in LDS, wave w owns the 32 query positions i of block w and
// chains two GEMMs whose results are kept TRANSPOSED so that the accumulator layout of the first (column = lane & 31,
// 16 rows per lane) is directly the B operand of the second -- no LDS round trip, no shuffles:
//   E^T[j][i] = sum_d kT[d][j] q[i][d]        A = kT (LDS), B = q (8 registers)  -> per j-block a lane holds 16 j's of ONE i
//   softmax over j: in-register max / sum + one shfl_xor(32) (the other 16 j's of each block live in the other lane half)
//   O^T[c][i] = sum_j v[j][c] p[i][j]         A = v rows (LDS), B = the p registers
// An MFMA contracts two k values per instruction, one from each lane half; WHICH two is free as long as A and B agree, and the
// accumulator layout hands each half its own rows (4 fh + 8 g + r): that pairing is used for the second GEMM.
// SPLIT: x and y in the split-fp16 activation layout (values * kActScale), otherwise fp32 NHWC.
typedef float v16f __attribute__((ext_vector_type(16)));

template <int C, bool SPLIT>
__global__ void __launch_bounds__(512, 2) attention_core_kernel(const float *__restrict__ qkv, const char *__restrict__ x, char *__restrict__ y, float gamma,
                                                                int *__restrict__ sat_flag)
{
    constexpr int DK = C / 8, T = 256, QKV = 2 * DK + C, CB = C / 32;
    extern __shared__ __attribute__((aligned(16))) float att_smem[];
    float *vs = att_smem;                 // [T][C]
    float *kT = att_smem + T * C;         // [DK][T]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    const float *base = qkv + (int64_t)blockIdx.x * T * QKV;
    for (int idx = tid; idx < T * C / 4; idx += 512) {
        const int j = idx / (C / 4), c4 = idx % (C / 4);
        *reinterpret_cast<float4 *>(vs + j * C + c4 * 4) = *reinterpret_cast<const float4 *>(base + (int64_t)j * QKV + 2 * DK + c4 * 4);
    }
    for (int idx = tid; idx < T * DK; idx += 512) {
        const int j = idx / DK, d = idx % DK;
        kT[d * T + j] = base[(int64_t)j * QKV + DK + d];
    }
    __syncthreads();

    const int i0 = wave * 32;
    float qreg[DK / 2];
#pragma unroll
    for (int s = 0; s < DK / 2; ++s) qreg[s] = base[(int64_t)(i0 + frow) * QKV + 2 * s + fh];
    // energies, transposed: e[jb][r] = energy(i = i0 + frow, j = jb * 32 + 4 fh + 8 (r >> 2) + (r & 3))
    v16f e[T / 32];
#pragma unroll
    for (int jb = 0; jb < T / 32; ++jb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) e[jb][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < DK / 2; ++s) e[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kT[(2 * s + fh) * T + jb * 32 + frow], qreg[s], e[jb], 0, 0, 0);
    }
    float m = -__builtin_inff();
#pragma unroll
    for (int jb = 0; jb < T / 32; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, e[jb][r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float ssum = 0.0f;
#pragma unroll
    for (int jb = 0; jb < T / 32; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { e[jb][r] = __expf(e[jb][r] - m); ssum += e[jb][r]; }
    ssum += __shfl_xor(ssum, 32, 64);
    const float sc = gamma / ssum;          // 1 / sum and gamma are applied to the C results instead of the 256 probabilities
    v16f o[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cb][r] = 0.0f;
#pragma unroll
    for (int jb = 0; jb < T / 32; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float *vj = vs + (jb * 32 + 4 * fh + 8 * (r >> 2) + (r & 3)) * C + frow;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) o[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(vj[cb * 32], e[jb][r], o[cb], 0, 0, 0);
        }
    // o[cb][r] = O^T[c = cb * 32 + 4 fh + 8 (r >> 2) + (r & 3)][i = i0 + frow]: 4 consecutive channels per (cb, g = r >> 2)
    const int64_t pos = (int64_t)blockIdx.x * T + i0 + frow;
    bool saturated = false;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = cb * 32 + 8 * g + 4 * fh;
            if constexpr (SPLIT) {
                typedef _Float16 v4h __attribute__((ext_vector_type(4)));
                const int64_t off = pos * C * 4 + (c0 >> 5) * 128 + (c0 & 31) * 2;
                const v4h xh = *reinterpret_cast<const v4h *>(x + off), xl = *reinterpret_cast<const v4h *>(x + off + 64);
                v4h hi, lo;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t = fmaf(sc * kActScale, o[cb][4 * g + r], (float)xh[r] + (float)xl[r]);   // stored values carry the factor kActScale
                    const float cl = fminf(fmaxf(t, -65504.0f), 65504.0f);
                    saturated |= (cl != t);
                    hi[r] = (_Float16)cl;
                    lo[r] = (_Float16)(cl - (float)hi[r]);
                }
                *reinterpret_cast<v4h *>(y + off) = hi;
                *reinterpret_cast<v4h *>(y + off + 64) = lo;
            } else {
                const float4 xv = *reinterpret_cast<const float4 *>(x + (pos * C + c0) * 4);
                float4 out;
                out.x = fmaf(sc, o[cb][4 * g + 0], xv.x);
                out.y = fmaf(sc, o[cb][4 * g + 1], xv.y);
                out.z = fmaf(sc, o[cb][4 * g + 2], xv.z);
                out.w = fmaf(sc, o[cb][4 * g + 3], xv.w);
                *reinterpret_cast<float4 *>(y + (pos * C + c0) * 4) = out;
            }
        }
    if (SPLIT && __any(saturated) && lane == 0) atomicAdd(sat_flag, 1);
}

template <int C>
int launch_attention_core(gl_ctx *ctx, bool split, const float *qkv, const float *x, float *y, int64_t images, float gamma)
{
    constexpr int lds = (256 * C + (C / 8) * 256) * 4;
    GL_ONCE_PER_DEVICE(ctx, \
        GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(attention_core_kernel<C, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(attention_core_kernel<C, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)););
    if (split)
        hipLaunchKernelGGL((attention_core_kernel<C, true>), dim3((unsigned)images), dim3(512), lds, ctx->stream, qkv, reinterpret_cast<const char *>(x),
                           reinterpret_cast<char *>(y), gamma, ctx->h3_sat);
    else
        hipLaunchKernelGGL((attention_core_kernel<C, false>), dim3((unsigned)images), dim3(512), lds, ctx->stream, qkv, reinterpret_cast<const char *>(x),
                           reinterpret_cast<char *>(y), gamma, ctx->h3_sat);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// ---- spectral normalisation on the device (SpectralNorm._update_u_v, gan_models/vaegan/ops.py:32-44), W = w_bar.view(H, -1) ----
// v_raw[j] = sum_i W[i][j] u[i]
__global__ void __launch_bounds__(256) sn_wt_u_kernel(const float *__restrict__ W, const float *__restrict__ u, int H, int Wd, float *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Wd) return;
    float acc = 0.0f;
    for (int i = 0; i < H; ++i) acc = fmaf(W[(int64_t)i * Wd + j], u[i], acc);
    out[j] = acc;
}

__device__ __forceinline__ double block_sum256(double v, double *red)
{
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const double s = red[0];
    __syncthreads();
    return s;
}

// x <- x / (|x| + 1e-12)   (l2normalize, ops.py:19-20); one workgroup
__global__ void __launch_bounds__(256) sn_normalize_kernel(float *__restrict__ x, int n)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)x[i] * (double)x[i];
    const float inv = 1.0f / ((float)sqrt(block_sum256(s, red)) + 1e-12f);
    for (int i = threadIdx.x; i < n; i += 256) x[i] *= inv;
}

// wv[i] = sum_j W[i][j] v[j]; one workgroup per row
__global__ void __launch_bounds__(256) sn_w_v_kernel(const float *__restrict__ W, const float *__restrict__ v, int Wd, float *__restrict__ out)
{
    __shared__ double red[256];
    const float *row = W + (int64_t)blockIdx.x * Wd;
    float acc = 0.0f;
    for (int j = threadIdx.x; j < Wd; j += 256) acc = fmaf(row[j], v[j], acc);
    const double s = block_sum256((double)acc, red);
    if (threadIdx.x == 0) out[blockIdx.x] = (float)s;
}

// u = wv / (|wv| + 1e-12);  with `last`: sigma = u . wv, scale[c] = bn_scale[c] / sigma (and the split-path copy * 2^-wexp).  One workgroup.
__global__ void __launch_bounds__(256) sn_finish_kernel(const float *__restrict__ wv, int H, float *__restrict__ u, int last, const float *__restrict__ bns, int C,
                                                        float *__restrict__ scale, float *__restrict__ scale_h3, float h3_factor)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < H; i += 256) s += (double)wv[i] * (double)wv[i];
    const double ss = block_sum256(s, red);
    const float inv = 1.0f / ((float)sqrt(ss) + 1e-12f);
    for (int i = threadIdx.x; i < H; i += 256) u[i] = wv[i] * inv;
    if (!last) return;
    const float sigma = (float)(ss * (double)inv);                       // u . wv = |wv|^2 / (|wv| + eps)
    for (int c = threadIdx.x; c < C; c += 256) {
        const float sc = bns[c] / sigma;                                 // conv(x, w_bar / sigma) = conv(x, w_bar) / sigma
        scale[c] = sc;
        scale_h3[c] = sc * h3_factor;
    }
}

int ensure_workspace(gl_dcgan *g, int64_t n)
{
    int64_t want = g->chunk > 0 ? g->chunk : 4096;
    // one activation tensor is addressed through a 32-bit buffer descriptor: keep the largest (32 x 32 x C3 values of 4 bytes) under 3 GiB
    const int64_t cap = (int64_t)(0xB0000000ull / ((uint64_t)1024 * g->cout[3] * 4));
    if (want > cap) want = cap;
    if (n < want) want = n;
    if (want <= g->ws_chunk) return GL_OK;
    GL_HIP(hipStreamSynchronize(g->ctx->stream));
    (void)hipFree(g->ws_z);
    g->ws_z = nullptr;
    for (int l = 0; l < 4; ++l) { (void)hipFree(g->ws_a[l]); g->ws_a[l] = nullptr; }
    (void)hipFree(g->ws_p);
    g->ws_p = nullptr;
    (void)hipFree(g->ws_att);
    (void)hipFree(g->ws_qkv);
    g->ws_att = g->ws_qkv = nullptr;
    g->ws_chunk = 0;
    GL_HIP(gl_device_alloc(g->ctx, (void **)&g->ws_z, (size_t)want * g->z_pad * 4));
    int hw = 16;
    for (int l = 0; l < 4; ++l) {
        GL_HIP(gl_device_alloc(g->ctx, (void **)&g->ws_a[l], (size_t)want * hw * g->cout[l] * 4));
        hw *= 4;
    }
    GL_HIP(gl_device_alloc(g->ctx, (void **)&g->ws_p, (size_t)want * 32 * 32 * 16 * g->nc * 4));
    if (g->have_att) {
        GL_HIP(gl_device_alloc(g->ctx, (void **)&g->ws_att, (size_t)want * 256 * g->cout[2] * 4));
        GL_HIP(gl_device_alloc(g->ctx, (void **)&g->ws_qkv, (size_t)want * 256 * (g->cout[2] + g->cout[2] / 4) * 4));
    }
    g->ws_chunk = want;
    return GL_OK;
}

}  // namespace

extern "C" {

int gl_dcgan_create(gl_ctx *ctx, int z_dim, int channels_img, int features_g, gl_dcgan **out)
{
    gl_make_current(ctx);
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
    g->tail_w = nullptr;
    g->fuse_tail = true;
    for (int l = 0; l < 5; ++l) { g->wsplit[l] = nullptr; g->wexp[l] = 0; g->scale_h3[l] = g->shift_h3[l] = nullptr; }
    g->have_att = false;
    g->sn_iters = 1;
    g->sn_hold = false;
    for (int l = 0; l < 4; ++l) { g->have_sn[l] = false; g->sn_w[l] = g->sn_u[l] = g->sn_v[l] = g->sn_wv[l] = g->sn_bns[l] = nullptr; }
    g->att_w = g->att_wsplit = g->att_bias = g->att_ones = g->att_scale_h3 = nullptr;
    g->att_gamma = 0.0f;
    g->att_cols = g->att_cols_pad = g->att_wexp = 0;
    g->ws_att = g->ws_qkv = nullptr;
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
    gl_make_current(g ? g->ctx : nullptr);
    if (!g) return GL_OK;
    (void)hipStreamSynchronize(g->ctx->stream);
    for (int l = 0; l < 5; ++l) (void)hipFree(g->wpack[l]);
    for (int l = 0; l < 4; ++l) { (void)hipFree(g->scale[l]); (void)hipFree(g->shift[l]); (void)hipFree(g->ws_a[l]); }
    (void)hipFree(g->bias_out);
    (void)hipFree(g->ws_z);
    (void)hipFree(g->ws_p);
    (void)hipFree(g->ws_att);
    for (int l = 0; l < 5; ++l) { (void)hipFree(g->wsplit[l]); (void)hipFree(g->scale_h3[l]); (void)hipFree(g->shift_h3[l]); }
    (void)hipFree(g->att_w); (void)hipFree(g->att_wsplit); (void)hipFree(g->att_bias); (void)hipFree(g->att_ones); (void)hipFree(g->att_scale_h3);
    (void)hipFree(g->ws_qkv);
    (void)hipFree(g->ident_scale);
    (void)hipFree(g->ident_shift);
    (void)hipFree(g->tail_w);
    for (int l = 0; l < 4; ++l) { (void)hipFree(g->sn_w[l]); (void)hipFree(g->sn_u[l]); (void)hipFree(g->sn_v[l]); (void)hipFree(g->sn_wv[l]); (void)hipFree(g->sn_bns[l]); }
    delete g;
    return GL_OK;
}

int gl_dcgan_set_chunk(gl_dcgan *g, int64_t images_per_pass)
{
    gl_make_current(g ? g->ctx : nullptr);
    GL_REQUIRE(g && images_per_pass >= 0, "gl_dcgan_set_chunk: bad argument");
    g->chunk = images_per_pass;
    return GL_OK;
}

int gl_dcgan_set_conv_weight(gl_dcgan *g, int layer, const float *w)
{
    gl_make_current(g ? g->ctx : nullptr);
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
        if (layer == 4 && (ci_n == 128 || ci_n == 64) && 16 * co_n == 48) {
            // the same 48 x C weights as the epilogue operand of layer 3 (fused tail, gl_conv.h)
            std::vector<float> img(24 * 1024 / 4);
            gl_pack_tail_weights_host(pk.data(), ci_n, std::ldexp(1.0f, e), img.data());
            if (!g->tail_w) GL_HIP(hipMalloc(&g->tail_w, 24 * 1024));
            GL_HIP(hipMemcpy(g->tail_w, img.data(), 24 * 1024, hipMemcpyHostToDevice));
        }
        g->h3_dirty = true;
    }
    g->have_w[layer] = true;
    return GL_OK;
}

int gl_dcgan_set_bn(gl_dcgan *g, int layer, const float *gamma, const float *beta, const float *mean, const float *var, float eps)
{
    gl_make_current(g ? g->ctx : nullptr);
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
    gl_make_current(g ? g->ctx : nullptr);
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
    gl_make_current(g ? g->ctx : nullptr);
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

/* Spectral normalisation of layer 0..3 (VAEGAN's SpectralNorm(ConvTranspose2d), gan_models/vaegan/ops.py:23-75, train.py:112-123): w_bar
 * [C_in][C_out][4][4] (also installs the convolution weights), the power-iteration vectors u [C_in], v [C_out * 16], the folded BatchNorm
 * scale gamma / sqrt(var + eps) and the folded shift (bias - mean) * scale + beta per output channel.  From then on EVERY gl_dcgan_forward
 * first runs `power_iterations` steps of v = normalise(W^T u), u = normalise(W v) on the device, sigma = u . W v, and uses scale / sigma in
 * the layer's epilogue -- the reference does this on every forward, also in eval mode. */
int gl_dcgan_set_spectral_norm(gl_dcgan *g, int layer, const float *w_bar, const float *u, const float *v, const float *bn_scale, const float *shift,
                               int power_iterations)
{
    gl_make_current(g ? g->ctx : nullptr);
    GL_REQUIRE(g && w_bar && u && v && bn_scale && shift && layer >= 0 && layer < 4 && power_iterations >= 1, "gl_dcgan_set_spectral_norm: bad argument");
    int rc = gl_dcgan_set_conv_weight(g, layer, w_bar);
    if (rc != GL_OK) return rc;
    const int H = g->cin[layer], C = g->cout[layer], Wd = C * 16;
    rc = upload(g->ctx, &g->sn_w[layer], std::vector<float>(w_bar, w_bar + (size_t)H * Wd));
    if (rc == GL_OK) rc = upload(g->ctx, &g->sn_u[layer], std::vector<float>(u, u + H));
    if (rc == GL_OK) rc = upload(g->ctx, &g->sn_v[layer], std::vector<float>(v, v + Wd));
    if (rc == GL_OK) rc = upload(g->ctx, &g->sn_wv[layer], std::vector<float>(H, 0.0f));
    if (rc == GL_OK) rc = upload(g->ctx, &g->sn_bns[layer], std::vector<float>(bn_scale, bn_scale + C));
    if (rc == GL_OK) rc = gl_dcgan_set_affine(g, layer, bn_scale, shift);       // shift is final; scale is rewritten by every forward
    if (rc != GL_OK) return rc;
    g->sn_iters = power_iterations;
    g->have_sn[layer] = true;
    return GL_OK;
}

int gl_dcgan_set_spectral_hold(gl_dcgan *g, int hold)
{
    gl_make_current(g ? g->ctx : nullptr);
    GL_REQUIRE(g, "gl_dcgan_set_spectral_hold: NULL generator");
    g->sn_hold = hold != 0;
    return GL_OK;
}

int gl_dcgan_get_spectral_state(gl_dcgan *g, int layer, float *u_host, float *v_host)
{
    gl_make_current(g ? g->ctx : nullptr);
    GL_REQUIRE(g && layer >= 0 && layer < 4 && g->have_sn[layer] && u_host && v_host, "gl_dcgan_get_spectral_state: bad argument / layer has no spectral norm");
    GL_HIP(hipMemcpyAsync(u_host, g->sn_u[layer], (size_t)g->cin[layer] * 4, hipMemcpyDeviceToHost, g->ctx->stream));
    GL_HIP(hipMemcpyAsync(v_host, g->sn_v[layer], (size_t)g->cout[layer] * 16 * 4, hipMemcpyDeviceToHost, g->ctx->stream));
    GL_HIP(hipStreamSynchronize(g->ctx->stream));
    return GL_OK;
}

/* SelfAttention(in_dim = C2) on the output of layer 2 (16 x 16 x C2, C2 = features_g * 4): query/key conv weights [C2/8][C2],
 * value conv weight [C2][C2] (1x1 convs), biases, gamma.  gan_models/vaegan/ops.py:86-120. */
int gl_dcgan_set_attention(gl_dcgan *g, const float *wq, const float *bq, const float *wk, const float *bk, const float *wv, const float *bv, float gamma)
{
    gl_make_current(g ? g->ctx : nullptr);
    GL_REQUIRE(g && wq && bq && wk && bk && wv && bv, "gl_dcgan_set_attention: NULL argument");
    const int C = g->cout[2], DK = C / 8;
    GL_REQUIRE(C == 64 || C == 128, "gl_dcgan_set_attention: attention width %d unsupported (64 or 128)", C);
    // rows [q (DK) | k (DK) | v (C)] x K = C (one tap: the K order of gl_conv_k_index is the identity), padded to whole 128-column tiles
    const int cols = 2 * DK + C, cols_pad = (int)gl_ceil_div(cols, 128) * 128;
    std::vector<float> pk((size_t)cols_pad * C, 0.0f), bias(cols_pad, 0.0f);
    std::copy(wq, wq + (size_t)DK * C, pk.begin());
    std::copy(wk, wk + (size_t)DK * C, pk.begin() + (size_t)DK * C);
    std::copy(wv, wv + (size_t)C * C, pk.begin() + (size_t)2 * DK * C);
    std::copy(bq, bq + DK, bias.begin());
    std::copy(bk, bk + DK, bias.begin() + DK);
    std::copy(bv, bv + C, bias.begin() + 2 * DK);
    float mx = 0.0f;
    for (float v : pk) mx = std::fmax(mx, std::fabs(v));
    int e = mx > 0.0f ? (int)std::floor(std::log2(8191.0f / mx)) : 0;
    e = e > 30 ? 30 : (e < -30 ? -30 : e);
    std::vector<float> split(pk.size());
    gl_split_weights_host(pk.data(), (size_t)cols_pad, (size_t)C, std::ldexp(1.0f, e), split.data());
    // split path: acc = 2^e * kActScale * (W x)  ->  true q / k / v = acc * 2^-e / kActScale + bias
    int rc = upload(g->ctx, &g->att_w, pk);
    if (rc == GL_OK) rc = upload(g->ctx, &g->att_wsplit, split);
    if (rc == GL_OK) rc = upload(g->ctx, &g->att_bias, bias);
    if (rc == GL_OK) rc = upload(g->ctx, &g->att_ones, std::vector<float>(cols_pad, 1.0f));
    if (rc == GL_OK) rc = upload(g->ctx, &g->att_scale_h3, std::vector<float>(cols_pad, std::ldexp(1.0f / kActScale, -e)));
    if (rc != GL_OK) return rc;
    g->att_cols = cols;
    g->att_cols_pad = cols_pad;
    g->att_wexp = e;
    if (!g->have_att) g->ws_chunk = 0;            // the attention workspaces are allocated with the others
    g->att_gamma = gamma;
    g->have_att = true;
    return GL_OK;
}

/* 0 = fp32 MFMA (every product exact in fp32), 1 = split-fp16 (default): operands carried as hi + lo halves (~22 bits), three fp16
 * MFMAs per product, fp32 accumulation.  Both meet the 1e-4 parity bound; 1 is ~2-3x faster. */
/* on (default): with split-fp16 arithmetic and a last hidden layer of 64 or 128 channels, the 3-channel output layer is evaluated in that layer's
 * epilogue and its activations are never stored; off: the two layers run as separate launches (same values up to fp32 summation order). */
int gl_dcgan_set_fuse_tail(gl_dcgan *g, int on)
{
    gl_make_current(g ? g->ctx : nullptr);
    GL_REQUIRE(g, "gl_dcgan_set_fuse_tail: NULL generator");
    g->fuse_tail = on != 0;
    return GL_OK;
}

int gl_dcgan_set_precision(gl_dcgan *g, int mode)
{
    gl_make_current(g ? g->ctx : nullptr);
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
    gl_make_current(g ? g->ctx : nullptr);
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
    const bool h3 = g->precision == 1;
    const bool fuse_tail = h3 && g->tail_w != nullptr && g->fuse_tail;
    if (h3) {
        rc = dcgan_prepare_h3(g);
        if (rc != GL_OK) return rc;
    }
    for (int l = 0; l < 4; ++l) {
        if (!g->have_sn[l] || g->sn_hold) continue;
        if (!g->scale_h3[l]) {                                      // fp32 mode never built the split-path copies
            rc = dcgan_prepare_h3(g);
            if (rc != GL_OK) return rc;
        }
        const int H = g->cin[l], C = g->cout[l], Wd = C * 16;
        for (int it = 0; it < g->sn_iters; ++it) {
            hipLaunchKernelGGL(sn_wt_u_kernel, dim3((unsigned)gl_ceil_div(Wd, 256)), dim3(256), 0, ctx->stream, g->sn_w[l], g->sn_u[l], H, Wd, g->sn_v[l]);
            hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(256), 0, ctx->stream, g->sn_v[l], Wd);
            hipLaunchKernelGGL(sn_w_v_kernel, dim3((unsigned)H), dim3(256), 0, ctx->stream, g->sn_w[l], g->sn_v[l], Wd, g->sn_wv[l]);
            hipLaunchKernelGGL(sn_finish_kernel, dim3(1), dim3(256), 0, ctx->stream, g->sn_wv[l], H, g->sn_u[l], it + 1 == g->sn_iters ? 1 : 0, g->sn_bns[l], C,
                               g->scale[l], g->scale_h3[l], std::ldexp(1.0f, -g->wexp[l]));
        }
        GL_LAUNCH_CHECK();
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

    // passes: the 4 x 4 -> 8 x 8 layer has one workgroup tile per 2 images and phase-column pair, i.e. images / 2 workgroups per launch, the
    // next layer twice that: passes of a multiple of 512 images fill the 256 CUs a whole number of times, and only the last pass is ragged
    // (12 496 images -- one rank's shard of 8 -- as 3 x 4096 + 208: 25 rounds of the first layer instead of the 28 of four equal passes).
    // Workspaces smaller than that: equal passes.
    const int64_t passes = gl_ceil_div(n, g->ws_chunk);
    const int64_t per_pass = g->ws_chunk >= 512 ? g->ws_chunk - g->ws_chunk % 512 : gl_ceil_div(n, passes);
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
            if (l == 3 && fuse_tail) {
                // layer 4 (128 -> 3, scatter form) rides in this layer's epilogue: the 32 x 32 x 128 activations are never stored
                p.tail_w = g->tail_w; p.tail_out = g->ws_p; p.tail_ld = m * 4 * hw * hw; p.tail_scale = std::ldexp(1.0f / kActScale, -g->wexp[4]);
            }
            rc = launch(p, l, 4);
            if (rc != GL_OK) return rc;
            hw *= 2;
            if (l == 2 && g->have_att) {
                GL_REQUIRE(hw == 16, "gl_dcgan_forward: attention expects a 16 x 16 map");
                GlGatherConv a = {};
                a.in = g->ws_a[2]; a.positions = m * 256; a.H = 16; a.W = 16; a.Cin = g->cout[2];
                a.cols = g->att_cols; a.cols_pad = g->att_cols_pad; a.ntaps = 1; a.tap_dy[0] = 1; a.tap_dx[0] = 1;
                a.out = g->ws_qkv; a.Ho = 16; a.Wo = 16; a.omul = 1; a.oy[0] = 0; a.ox[0] = 0;
                a.shift = g->att_bias; a.cmod = g->att_cols_pad; a.act = 0; a.zero = ctx->zero_page;
                if (h3) { a.wpack = g->att_wsplit; a.scale = g->att_scale_h3; rc = gl_launch_gather_conv_h3(ctx, a, 1); }
                else { a.wpack = g->att_w; a.scale = g->att_ones; rc = gl_launch_gather_conv(ctx, a, 1); }
                if (rc != GL_OK) return rc;
                rc = g->cout[2] == 128 ? launch_attention_core<128>(ctx, h3, g->ws_qkv, g->ws_a[2], g->ws_att, m, g->att_gamma)
                                       : launch_attention_core<64>(ctx, h3, g->ws_qkv, g->ws_a[2], g->ws_att, m, g->att_gamma);
                if (rc != GL_OK) return rc;
            }
        }
        // layer 4: ConvT k4 s2 p1 -> 3 channels: scatter-form GEMM (48 columns) on the matrix cores,
        // then col2im + bias + tanh (+ quantise)
        if (!fuse_tail) {
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
