// LPIPS (v0.1, net-lin, VGG16) feature extractor and the 0.2*LPIPS + L2 nearest-neighbour search.
//   PNetLin.forward   attack_models/lpips_pytorch/models/networks_basic.py:134-181
//   vgg16 slices      attack_models/lpips_pytorch/models/pretrained_networks.py:96-134
//   normalize_tensor  attack_models/lpips_pytorch/util/util.py:70-73
//   Loss('l2-lpips')  attack_models/utils.py:166-176
//
// Reformulation: features are computed ONCE per image instead of once per
// (query, batch) as the reference does, and the distance becomes one dense contraction:
//   d(q,n) = 0.2 * sum_l mean_hw sum_c w_lc (f^q - f^n)^2 + mean_k (x_q - x_n)^2  =  |V_q - V_n|^2
//   V = [ sqrt(0.2 w_lc / (H_l W_l)) * f_lc(h,w) / (|f_l(h,w)|_c + 1e-10)  for every tap l, position, channel ;  x_k / sqrt(D) ]
//   (length 499 712 + 12 288 = 512 000 at 64x64), and |V_q - V_n|^2 = |V_q|^2 + |V_n|^2 - 2 V_q.V_n .
// The convolutions run on the fp32 matrix cores.  V is stored in the split-fp16 layout of gl_conv_h3.hip (every 32 values =
// 32 hi halves + 32 lo halves of V * 2^14, the same 4 bytes per value as fp32) and the contraction runs as three fp16 MFMAs
// per product with fp32 accumulation (~22-bit operands): fp32-class accuracy at ~2.7x the fp32-MFMA rate.
//
// Convolutions reuse gather_conv_kernel (gl_conv.hip): 3x3 p1 = 9 taps; the first layer (3 input channels)
// is im2col'ed by the input kernel into one 32-wide K slice (27 values + 5 zeros).
#include "gl_conv.h"
#include "gl_pair256.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int kNumConv = 13;
const int kCout[kNumConv] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
const int kCin[kNumConv] = {3, 64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512};
// after conv index i: 1 = LPIPS tap, 2 = tap then 2x2 max-pool
const int kAfter[kNumConv] = {0, 2, 0, 2, 0, 0, 2, 0, 0, 2, 0, 0, 1};
const int kTapC[5] = {64, 128, 256, 512, 512};

__constant__ float c_shift[3] = {-.030f, -.088f, -.188f};   // networks_basic.py:115
__constant__ float c_scale[3] = {.458f, .448f, .450f};      // networks_basic.py:116

__device__ __forceinline__ float load_pixel(const uint8_t *p, const float *lut) { return lut[*p]; }
__device__ __forceinline__ float load_pixel(const float *p, const float *) { return *p; }

// 2x2 / stride 2 max-pool, NHWC, C % 4 == 0
__global__ void __launch_bounds__(256) maxpool2_nhwc_kernel(const float *__restrict__ in, int64_t n, int H, int W, int C, float *__restrict__ out)
{
    const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
    const int64_t total = n * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        int64_t r = i / C4;
        const int xo = (int)(r % Wo);
        r /= Wo;
        const int yo = (int)(r % Ho);
        const int64_t im = r / Ho;
        const float4 *p = reinterpret_cast<const float4 *>(in + (((im * H + 2 * yo) * W + 2 * xo) * (int64_t)C)) + c4;
        const float4 a = p[0], b = p[C4], c = p[(int64_t)W * C4], d = p[(int64_t)W * C4 + C4];
        float4 m;
        m.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x));
        m.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
        m.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z));
        m.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
        reinterpret_cast<float4 *>(out)[i] = m;
    }
}

constexpr float kVScale = 16384.0f;     // V is stored as halves of V * 2^14 (values of 1e-4 .. 1e-1 stay out of the fp16 subnormals)

// "Lattice" search rows (8-bit images only): a pixel 2 c / 255 - 1 is m / 255 with m = 2 c - 255 an odd integer of 9 bits, so with the row
// scale u = 255 sqrt(D) 2^e the image part of V * u is m * 2^e -- EXACT in one fp16, no hi / lo pair, one K segment of D halves instead of
// three.  e is chosen so that u lies in (2^13, 2^14] like kVScale (u = 14 133.5 for every square power-of-two image size); the LPIPS part
// is V * u rounded to fp16 as before.  The L2 term of the distance is then exact up to the fp32 accumulation, and the contraction is
// K_lp + D long (the algorithmic length) instead of K_lp + 3 D.
static inline int lp_lattice_exp(int64_t D) { return (int)std::floor(std::log2(16384.0 / (255.0 * std::sqrt((double)D)))); }
static inline double lp_lattice_scale(int64_t D) { return 255.0 * std::sqrt((double)D) * std::ldexp(1.0, lp_lattice_exp(D)); }

// element k of a split-layout row -> byte offset of its hi half (lo half: +64)
__device__ __forceinline__ int64_t split_off(int64_t k) { return (k >> 5) * 128 + (k & 31) * 2; }

__device__ __forceinline__ void split_store2(char *row, int64_t k, float a, float b)   // two consecutive elements, k even
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 hi, lo;
    hi[0] = (_Float16)a; hi[1] = (_Float16)b;
    lo[0] = (_Float16)(a - (float)hi[0]); lo[1] = (_Float16)(b - (float)hi[1]);
    char *dst = row + split_off(k);
    *reinterpret_cast<h2 *>(dst) = hi;
    *reinterpret_cast<h2 *>(dst + 64) = lo;
}

// "search" rows (H1 = true): one fp16 per value (V * 2^14 rounded once), K contiguous -- the operand of feat_knn_h1_kernel
template <bool H1>
__device__ __forceinline__ void v_store2(char *V, int64_t ldv, int64_t row, int64_t k, float a, float b)
{
    if constexpr (H1) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 h;
        h[0] = (_Float16)a; h[1] = (_Float16)b;
        *reinterpret_cast<h2 *>(gl_vrow_elem(V, ldv, row, k)) = h;
    } else {
        split_store2(V + row * ldv, k, a, b);
    }
}

// one wave per position: f / (sqrt(sum_c f^2) + 1e-10) * coef_c  ->  V[img][off + pos*C + c]   (C % 64 == 0), split layout
template <bool H1>
__global__ void __launch_bounds__(256) lpips_tap_kernel(const float *__restrict__ f, int64_t n, int HW, int C, const float *__restrict__ coef,
                                                        char *__restrict__ V, int64_t ldv_bytes, int64_t off, int64_t row0)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t total = n * HW;
    for (int64_t pos = wave; pos < total; pos += nwaves) {
        const float *src = f + pos * C;
        float ss = 0.0f;
        for (int c = 2 * lane; c < C; c += 128) { const float2 t = *reinterpret_cast<const float2 *>(src + c); ss = fmaf(t.x, t.x, ss); ss = fmaf(t.y, t.y, ss); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float inv = kVScale / (sqrtf(ss) + 1e-10f);        // eps outside the sqrt (util.py:72-73)
        const int64_t im = pos / HW;
        const int64_t k0 = off + (pos - im * HW) * C;
        for (int c = 2 * lane; c < C; c += 128) {
            const float2 t = *reinterpret_cast<const float2 *>(src + c);
            v_store2<H1>(V, ldv_bytes, row0 + im, k0 + c, t.x * inv * coef[c], t.y * inv * coef[c + 1]);
        }
    }
}

// image part of V: x_k / sqrt(D), split layout; one thread per pair of elements (D even)
template <typename T>
__global__ void __launch_bounds__(256) image_part_kernel(const T *__restrict__ img, int64_t n, int64_t D, float inv_sqrt_d, char *__restrict__ V, int64_t ldv_bytes,
                                                         int64_t off)
{
    __shared__ float lut[256];
    lut[threadIdx.x] = (float)(2.0 * ((double)threadIdx.x / 255.0) - 1.0);
    __syncthreads();
    const int64_t total = n * D / 2;
    const float sc = inv_sqrt_d * kVScale;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = 2 * i;
        const int64_t im = e / D;
        const int64_t k = e - im * D;
        split_store2(V + im * ldv_bytes, off + k, load_pixel(img + e, lut) * sc, load_pixel(img + e + 1, lut) * sc);
    }
}

// image part of a search row: x_k / sqrt(D) * 2^14 = hi + lo, stored as three segments of Dp halves (Dp = D rounded up to 64, zero
// padded) so that a plain fp16 dot of a query row and a bank row yields hi_q hi_n + hi_q lo_n + lo_q hi_n:
//   query rows  [hi | hi | lo]      bank rows  [hi | lo | hi]        (lo_seg = 2 for queries, 1 for bank rows)
// (8-bit images take only 256 values, so rounding x to ONE half gives a systematic ~1e-4 error in the L2 term; the LPIPS values do not)
template <typename T>
__global__ void __launch_bounds__(256) image_part_h1_kernel(const T *__restrict__ img, int64_t n, int64_t D, int64_t Dp, float inv_sqrt_d, char *__restrict__ V,
                                                            int64_t ldv_bytes, int64_t off, int lo_seg, int pad, int64_t row0)
{
    __shared__ float lut[256];
    lut[threadIdx.x] = (float)(2.0 * ((double)threadIdx.x / 255.0) - 1.0);
    __syncthreads();
    const int64_t total = n * Dp;
    const float sc = inv_sqrt_d * kVScale;
    const int hi2_seg = 3 - lo_seg;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t im = i / Dp;
        const int64_t k = i - im * Dp;
        const float v = k < D ? load_pixel(img + im * D + k, lut) * sc : 0.0f;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        auto at = [&](int64_t kk) -> _Float16 & { return *reinterpret_cast<_Float16 *>(gl_vrow_elem(V, ldv_bytes, row0 + im, off + kk)); };
        at(k) = hi;
        at(hi2_seg * Dp + k) = hi;
        at(lo_seg * Dp + k) = lo;
        if (k < pad) at(3 * Dp + k) = (_Float16)0.0f;       // the zero tail of a padded row (lp_search_pad)
    }
}

// image part of a lattice search row: (2 code - 255) * 2^e, exact in fp16; zero padded to Dp + pad
__global__ void __launch_bounds__(256) image_part_lattice_kernel(const uint8_t *__restrict__ img, int64_t n, int64_t D, int64_t Dp_pad, float two_e,
                                                                 char *__restrict__ V, int64_t ldv_bytes, int64_t off, int64_t row0)
{
    const int64_t total = n * Dp_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t im = i / Dp_pad;
        const int64_t k = i - im * Dp_pad;
        const float v = k < D ? (float)(2 * (int)img[im * D + k] - 255) * two_e : 0.0f;
        *reinterpret_cast<_Float16 *>(gl_vrow_elem(V, ldv_bytes, row0 + im, off + k)) = (_Float16)v;
    }
}

// |row|^2 of a search row (unscaled): sum over the LPIPS halves of h^2 + sum over the image part of (hi + lo)^2
constexpr int kNormSeg = 32768;          // halves of a row per workgroup
__global__ void __launch_bounds__(256) row_sqnorm_h1_part_kernel(const char *__restrict__ V, int64_t ldv_bytes, int64_t K_lp, int64_t Dp, int lo_seg, int nseg,
                                                                  double *__restrict__ part, int64_t row0)
{
    // grid (segment, row): a 17 MB row of a 256 x 256 image is summed by 262 workgroups, not one (passes of that size have 128 rows); the
    // segmentation depends on the row length only, so a row's norm does not depend on the pass it is computed in
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    __shared__ double red[256];
    const int64_t r = blockIdx.y;
    char *Vm = const_cast<char *>(V);
    const int64_t k0 = (int64_t)blockIdx.x * kNormSeg;
    const int64_t kend = k0 + kNormSeg < K_lp + Dp ? k0 + kNormSeg : K_lp + Dp;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int64_t k = k0 + (int64_t)threadIdx.x * 8; k < kend; k += 256 * 8) {
        const h8 h = *reinterpret_cast<const h8 *>(gl_vrow_elem(Vm, ldv_bytes, row0 + r, k));
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
        if (lo_seg > 0 && k >= K_lp) {
            const h8 l = *reinterpret_cast<const h8 *>(gl_vrow_elem(Vm, ldv_bytes, row0 + r, k + lo_seg * Dp));
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)l[j];
        }
        s0 = fmaf(v[0], v[0], s0); s1 = fmaf(v[1], v[1], s1); s2 = fmaf(v[2], v[2], s2); s3 = fmaf(v[3], v[3], s3);
        s0 = fmaf(v[4], v[4], s0); s1 = fmaf(v[5], v[5], s1); s2 = fmaf(v[6], v[6], s2); s3 = fmaf(v[7], v[7], s3);
    }
    red[threadIdx.x] = ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[r * nseg + blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(256) row_sqnorm_h1_final_kernel(const double *__restrict__ part, int64_t n, int nseg, double inv_scale2, float *__restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double s = 0.0;
    for (int i = 0; i < nseg; ++i) s += part[r * nseg + i];
    out[r] = (float)(s * inv_scale2);
}

// 8 consecutive elements of a split row (k % 8 == 0) as floats (still multiplied by kVScale)
__device__ __forceinline__ void split_load8(const char *row, int64_t k, float (&v)[8])
{
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const char *src = row + (k >> 5) * 128 + (k & 31) * 2;
    const h8 hi = *reinterpret_cast<const h8 *>(src), lo = *reinterpret_cast<const h8 *>(src + 64);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)hi[j] + (float)lo[j];
}

// |V_row|^2 (unscaled): one workgroup per row, fp32 chains per thread, fp64 combine
__global__ void __launch_bounds__(256) row_sqnorm_kernel(const char *__restrict__ V, int64_t n, int64_t K, float *__restrict__ out)
{
    __shared__ double red[256];
    for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
        const char *row = V + r * K * 4;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int64_t k = (int64_t)threadIdx.x * 8; k < K; k += 256 * 8) {
            float v[8];
            split_load8(row, k, v);
            s0 = fmaf(v[0], v[0], s0); s1 = fmaf(v[1], v[1], s1); s2 = fmaf(v[2], v[2], s2); s3 = fmaf(v[3], v[3], s3);
            s0 = fmaf(v[4], v[4], s0); s1 = fmaf(v[5], v[5], s1); s2 = fmaf(v[6], v[6], s2); s3 = fmaf(v[7], v[7], s3);
        }
        red[threadIdx.x] = ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[r] = (float)(red[0] / ((double)kVScale * (double)kVScale));
        __syncthreads();
    }
}

// per-row |V_a - V_b|^2 split at K_lp: out_lp = (sum over the LPIPS part) / 0.2, out_l2 = sum over the image part.
// (Loss.forward's loss_lpips / loss_l2 vectors, attack_models/utils.py:173-176).  K_lp and K are multiples of 8.
__global__ void __launch_bounds__(256) feat_rows_dist_kernel(const char *__restrict__ Va, int64_t b, const char *__restrict__ Vb, int64_t b_gt, int64_t K,
                                                             int64_t K_lp, float *__restrict__ out_lp, float *__restrict__ out_l2)
{
    __shared__ double red[2][256];
    for (int64_t r = blockIdx.x; r < b; r += gridDim.x) {
        const char *pa = Va + r * K * 4;
        const char *pb = Vb + (b_gt == 1 ? 0 : r) * K * 4;
        float s_lp = 0.f, s_l2 = 0.f;
        for (int64_t k = (int64_t)threadIdx.x * 8; k < K; k += 256 * 8) {
            float va[8], vb[8];
            split_load8(pa, k, va);
            split_load8(pb, k, vb);
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float t = vb[j] - va[j]; acc = fmaf(t, t, acc); }
            if (k < K_lp) s_lp += acc; else s_l2 += acc;
        }
        red[0][threadIdx.x] = s_lp;
        red[1][threadIdx.x] = s_l2;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
            __syncthreads();
        }
        const double s2 = (double)kVScale * (double)kVScale;
        if (threadIdx.x == 0) { out_lp[r] = (float)(red[0][0] / s2 / 0.2); out_l2[r] = (float)(red[1][0] / s2); }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// split-fp16 ("h3") flavour of the VGG16 pipeline: activations are stored as halves of (value * kVggAct) in the split
// layout of gl_conv_h3.hip, so the 13 convolutions run as three fp16 MFMAs per product.
// ---------------------------------------------------------------------------------------------
constexpr float kVggAct = 4.0f;

// conv1_1 (3 -> 64 channels, 3x3, pad 1) + ReLU straight from the image, on the fp32-input matrix instruction (v_mfma_f32_32x32x2_f32: exact
// fp32, an fmaf chain over k): the scaled input tensor is never materialised.  A = weights [64 channels][k], k = (ky*3+kx)*3 + c (28 of
// the packed 32 columns), held in registers; B = the 27 scaled taps of 32 consecutive positions, gathered from the image; C: lane = position,
// registers = channels, turned through LDS so that every lane stores 16 contiguous bytes.  One wave owns 32 positions per iteration.
// SPLIT: activations * kVggAct in the split layout, else fp32 NHWC (256 bytes per position either way).
__device__ __forceinline__ float conv1_tap(uint8_t code, int c, const float (*lut)[256], float) { return lut[c][code]; }
__device__ __forceinline__ float conv1_tap(float v, int c, const float (*)[256], float act) { return __fdiv_rn(__fsub_rn(v, c_shift[c]), c_scale[c]) * act; }

template <typename T, bool SPLIT>
__global__ void __launch_bounds__(256, 3) vgg_conv1_kernel(const T *__restrict__ img, int n, int H, int W, const float *__restrict__ wpack,
                                                           const float *__restrict__ bias, char *__restrict__ out, int *__restrict__ sat_flag, float act)
{
    typedef float v16f __attribute__((ext_vector_type(16)));
    typedef _Float16 v4h __attribute__((ext_vector_type(4)));
    constexpr int kRow = 272;                                   // bytes of LDS per position (256 + padding)
    __shared__ float lut[3][256];                               // code -> ((2 code / 255 - 1) - shift_c) / scale_c * act (attack_models/utils.py:82)
    __shared__ __attribute__((aligned(16))) char turn[4][32 * kRow];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        lut[c][threadIdx.x] = __fdiv_rn(__fsub_rn((float)(2.0 * ((double)threadIdx.x / 255.0) - 1.0), c_shift[c]), c_scale[c]) * act;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    float wa[2][14];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 14; ++s) wa[t][s] = wpack[(t * 32 + col) * 32 + 2 * s + half];
    // k = 27 (the first of the packed row's zero columns) carries the bias: its "tap" is the constant act
    if (half) { wa[0][13] = bias[col]; wa[1][13] = bias[32 + col]; }
    const int HW = H * W;
    const int total = n * HW;                                   // positions of the pass (host-checked: n * 3 * H * W < 2^31)
    const int groups = (total + 31) / 32;
    const int stride = (int)gridDim.x * 4;
    char *mine = turn[wave];
    bool saturated = false;

    // the 14 taps of this lane (k = 2 s + half) for position group g: unconditional loads from clamped offsets, all in flight together;
    // bit s of okm: the tap lies inside the image
    auto fetch = [&](int g, T (&raw)[14], uint32_t &okm) {
        const int pos = g * 32 + col;
        const bool live = (g < groups) & (pos < total);
        const int im = live ? pos / HW : 0;
        const int rem = live ? pos - im * HW : 0;
        const int y = rem / W, x = rem - y * W;
        const int at = im * 3 * HW + rem;
        okm = 0;
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int k0 = 2 * s, k1 = 2 * s + 1;
            const int c0 = k0 % 3, ky0 = (k0 / 3) / 3, kx0 = (k0 / 3) % 3;
            const int c1 = k1 % 3, ky1 = (k1 / 3) / 3, kx1 = (k1 / 3) % 3;
            const int c = half ? c1 : c0, dy = (half ? ky1 : ky0) - 1, dx = (half ? kx1 : kx0) - 1;
            const bool real = half ? (k1 < 27) : (k0 < 27);
            const int yy = y + dy, xx = x + dx;
            const bool ok = live & real & (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W);
            okm |= ok ? 1u << s : 0u;
            raw[s] = img[ok ? at + c * HW + dy * W + dx : 0];
        }
        if (live & (half != 0)) okm |= 1u << 14;                // the bias tap
    };

    T raw[14];
    uint32_t okm;
    int g = (int)blockIdx.x * 4 + wave;
    fetch(g, raw, okm);
    for (; g < groups; g += stride) {
        float b[14];
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int c = half ? (2 * s + 1) % 3 : (2 * s) % 3;
            b[s] = conv1_tap(raw[s], c, lut, act) * ((okm >> s) & 1u ? 1.0f : 0.0f);
        }
        if ((okm >> 14) & 1u) b[13] = act;
        fetch(g + stride, raw, okm);                            // the next group's taps travel while this one multiplies
        v16f acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[0][s], b[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[1][s], b[s], acc[1], 0, 0, 0);
        }
        // C: column (position) = lane & 31, row (channel within the tile) = (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = t * 32 + 8 * q + 4 * half;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[t][4 * q + r], 0.0f);
                if constexpr (SPLIT) {
                    char *dst = mine + col * kRow + t * 128 + (ch & 31) * 2;
                    v4h hi, lo;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float cl = fminf(v[r], 65504.0f);
                        saturated |= cl != v[r];
                        hi[r] = (_Float16)cl;
                        lo[r] = (_Float16)__fsub_rn(cl, (float)hi[r]);
                    }
                    *reinterpret_cast<v4h *>(dst) = hi;
                    *reinterpret_cast<v4h *>(dst + 64) = lo;
                } else {
                    *reinterpret_cast<float4 *>(mine + col * kRow + ch * 4) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int pp = it * 4 + (lane >> 4), chunk = lane & 15;
            const uint4 v = *reinterpret_cast<const uint4 *>(mine + pp * kRow + chunk * 16);
            if (g * 32 + pp < total) *reinterpret_cast<uint4 *>(out + (int64_t)(g * 32 + pp) * 256 + chunk * 16) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (SPLIT && __any(saturated) && lane == 0) atomicAdd(sat_flag, 1);
}

// 2x2 / stride 2 max-pool on the split layout; one thread per 8 channels of an output position (C % 32 == 0)
__global__ void __launch_bounds__(256) maxpool2_split_kernel(const char *__restrict__ in, int64_t n, int H, int W, int C, char *__restrict__ out)
{
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const int Ho = H / 2, Wo = W / 2, C8 = C / 8;
    const int64_t total = n * Ho * Wo * C8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8) * 8;
        int64_t r = i / C8;
        const int xo = (int)(r % Wo);
        r /= Wo;
        const int yo = (int)(r % Ho);
        const int64_t im = r / Ho;
        const int64_t coff = (c >> 5) * 128 + (c & 31) * 2;
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -__builtin_inff();
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const char *src = in + (((im * H + 2 * yo + dy) * W + 2 * xo + dx) * (int64_t)C) * 4 + coff;
                const h8 hi = *reinterpret_cast<const h8 *>(src), lo = *reinterpret_cast<const h8 *>(src + 64);
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], (float)hi[j] + (float)lo[j]);
            }
        h8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) { hi[j] = (_Float16)m[j]; lo[j] = (_Float16)(m[j] - (float)hi[j]); }
        char *dst = out + (((im * Ho + yo) * Wo + xo) * (int64_t)C) * 4 + coff;
        *reinterpret_cast<h8 *>(dst) = hi;
        *reinterpret_cast<h8 *>(dst + 64) = lo;
    }
}

// Sum of squares over the channels of a position in the convolution epilogue's canonical order (gl_conv_h3_epi.h), so that a tap taken
// in the epilogue and one taken here from the stored activation give the same bits: per 16-channel tile, 4 groups of 4 consecutive
// channels, each an fmaf chain from 0, combined as (g0 + g1) + (g2 + g3); then a balanced binary tree over the tiles.  A lane owns 8
// channels = 2 groups; lanes 2 t and 2 t + 1 hold tile t.
constexpr float kTapEps1 = 1e-10f;          // normalize_tensor's eps (util/util.py:72); activations stored as A f use A eps: (A f) / (|A f| + A eps) = f / (|f| + eps)
__device__ __forceinline__ float tap_sumsq8(const float (&v)[8])
{
    float a = 0.0f, b = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a = fmaf(v[j], v[j], a); b = fmaf(v[4 + j], v[4 + j], b); }
    return __fadd_rn(a, b);
}
template <int G>
__device__ __forceinline__ float tap_sumsq_across(float s)
{
#pragma unroll
    for (int o = 1; o < G; o <<= 1) s = __fadd_rn(s, __shfl_xor(s, o, 64));
    return s;
}

// lpips_tap_kernel for split-layout activations (values carry the factor kVggAct, which cancels in the normalisation).
// A lane owns 8 consecutive channels (16 B of hi halves + 16 B of lo halves); C / 8 lanes share a position, so a wave covers
// 512 / C positions per pass with 2 KiB of contiguous reads and C * 2 (or 4) bytes of contiguous writes per position.
template <bool H1, int C>
__global__ void __launch_bounds__(256) lpips_tap_split_kernel(const char *__restrict__ f, int64_t n, int HW, const float *__restrict__ coef,
                                                              char *__restrict__ V, int64_t ldv_bytes, int64_t off, float tap_eps, int64_t row0)
{
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    constexpr int G = C / 8;            // lanes per position
    constexpr int P = 64 / G;           // positions per wave and pass
    const int lane = threadIdx.x & 63;
    const int cb = lane % G, pl = lane / G;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t total = n * HW;
    float cf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cf[j] = coef[cb * 8 + j];
    const int in_off = (cb >> 2) * 128 + (cb & 3) * 16;          // hi halves of channels 8 cb .. 8 cb + 7 (lo: + 64)
    for (int64_t p0 = wave * P; p0 < total; p0 += nwaves * P) {
        const int64_t pos = p0 + pl;
        const bool live = pos < total;
        float v[8];
        float ss = 0.0f;
        if (live) {
            const char *src = f + pos * C * 4 + in_off;
            const h8 hi = *reinterpret_cast<const h8 *>(src), lo = *reinterpret_cast<const h8 *>(src + 64);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __fadd_rn((float)hi[j], (float)lo[j]);
            ss = tap_sumsq8(v);
        }
        ss = tap_sumsq_across<G>(ss);
        if (!live) continue;
        const float inv = __fdiv_rn(kVScale, __fadd_rn(__fsqrt_rn(ss), tap_eps));        // (A f) / (|A f| + A eps) = f / (|f| + eps)
        const int64_t im = pos / HW;
        const int64_t k = off + (pos - im * HW) * C + cb * 8;
        h8 oh, ol;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = gl_tap_value(v[j], inv, cf[j]);
            oh[j] = (_Float16)t;
            ol[j] = (_Float16)__fsub_rn(t, (float)oh[j]);
        }
        if constexpr (H1) {
            *reinterpret_cast<h8 *>(gl_vrow_elem(V, ldv_bytes, row0 + im, k)) = oh;
        } else {
            char *dst = V + (row0 + im) * ldv_bytes + split_off(k);
            *reinterpret_cast<h8 *>(dst) = oh;
            *reinterpret_cast<h8 *>(dst + 64) = ol;
        }
    }
}

// lpips_tap_split_kernel + maxpool2_split_kernel in one pass over the activation (taps relu1_2, 2_2, 3_3, 4_3 are each followed by a
// 2x2 max-pool, pretrained_networks.py:108-115): a lane owns 8 channels of one 2 x 2 window, i.e. of 4 positions; per position the
// same loads, the same sum of squares (same shuffle order) and the same V values as the tap kernel, plus max over the 4 positions
// -> the pooled activation.  The activation (1 MB per image at relu1_2) is read once instead of twice.
template <bool H1, int C>
__global__ void __launch_bounds__(256) lpips_tap_pool_split_kernel(const char *__restrict__ f, int64_t n, int H, int W, const float *__restrict__ coef,
                                                                   char *__restrict__ V, int64_t ldv_bytes, int64_t off, char *__restrict__ pooled, float tap_eps, int64_t row0)
{
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    constexpr int G = C / 8;            // lanes per window
    constexpr int P = 64 / G;           // windows per wave and pass
    const int lane = threadIdx.x & 63;
    const int cb = lane % G, pl = lane / G;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int Ho = H / 2, Wo = W / 2;
    const int64_t HW = (int64_t)H * W, total = n * Ho * Wo;
    float cf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cf[j] = coef[cb * 8 + j];
    const int in_off = (cb >> 2) * 128 + (cb & 3) * 16;          // hi halves of channels 8 cb .. 8 cb + 7 (lo: + 64)
    for (int64_t w0 = wave * P; w0 < total; w0 += nwaves * P) {
        const int64_t win = w0 + pl;
        const bool live = win < total;
        const int64_t im = live ? win / (Ho * Wo) : 0;
        const int rem = live ? (int)(win - im * (Ho * Wo)) : 0;
        const int yo = rem / Wo, xo = rem - yo * Wo;
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -__builtin_inff();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t pin = (int64_t)(2 * yo + (q >> 1)) * W + 2 * xo + (q & 1);      // position inside the image
            float v[8];
            float ss = 0.0f;
            if (live) {
                const char *src = f + (im * HW + pin) * C * 4 + in_off;
                const h8 hi = *reinterpret_cast<const h8 *>(src), lo = *reinterpret_cast<const h8 *>(src + 64);
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[j] = __fadd_rn((float)hi[j], (float)lo[j]); m[j] = fmaxf(m[j], v[j]); }
                ss = tap_sumsq8(v);
            }
            ss = tap_sumsq_across<G>(ss);
            if (!live) continue;
            const float inv = __fdiv_rn(kVScale, __fadd_rn(__fsqrt_rn(ss), tap_eps));        // (A f) / (|A f| + A eps) = f / (|f| + eps)
            const int64_t k = off + pin * C + cb * 8;
            h8 oh, ol;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = gl_tap_value(v[j], inv, cf[j]);
                oh[j] = (_Float16)t;
                ol[j] = (_Float16)__fsub_rn(t, (float)oh[j]);
            }
            if constexpr (H1) {
                *reinterpret_cast<h8 *>(gl_vrow_elem(V, ldv_bytes, row0 + im, k)) = oh;
            } else {
                char *dst = V + (row0 + im) * ldv_bytes + split_off(k);
                *reinterpret_cast<h8 *>(dst) = oh;
                *reinterpret_cast<h8 *>(dst + 64) = ol;
            }
        }
        if (!live) continue;
        h8 ph, plo;
#pragma unroll
        for (int j = 0; j < 8; ++j) { ph[j] = (_Float16)m[j]; plo[j] = (_Float16)__fsub_rn(m[j], (float)ph[j]); }
        char *dst = pooled + (win * C) * 4 + in_off;
        *reinterpret_cast<h8 *>(dst) = ph;
        *reinterpret_cast<h8 *>(dst + 64) = plo;
    }
}

int stream_blocks(int64_t items);

template <bool H1>
void launch_tap_split(hipStream_t st, const char *f, int64_t n, int HW, int C, const float *coef, char *V, int64_t ldv, int64_t off, float eps, int64_t row0)
{
    const dim3 grid((unsigned)stream_blocks(n * HW * (C / 8)));
    switch (C) {
    case 64: hipLaunchKernelGGL((lpips_tap_split_kernel<H1, 64>), grid, dim3(256), 0, st, f, n, HW, coef, V, ldv, off, eps, row0); break;
    case 128: hipLaunchKernelGGL((lpips_tap_split_kernel<H1, 128>), grid, dim3(256), 0, st, f, n, HW, coef, V, ldv, off, eps, row0); break;
    case 256: hipLaunchKernelGGL((lpips_tap_split_kernel<H1, 256>), grid, dim3(256), 0, st, f, n, HW, coef, V, ldv, off, eps, row0); break;
    default: hipLaunchKernelGGL((lpips_tap_split_kernel<H1, 512>), grid, dim3(256), 0, st, f, n, HW, coef, V, ldv, off, eps, row0); break;
    }
}

template <bool H1>
void launch_tap_pool_split(hipStream_t st, const char *f, int64_t n, int H, int W, int C, const float *coef, char *V, int64_t ldv, int64_t off, char *pooled, float eps, int64_t row0)
{
    const dim3 grid((unsigned)stream_blocks(n * (H / 2) * (W / 2) * (C / 8)));
    switch (C) {
    case 64: hipLaunchKernelGGL((lpips_tap_pool_split_kernel<H1, 64>), grid, dim3(256), 0, st, f, n, H, W, coef, V, ldv, off, pooled, eps, row0); break;
    case 128: hipLaunchKernelGGL((lpips_tap_pool_split_kernel<H1, 128>), grid, dim3(256), 0, st, f, n, H, W, coef, V, ldv, off, pooled, eps, row0); break;
    case 256: hipLaunchKernelGGL((lpips_tap_pool_split_kernel<H1, 256>), grid, dim3(256), 0, st, f, n, H, W, coef, V, ldv, off, pooled, eps, row0); break;
    default: hipLaunchKernelGGL((lpips_tap_pool_split_kernel<H1, 512>), grid, dim3(256), 0, st, f, n, H, W, coef, V, ldv, off, pooled, eps, row0); break;
    }
}

// generic fp32 rows for the split-fp16 search (gl_rows_knn_split): one workgroup per row.
//   e = the largest power of two with max|x| * 2^e <= 2^15   (per ROW, so rows of very different magnitude keep ~22 bits each)
//   V[row] = hi + lo halves of x * 2^e in the split layout, zero padded to Kp;  norms[row] = sum x^2 / d;  scales[row] = 2^-e
__global__ void __launch_bounds__(256) rows_split_kernel(const float *__restrict__ x, int64_t n, int64_t d, int64_t Kp, char *__restrict__ V,
                                                         float *__restrict__ norms, float *__restrict__ scales)
{
    __shared__ double red[256];
    __shared__ float redm[256];
    for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
        const float *row = x + r * d;
        float m = 0.0f, s0 = 0.0f, s1 = 0.0f;
        for (int64_t k = threadIdx.x; k < d; k += 512) {
            const float a = row[k], b = (k + 256 < d) ? row[k + 256] : 0.0f;
            m = fmaxf(m, fmaxf(fabsf(a), fabsf(b)));
            s0 = fmaf(a, a, s0);
            s1 = fmaf(b, b, s1);
        }
        red[threadIdx.x] = (double)s0 + (double)s1;
        redm[threadIdx.x] = m;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) { red[threadIdx.x] += red[threadIdx.x + o]; redm[threadIdx.x] = fmaxf(redm[threadIdx.x], redm[threadIdx.x + o]); }
            __syncthreads();
        }
        const float mx = redm[0];
        int e = 0;
        if (mx > 0.0f && mx < __builtin_inff()) {
            int ex;
            (void)frexpf(mx, &ex);                 // mx = f * 2^ex, f in [0.5, 1)
            e = 15 - ex;                           // mx * 2^e in [2^14, 2^15)
            e = e > 100 ? 100 : (e < -100 ? -100 : e);
        }
        const float up = ldexpf(1.0f, e);
        if (threadIdx.x == 0) { norms[r] = (float)(red[0] / (double)d); scales[r] = ldexpf(1.0f, -e); }
        char *dst = V + r * Kp * 4;
        for (int64_t k = threadIdx.x; k < Kp; k += 256) {
            const float v = k < d ? row[k] * up : 0.0f;
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            char *p = dst + split_off(k);
            *reinterpret_cast<_Float16 *>(p) = hi;
            *reinterpret_cast<_Float16 *>(p + 64) = lo;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// pairwise |V_q - V_n|^2 + argmin on the fp16 matrix cores, split operands (see gl_conv_h3.hip):
// C[n][q] = sum_k (hi_n hi_q + hi_n lo_q + lo_n hi_q) = 2^28 * V_n . V_q   (v_mfma_f32_16x16x32_f16, fp32 accumulate)
// tile 128 bank rows x 128 queries, 4 waves as 2 x 2, each 64 x 64 = 4 x 4 tiles of 16 x 16; K slices of 32 values
// (128 bytes) double buffered in LDS via global_load_lds (rows may lie beyond 4 GiB: no buffer descriptor here);
// epilogue: dist = max(|V_q|^2 + |V_n|^2 - 2 C / 2^28, 0), key = float_bits(dist) << 32 | global index, atomicMin.
// ---------------------------------------------------------------------------------------------
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int FT = 128, FROW = 128, FOPER = FT * FROW;
constexpr int kSplitSeg = 2048;          // slices (of 32 values) per accumulation segment

__global__ void __launch_bounds__(256, 2)
feat_knn_kernel(const char *__restrict__ bank, const float *__restrict__ bank_norm, int64_t n_rows, int64_t index_base,
                const char *__restrict__ query, const float *__restrict__ query_norm, int64_t nq, int64_t K,
                unsigned long long *__restrict__ keys, int q_tiles, int n_tiles, float inv_s2, const float *__restrict__ bank_scale,
                const float *__restrict__ query_scale)
{
    // inv_s2: 1 / (scale of the stored halves)^2; bank_scale / query_scale (optional): per-row factors 2^-e of rows stored as x * 2^e
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned id = gl_xcd_remap(blockIdx.x, (unsigned)q_tiles * (unsigned)n_tiles);
    int qt, nt;
    gl_strip_order(id, q_tiles, n_tiles, qt, nt);
    const int64_t n0 = (int64_t)nt * FT, q0 = (int64_t)qt * FT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wq = wave & 1;
    const int rsub = lane >> 3, slot = lane & 7;
    const int64_t row_bytes = K * 4;

    const char *a_src[4], *b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + rsub;
        int64_t gn = n0 + r, gq = q0 + r;
        if (gn >= n_rows) gn = n_rows - 1;      // clamped duplicates are masked in the epilogue
        if (gq >= nq) gq = nq - 1;
        a_src[i] = bank + gn * row_bytes + (slot ^ (r & 7)) * 16;
        b_src[i] = query + gq * row_bytes + (slot ^ (r & 7)) * 16;
    }
    auto stage = [&](int64_t kt, char *buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) gl_glds16(a_src[i] + kt * FROW, buf + (wave * 4 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) gl_glds16(b_src[i] + kt * FROW, buf + FOPER + (wave * 4 + i) * 1024);
    };

    v4f acc[4][4], tot[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = tot[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

    const int64_t nk = K / 32;
    stage(0, smem);
    const int frow = lane & 15, fk = lane >> 4;
    for (int64_t kt = 0; kt < nk; ++kt) {
        __syncthreads();
        char *cur = smem + (kt & 1) * 2 * FOPER;
        if (kt + 1 < nk) stage(kt + 1, smem + ((kt + 1) & 1) * 2 * FOPER);
        const char *la = cur + (wn * 64) * FROW;
        const char *lb = cur + FOPER + (wq * 64) * FROW;
        v8h a_hi[4], a_lo[4], b_hi[4], b_lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 16 + frow;
            a_hi[i] = *reinterpret_cast<const v8h *>(la + r * FROW + ((fk ^ (r & 7)) << 4));
            a_lo[i] = *reinterpret_cast<const v8h *>(la + r * FROW + (((4 + fk) ^ (r & 7)) << 4));
            b_hi[i] = *reinterpret_cast<const v8h *>(lb + r * FROW + ((fk ^ (r & 7)) << 4));
            b_lo[i] = *reinterpret_cast<const v8h *>(lb + r * FROW + (((4 + fk) ^ (r & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[i], b_hi[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[i], b_lo[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[i], b_hi[j], acc[i][j], 0, 0, 0);
            }
        if ((kt & (kSplitSeg - 1)) == kSplitSeg - 1) {
            // two-level sum: a row of a 256 x 256 image is 8.2 M values, and one fp32 chain that long loses ~2e-5 of a distance
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { tot[i][j] += acc[i][j]; acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f}; }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += tot[i][j];

    // epilogue: C tile 16x16: column (query) = lane & 15, row (bank) = 4 * (lane >> 4) + reg
    const int64_t nbase = n0 + wn * 64 + fk * 4;
    float bn[4][4], bs[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t n = nbase + i * 16 + r;
            bn[i][r] = n < n_rows ? bank_norm[n] : 0.0f;
            bs[i][r] = (bank_scale && n < n_rows) ? bank_scale[n] : 1.0f;
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t q = q0 + wq * 64 + j * 16 + frow;
        const float qn = q < nq ? query_norm[q] : 0.0f;
        const float qs = -2.0f * inv_s2 * ((query_scale && q < nq) ? query_scale[q] : 1.0f);
        unsigned long long best = ~0ull;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t n = nbase + i * 16 + r;
                const float d = fmaxf(fmaf(qs * bs[i][r], acc[i][j][r], __fadd_rn(qn, bn[i][r])), 0.0f);
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(index_base + n);
                if (n < n_rows && key < best) best = key;
            }
        unsigned long long o = __shfl_xor(best, 16, 64);
        best = o < best ? o : best;
        o = __shfl_xor(best, 32, 64);
        best = o < best ? o : best;
        if (fk == 0 && q < nq && best != ~0ull) atomicMin(&keys[q], best);
    }
}

// ---------------------------------------------------------------------------------------------
// feat_knn_h1_kernel: the same search on "search rows" (one half per value, gl_lpips_search_features_*): a plain fp16 GEMM,
// one v_mfma_f32_16x16x32_f16 per product instead of three and half the operand bytes.
// The 128 x 128 kernel above is fed from beyond L2 at ~19 B/clk/CU (the Infinity-Cache gather rate) and that, not the
// matrix pipe, sets its time; so this one uses a 256 x 256 tile (bytes per MFMA halved): 8 waves as 2 (bank) x 4 (query),
// each 128 bank rows x 64 queries = 8 x 4 tiles of 16 x 16; K slices of 64 halves (128 B per row, 64 KiB per slice for both
// operands) double buffered in 128 KiB of LDS, one workgroup per CU.  Block order: strips of 4 bank tiles with the bank tile
// fastest, so the 32 workgroups of an XCD cover 4 bank x 8 query tiles and share operand panels in its L2.
// ---------------------------------------------------------------------------------------------
constexpr int GT = 256, GOPER = GT * FROW;

#ifdef GL_TUNING      // the round-1 kernel and the one-workgroup-per-tile pipelined kernel: A/B material of tools/bench_pairwise.py only
__global__ void __launch_bounds__(512, 2)
feat_knn_h1_kernel(const char *__restrict__ bank, const float *__restrict__ bank_norm, int64_t n_rows, int64_t index_base,
                   const char *__restrict__ query, const float *__restrict__ query_norm, int64_t nq, int64_t K1,
                   unsigned long long *__restrict__ keys, int q_tiles, int n_tiles, float inv_s2)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][bank 32 KiB | query 32 KiB]
    const unsigned id = gl_xcd_remap(blockIdx.x, (unsigned)q_tiles * (unsigned)n_tiles);
    int qt, nt;
    {
        constexpr int STRIP = 4;
        const unsigned per_strip = (unsigned)STRIP * (unsigned)q_tiles;
        const int strip = (int)(id / per_strip);
        const unsigned r = id % per_strip;
        const int width = n_tiles - strip * STRIP < STRIP ? n_tiles - strip * STRIP : STRIP;
        nt = strip * STRIP + (int)(r % (unsigned)width);
        qt = (int)(r / (unsigned)width);
    }
    const int64_t n0 = (int64_t)nt * GT, q0 = (int64_t)qt * GT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wq = wave & 3;
    const int rsub = lane >> 3, slot = lane & 7;
    const int64_t row_bytes = K1 * 2;

    // staging: a slice is 512 rows x 128 B = 64 pieces of 1 KiB (8 rows each); wave w loads bank pieces 4w..4w+3 and query pieces 4w..4w+3
    const char *a_src[4], *b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + rsub;
        int64_t gn = n0 + r, gq = q0 + r;
        if (gn >= n_rows) gn = n_rows - 1;      // clamped duplicates are masked in the epilogue
        if (gq >= nq) gq = nq - 1;
        a_src[i] = bank + gn * row_bytes + (slot ^ (r & 7)) * 16;
        b_src[i] = query + gq * row_bytes + (slot ^ (r & 7)) * 16;
    }
    auto stage = [&](int64_t kt, char *buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) gl_glds16(a_src[i] + kt * FROW, buf + (wave * 4 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) gl_glds16(b_src[i] + kt * FROW, buf + GOPER + (wave * 4 + i) * 1024);
    };

    v4f acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

    const int64_t nk = K1 / 64;
    stage(0, smem);
    const int frow = lane & 15, fk = lane >> 4;
    for (int64_t kt = 0; kt < nk; ++kt) {
        __syncthreads();
        const char *cur = smem + (kt & 1) * 2 * GOPER;
        if (kt + 1 < nk) stage(kt + 1, smem + ((kt + 1) & 1) * 2 * GOPER);
        const char *la = cur + (wn * 128) * FROW;
        const char *lb = cur + GOPER + (wq * 64) * FROW;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v8h a[8], b[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = i * 16 + frow;
                a[i] = *reinterpret_cast<const v8h *>(la + r * FROW + (((4 * ks + fk) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = j * 16 + frow;
                b[j] = *reinterpret_cast<const v8h *>(lb + r * FROW + (((4 * ks + fk) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: C tile 16x16: column (query) = lane & 15, row (bank) = 4 * (lane >> 4) + reg
    const int64_t nbase = n0 + wn * 128 + fk * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t q = q0 + wq * 64 + j * 16 + frow;
        const float qn = q < nq ? query_norm[q] : 0.0f;
        unsigned long long best = ~0ull;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t n = nbase + i * 16 + r;
                const float bn = n < n_rows ? bank_norm[n] : 0.0f;
                const float d = fmaxf(fmaf(-2.0f * inv_s2, acc[i][j][r], __fadd_rn(qn, bn)), 0.0f);
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(index_base + n);
                if (n < n_rows && key < best) best = key;
            }
        unsigned long long o = __shfl_xor(best, 16, 64);
        best = o < best ? o : best;
        o = __shfl_xor(best, 32, 64);
        best = o < best ? o : best;
        if (fk == 0 && q < nq && best != ~0ull) atomicMin(&keys[q], best);
    }
}

// the same tile on the shared software-pipelined main loop (gl_pair256.h)
template <int SPREAD>
__global__ void __launch_bounds__(512, 2)
feat_knn_h1p_kernel(const char *__restrict__ bank, const float *__restrict__ bank_norm, int64_t n_rows, int64_t index_base,
                    const char *__restrict__ query, const float *__restrict__ query_norm, int64_t nq, int64_t K1,
                    unsigned long long *__restrict__ keys, int q_tiles, int n_tiles, float inv_s2)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned id = gl_xcd_remap(blockIdx.x, (unsigned)q_tiles * (unsigned)n_tiles);
    int qt, nt;
    {
        constexpr int STRIP = 4;
        const unsigned per_strip = (unsigned)STRIP * (unsigned)q_tiles;
        const int strip = (int)(id / per_strip);
        const unsigned r = id % per_strip;
        const int width = n_tiles - strip * STRIP < STRIP ? n_tiles - strip * STRIP : STRIP;
        nt = strip * STRIP + (int)(r % (unsigned)width);
        qt = (int)(r / (unsigned)width);
    }
    const int64_t n0 = (int64_t)nt * GT, q0 = (int64_t)qt * GT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wq = wave & 3;
    const int frow = lane & 15, fk = lane >> 4;
    const gl_pair256::Source sa = gl_pair256::make_source(bank, n0, n_rows, K1 * 2, wave, lane);
    const gl_pair256::Source sb = gl_pair256::make_source(query, q0, nq, K1 * 2, wave, lane);
    v4f acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    gl_pair256::mainloop<v8h, 0, SPREAD>(sa, sb, K1 / 64, smem, acc, wave, lane,
                              [](const v8h &a, const v8h &b, const v4f &c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); });

    const int64_t nbase = n0 + wn * 128 + fk * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t q = q0 + wq * 64 + j * 16 + frow;
        const float qn = q < nq ? query_norm[q] : 0.0f;
        unsigned long long best = ~0ull;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t n = nbase + i * 16 + r;
                const float bn = n < n_rows ? bank_norm[n] : 0.0f;
                const float d = fmaxf(fmaf(-2.0f * inv_s2, acc[i][j][r], __fadd_rn(qn, bn)), 0.0f);
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(index_base + n);
                if (n < n_rows && key < best) best = key;
            }
        unsigned long long o = __shfl_xor(best, 16, 64);
        best = o < best ? o : best;
        o = __shfl_xor(best, 32, 64);
        best = o < best ? o : best;
        if (fk == 0 && q < nq && best != ~0ull) atomicMin(&keys[q], best);
    }
}

#endif  // GL_TUNING

// ---------------------------------------------------------------------------------------------
// feat_knn_h1c_kernel: the same search as ONE persistent launch of `clusters x members` workgroups (8 x 32 on an MI355X: one
// workgroup per CU).  Why: K is 536 576 halves at 64 x 64 (8 384 slices, ~9 ms per tile) and 8.6 M at 256 x 256; with one
// workgroup per tile the 32 tiles that share an XCD's L2 start at different times once the first wave of tiles has finished, sit
// at unrelated K positions and stop sharing operand panels (measured: 3.5 TB fetched beyond L2 per launch at 10k x 100k, against
// 1.6 TB if every 4 x 8 group of tiles walked K together).  Here the workgroups with the same blockIdx & 7 (dealt to one XCD by
// the dispatcher; a different placement costs speed, never correctness) form a cluster that takes a super-tile of 4 bank x 8 query
// tiles at a time and meets at a counter before every super-tile and every K segment.  The counter carries no data (every
// workgroup's result goes to keys[] by atomicMin as before), the wait is bounded, so a missing member delays and cannot hang.
//
// K segments: after every SEG slices the fp32 accumulators are added into per-workgroup totals in HBM and cleared, which turns the
// 266 000-step accumulation chain of a 256 x 256 image pair into a two-level sum (error of a distance 1.8e-5 -> ~1e-6).
// ---------------------------------------------------------------------------------------------
constexpr int kClusters = 8, kSuperN = 4, kSuperQ = 8;
constexpr int kSegSlices = 2048;                 // 128 Ki halves of K per segment
constexpr size_t kTotalsPerWg = 8 * 32 * 64 * sizeof(v4f);     // 256 KiB: 8 waves x 32 accumulator tiles x 64 lanes x 4 floats

__device__ __forceinline__ void cluster_meet(unsigned *counter, unsigned target)
{
    // one lane arrives and polls; the counter only orders time (L2 sharing), no memory is handed over
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spin = 0; spin < 40000; ++spin) {
            if ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) break;
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(512, 2)
feat_knn_h1c_kernel(const char *__restrict__ bank, const float *__restrict__ bank_norm, int64_t n_rows, int64_t index_base,
                    const char *__restrict__ query, const float *__restrict__ query_norm, int64_t nq, int64_t K1,
                    unsigned long long *__restrict__ keys, int q_tiles, int n_tiles, char *__restrict__ scratch, int members, float inv_s2, int blocked)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t kstep = blocked ? 32768 : 128;        // bytes between consecutive K slices of a tile's rows (K-blocked rows: gl_conv.h)
    const int cluster = blockIdx.x & (kClusters - 1), member = blockIdx.x >> 3;
    unsigned *counter = reinterpret_cast<unsigned *>(scratch) + cluster * 32;           // 128 B apart
    v4f *totals = reinterpret_cast<v4f *>(scratch + 4096 + (size_t)blockIdx.x * kTotalsPerWg);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wq = wave & 3;
    const int frow = lane & 15, fk = lane >> 4;
    const int sup_n = (n_tiles + kSuperN - 1) / kSuperN, sup_q = (q_tiles + kSuperQ - 1) / kSuperQ;
    const int64_t nk = K1 / 64;
    const int nseg = (int)((nk + kSegSlices - 1) / kSegSlices);
    v4f *my_tot = totals + (size_t)wave * 32 * 64 + lane;
    unsigned episode = 0;

    for (int s = cluster; s < sup_n * sup_q; s += kClusters) {
        // the 8 clusters work on consecutive bank groups of one query group: the query panels of the moment are then shared chip-wide
        const int sq = s / sup_n, sn = s % sup_n;
        const int nt = sn * kSuperN + (member & (kSuperN - 1)), qt = sq * kSuperQ + (member >> 2);
        const bool active = member < kSuperN * kSuperQ && nt < n_tiles && qt < q_tiles;
        const int64_t n0 = (int64_t)nt * GT, q0 = (int64_t)qt * GT;
        gl_pair256::Source sa = {}, sb = {};
        if (active) {
            if (blocked) {
                sa = gl_pair256::make_source_blocked(bank, n0, nk, wave, lane);
                sb = gl_pair256::make_source_blocked(query, q0, nk, wave, lane);
            } else {
                sa = gl_pair256::make_source(bank, n0, n_rows, K1 * 2, wave, lane);
                sb = gl_pair256::make_source(query, q0, nq, K1 * 2, wave, lane);
            }
        }
        v4f acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
        for (int seg = 0; seg < nseg; ++seg) {
            cluster_meet(counter, (unsigned)members * ++episode);
            if (!active) continue;
            const int64_t k0 = (int64_t)seg * kSegSlices;
            const int64_t len = nk - k0 < kSegSlices ? nk - k0 : kSegSlices;
            gl_pair256::mainloop<v8h>(sa, sb, len, smem, acc, wave, lane,
                                      [](const v8h &a, const v8h &b, const v4f &c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }, k0 * kstep, kstep);
            __syncthreads();                             // all fragment reads of the segment are done before its buffers are refilled
            if (nseg > 1) {
                // totals (+)= accumulators; the last segment leaves the sum in the accumulators
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v4f *t = my_tot + (i * 4 + j) * 64;
                        if (seg > 0) acc[i][j] += *t;
                        if (seg + 1 < nseg) { *t = acc[i][j]; acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f}; }
                        __builtin_amdgcn_sched_barrier(0);       // one tile at a time: 32 loads in flight at once would not fit the register file
                    }
            }
        }
        if (!active) continue;
        const int64_t nbase = n0 + wn * 128 + fk * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t q = q0 + wq * 64 + j * 16 + frow;
            const float qn = q < nq ? query_norm[q] : 0.0f;
            unsigned long long best = ~0ull;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t n = nbase + i * 16 + r;
                    const float bn = n < n_rows ? bank_norm[n] : 0.0f;
                    const float d = fmaxf(fmaf(-2.0f * inv_s2, acc[i][j][r], __fadd_rn(qn, bn)), 0.0f);
                    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(index_base + n);
                    if (n < n_rows && key < best) best = key;
                }
            unsigned long long o = __shfl_xor(best, 16, 64);
            best = o < best ? o : best;
            o = __shfl_xor(best, 32, 64);
            best = o < best ? o : best;
            if (fk == 0 && q < nq && best != ~0ull) atomicMin(&keys[q], best);
        }
    }
}


// The same arithmetic for a device that cannot hold a cluster's 4 x 8 super-tile (fewer than 256 compute units: a partitioned MI355X):
// a persistent launch of one workgroup per CU that walks the tiles in strip order without meeting anyone.  Same main loop, same K
// segments, same order of the segment totals, so a distance is the same bits on every device and in both kernels.
__global__ void __launch_bounds__(512, 2)
feat_knn_h1s_kernel(const char *__restrict__ bank, const float *__restrict__ bank_norm, int64_t n_rows, int64_t index_base,
                    const char *__restrict__ query, const float *__restrict__ query_norm, int64_t nq, int64_t K1,
                    unsigned long long *__restrict__ keys, int q_tiles, int n_tiles, char *__restrict__ scratch, float inv_s2, int blocked)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t kstep = blocked ? 32768 : 128;
    v4f *totals = reinterpret_cast<v4f *>(scratch + 4096 + (size_t)blockIdx.x * kTotalsPerWg);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wq = wave & 3;
    const int frow = lane & 15, fk = lane >> 4;
    const int64_t nk = K1 / 64;
    const int nseg = (int)((nk + kSegSlices - 1) / kSegSlices);
    v4f *my_tot = totals + (size_t)wave * 32 * 64 + lane;
    const unsigned tiles = (unsigned)q_tiles * (unsigned)n_tiles;
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        int qt, nt;
        {
            constexpr int STRIP = 4;
            const unsigned per_strip = (unsigned)STRIP * (unsigned)q_tiles;
            const int strip = (int)(t / per_strip);
            const unsigned r = t % per_strip;
            const int width = n_tiles - strip * STRIP < STRIP ? n_tiles - strip * STRIP : STRIP;
            nt = strip * STRIP + (int)(r % (unsigned)width);
            qt = (int)(r / (unsigned)width);
        }
        const int64_t n0 = (int64_t)nt * GT, q0 = (int64_t)qt * GT;
        const gl_pair256::Source sa = blocked ? gl_pair256::make_source_blocked(bank, n0, nk, wave, lane) : gl_pair256::make_source(bank, n0, n_rows, K1 * 2, wave, lane);
        const gl_pair256::Source sb = blocked ? gl_pair256::make_source_blocked(query, q0, nk, wave, lane) : gl_pair256::make_source(query, q0, nq, K1 * 2, wave, lane);
        v4f acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
        for (int seg = 0; seg < nseg; ++seg) {
            const int64_t k0 = (int64_t)seg * kSegSlices;
            const int64_t len = nk - k0 < kSegSlices ? nk - k0 : kSegSlices;
            gl_pair256::mainloop<v8h>(sa, sb, len, smem, acc, wave, lane,
                                      [](const v8h &a, const v8h &b, const v4f &c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }, k0 * kstep, kstep);
            __syncthreads();
            if (nseg > 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v4f *tt = my_tot + (i * 4 + j) * 64;
                        if (seg > 0) acc[i][j] += *tt;
                        if (seg + 1 < nseg) { *tt = acc[i][j]; acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f}; }
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
        }
        const int64_t nbase = n0 + wn * 128 + fk * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t q = q0 + wq * 64 + j * 16 + frow;
            const float qn = q < nq ? query_norm[q] : 0.0f;
            unsigned long long best = ~0ull;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t n = nbase + i * 16 + r;
                    const float bn = n < n_rows ? bank_norm[n] : 0.0f;
                    const float d = fmaxf(fmaf(-2.0f * inv_s2, acc[i][j][r], __fadd_rn(qn, bn)), 0.0f);
                    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(index_base + n);
                    if (n < n_rows && key < best) best = key;
                }
            unsigned long long o = __shfl_xor(best, 16, 64);
            best = o < best ? o : best;
            o = __shfl_xor(best, 32, 64);
            best = o < best ? o : best;
            if (fk == 0 && q < nq && best != ~0ull) atomicMin(&keys[q], best);
        }
        __syncthreads();           // the next tile's prologue refills the slice buffers
    }
}

}  // namespace

struct gl_lpips {
    gl_ctx *ctx;
    float *w[kNumConv];       // packed conv weights
    float *bias[kNumConv];
    float *ones;              // 512 ones (epilogue scale)
    float *lin[5];            // raw lin weights (host-checked >= 0)
    std::vector<float> lin_host[5];
    bool have_w[kNumConv], have_lin[5];
    // split-fp16 path: weights in the split layout scaled by 2^wexp, epilogue constants 2^-wexp and bias * kVggAct
    int precision;
    float *wsplit[kNumConv], *scale_h3[kNumConv], *bias_h3[kNumConv];
    int wexp[kNumConv];
    // the output of convolution i is stored as (value * act[i]) in the split layout; act[i] = 2^aexp[i], chosen per layer by a calibration
    // pass on fixed synthetic images the first time the split path runs (lp_calibrate), kVggAct before that.  Powers of two: a value's halves
    // only move in exponent, so wherever nothing clamps or falls into the fp16 subnormals the results do not depend on the choice.
    float act[kNumConv];
    std::vector<float> bias_host[kNumConv];
    bool calibrated, calibrate;
    unsigned *calib_max;      // device, kNumConv words: max |activation| per layer as float bits (set only during the calibration pass)
    // workspace for `chunk` images of H x W
    int64_t chunk, ws_imgs;
    int ws_H, ws_W;
    float *ws_a, *ws_b, *ws_coef;
};

namespace {

int lp_upload(gl_ctx *ctx, float **dev, const std::vector<float> &host)
{
    if (!*dev) GL_HIP(hipMalloc((void **)dev, host.size() * sizeof(float)));
    GL_HIP(hipMemcpyAsync(*dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    GL_HIP(hipStreamSynchronize(ctx->stream));
    return GL_OK;
}

// Search rows whose byte length is a multiple of 32 KiB (128 x 128 and 256 x 256 images: 2^15 x 131 and 2^17 x 131 bytes) get 64 zero halves
// appended: with such a stride the same K slice of every row of a tile falls on the same few memory channels (measured on the pairwise kernel
// at 256 x 256: 971 -> 1020 TFLOP/s with the pad; rows of 64 x 64 images, 2^13 x 131 bytes, do not need it).
int64_t lp_search_pad(int64_t K_lp, int64_t Dp, int img_segs = 3) { return ((K_lp + img_segs * Dp) * 2) % 32768 == 0 ? 64 : 0; }

int lp_workspace(gl_lpips *l, int64_t n, int H, int W)
{
    int64_t want = l->chunk > 0 ? l->chunk : 2048;
    // the convolution kernels address one activation tensor through a 32-bit buffer descriptor: keep the largest (H x W x 64 values of 4 bytes) under 3 GiB
    const int64_t cap = (int64_t)(0xB0000000ull / ((uint64_t)H * W * 64 * 4));
    if (want > cap) want = cap;
    // the deepest block (conv5_x: 2 column tiles x H W / 65 536 position tiles per image) should fill the 256 CUs a whole number of times:
    // 2048 images of 64 x 64, 512 of 128 x 128, 128 of 256 x 256 (176 would leave conv5_x at 1.4 rounds)
    const int64_t whole = (int64_t)((8ull << 20) / ((uint64_t)H * W));
    if (l->chunk <= 0 && whole >= 1 && want >= whole) want -= want % whole;
    GL_REQUIRE(want >= 1, "gl_lpips_features: %d x %d images are too large for one pass", H, W);
    if (n < want) want = n;
    if (want <= l->ws_imgs && H == l->ws_H && W == l->ws_W) return GL_OK;
    GL_HIP(hipStreamSynchronize(l->ctx->stream));
    (void)hipFree(l->ws_a); (void)hipFree(l->ws_b);
    l->ws_a = l->ws_b = nullptr;
    l->ws_imgs = 0;
    const size_t px = (size_t)want * H * W;
    GL_HIP(gl_device_alloc(l->ctx, (void **)&l->ws_a, px * 64 * 4));      // largest activation: H x W x 64
    GL_HIP(gl_device_alloc(l->ctx, (void **)&l->ws_b, px * 64 * 4));
    l->ws_imgs = want; l->ws_H = H; l->ws_W = W;
    return GL_OK;
}

int stream_blocks(int64_t items)
{
    int64_t b = gl_ceil_div(items, 256);
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// max |x| over an fp32 activation tensor (non-negative after ReLU; the bits of a non-negative float order like unsigned integers)
__global__ void __launch_bounds__(256) absmax_f32_kernel(const float *__restrict__ x, int64_t count, unsigned *__restrict__ out)
{
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// (re)build the split path's epilogue constants from the current act[]: acc = 2^wexp * act_in * conv  ->  stored out = acc * scale + shift with
// scale = act_out / (2^wexp * act_in) and shift = bias * act_out  (all powers of two times the fp32 bias: exact)
int lp_apply_scales(gl_lpips *l)
{
    for (int ci = 0; ci < kNumConv; ++ci) {
        if (!l->have_w[ci]) continue;
        const int co_n = kCout[ci];
        const float a_in = ci == 0 ? 1.0f : l->act[ci - 1], a_out = l->act[ci];
        std::vector<float> sc(co_n, std::ldexp(1.0f, -l->wexp[ci]) * (a_out / a_in)), bh(co_n);
        for (int c = 0; c < co_n; ++c) bh[c] = l->bias_host[ci][c] * a_out;
        int rc = lp_upload(l->ctx, &l->scale_h3[ci], sc);
        if (rc == GL_OK) rc = lp_upload(l->ctx, &l->bias_h3[ci], bh);
        if (rc != GL_OK) return rc;
    }
    return GL_OK;
}

template <typename T>
int lpips_features_impl(gl_lpips *l, const T *img_dev, int64_t n, int H, int W, void *V_dev, float *norms_dev, int fmt);

// Per-layer activation scales for the split path.  The fp32 pipeline runs once on 16 fixed synthetic 64 x 64 images (smooth fields, uniform
// noise, flat black / white, stripes and a checkerboard, smooth + noise: generated here from a fixed LCG, so every context of every process
// derives the same scales from the same weights -- a shard's features do not depend on which images it happened to see first), the largest
// activation of every layer is taken, and act[i] = the power of two that puts it in [1024, 2048): 32 - 64 x of headroom below the fp16
// maximum for images that excite a layer more than the calibration set does (a clamp is still counted, gl_ctx_h3_saturations, and the
// Python callers then redo the pass with fp32 products), and full 22-bit operands for every value above ~6e-5 of that maximum.
int lp_calibrate(gl_lpips *l)
{
    gl_ctx *ctx = l->ctx;
    constexpr int N = 16, R = 64, D = 3 * R * R;
    std::vector<uint8_t> img((size_t)N * D);
    uint32_t lcg = 0x9E3779B9u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (lcg >> 8) & 0xFFFFu; };       // 16 bits
    for (int i = 0; i < N; ++i) {
        float grid[3][9][9];
        for (auto &c : grid) for (auto &r : c) for (float &v : r) v = (float)(rnd() & 255u);
        for (int c = 0; c < 3; ++c)
            for (int y = 0; y < R; ++y)
                for (int x = 0; x < R; ++x) {
                    const int gy = y / 8, gx = x / 8;
                    const float fy = (y % 8) / 8.0f, fx = (x % 8) / 8.0f;
                    const float smooth = (grid[c][gy][gx] * (1 - fx) + grid[c][gy][gx + 1] * fx) * (1 - fy) + (grid[c][gy + 1][gx] * (1 - fx) + grid[c][gy + 1][gx + 1] * fx) * fy;
                    const float noise = (float)(rnd() & 255u);
                    float v;
                    switch (i % 8) {
                    case 0: case 1: v = smooth; break;
                    case 2: case 3: v = noise; break;
                    case 4: v = i < 8 ? 0.0f : 255.0f; break;
                    case 5: v = ((i < 8 ? x : y) / (1 + c)) % 2 ? 255.0f : 0.0f; break;
                    case 6: v = ((x / 4 + y / 4) % 2) ? 255.0f : 0.0f; break;
                    default: v = 0.6f * smooth + 0.4f * noise; break;
                    }
                    img[(size_t)i * D + ((size_t)c * R + y) * R + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
                }
    }
    int64_t K_lp = 0;
    { int h = R; for (int t = 0; t < 5; ++t) { K_lp += (int64_t)kTapC[t] * h * h; h /= 2; } }
    const int64_t K = K_lp + D;
    uint8_t *img_dev = nullptr;
    float *V = nullptr, *norms = nullptr;
    int rc = GL_OK;
    const int saved_precision = l->precision;
    const int64_t saved_chunk = l->chunk;
    unsigned host_max[kNumConv];
    GL_HIP(hipMalloc((void **)&img_dev, img.size()));
    if (hipMalloc((void **)&V, (size_t)N * K * 4) != hipSuccess || hipMalloc((void **)&norms, N * 4) != hipSuccess ||
        hipMalloc((void **)&l->calib_max, kNumConv * sizeof(unsigned)) != hipSuccess) {
        gl_set_error("lp_calibrate: out of device memory");
        rc = GL_ERR_HIP;
    }
    if (rc == GL_OK && (hipMemcpyAsync(img_dev, img.data(), img.size(), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                        hipMemsetAsync(l->calib_max, 0, kNumConv * sizeof(unsigned), ctx->stream) != hipSuccess)) {
        gl_set_error("lp_calibrate: upload failed");
        rc = GL_ERR_HIP;
    }
    if (rc == GL_OK) {
        l->precision = 0;
        l->chunk = N;
        rc = lpips_features_impl<uint8_t>(l, img_dev, N, R, R, V, norms, 0);
        l->precision = saved_precision;
        l->chunk = saved_chunk;
    }
    if (rc == GL_OK && (hipMemcpyAsync(host_max, l->calib_max, sizeof(host_max), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                        hipStreamSynchronize(ctx->stream) != hipSuccess)) {
        gl_set_error("lp_calibrate: read-back failed");
        rc = GL_ERR_HIP;
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(img_dev); (void)hipFree(V); (void)hipFree(norms); (void)hipFree(l->calib_max);
    l->calib_max = nullptr;
    // the calibration pass sized the workspace for 16 small images: let the next call size it for its own
    if (rc != GL_OK) return rc;
    for (int ci = 0; ci < kNumConv; ++ci) {
        float mx;
        memcpy(&mx, &host_max[ci], 4);
        int e = 2;                                            // a dead layer keeps the default
        if (mx > 0.0f && std::isfinite(mx)) e = (int)std::floor(std::log2(2047.0f / mx));
        e = e > 40 ? 40 : (e < -40 ? -40 : e);
        l->act[ci] = std::ldexp(1.0f, e);
    }
    return lp_apply_scales(l);
}

template <typename T>
int lpips_features_impl(gl_lpips *l, const T *img_dev, int64_t n, int H, int W, void *V_dev, float *norms_dev, int fmt)
{
    // fmt 0: split rows of K values (4 K bytes);  1 / 2: search rows for queries / bank rows (K_lp + 3 Dp halves)
    GL_REQUIRE(l && n >= 0, "gl_lpips_features: bad argument");
    GL_REQUIRE(H >= 16 && W >= 16 && H % 16 == 0 && W % 16 == 0, "gl_lpips_features: H, W must be multiples of 16 (four 2x2 pools), got %dx%d", H, W);
    for (int i = 0; i < kNumConv; ++i)
        if (!l->have_w[i]) { gl_set_error("gl_lpips_features: VGG16 conv %d not loaded", i); return GL_ERR_STATE; }
    for (int i = 0; i < 5; ++i)
        if (!l->have_lin[i]) { gl_set_error("gl_lpips_features: lin%d not loaded", i); return GL_ERR_STATE; }
    if (n == 0) return GL_OK;
    GL_REQUIRE(img_dev && V_dev && norms_dev, "gl_lpips_features: NULL device pointer");
    gl_ctx *ctx = l->ctx;
    int rc = GL_OK;
    if (l->precision == 1 && !l->calibrated) {
        l->calibrated = true;                                 // (the calibration pass comes back through this function with fp32 products)
        rc = lp_calibrate(l);
        if (rc != GL_OK) { l->calibrated = false; return rc; }
    }
    rc = lp_workspace(l, n, H, W);
    if (rc != GL_OK) return rc;
    const int64_t D = 3ll * H * W;
    int64_t K_lp = 0;
    {
        int h = H, w = W;
        for (int t = 0; t < 5; ++t) { K_lp += (int64_t)kTapC[t] * h * w; h /= 2; w /= 2; }
    }
    const int64_t K = K_lp + D;
    const int64_t Dp = gl_ceil_div(D, 64) * 64;
    // fmt 3: lattice search rows (8-bit images): K_lp + Dp halves, image part exact, row scale lp_lattice_scale(D) instead of kVScale
    const bool lattice = fmt == 3;
    GL_REQUIRE(!lattice || sizeof(T) == 1, "gl_lpips_features: lattice search rows need 8-bit images");
    const int img_segs = lattice ? 1 : 3;
    const int64_t pad = fmt ? lp_search_pad(K_lp, Dp, img_segs) : 0;
    const int64_t ldv = fmt ? (K_lp + img_segs * Dp + pad) * 2 : K * 4;        // bytes per row of V
    const int lo_seg = lattice ? 0 : (fmt == 1 ? 2 : 1);
    const double row_scale = lattice ? (double)(float)lp_lattice_scale(D) : (double)kVScale;      // the float the search kernel is given
    // per-tap coefficients sqrt(0.2 * w_c / (h*w))
    {
        std::vector<float> coef;
        int h = H, w = W;
        for (int t = 0; t < 5; ++t) {
            // the tap kernels multiply by kVScale; rows of another scale get the ratio here
            for (int c = 0; c < kTapC[t]; ++c) coef.push_back((float)(std::sqrt(0.2 * (double)l->lin_host[t][c] / ((double)h * w)) * (row_scale / (double)kVScale)));
            h /= 2; w /= 2;
        }
        rc = lp_upload(ctx, &l->ws_coef, coef);
        if (rc != GL_OK) return rc;
    }

    for (int64_t i0 = 0; i0 < n; i0 += l->ws_imgs) {
        const int64_t m = (n - i0 < l->ws_imgs) ? n - i0 : l->ws_imgs;
        char *Vc = reinterpret_cast<char *>(V_dev) + i0 * ldv;       // split rows (fmt 0): this pass's first row
        // fp16 search rows: the writers address the whole buffer by (row, k) -- row-major, or K-blocked for long rows (gl_conv.h gl_vrow_elem)
        char *Vw = fmt ? reinterpret_cast<char *>(V_dev) : Vc;
        const int64_t rw = fmt ? i0 : 0;
        const int64_t ldw = (fmt && gl_vrow_blocked(ldv / 2)) ? -(ldv / 128) : ldv;
        const bool h3 = l->precision == 1;
        const float *cur = nullptr;
        float *bufs[2] = {l->ws_a, l->ws_b};
        int which = 0, h = H, w = W;
        int64_t off = 0, coef_off = 0;
        for (int ci = 0; ci < kNumConv; ++ci) {
            if (ci == 0) {
                // conv1_1 reads the image itself
                GL_REQUIRE(m * 3 * h * w < (1ll << 31), "gl_lpips_features: a pass of %lld images of %d x %d is too large for conv1_1's 32-bit offsets", (long long)m, h, w);
                const unsigned nb = (unsigned)std::min<int64_t>(gl_ceil_div(m * h * w, 128), (int64_t)ctx->num_cu * 8);
                if (h3)
                    hipLaunchKernelGGL((vgg_conv1_kernel<T, true>), dim3(nb), dim3(256), 0, ctx->stream, img_dev + i0 * D, (int)m, h, w, l->w[0], l->bias[0],
                                       reinterpret_cast<char *>(bufs[which]), ctx->h3_sat, l->act[0]);
                else
                    hipLaunchKernelGGL((vgg_conv1_kernel<T, false>), dim3(nb), dim3(256), 0, ctx->stream, img_dev + i0 * D, (int)m, h, w, l->w[0], l->bias[0],
                                       reinterpret_cast<char *>(bufs[which]), ctx->h3_sat, 1.0f);
                GL_LAUNCH_CHECK();
                cur = bufs[which];
                if (!h3 && l->calib_max) {
                    hipLaunchKernelGGL(absmax_f32_kernel, dim3((unsigned)stream_blocks(m * h * w * 64)), dim3(256), 0, ctx->stream, cur, m * h * w * 64, l->calib_max);
                    GL_LAUNCH_CHECK();
                }
                which ^= 1;
                continue;
            }
            GlGatherConv p = {};
            p.in = cur; p.positions = m * h * w; p.H = h; p.W = w;
            p.wpack = l->w[ci]; p.cols = kCout[ci]; p.cols_pad = kCout[ci];
            {
                p.Cin = kCin[ci]; p.ntaps = 9;
                uint32_t dy = 0, dx = 0;
                for (int t = 0; t < 9; ++t) { dy |= (uint32_t)(t / 3) << (2 * t); dx |= (uint32_t)(t % 3) << (2 * t); }
                p.tap_dy[0] = dy; p.tap_dx[0] = dx;
            }
            p.out = bufs[which]; p.Ho = h; p.Wo = w; p.omul = 1; p.oy[0] = 0; p.ox[0] = 0;
            p.scale = l->ones; p.shift = l->bias[ci]; p.cmod = kCout[ci]; p.act = 1; p.zero = ctx->zero_page;
            if (h3) {
                p.wpack = l->wsplit[ci]; p.cols_pad = (int)gl_ceil_div(kCout[ci], 128) * 128;
                p.scale = l->scale_h3[ci]; p.shift = l->bias_h3[ci]; p.out_mode = 2;
                if (kAfter[ci] == 2 && gl_conv_h3_tap_fusable(p, 1)) {
                    // tap + 2x2 max-pool in the convolution's epilogue: the full-resolution activation is never stored
                    const int C = kCout[ci];
                    p.tap_V = Vw; p.tap_coef = l->ws_coef + coef_off; p.tap_ldv = ldw; p.tap_row0 = rw; p.tap_off = off; p.tap_fmt = fmt ? 1 : 0;
                    p.tap_pool = reinterpret_cast<char *>(bufs[which]); p.tap_scale = kVScale; p.tap_eps = kTapEps1 * l->act[ci];
                    rc = gl_launch_gather_conv_h3(ctx, p, 1);
                    if (rc != GL_OK) return rc;
                    off += (int64_t)C * h * w;
                    coef_off += C;
                    cur = bufs[which];
                    which ^= 1;
                    h /= 2; w /= 2;
                    continue;
                }
                rc = gl_launch_gather_conv_h3(ctx, p, 1);
            } else {
                rc = gl_launch_gather_conv(ctx, p, 1);
            }
            if (rc != GL_OK) return rc;
            cur = bufs[which];
            which ^= 1;
            if (!h3 && l->calib_max) {
                hipLaunchKernelGGL(absmax_f32_kernel, dim3((unsigned)stream_blocks(m * h * w * kCout[ci])), dim3(256), 0, ctx->stream, cur, m * h * w * kCout[ci],
                                   l->calib_max + ci);
                GL_LAUNCH_CHECK();
            }
            if (kAfter[ci] == 2 && h3) {
                // tap + 2x2 max-pool in one pass over the activation
                const int C = kCout[ci];
                if (fmt)
                    launch_tap_pool_split<true>(ctx->stream, reinterpret_cast<const char *>(cur), m, h, w, C, l->ws_coef + coef_off, Vw, ldw, off,
                                                reinterpret_cast<char *>(bufs[which]), kTapEps1 * l->act[ci], rw);
                else
                    launch_tap_pool_split<false>(ctx->stream, reinterpret_cast<const char *>(cur), m, h, w, C, l->ws_coef + coef_off, Vw, ldw, off,
                                                 reinterpret_cast<char *>(bufs[which]), kTapEps1 * l->act[ci], rw);
                GL_LAUNCH_CHECK();
                off += (int64_t)C * h * w;
                coef_off += C;
                cur = bufs[which];
                which ^= 1;
                h /= 2; w /= 2;
                continue;
            }
            if (kAfter[ci] >= 1) {
                const int C = kCout[ci];
                const dim3 tg((unsigned)stream_blocks(m * h * w * 64));
                if (h3 && fmt)
                    launch_tap_split<true>(ctx->stream, reinterpret_cast<const char *>(cur), m, h * w, C, l->ws_coef + coef_off, Vw, ldw, off, kTapEps1 * l->act[ci], rw);
                else if (h3)
                    launch_tap_split<false>(ctx->stream, reinterpret_cast<const char *>(cur), m, h * w, C, l->ws_coef + coef_off, Vw, ldw, off, kTapEps1 * l->act[ci], rw);
                else if (fmt)
                    hipLaunchKernelGGL(lpips_tap_kernel<true>, tg, dim3(256), 0, ctx->stream, cur, m, h * w, C, l->ws_coef + coef_off, Vw, ldw, off, rw);
                else
                    hipLaunchKernelGGL(lpips_tap_kernel<false>, tg, dim3(256), 0, ctx->stream, cur, m, h * w, C, l->ws_coef + coef_off, Vw, ldw, off, rw);
                GL_LAUNCH_CHECK();
                off += (int64_t)C * h * w;
                coef_off += C;
            }
            if (kAfter[ci] == 2) {
                const int C = kCout[ci];
                if (h3)
                    hipLaunchKernelGGL(maxpool2_split_kernel, dim3((unsigned)stream_blocks(m * (h / 2) * (w / 2) * (C / 8))), dim3(256), 0, ctx->stream,
                                       reinterpret_cast<const char *>(cur), m, h, w, C, reinterpret_cast<char *>(bufs[which]));
                else
                    hipLaunchKernelGGL(maxpool2_nhwc_kernel, dim3((unsigned)stream_blocks(m * (h / 2) * (w / 2) * (C / 4))), dim3(256), 0, ctx->stream, cur, m, h, w, C,
                                       bufs[which]);
                GL_LAUNCH_CHECK();
                cur = bufs[which];
                which ^= 1;
                h /= 2; w /= 2;
            }
        }
        if (fmt) {
            if (lattice) {
                if constexpr (sizeof(T) == 1)
                    hipLaunchKernelGGL(image_part_lattice_kernel, dim3((unsigned)stream_blocks(m * (Dp + pad))), dim3(256), 0, ctx->stream,
                                       reinterpret_cast<const uint8_t *>(img_dev) + i0 * D, m, D, Dp + pad, (float)std::ldexp(1.0, lp_lattice_exp(D)), Vw, ldw, K_lp, rw);
            } else {
                hipLaunchKernelGGL(image_part_h1_kernel<T>, dim3((unsigned)stream_blocks(m * Dp)), dim3(256), 0, ctx->stream, img_dev + i0 * D, m, D, Dp,
                                   (float)(1.0 / std::sqrt((double)D)), Vw, ldw, K_lp, lo_seg, (int)pad, rw);
            }
            GL_LAUNCH_CHECK();
            {
                // partial sums go to the first activation buffer, which is free by now (m * nseg doubles)
                const int nseg = (int)gl_ceil_div(K_lp + Dp, kNormSeg);
                double *part = reinterpret_cast<double *>(l->ws_a);
                hipLaunchKernelGGL(row_sqnorm_h1_part_kernel, dim3((unsigned)nseg, (unsigned)m), dim3(256), 0, ctx->stream, Vw, ldw, K_lp, Dp, lo_seg, nseg, part, rw);
                GL_LAUNCH_CHECK();
                hipLaunchKernelGGL(row_sqnorm_h1_final_kernel, dim3((unsigned)gl_ceil_div(m, 256)), dim3(256), 0, ctx->stream, part, m, nseg,
                                   1.0 / (row_scale * row_scale), norms_dev + i0);
            }
        } else {
            hipLaunchKernelGGL(image_part_kernel<T>, dim3((unsigned)stream_blocks(m * D)), dim3(256), 0, ctx->stream, img_dev + i0 * D, m, D,
                               (float)(1.0 / std::sqrt((double)D)), Vc, ldv, K_lp);
            GL_LAUNCH_CHECK();
            hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)(m < 2048 ? m : 2048)), dim3(256), 0, ctx->stream, Vc, m, K, norms_dev + i0);
        }
        GL_LAUNCH_CHECK();
    }
    return GL_OK;
}

}  // namespace

extern "C" {

int gl_lpips_create(gl_ctx *ctx, gl_lpips **out)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && out, "gl_lpips_create: NULL argument");
    gl_lpips *l = new gl_lpips();
    l->ctx = ctx;
    for (int i = 0; i < kNumConv; ++i) { l->w[i] = l->bias[i] = nullptr; l->have_w[i] = false; l->wsplit[i] = l->scale_h3[i] = l->bias_h3[i] = nullptr; l->wexp[i] = 0; l->act[i] = kVggAct; }
    l->calibrated = false; l->calibrate = true; l->calib_max = nullptr;
    l->precision = 1;
    for (int i = 0; i < 5; ++i) { l->lin[i] = nullptr; l->have_lin[i] = false; }
    l->ones = nullptr;
    l->chunk = 0; l->ws_imgs = 0; l->ws_H = l->ws_W = 0;
    l->ws_a = l->ws_b = l->ws_coef = nullptr;
    std::vector<float> one(512, 1.0f);
    int rc = lp_upload(ctx, &l->ones, one);
    if (rc != GL_OK) { delete l; return rc; }
    *out = l;
    return GL_OK;
}

int gl_lpips_destroy(gl_lpips *l)
{
    gl_make_current(l ? l->ctx : nullptr);
    if (!l) return GL_OK;
    (void)hipStreamSynchronize(l->ctx->stream);
    for (int i = 0; i < kNumConv; ++i) { (void)hipFree(l->w[i]); (void)hipFree(l->bias[i]); (void)hipFree(l->wsplit[i]); (void)hipFree(l->scale_h3[i]); (void)hipFree(l->bias_h3[i]); }
    for (int i = 0; i < 5; ++i) (void)hipFree(l->lin[i]);
    (void)hipFree(l->ones); (void)hipFree(l->ws_a); (void)hipFree(l->ws_b); (void)hipFree(l->ws_coef);
    delete l;
    return GL_OK;
}

int gl_lpips_set_chunk(gl_lpips *l, int64_t images_per_pass)
{
    gl_make_current(l ? l->ctx : nullptr);
    GL_REQUIRE(l && images_per_pass >= 0, "gl_lpips_set_chunk: bad argument");
    l->chunk = images_per_pass;
    return GL_OK;
}

int gl_lpips_set_precision(gl_lpips *l, int mode)
{
    gl_make_current(l ? l->ctx : nullptr);
    GL_REQUIRE(l && (mode == 0 || mode == 1), "gl_lpips_set_precision: mode must be 0 or 1");
    l->precision = mode;
    return GL_OK;
}

int gl_lpips_set_calibration(gl_lpips *l, int enabled)
{
    gl_make_current(l ? l->ctx : nullptr);
    GL_REQUIRE(l && (enabled == 0 || enabled == 1), "gl_lpips_set_calibration: enabled must be 0 or 1");
    l->calibrate = enabled != 0;
    for (int i = 0; i < kNumConv; ++i) l->act[i] = kVggAct;
    l->calibrated = !l->calibrate;                    // disabled: the defaults stand; enabled: the next split-path call calibrates
    return lp_apply_scales(l);
}

int gl_lpips_set_conv(gl_lpips *l, int conv_index, const float *w, const float *bias)
{
    gl_make_current(l ? l->ctx : nullptr);
    GL_REQUIRE(l && w && bias && conv_index >= 0 && conv_index < kNumConv, "gl_lpips_set_conv: bad argument");
    const int co_n = kCout[conv_index], ci_n = kCin[conv_index];
    auto Wt = [&](int co, int ci, int ky, int kx) { return w[(((int64_t)co * ci_n + ci) * 3 + ky) * 3 + kx]; };
    std::vector<float> pk;
    if (conv_index == 0) {
        pk.assign((size_t)co_n * 32, 0.0f);     // K = 32: (ky*3+kx)*3 + c, as vgg_conv1_kernel walks it
        for (int co = 0; co < co_n; ++co)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx)
                    for (int c = 0; c < 3; ++c) pk[(size_t)co * 32 + (ky * 3 + kx) * 3 + c] = Wt(co, c, ky, kx);
    } else {
        const int K = 9 * ci_n;
        pk.assign((size_t)co_n * K, 0.0f);
        for (int co = 0; co < co_n; ++co)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx)
                    for (int ci = 0; ci < ci_n; ++ci) pk[(size_t)co * K + gl_conv_k_index(ky * 3 + kx, ci, 9)] = Wt(co, ci, ky, kx);
    }
    int rc = lp_upload(l->ctx, &l->w[conv_index], pk);
    if (rc != GL_OK) return rc;
    std::vector<float> b(bias, bias + co_n);
    rc = lp_upload(l->ctx, &l->bias[conv_index], b);
    if (rc != GL_OK) return rc;
    {
        // split-fp16 copy: rows padded to a multiple of 128, values * 2^wexp with max |w| * 2^wexp in [2^12, 2^13)
        const size_t Kl = conv_index == 0 ? 32 : (size_t)9 * ci_n;
        const size_t rows128 = (size_t)gl_ceil_div(co_n, 128) * 128;
        float mx = 0.0f;
        for (float v : pk) mx = std::fmax(mx, std::fabs(v));
        int e = mx > 0.0f ? (int)std::floor(std::log2(8191.0f / mx)) : 0;
        e = e > 30 ? 30 : (e < -30 ? -30 : e);
        l->wexp[conv_index] = e;
        std::vector<float> padded(rows128 * Kl, 0.0f), split(rows128 * Kl);
        std::copy(pk.begin(), pk.end(), padded.begin());
        gl_split_weights_host(padded.data(), rows128, Kl, std::ldexp(1.0f, e), split.data());
        rc = lp_upload(l->ctx, &l->wsplit[conv_index], split);
        if (rc != GL_OK) return rc;
        l->bias_host[conv_index].assign(bias, bias + co_n);
    }
    l->have_w[conv_index] = true;
    // new weights: back to the default activation scales until the next split-path call calibrates again
    for (int i = 0; i < kNumConv; ++i) l->act[i] = kVggAct;
    l->calibrated = !l->calibrate;
    return lp_apply_scales(l);
}

int gl_lpips_set_lin(gl_lpips *l, int layer, const float *w)
{
    gl_make_current(l ? l->ctx : nullptr);
    GL_REQUIRE(l && w && layer >= 0 && layer < 5, "gl_lpips_set_lin: bad argument");
    for (int c = 0; c < kTapC[layer]; ++c)
        GL_REQUIRE(w[c] >= 0.0f, "gl_lpips_set_lin: lin%d weight %d is negative (%g); the |V_q - V_n|^2 form needs w >= 0", layer, c, (double)w[c]);
    l->lin_host[layer].assign(w, w + kTapC[layer]);
    l->have_lin[layer] = true;
    return GL_OK;
}

int64_t gl_lpips_feature_dim(int H, int W)
{
    if (H < 16 || W < 16 || H % 16 || W % 16) return -1;
    int64_t k = 3ll * H * W;
    int h = H, w = W;
    for (int t = 0; t < 5; ++t) { k += (int64_t)kTapC[t] * h * w; h /= 2; w /= 2; }
    return k;
}

int gl_lpips_features_u8(gl_lpips *l, const uint8_t *img_u8_dev, int64_t n, int H, int W, float *V_dev, float *norms_dev)
{
    gl_make_current(l ? l->ctx : nullptr);
    return lpips_features_impl<uint8_t>(l, img_u8_dev, n, H, W, V_dev, norms_dev, 0);
}

int gl_lpips_features_f32(gl_lpips *l, const float *img_f32_dev, int64_t n, int H, int W, float *V_dev, float *norms_dev)
{
    gl_make_current(l ? l->ctx : nullptr);
    return lpips_features_impl<float>(l, img_f32_dev, n, H, W, V_dev, norms_dev, 0);
}

int64_t gl_lpips_search_dim(int H, int W)
{
    const int64_t k = gl_lpips_feature_dim(H, W);
    if (k < 0) return -1;
    const int64_t D = 3ll * H * W;
    return k - D + 3 * (gl_ceil_div(D, 64) * 64) + lp_search_pad(k - D, gl_ceil_div(D, 64) * 64);
}

int64_t gl_lpips_lattice_dim(int H, int W)
{
    const int64_t k = gl_lpips_feature_dim(H, W);
    if (k < 0) return -1;
    const int64_t D = 3ll * H * W, Dp = gl_ceil_div(D, 64) * 64;
    return k - D + Dp + lp_search_pad(k - D, Dp, 1);
}

int64_t gl_lpips_search_rows_capacity(int64_t n, int64_t K1) { return n < 0 || K1 <= 0 ? -1 : gl_vrow_capacity(n, K1); }

float gl_lpips_lattice_scale(int H, int W) { return H > 0 && W > 0 ? (float)lp_lattice_scale(3ll * H * W) : 0.0f; }

int gl_lpips_lattice_features_u8(gl_lpips *l, const uint8_t *img_u8_dev, int64_t n, int H, int W, void *V16_dev, float *norms_dev)
{
    gl_make_current(l ? l->ctx : nullptr);
    return lpips_features_impl<uint8_t>(l, img_u8_dev, n, H, W, V16_dev, norms_dev, 3);
}

int gl_lpips_search_features_u8(gl_lpips *l, const uint8_t *img_u8_dev, int64_t n, int H, int W, int role, void *V16_dev, float *norms_dev)
{
    gl_make_current(l ? l->ctx : nullptr);
    GL_REQUIRE(role == 0 || role == 1, "gl_lpips_search_features: role must be 0 (query rows) or 1 (bank rows)");
    return lpips_features_impl<uint8_t>(l, img_u8_dev, n, H, W, V16_dev, norms_dev, 1 + role);
}

int gl_lpips_search_features_f32(gl_lpips *l, const float *img_f32_dev, int64_t n, int H, int W, int role, void *V16_dev, float *norms_dev)
{
    gl_make_current(l ? l->ctx : nullptr);
    GL_REQUIRE(role == 0 || role == 1, "gl_lpips_search_features: role must be 0 (query rows) or 1 (bank rows)");
    return lpips_features_impl<float>(l, img_f32_dev, n, H, W, V16_dev, norms_dev, 1 + role);
}

int gl_feat_knn_h1_scaled(gl_ctx *ctx, const void *bank_V16_dev, const float *bank_norm_dev, int64_t n_rows, int64_t index_base, const void *query_V16_dev,
                          const float *query_norm_dev, int64_t nq, int64_t K1, uint64_t *keys_dev, float row_scale)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && n_rows >= 0 && nq >= 0 && K1 > 0 && K1 % 64 == 0, "gl_feat_knn_h1: bad sizes (K1 must be a multiple of 64)");
    GL_REQUIRE(row_scale > 0.0f, "gl_feat_knn_h1: the row scale must be positive");
    const float inv_s2 = 1.0f / (row_scale * row_scale);
    GL_REQUIRE(index_base >= 0 && index_base + n_rows <= 0xFFFFFFFFll, "gl_feat_knn_h1: global index does not fit 32 bits");
    if (n_rows == 0 || nq == 0) return GL_OK;
    GL_REQUIRE(bank_V16_dev && bank_norm_dev && query_V16_dev && query_norm_dev && keys_dev, "gl_feat_knn_h1: NULL device pointer");
    GL_REQUIRE(((reinterpret_cast<uintptr_t>(bank_V16_dev) | reinterpret_cast<uintptr_t>(query_V16_dev)) & 15) == 0, "gl_feat_knn_h1: rows must be 16-byte aligned");
    const int64_t q_tiles = gl_ceil_div(nq, GT), n_tiles = gl_ceil_div(n_rows, GT);
    GL_REQUIRE(q_tiles * n_tiles < (1ll << 31), "gl_feat_knn_h1: grid too large");
    const int lds = 4 * GOPER;
    gl_prof_scope prof_(ctx, GL_PROF_FEAT_KNN);
    // variant 3: the persistent cluster form (needs 32 workgroup slots per cluster, i.e. a whole MI355X); 5: the persistent form without clusters
    // (what a device with fewer compute units gets -- the same bits).  Tuning builds add 0 / 1 / 2 / 4: the one-workgroup-per-tile kernels.
    // long rows are K-blocked (gl_conv.h gl_vrow_blocked: decided by the row length alone, the same rule the feature writers use); the buffers
    // then hold whole blocks of 256 rows
    const int blocked = gl_vrow_blocked(K1) ? 1 : 0;
    const int variant = blocked ? 3 : gl_tuning_int("GL_PAIR_VARIANT", 3);
    const int members = ctx->num_cu / kClusters;
    const bool clustered = variant == 3 && members >= kSuperN * kSuperQ;
    if (clustered || variant == 3 || variant == 5) {
        const int grid = clustered ? kClusters * members : (ctx->num_cu > 0 ? ctx->num_cu : 256);
        const size_t need = 4096 + (size_t)grid * kTotalsPerWg;
        if (ctx->pair_scratch_bytes < need) {
            GL_HIP(hipStreamSynchronize(ctx->stream));
            (void)hipFree(ctx->pair_scratch);
            ctx->pair_scratch = nullptr; ctx->pair_scratch_bytes = 0;
            GL_HIP(gl_device_alloc(ctx, (void **)&ctx->pair_scratch, need));
            ctx->pair_scratch_bytes = need;
        }
        GL_ONCE_PER_DEVICE(ctx, \
            GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(feat_knn_h1c_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
            GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(feat_knn_h1s_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds)););
        if (clustered) {
            GL_HIP(hipMemsetAsync(ctx->pair_scratch, 0, 4096, ctx->stream));         // the cluster counters
            hipLaunchKernelGGL(feat_knn_h1c_kernel, dim3((unsigned)grid), dim3(512), lds, ctx->stream, reinterpret_cast<const char *>(bank_V16_dev),
                               bank_norm_dev, n_rows, index_base, reinterpret_cast<const char *>(query_V16_dev), query_norm_dev, nq, K1,
                               reinterpret_cast<unsigned long long *>(keys_dev), (int)q_tiles, (int)n_tiles, ctx->pair_scratch, members, inv_s2, blocked);
        } else {
            hipLaunchKernelGGL(feat_knn_h1s_kernel, dim3((unsigned)grid), dim3(512), lds, ctx->stream, reinterpret_cast<const char *>(bank_V16_dev),
                               bank_norm_dev, n_rows, index_base, reinterpret_cast<const char *>(query_V16_dev), query_norm_dev, nq, K1,
                               reinterpret_cast<unsigned long long *>(keys_dev), (int)q_tiles, (int)n_tiles, ctx->pair_scratch, inv_s2, blocked);
        }
        GL_LAUNCH_CHECK();
        return GL_OK;
    }
#ifdef GL_TUNING
    GL_ONCE_PER_DEVICE(ctx, \
        GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(feat_knn_h1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(feat_knn_h1p_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(feat_knn_h1p_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(feat_knn_h1p_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)););
    auto kern = variant == 0 ? feat_knn_h1_kernel : variant == 4 ? feat_knn_h1p_kernel<4> : variant == 2 ? feat_knn_h1p_kernel<1> : feat_knn_h1p_kernel<8>;
    hipLaunchKernelGGL(kern, dim3((unsigned)(q_tiles * n_tiles)), dim3(512), lds, ctx->stream, reinterpret_cast<const char *>(bank_V16_dev), bank_norm_dev, n_rows,
                       index_base, reinterpret_cast<const char *>(query_V16_dev), query_norm_dev, nq, K1, reinterpret_cast<unsigned long long *>(keys_dev), (int)q_tiles,
                       (int)n_tiles, inv_s2);
    GL_LAUNCH_CHECK();
    return GL_OK;
#else
    gl_set_error("gl_feat_knn_h1: unreachable dispatch");
    return GL_ERR_STATE;
#endif
}

int gl_feat_knn_h1(gl_ctx *ctx, const void *bank_V16_dev, const float *bank_norm_dev, int64_t n_rows, int64_t index_base, const void *query_V16_dev,
                   const float *query_norm_dev, int64_t nq, int64_t K1, uint64_t *keys_dev)
{
    return gl_feat_knn_h1_scaled(ctx, bank_V16_dev, bank_norm_dev, n_rows, index_base, query_V16_dev, query_norm_dev, nq, K1, keys_dev, kVScale);
}

int gl_feat_knn(gl_ctx *ctx, const float *bank_V_dev, const float *bank_norm_dev, int64_t n_rows, int64_t index_base, const float *query_V_dev,
                    const float *query_norm_dev, int64_t nq, int64_t K, uint64_t *keys_dev)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && n_rows >= 0 && nq >= 0 && K > 0 && K % 32 == 0, "gl_feat_knn: bad sizes (K must be a multiple of 32)");
    GL_REQUIRE(index_base >= 0 && index_base + n_rows <= 0xFFFFFFFFll, "gl_feat_knn: global index does not fit 32 bits");
    if (n_rows == 0 || nq == 0) return GL_OK;
    GL_REQUIRE(bank_V_dev && bank_norm_dev && query_V_dev && query_norm_dev && keys_dev, "gl_feat_knn: NULL device pointer");
    GL_REQUIRE(((reinterpret_cast<uintptr_t>(bank_V_dev) | reinterpret_cast<uintptr_t>(query_V_dev)) & 15) == 0, "gl_feat_knn: rows must be 16-byte aligned");
    const int64_t q_tiles = gl_ceil_div(nq, FT), n_tiles = gl_ceil_div(n_rows, FT);
    GL_REQUIRE(q_tiles * n_tiles < (1ll << 31), "gl_feat_knn: grid too large");
    const int lds = 4 * FOPER;
    GL_ONCE_PER_DEVICE(ctx, \
        GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(feat_knn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds)););
    gl_prof_scope prof_(ctx, GL_PROF_FEAT_KNN);
    hipLaunchKernelGGL(feat_knn_kernel, dim3((unsigned)(q_tiles * n_tiles)), dim3(256), lds, ctx->stream, reinterpret_cast<const char *>(bank_V_dev), bank_norm_dev, n_rows,
                       index_base, reinterpret_cast<const char *>(query_V_dev), query_norm_dev, nq, K, reinterpret_cast<unsigned long long *>(keys_dev), (int)q_tiles, (int)n_tiles,
                       1.0f / (kVScale * kVScale), (const float *)nullptr, (const float *)nullptr);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int64_t gl_rows_split_dim(int64_t d) { return d <= 0 ? 0 : gl_ceil_div(d, 32) * 32; }

int gl_rows_split_f32(gl_ctx *ctx, const float *rows_f32_dev, int64_t n, int64_t d, void *V_dev, float *norms_dev, float *scales_dev)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && n >= 0 && d > 0, "gl_rows_split_f32: bad sizes");
    if (n == 0) return GL_OK;
    GL_REQUIRE(rows_f32_dev && V_dev && norms_dev && scales_dev, "gl_rows_split_f32: NULL device pointer");
    GL_REQUIRE((reinterpret_cast<uintptr_t>(V_dev) & 15) == 0, "gl_rows_split_f32: V must be 16-byte aligned");
    hipLaunchKernelGGL(rows_split_kernel, dim3((unsigned)(n < 4096 ? n : 4096)), dim3(256), 0, ctx->stream, rows_f32_dev, n, d, gl_rows_split_dim(d),
                       reinterpret_cast<char *>(V_dev), norms_dev, scales_dev);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int gl_rows_knn_split(gl_ctx *ctx, const void *bank_V_dev, const float *bank_norm_dev, const float *bank_scale_dev, int64_t n_rows, int64_t index_base,
                      const void *query_V_dev, const float *query_norm_dev, const float *query_scale_dev, int64_t nq, int64_t d, uint64_t *keys_dev)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && n_rows >= 0 && nq >= 0 && d > 0, "gl_rows_knn_split: bad sizes");
    GL_REQUIRE(index_base >= 0 && index_base + n_rows <= 0xFFFFFFFFll, "gl_rows_knn_split: global index does not fit 32 bits");
    if (n_rows == 0 || nq == 0) return GL_OK;
    GL_REQUIRE(bank_V_dev && bank_norm_dev && bank_scale_dev && query_V_dev && query_norm_dev && query_scale_dev && keys_dev, "gl_rows_knn_split: NULL device pointer");
    GL_REQUIRE(((reinterpret_cast<uintptr_t>(bank_V_dev) | reinterpret_cast<uintptr_t>(query_V_dev)) & 15) == 0, "gl_rows_knn_split: rows must be 16-byte aligned");
    const int64_t q_tiles = gl_ceil_div(nq, FT), n_tiles = gl_ceil_div(n_rows, FT);
    GL_REQUIRE(q_tiles * n_tiles < (1ll << 31), "gl_rows_knn_split: grid too large");
    const int lds = 4 * FOPER;
    GL_ONCE_PER_DEVICE(ctx, \
        GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(feat_knn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds)););
    gl_prof_scope prof_(ctx, GL_PROF_FEAT_KNN);
    hipLaunchKernelGGL(feat_knn_kernel, dim3((unsigned)(q_tiles * n_tiles)), dim3(256), lds, ctx->stream, reinterpret_cast<const char *>(bank_V_dev), bank_norm_dev, n_rows,
                       index_base, reinterpret_cast<const char *>(query_V_dev), query_norm_dev, nq, gl_rows_split_dim(d), reinterpret_cast<unsigned long long *>(keys_dev),
                       (int)q_tiles, (int)n_tiles, (float)(1.0 / (double)d), bank_scale_dev, query_scale_dev);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int gl_feat_rows_dist(gl_ctx *ctx, const float *V_hat_dev, int64_t b, const float *V_gt_dev, int64_t b_gt, int64_t K, int64_t K_lp, float *out_lpips_dev,
                      float *out_l2_dev)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && b >= 0 && K > 0 && K_lp >= 0 && K_lp <= K && K % 8 == 0 && K_lp % 8 == 0, "gl_feat_rows_dist: bad sizes");
    GL_REQUIRE(b_gt == 1 || b_gt == b, "gl_feat_rows_dist: x_gt must hold 1 row or %lld rows, got %lld", (long long)b, (long long)b_gt);
    if (b == 0) return GL_OK;
    GL_REQUIRE(V_hat_dev && V_gt_dev && out_lpips_dev && out_l2_dev, "gl_feat_rows_dist: NULL device pointer");
    hipLaunchKernelGGL(feat_rows_dist_kernel, dim3((unsigned)(b < 2048 ? b : 2048)), dim3(256), 0, ctx->stream, reinterpret_cast<const char *>(V_hat_dev), b, reinterpret_cast<const char *>(V_gt_dev), b_gt, K, K_lp,
                       out_lpips_dev, out_l2_dev);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

/* custom_knn with Loss('l2-lpips') for whole query sets held in HOST memory (the reference's default fbb distance, attack_models/fbb.py:148),
 * as one call for foreign callers: BATCH_SIZE truncation (fbb.py:77), search rows, the bank streamed through HBM in chunks so that
 * at most ~max_device_bytes of feature rows are resident (0 = 64 GiB). */
int gl_fbb_knn_lpips_host(gl_ctx *ctx, gl_lpips *l, const uint8_t *bank_u8_host, int64_t n_bank, const uint8_t *queries_u8_host, int64_t nq, int H, int W,
                          int64_t batch_size, int64_t max_device_bytes, float *dist_host, int64_t *idx_host)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && l && l->ctx == ctx && n_bank >= 0 && nq >= 0 && batch_size > 0, "gl_fbb_knn_lpips_host: bad argument");
    const int64_t K1 = gl_lpips_lattice_dim(H, W);          // 8-bit images on both sides: lattice search rows
    GL_REQUIRE(K1 > 0, "gl_fbb_knn_lpips_host: H, W must be multiples of 16, got %dx%d", H, W);
    const int64_t n_eff = (n_bank / batch_size) * batch_size;
    if (n_eff == 0) {
        gl_set_error("gl_fbb_knn_lpips_host: bank of %lld rows holds no full batch of %lld (reference: ValueError at fbb.py:83)", (long long)n_bank, (long long)batch_size);
        return GL_ERR_EMPTY_BANK;
    }
    if (nq == 0) return GL_OK;
    GL_REQUIRE(bank_u8_host && queries_u8_host && dist_host && idx_host, "gl_fbb_knn_lpips_host: NULL host pointer");
    const int64_t D = 3ll * H * W, row = K1 * 2;
    const int64_t budget = max_device_bytes > 0 ? max_device_bytes : (64ll << 30);
    GL_REQUIRE(nq * row <= budget, "gl_fbb_knn_lpips_host: %lld query rows of %lld bytes exceed the device budget; call with fewer queries", (long long)nq, (long long)row);
    int64_t chunk = budget / row;
    if (chunk > n_eff) chunk = n_eff;
    if (chunk < 1) chunk = 1;
    uint8_t *raw = nullptr;
    void *qV = nullptr, *bV = nullptr;
    float *qn = nullptr, *bn = nullptr, *dist = nullptr;
    uint64_t *keys = nullptr;
    int64_t *idx = nullptr;
    int rc = GL_OK;
    const int precision_in = l->precision;
    // split-fp16 VGG16 stores clamp at the fp16 range and count it: a chunk whose features saturated is recomputed with fp32 products (and so is
    // everything after it), as LpipsModel.features does on the Python side
    auto features = [&](int64_t n, int role, void *V, float *norms) -> int {
        int64_t sat = 0;
        int r = gl_ctx_h3_saturations(ctx, &sat);                      // clear what earlier calls on this context left behind
        if (r != GL_OK) return r;
        r = gl_lpips_lattice_features_u8(l, raw, n, H, W, V, norms);
        if (r != GL_OK || l->precision == 0) return r;
        r = gl_ctx_h3_saturations(ctx, &sat);
        if (r != GL_OK || sat == 0) return r;
        l->precision = 0;
        return gl_lpips_lattice_features_u8(l, raw, n, H, W, V, norms);
    };
#define GL_TRY(e) do { rc = (e); if (rc != GL_OK) goto done; } while (0)
    GL_TRY(gl_malloc(ctx, (size_t)((chunk > nq ? chunk : nq) * D), (void **)&raw));
    GL_TRY(gl_malloc(ctx, (size_t)(gl_vrow_capacity(nq, K1) * row), &qV));          // long rows are K-blocked: whole blocks of 256 rows
    GL_TRY(gl_malloc(ctx, (size_t)(gl_vrow_capacity(chunk, K1) * row), &bV));
    GL_TRY(gl_malloc(ctx, (size_t)nq * 4, (void **)&qn));
    GL_TRY(gl_malloc(ctx, (size_t)chunk * 4, (void **)&bn));
    GL_TRY(gl_malloc(ctx, (size_t)nq * 8, (void **)&keys));
    GL_TRY(gl_malloc(ctx, (size_t)nq * 4, (void **)&dist));
    GL_TRY(gl_malloc(ctx, (size_t)nq * 8, (void **)&idx));
    GL_TRY(gl_memcpy_h2d(ctx, raw, queries_u8_host, (size_t)(nq * D)));
    GL_TRY(features(nq, 0, qV, qn));
    GL_TRY(gl_keys_init(ctx, keys, nq));
    for (int64_t lo = 0; lo < n_eff; lo += chunk) {
        const int64_t m = n_eff - lo < chunk ? n_eff - lo : chunk;
        GL_TRY(gl_ctx_sync(ctx));                                  // raw is reused
        GL_TRY(gl_memcpy_h2d(ctx, raw, bank_u8_host + lo * D, (size_t)(m * D)));
        GL_TRY(features(m, 1, bV, bn));
        GL_TRY(gl_feat_knn_h1_scaled(ctx, bV, bn, m, lo, qV, qn, nq, K1, keys, gl_lpips_lattice_scale(H, W)));
    }
    GL_TRY(gl_keys_unpack_f32(ctx, keys, nq, dist, idx));
    GL_TRY(gl_memcpy_d2h(ctx, dist_host, dist, (size_t)nq * 4));
    GL_TRY(gl_memcpy_d2h(ctx, idx_host, idx, (size_t)nq * 8));
#undef GL_TRY
done:
    l->precision = precision_in;
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(raw); (void)hipFree(qV); (void)hipFree(bV); (void)hipFree(qn); (void)hipFree(bn); (void)hipFree(keys); (void)hipFree(dist); (void)hipFree(idx);
    return rc;
}

}  // extern "C"
