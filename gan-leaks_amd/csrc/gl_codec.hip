// 8-bit image codec and the small HBM-bound kernels around the L2 nearest-neighbour kernel.
// All are streaming kernels: 16 B per lane where the layout allows, grid-stride, <= 2048 blocks.
#include "gl_common.h"

namespace {

constexpr int kThreads = 256;

inline int stream_grid(int64_t work_items)
{
    int64_t b = gl_ceil_div(work_items, kThreads);
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

// x == fl32(2*(u/255.)-1)?  table built once per block from the same float64 expression the
// reference evaluates (attack_models/utils.py:82), so the comparison is exact.
// INT = true: the integer lattice x == (float)u, u in 0..255 (binary / count tables such as medGAN's thresholded rows)
template <bool INT>
__global__ void __launch_bounds__(kThreads) encode_lattice_kernel(const float *__restrict__ x, int64_t count,
                                                                  uint8_t *__restrict__ out, int32_t *__restrict__ off)
{
    __shared__ float lut[256];
    if (threadIdx.x < 256) lut[threadIdx.x] = INT ? (float)threadIdx.x : (float)(2.0 * ((double)threadIdx.x / 255.0) - 1.0);
    __syncthreads();
    int bad = 0;
    const int64_t n4 = count >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        const float f[4] = {v.x, v.y, v.z, v.w};
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float t = INT ? rintf(f[k]) : rintf((f[k] + 1.0f) * 127.5f);
            t = fminf(fmaxf(t, 0.0f), 255.0f);
            const int u = (int)t;
            bad += (lut[u] != f[k]);   // NaN compares unequal -> counted
            packed |= (uint32_t)u << (8 * k);
        }
        reinterpret_cast<uint32_t *>(out)[i] = packed;
    }
    // tail (count % 4) handled by block 0
    if (blockIdx.x == 0) {
        for (int64_t i = (n4 << 2) + threadIdx.x; i < count; i += blockDim.x) {
            float t = INT ? rintf(x[i]) : rintf((x[i] + 1.0f) * 127.5f);
            t = fminf(fmaxf(t, 0.0f), 255.0f);
            const int u = (int)t;
            bad += (lut[u] != x[i]);
            out[i] = (uint8_t)u;
        }
    }
    if (bad) atomicAdd(off, bad);
}

template <bool INT>
__global__ void __launch_bounds__(kThreads) decode_u8_kernel(const uint8_t *__restrict__ u8, int64_t count, float *__restrict__ x)
{
    __shared__ float lut[256];
    if (threadIdx.x < 256) lut[threadIdx.x] = INT ? (float)threadIdx.x : (float)(2.0 * ((double)threadIdx.x / 255.0) - 1.0);
    __syncthreads();
    const int64_t n4 = count >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const uint32_t p = reinterpret_cast<const uint32_t *>(u8)[i];
        float4 v;
        v.x = lut[p & 255u];
        v.y = lut[(p >> 8) & 255u];
        v.z = lut[(p >> 16) & 255u];
        v.w = lut[p >> 24];
        reinterpret_cast<float4 *>(x)[i] = v;
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < count; i += blockDim.x) x[i] = lut[u8[i]];
}

// the generate branch's float -> byte step, same fp32 operation order as the reference:
// mode 0: t = (x + 1) / 2 ; mode 1: t = x*0.5 + 0.5 (two roundings, no FMA contraction);
// u = trunc(t * 255) clamped to [0,255].
__device__ __forceinline__ uint32_t quantize_one(float x, int mode)
{
    float t;
    if (mode == 0) t = __fdiv_rn(__fadd_rn(x, 1.0f), 2.0f);
    else t = __fadd_rn(__fmul_rn(x, 0.5f), 0.5f);
    t = __fmul_rn(t, 255.0f);
    t = fminf(fmaxf(truncf(t), 0.0f), 255.0f);
    return (uint32_t)(int)t;
}

__global__ void __launch_bounds__(kThreads) quantize_kernel(const float *__restrict__ x, int64_t count, int mode, uint8_t *__restrict__ out)
{
    const int64_t n4 = count >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        const uint32_t p = quantize_one(v.x, mode) | (quantize_one(v.y, mode) << 8) | (quantize_one(v.z, mode) << 16) |
                           (quantize_one(v.w, mode) << 24);
        reinterpret_cast<uint32_t *>(out)[i] = p;
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < count; i += blockDim.x) out[i] = (uint8_t)quantize_one(x[i], mode);
}

// one wave per row: u8 -> (u ^ 0x80) as int8, zero padded to `stride` bytes; norm = sum (u-128)^2.
// v_dot4_i32_i8 does the four squares of a dword in one instruction.
__global__ void __launch_bounds__(kThreads) l2_prepare_kernel(const uint8_t *__restrict__ rows, int64_t count, int64_t d, int64_t stride,
                                                              int8_t *__restrict__ out, int32_t *__restrict__ norms)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const bool vec = ((d & 15) == 0) && ((reinterpret_cast<uintptr_t>(rows) & 15) == 0);
    for (int64_t r = wave; r < count; r += nwaves) {
        const uint8_t *src = rows + r * d;
        int8_t *dst = out + r * stride;
        int acc = 0;                       // sum (u-128)^2 <= 16384 d < 2^32: read as unsigned when d > 131071
        if (vec) {
            for (int64_t k = (int64_t)lane * 16; k < stride; k += 64 * 16) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (k < d) {
                    v = *reinterpret_cast<const uint4 *>(src + k);
                    v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
                    acc = __builtin_amdgcn_sdot4((int)v.x, (int)v.x, acc, false);
                    acc = __builtin_amdgcn_sdot4((int)v.y, (int)v.y, acc, false);
                    acc = __builtin_amdgcn_sdot4((int)v.z, (int)v.z, acc, false);
                    acc = __builtin_amdgcn_sdot4((int)v.w, (int)v.w, acc, false);
                }
                *reinterpret_cast<uint4 *>(dst + k) = v;
            }
        } else {
            for (int64_t k = lane; k < stride; k += 64) {
                int b = 0;
                if (k < d) { b = (int)src[k] - 128; acc += b * b; }
                dst[k] = (int8_t)b;
            }
        }
        unsigned total = (unsigned)acc;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
        if (lane == 0) norms[r] = (int32_t)total;
    }
}

// per-row squared L2 between x_hat rows and x_gt (one row broadcast, or one per x_hat row): the [B]
// vector Loss('l2').forward returns (attack_models/utils.py:163,169,176).  One wave per row,
// S = sum a^2 + sum b^2 - 2 sum ab with the unsigned 8-bit dot instruction, exact in uint32.
__global__ void __launch_bounds__(kThreads) l2_rows_u8_kernel(const uint8_t *__restrict__ xh, int64_t b, const uint8_t *__restrict__ xg, int64_t b_gt,
                                                              int64_t d, double scale, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < b; r += nwaves) {
        const uint8_t *pa = xh + r * d;
        const uint8_t *pb = xg + (b_gt == 1 ? 0 : r) * d;
        unsigned saa = 0, sbb = 0, sab = 0;
        if (((d & 3) == 0) && (((reinterpret_cast<uintptr_t>(pa) | reinterpret_cast<uintptr_t>(pb)) & 3) == 0)) {
            for (int64_t k = (int64_t)lane * 4; k < d; k += 256) {
                const unsigned a = *reinterpret_cast<const unsigned *>(pa + k), c = *reinterpret_cast<const unsigned *>(pb + k);
                saa = __builtin_amdgcn_udot4(a, a, saa, false);
                sbb = __builtin_amdgcn_udot4(c, c, sbb, false);
                sab = __builtin_amdgcn_udot4(a, c, sab, false);
            }
        } else {
            for (int64_t k = lane; k < d; k += 64) {
                const unsigned a = pa[k], c = pb[k];
                saa += a * a; sbb += c * c; sab += a * c;
            }
        }
        long long s = (long long)saa + (long long)sbb - 2ll * (long long)sab;   // per-lane parts fit 32 unsigned bits (d / 64 values each)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) out[r] = (float)((double)s * scale);
    }
}

// INT: dist = fl32(S / d) -- the double quotient rounded once more to fp32 is the correctly rounded fp32 quotient (53 >= 2*24+2),
// i.e. what an fp32 `sum / d` gives while the sum is exact
template <bool INT>
__global__ void __launch_bounds__(kThreads) keys_unpack_kernel(const uint64_t *__restrict__ keys, int64_t nq, double scale, int shift,
                                                               float *__restrict__ dist, int64_t *__restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const uint64_t k = keys[i];
    dist[i] = INT ? (float)((double)(k >> shift) / scale) : (float)((double)(k >> shift) * scale);   // fl32(S * 4/(255^2 D)), same expression as the oracle
    idx[i] = (int64_t)(k & ((1ull << shift) - 1ull));
}

}  // namespace

extern "C" {

static int encode_impl(gl_ctx *ctx, const float *x_dev, int64_t count, uint8_t *u8_dev, int32_t *off_lattice_dev, bool integers)
{
    GL_REQUIRE(ctx && count >= 0, "gl_encode_*_f32: bad ctx/count");
    if (count == 0) return GL_OK;
    GL_REQUIRE(x_dev && u8_dev && off_lattice_dev, "gl_encode_*_f32: NULL device pointer");
    GL_REQUIRE((reinterpret_cast<uintptr_t>(x_dev) & 15) == 0 && (reinterpret_cast<uintptr_t>(u8_dev) & 3) == 0,
               "gl_encode_*_f32: x_dev must be 16-byte and u8_dev 4-byte aligned");
    if (integers)
        hipLaunchKernelGGL(encode_lattice_kernel<true>, dim3(stream_grid(count / 4)), dim3(kThreads), 0, ctx->stream, x_dev, count, u8_dev, off_lattice_dev);
    else
        hipLaunchKernelGGL(encode_lattice_kernel<false>, dim3(stream_grid(count / 4)), dim3(kThreads), 0, ctx->stream, x_dev, count, u8_dev, off_lattice_dev);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

static int decode_impl(gl_ctx *ctx, const uint8_t *u8_dev, int64_t count, float *x_dev, bool integers)
{
    GL_REQUIRE(ctx && count >= 0, "gl_decode_u8*: bad ctx/count");
    if (count == 0) return GL_OK;
    GL_REQUIRE(x_dev && u8_dev, "gl_decode_u8*: NULL device pointer");
    GL_REQUIRE((reinterpret_cast<uintptr_t>(x_dev) & 15) == 0 && (reinterpret_cast<uintptr_t>(u8_dev) & 3) == 0,
               "gl_decode_u8*: x_dev must be 16-byte and u8_dev 4-byte aligned");
    if (integers) hipLaunchKernelGGL(decode_u8_kernel<true>, dim3(stream_grid(count / 4)), dim3(kThreads), 0, ctx->stream, u8_dev, count, x_dev);
    else hipLaunchKernelGGL(decode_u8_kernel<false>, dim3(stream_grid(count / 4)), dim3(kThreads), 0, ctx->stream, u8_dev, count, x_dev);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int gl_encode_lattice_f32(gl_ctx *ctx, const float *x_dev, int64_t count, uint8_t *u8_dev, int32_t *off_lattice_dev)
{
    gl_make_current(ctx);
    return encode_impl(ctx, x_dev, count, u8_dev, off_lattice_dev, false);
}

int gl_encode_integers_f32(gl_ctx *ctx, const float *x_dev, int64_t count, uint8_t *u8_dev, int32_t *off_lattice_dev)
{
    gl_make_current(ctx);
    return encode_impl(ctx, x_dev, count, u8_dev, off_lattice_dev, true);
}

int gl_decode_u8(gl_ctx *ctx, const uint8_t *u8_dev, int64_t count, float *x_dev)
{
    gl_make_current(ctx);
    return decode_impl(ctx, u8_dev, count, x_dev, false);
}

int gl_decode_u8_integers(gl_ctx *ctx, const uint8_t *u8_dev, int64_t count, float *x_dev)
{
    gl_make_current(ctx);
    return decode_impl(ctx, u8_dev, count, x_dev, true);
}

int gl_quantize_f32(gl_ctx *ctx, const float *x_dev, int64_t count, int mode, uint8_t *u8_dev)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && count >= 0 && (mode == 0 || mode == 1), "gl_quantize_f32: bad ctx/count/mode");
    if (count == 0) return GL_OK;
    GL_REQUIRE(x_dev && u8_dev, "gl_quantize_f32: NULL device pointer");
    GL_REQUIRE((reinterpret_cast<uintptr_t>(x_dev) & 15) == 0 && (reinterpret_cast<uintptr_t>(u8_dev) & 3) == 0,
               "gl_quantize_f32: x_dev must be 16-byte and u8_dev 4-byte aligned");
    hipLaunchKernelGGL(quantize_kernel, dim3(stream_grid(count / 4)), dim3(kThreads), 0, ctx->stream, x_dev, count, mode, u8_dev);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int64_t gl_l2_row_stride(int64_t d) { return d <= 0 ? 0 : gl_ceil_div(d, 128) * 128; }

int gl_l2_prepare(gl_ctx *ctx, const uint8_t *rows_u8_dev, int64_t count, int64_t d, int8_t *rows_i8_dev, int32_t *norms_dev)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && count >= 0 && d > 0, "gl_l2_prepare: bad ctx/count/d");
    // sum (u-128)^2 <= 128^2 * d must fit int32 together with the cross term (see gl_l2knn.hip)
    GL_REQUIRE(d <= GL_L2_MAX_D, "gl_l2_prepare: d=%lld exceeds the exact-integer limit %lld", (long long)d, (long long)GL_L2_MAX_D);
    if (count == 0) return GL_OK;
    GL_REQUIRE(rows_u8_dev && rows_i8_dev && norms_dev, "gl_l2_prepare: NULL device pointer");
    GL_REQUIRE((reinterpret_cast<uintptr_t>(rows_i8_dev) & 15) == 0, "gl_l2_prepare: rows_i8_dev must be 16-byte aligned");
    const int64_t waves_needed = count;
    int64_t blocks = gl_ceil_div(waves_needed, kThreads / 64);
    if (blocks > 4096) blocks = 4096;
    gl_prof_scope prof_(ctx, GL_PROF_L2_PREPARE);
    hipLaunchKernelGGL(l2_prepare_kernel, dim3((int)blocks), dim3(kThreads), 0, ctx->stream, rows_u8_dev, count, d, gl_l2_row_stride(d),
                       rows_i8_dev, norms_dev);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int gl_l2_rows_u8(gl_ctx *ctx, const uint8_t *x_hat_u8_dev, int64_t b, const uint8_t *x_gt_u8_dev, int64_t b_gt, int64_t d, float *out_dev)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && b >= 0 && d > 0 && d <= GL_L2_MAX_D, "gl_l2_rows_u8: bad ctx/b/d");
    GL_REQUIRE(b_gt == 1 || b_gt == b, "gl_l2_rows_u8: x_gt must hold 1 row or %lld rows (broadcast rule of utils.py:163), got %lld", (long long)b,
               (long long)b_gt);
    if (b == 0) return GL_OK;
    GL_REQUIRE(x_hat_u8_dev && x_gt_u8_dev && out_dev, "gl_l2_rows_u8: NULL device pointer");
    int64_t blocks = gl_ceil_div(b, kThreads / 64);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(l2_rows_u8_kernel, dim3((int)blocks), dim3(kThreads), 0, ctx->stream, x_hat_u8_dev, b, x_gt_u8_dev, b_gt, d,
                       4.0 / (65025.0 * (double)d), out_dev);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int gl_keys_init(gl_ctx *ctx, uint64_t *keys_dev, int64_t nq)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && nq >= 0, "gl_keys_init: bad ctx/nq");
    if (nq == 0) return GL_OK;
    GL_REQUIRE(keys_dev, "gl_keys_init: NULL keys");
    GL_HIP(hipMemsetAsync(keys_dev, 0xFF, (size_t)nq * 8, ctx->stream));
    return GL_OK;
}

int gl_keys_unpack(gl_ctx *ctx, const uint64_t *keys_dev, int64_t nq, int64_t d, float *dist_dev, int64_t *idx_dev)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && nq >= 0 && d > 0, "gl_keys_unpack: bad ctx/nq/d");
    if (nq == 0) return GL_OK;
    GL_REQUIRE(keys_dev && dist_dev && idx_dev, "gl_keys_unpack: NULL device pointer");
    const double scale = 4.0 / (65025.0 * (double)d);
    hipLaunchKernelGGL(keys_unpack_kernel<false>, dim3((int)gl_ceil_div(nq, kThreads)), dim3(kThreads), 0, ctx->stream, keys_dev, nq, scale, gl_l2_key_shift(d), dist_dev, idx_dev);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int gl_keys_unpack_integers(gl_ctx *ctx, const uint64_t *keys_dev, int64_t nq, int64_t d, float *dist_dev, int64_t *idx_dev)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && nq >= 0 && d > 0, "gl_keys_unpack_integers: bad ctx/nq/d");
    if (nq == 0) return GL_OK;
    GL_REQUIRE(keys_dev && dist_dev && idx_dev, "gl_keys_unpack_integers: NULL device pointer");
    hipLaunchKernelGGL(keys_unpack_kernel<true>, dim3((int)gl_ceil_div(nq, kThreads)), dim3(kThreads), 0, ctx->stream, keys_dev, nq, (double)d, gl_l2_key_shift(d), dist_dev,
                       idx_dev);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

}  // extern "C"
