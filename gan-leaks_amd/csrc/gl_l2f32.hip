// L2 nearest neighbour for ARBITRARY fp32 images (values not on the 8-bit lattice): the general form of
// attack_models/utils.py:163 `mean((y - x)**2, dim=[1,2,3])` inside custom_knn (attack_models/fbb.py:73-88).
//
// The reference's own fp32 result depends on its reduction order (different on its CPU and CUDA builds), so
// this path fixes ONE order, shared bit for bit with the oracle (oracle/fbb_oracle.c gl_oracle_l2_f32):
//     d_k   = fl32(y_k - x_k)
//     c_j   = fmaf chain over k = j, j+4, j+8, ...            (j = 0..3, ascending k)
//     dist  = fl32( fl32( fl32(c_0 + c_1) + fl32(c_2 + c_3) ) / D )
// Every pair is an independent chain, so the result does not depend on tiling, launch shape or shard count.
// key = (float bits of dist) << 32 | global index, merged with atomicMin like the integer path.
//
// VALU kernel (2 ops per element pair; the matrix cores cannot form y - x before squaring without losing
// the fp32 difference).  Tile: 64 queries x 64 bank rows per workgroup, 256 threads x (4 x 4) pairs,
// K slices of 32 floats staged through LDS (row stride 36 floats: conflict-free float4 reads).
#include "gl_common.h"

namespace {

constexpr int TQ = 64, TN = 64, KS = 32, LDS_STRIDE = KS + 4, THREADS = 256;

__device__ __forceinline__ float4 load_row4(const float *__restrict__ base, int64_t row, int64_t nrows, int64_t d, int64_t k, bool vec)
{
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row >= nrows) return v;
    const float *p = base + row * d + k;
    if (vec) {
        if (k + 3 < d) return *reinterpret_cast<const float4 *>(p);
    }
    if (k + 0 < d) v.x = p[0];
    if (k + 1 < d) v.y = p[1];
    if (k + 2 < d) v.z = p[2];
    if (k + 3 < d) v.w = p[3];
    return v;
}

__global__ void __launch_bounds__(THREADS) l2_knn_f32_kernel(const float *__restrict__ bank, int64_t n_rows, int64_t index_base,
                                                              const float *__restrict__ query, int64_t nq, int64_t d,
                                                              unsigned long long *__restrict__ keys, int q_tiles)
{
    __shared__ __attribute__((aligned(16))) float sq[TQ * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float sb[TN * LDS_STRIDE];
    const int qt = blockIdx.x % q_tiles, nt = blockIdx.x / q_tiles;
    const int64_t q0 = (int64_t)qt * TQ, n0 = (int64_t)nt * TN;
    const int tid = threadIdx.x;
    const int tq = tid >> 4, tn = tid & 15;          // 16 x 16 threads, each 4 queries x 4 bank rows
    const bool vec = ((d & 3) == 0) && (((reinterpret_cast<uintptr_t>(bank) | reinterpret_cast<uintptr_t>(query)) & 15) == 0);

    float acc[4][4][4];                               // [query][bank row][chain j]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[a][b][j] = 0.0f;

    for (int64_t k0 = 0; k0 < d; k0 += KS) {
        // stage: 64 rows x 8 float4 per operand = 512 float4, 2 per thread per operand (zero beyond d / beyond the rows)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * THREADS;
            const int r = e >> 3, c = (e & 7) * 4;
            *reinterpret_cast<float4 *>(&sq[r * LDS_STRIDE + c]) = load_row4(query, q0 + r, nq, d, k0 + c, vec);
            *reinterpret_cast<float4 *>(&sb[r * LDS_STRIDE + c]) = load_row4(bank, n0 + r, n_rows, d, k0 + c, vec);
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KS; kk += 4) {
            float4 qv[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) qv[a] = *reinterpret_cast<const float4 *>(&sq[(tq * 4 + a) * LDS_STRIDE + kk]);
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = *reinterpret_cast<const float4 *>(&sb[(tn + 16 * b) * LDS_STRIDE + kk]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    float t;
                    t = __fsub_rn(qv[a].x, bv[b].x); acc[a][b][0] = fmaf(t, t, acc[a][b][0]);
                    t = __fsub_rn(qv[a].y, bv[b].y); acc[a][b][1] = fmaf(t, t, acc[a][b][1]);
                    t = __fsub_rn(qv[a].z, bv[b].z); acc[a][b][2] = fmaf(t, t, acc[a][b][2]);
                    t = __fsub_rn(qv[a].w, bv[b].w); acc[a][b][3] = fmaf(t, t, acc[a][b][3]);
                }
        }
        __syncthreads();
    }

    const float fd = (float)d;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int64_t q = q0 + tq * 4 + a;
        unsigned long long best = ~0ull;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int64_t n = n0 + tn + 16 * b;
            const float s = __fadd_rn(__fadd_rn(acc[a][b][0], acc[a][b][1]), __fadd_rn(acc[a][b][2], acc[a][b][3]));
            const float dist = __fdiv_rn(s, fd);
            const unsigned long long key = ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned long long)(index_base + n);
            if (n < n_rows && key < best) best = key;
        }
        // the 16 threads tn = 0..15 (consecutive lanes) hold the other bank rows of this query
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o, 64);
            best = other < best ? other : best;
        }
        if (tn == 0 && q < nq && best != ~0ull) atomicMin(&keys[q], best);
    }
}

// per-row distances (Loss('l2').forward for float inputs), same chain definition; one wave per row pair is
// not possible (the chain is sequential), so one THREAD per row: only used for BATCH_SIZE-sized calls.
__global__ void l2_rows_f32_kernel(const float *__restrict__ xh, int64_t b, const float *__restrict__ xg, int64_t b_gt, int64_t d,
                                   float *__restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= b) return;
    const float *pa = xh + r * d;
    const float *pb = xg + (b_gt == 1 ? 0 : r) * d;
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t k = 0; k < d; ++k) {
        const float t = __fsub_rn(pb[k], pa[k]);
        c[k & 3] = fmaf(t, t, c[k & 3]);
    }
    out[r] = __fdiv_rn(__fadd_rn(__fadd_rn(c[0], c[1]), __fadd_rn(c[2], c[3])), (float)d);
}

__global__ void keys_unpack_f32_kernel(const unsigned long long *__restrict__ keys, int64_t nq, float *__restrict__ dist, int64_t *__restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const unsigned long long k = keys[i];
    dist[i] = __uint_as_float((unsigned)(k >> 32));
    idx[i] = (int64_t)(k & 0xFFFFFFFFull);
}

}  // namespace

extern "C" {

int gl_l2_knn_f32(gl_ctx *ctx, const float *bank_dev, int64_t n_rows, int64_t index_base, const float *query_dev, int64_t nq, int64_t d,
                  uint64_t *keys_dev)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && n_rows >= 0 && nq >= 0 && d > 0, "gl_l2_knn_f32: bad sizes");
    GL_REQUIRE(index_base >= 0 && index_base + n_rows <= 0xFFFFFFFFll, "gl_l2_knn_f32: global index does not fit 32 bits");
    if (n_rows == 0 || nq == 0) return GL_OK;
    GL_REQUIRE(bank_dev && query_dev && keys_dev, "gl_l2_knn_f32: NULL device pointer");
    const int64_t q_tiles = gl_ceil_div(nq, TQ), n_tiles = gl_ceil_div(n_rows, TN);
    GL_REQUIRE(q_tiles * n_tiles < (1ll << 31), "gl_l2_knn_f32: grid too large");
    hipLaunchKernelGGL(l2_knn_f32_kernel, dim3((unsigned)(q_tiles * n_tiles)), dim3(THREADS), 0, ctx->stream, bank_dev, n_rows, index_base, query_dev,
                       nq, d, reinterpret_cast<unsigned long long *>(keys_dev), (int)q_tiles);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int gl_l2_rows_f32(gl_ctx *ctx, const float *x_hat_dev, int64_t b, const float *x_gt_dev, int64_t b_gt, int64_t d, float *out_dev)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && b >= 0 && d > 0, "gl_l2_rows_f32: bad ctx/b/d");
    GL_REQUIRE(b_gt == 1 || b_gt == b, "gl_l2_rows_f32: x_gt must hold 1 row or %lld rows, got %lld", (long long)b, (long long)b_gt);
    if (b == 0) return GL_OK;
    GL_REQUIRE(x_hat_dev && x_gt_dev && out_dev, "gl_l2_rows_f32: NULL device pointer");
    hipLaunchKernelGGL(l2_rows_f32_kernel, dim3((unsigned)gl_ceil_div(b, 64)), dim3(64), 0, ctx->stream, x_hat_dev, b, x_gt_dev, b_gt, d, out_dev);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int gl_keys_unpack_f32(gl_ctx *ctx, const uint64_t *keys_dev, int64_t nq, float *dist_dev, int64_t *idx_dev)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && nq >= 0, "gl_keys_unpack_f32: bad ctx/nq");
    if (nq == 0) return GL_OK;
    GL_REQUIRE(keys_dev && dist_dev && idx_dev, "gl_keys_unpack_f32: NULL device pointer");
    hipLaunchKernelGGL(keys_unpack_f32_kernel, dim3((unsigned)gl_ceil_div(nq, 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const unsigned long long *>(keys_dev), nq, dist_dev, idx_dev);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

}  // extern "C"
