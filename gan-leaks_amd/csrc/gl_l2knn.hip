// Pairwise squared-L2 + per-query argmin over the sample bank: the |queries| x |bank| contraction
// of the fbb attack (attack_models/fbb.py:77-86 with attack_models/utils.py:163) as ONE kernel.
//
// Arithmetic: images live on the 8-bit lattice, so with a = u_bank-128, b = u_query-128 (int8)
//     S(q,n) = sum_k (b_k - a_k)^2 = |a_n|^2 + |b_q|^2 - 2 * sum_k a_nk b_qk
// is computed EXACTLY: the cross term on the int8 matrix cores (v_mfma_i32_16x16x64_i8, int32
// accumulate), the norms precomputed by gl_l2_prepare.  Ranges: 0 <= S <= 255^2 d.
//   d <= 66051 (S < 2^32): norms, cross term and S are evaluated modulo 2^32 (two's-complement wrap-around of the int32
//     accumulators is harmless), which is exact because the true S fits 32 unsigned bits;
//   d <= 262143 (BIG = true): the int32 accumulators are flushed into 64-bit totals every 64 KiB of K (|cross| <= 2^30 per
//     segment), S = |a|^2 + |b|^2 - 2 cross in 64 bits.
// key = S << gl_l2_key_shift(d) | global index (shift 32 up to d = 33025).
//
// Tiling (v1): 128 bank rows x 128 queries per workgroup, 4 waves as 2 x 2, each wave 64 x 64 =
// 4 x 4 MFMA tiles; K streamed in 128-byte slices, double buffered in LDS (64 KiB -> 2 WG/CU),
// filled by global_load_lds (16 B/lane).  LDS rows are 128 B; chunk c of row r is stored at slot
// c ^ (r & 7) (applied on the per-lane SOURCE address, the LDS write itself is lane-linear), which
// makes every ds_read_b128 of an MFMA operand conflict-free.
// Epilogue: S from the accumulators, min over the wave's 64 bank rows in registers + 2 shuffles,
// one 64-bit atomicMin per (query, wave) on key = S << 32 | global_index  -- smallest index wins
// ties, as torch.min does (fbb.py:86), independent of tile / shard order.
#include "gl_common.h"
#include "gl_pair256.h"
#include <cstdlib>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int TILE_N = 128;   // bank rows per workgroup   (MFMA "M")
constexpr int TILE_Q = 128;   // queries per workgroup     (MFMA "N")
constexpr int TILE_K = 128;   // bytes of K per slice
constexpr int THREADS = 256;
constexpr int OPER_BYTES = TILE_N * TILE_K;   // 16 KiB per operand per buffer

// stage one operand slice: 128 rows x 128 B.  Each wave-instruction writes 1 KiB = 8 rows.
// lane -> (row = lane / 8, slot = lane % 8); the slot holds global chunk slot ^ (row & 7).
__device__ __forceinline__ void stage_operand(const int8_t *__restrict__ base, int64_t row0, int64_t nrows_valid, int64_t stride, int64_t kbyte,
                                              char *lds_oper, int wave, int lane)
{
    const int rsub = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i;            // 16 pieces of 8 rows
        const int r = piece * 8 + rsub;            // row in tile
        int64_t gr = row0 + r;
        if (gr >= nrows_valid) gr = nrows_valid - 1;   // clamp: duplicates are masked in the epilogue
        const int chunk = slot ^ (r & 7);
        gl_glds16(base + gr * stride + kbyte + chunk * 16, lds_oper + piece * 1024);
    }
}

template <bool BIG>
__global__ void __launch_bounds__(THREADS, 2)
l2_knn_i8_kernel(const int8_t *__restrict__ bank, const int32_t *__restrict__ bank_norm, int64_t n_rows, int64_t index_base,
                 const int8_t *__restrict__ query, const int32_t *__restrict__ query_norm, int64_t nq, int64_t stride,
                 unsigned long long *__restrict__ keys, int q_tiles, int n_tiles, int shift)
{
    constexpr int FLUSH = 512;                        // slices per 64-bit flush (BIG): 64 KiB of K
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][bank 16 KiB | query 16 KiB]

    const unsigned nwg = (unsigned)q_tiles * (unsigned)n_tiles;
    const unsigned id = gl_xcd_remap(blockIdx.x, nwg);
    int qt, nt;
    gl_strip_order(id, q_tiles, n_tiles, qt, nt);     // co-resident blocks of an XCD: ~8 bank tiles x 8 query tiles
    const int64_t n0 = (int64_t)nt * TILE_N, q0 = (int64_t)qt * TILE_Q;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wq = wave & 1;           // wave position in the 2 x 2 grid
    const int frow = lane & 15, fk = lane >> 4;        // MFMA operand lane map: row, 16-byte k group

    v4i acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4i){0, 0, 0, 0};
    long long tot[BIG ? 4 : 1][BIG ? 4 : 1][4] = {};
    auto flush = [&]() {
        if constexpr (BIG) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { tot[i][j][r] += (long long)acc[i][j][r]; acc[i][j][r] = 0; }
        }
    };

    const int nk = (int)(stride / TILE_K);
    stage_operand(bank, n0, n_rows, stride, 0, smem, wave, lane);
    stage_operand(query, q0, nq, stride, 0, smem + OPER_BYTES, wave, lane);

    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();   // (vmcnt(0) + barrier) slice kt landed; everyone is done reading the other buffer
        char *cur = smem + (kt & 1) * 2 * OPER_BYTES;
        if (kt + 1 < nk) {
            char *nxt = smem + ((kt + 1) & 1) * 2 * OPER_BYTES;
            stage_operand(bank, n0, n_rows, stride, (int64_t)(kt + 1) * TILE_K, nxt, wave, lane);
            stage_operand(query, q0, nq, stride, (int64_t)(kt + 1) * TILE_K, nxt + OPER_BYTES, wave, lane);
        }
        const char *lb = cur + (wn * 64) * TILE_K;
        const char *lq = cur + OPER_BYTES + (wq * 64) * TILE_K;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {              // two 64-byte MFMA k-steps per slice
            const int chunk = ks * 4 + fk;
            v4i a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = i * 16 + frow;
                a[i] = *reinterpret_cast<const v4i *>(lb + r * TILE_K + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = j * 16 + frow;
                b[j] = *reinterpret_cast<const v4i *>(lq + r * TILE_K + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (BIG && (kt % FLUSH) == FLUSH - 1) flush();
    }
    flush();

    // ---- epilogue.  C layout of the 16x16 tile: column (query) = lane & 15, row (bank) = (lane>>4)*4 + reg.
    const int64_t nbase = n0 + wn * 64 + fk * 4;
    int bn[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t n = nbase + i * 16 + r;
            bn[i][r] = n < n_rows ? bank_norm[n] : 0;
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t q = q0 + wq * 64 + j * 16 + frow;
        const int qn = q < nq ? query_norm[q] : 0;
        unsigned long long best = ~0ull;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t n = nbase + i * 16 + r;
                unsigned long long s;
                if constexpr (BIG) s = (unsigned long long)((long long)(unsigned)bn[i][r] + (long long)(unsigned)qn - 2ll * tot[i][j][r]);
                else s = (unsigned)bn[i][r] + (unsigned)qn - 2u * (unsigned)acc[i][j][r];       // exact modulo 2^32, and S < 2^32
                const unsigned long long key = (s << shift) | (unsigned long long)(index_base + n);
                if (n < n_rows && key < best) best = key;
            }
        // the other three k-groups of lanes hold other bank rows of the same query
        unsigned long long o = __shfl_xor(best, 16, 64);
        best = o < best ? o : best;
        o = __shfl_xor(best, 32, 64);
        best = o < best ? o : best;
        if (fk == 0 && q < nq && best != ~0ull) atomicMin(&keys[q], best);
    }
}

// 256 bank rows x 256 queries per workgroup: half the operand bytes per MFMA of the 128 x 128 tile.  The bank and the queries
// are streamed from beyond L2 (a tile's two panels are 2 x 256 x d bytes; the co-resident tiles of an XCD cycle through more
// panels than its 4 MiB L2 holds), and with the int8 matrix rate twice the fp16 one that stream, not the matrix pipe, sets the
// time of the small tile.  8 waves as 2 (bank) x 4 (query), each 128 x 64 = 8 x 4 MFMA tiles; 128-byte K slices double buffered
// in 128 KiB of LDS (one workgroup per CU); strips of 4 bank tiles so the 32 workgroups of an XCD cover 4 x 8 tiles.
constexpr int BT = 256, BOPER = BT * TILE_K;

#ifdef GL_TUNING      // the round-1 form of the 256 x 256 tile (plain double buffering): A/B material and the bit-for-bit check of the pipelined kernel
__global__ void __launch_bounds__(512, 2)
l2_knn_i8_256_kernel(const int8_t *__restrict__ bank, const int32_t *__restrict__ bank_norm, int64_t n_rows, int64_t index_base,
                     const int8_t *__restrict__ query, const int32_t *__restrict__ query_norm, int64_t nq, int64_t stride,
                     unsigned long long *__restrict__ keys, int q_tiles, int n_tiles, int shift)
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
    const int64_t n0 = (int64_t)nt * BT, q0 = (int64_t)qt * BT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wq = wave & 3;
    const int rsub = lane >> 3, slot = lane & 7;
    const int frow = lane & 15, fk = lane >> 4;

    const int8_t *a_src[4], *b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + rsub;
        int64_t gn = n0 + r, gq = q0 + r;
        if (gn >= n_rows) gn = n_rows - 1;          // clamped duplicates are masked in the epilogue
        if (gq >= nq) gq = nq - 1;
        a_src[i] = bank + gn * stride + (slot ^ (r & 7)) * 16;
        b_src[i] = query + gq * stride + (slot ^ (r & 7)) * 16;
    }
    auto stage = [&](int kt, char *buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) gl_glds16(a_src[i] + (int64_t)kt * TILE_K, buf + (wave * 4 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) gl_glds16(b_src[i] + (int64_t)kt * TILE_K, buf + BOPER + (wave * 4 + i) * 1024);
    };

    v4i acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4i){0, 0, 0, 0};

    const int nk = (int)(stride / TILE_K);
    stage(0, smem);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        const char *cur = smem + (kt & 1) * 2 * BOPER;
        if (kt + 1 < nk) stage(kt + 1, smem + ((kt + 1) & 1) * 2 * BOPER);
        const char *la = cur + (wn * 128) * TILE_K;
        const char *lb = cur + BOPER + (wq * 64) * TILE_K;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = ks * 4 + fk;
            v4i a[8], b[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = i * 16 + frow;
                a[i] = *reinterpret_cast<const v4i *>(la + r * TILE_K + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = j * 16 + frow;
                b[j] = *reinterpret_cast<const v4i *>(lb + r * TILE_K + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    const int64_t nbase = n0 + wn * 128 + fk * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t q = q0 + wq * 64 + j * 16 + frow;
        const unsigned qn = q < nq ? (unsigned)query_norm[q] : 0u;
        unsigned long long best = ~0ull;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t n = nbase + i * 16 + r;
                const unsigned bn = n < n_rows ? (unsigned)bank_norm[n] : 0u;
                const unsigned long long s = bn + qn - 2u * (unsigned)acc[i][j][r];       // exact modulo 2^32, and S < 2^32
                const unsigned long long key = (s << shift) | (unsigned long long)(index_base + n);
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

// The same tile on the shared software-pipelined main loop (gl_pair256.h): fragment reads of the next MFMA block are issued before
// the current block, one barrier per slice.
template <int DIAG, int SPREAD>
__global__ void __launch_bounds__(512, 2)
l2_knn_i8_256p_kernel(const int8_t *__restrict__ bank, const int32_t *__restrict__ bank_norm, int64_t n_rows, int64_t index_base,
                      const int8_t *__restrict__ query, const int32_t *__restrict__ query_norm, int64_t nq, int64_t stride,
                      unsigned long long *__restrict__ keys, int q_tiles, int n_tiles, int shift)
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
    const int64_t n0 = (int64_t)nt * BT, q0 = (int64_t)qt * BT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wq = wave & 3;
    const int frow = lane & 15, fk = lane >> 4;

    const gl_pair256::Source sa = gl_pair256::make_source(reinterpret_cast<const char *>(bank), n0, n_rows, stride, wave, lane);
    const gl_pair256::Source sb = gl_pair256::make_source(reinterpret_cast<const char *>(query), q0, nq, stride, wave, lane);
    v4i acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4i){0, 0, 0, 0};
    gl_pair256::mainloop<v4i, DIAG, SPREAD>(sa, sb, stride / TILE_K, smem, acc, wave, lane,
                              [](const v4i &a, const v4i &b, const v4i &c) { return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); });

    const int64_t nbase = n0 + wn * 128 + fk * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t q = q0 + wq * 64 + j * 16 + frow;
        const unsigned qn = q < nq ? (unsigned)query_norm[q] : 0u;
        unsigned long long best = ~0ull;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t n = nbase + i * 16 + r;
                const unsigned bn = n < n_rows ? (unsigned)bank_norm[n] : 0u;
                const unsigned long long s = bn + qn - 2u * (unsigned)acc[i][j][r];       // exact modulo 2^32, and S < 2^32
                const unsigned long long key = (s << shift) | (unsigned long long)(index_base + n);
                if (n < n_rows && key < best) best = key;
            }
        unsigned long long o = __shfl_xor(best, 16, 64);
        best = o < best ? o : best;
        o = __shfl_xor(best, 32, 64);
        best = o < best ? o : best;
        if (fk == 0 && q < nq && best != ~0ull) atomicMin(&keys[q], best);
    }
}

}  // namespace

extern "C" {

int gl_l2_knn_i8(gl_ctx *ctx, const int8_t *bank_i8_dev, const int32_t *bank_norm_dev, int64_t n_rows, int64_t index_base,
                 const int8_t *query_i8_dev, const int32_t *query_norm_dev, int64_t nq, int64_t d, uint64_t *keys_dev)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx, "gl_l2_knn_i8: NULL ctx");
    GL_REQUIRE(n_rows >= 0 && nq >= 0 && d > 0 && d <= GL_L2_MAX_D, "gl_l2_knn_i8: bad sizes n_rows=%lld nq=%lld d=%lld (d <= %lld)", (long long)n_rows,
               (long long)nq, (long long)d, (long long)GL_L2_MAX_D);
    const int shift = gl_l2_key_shift(d);
    GL_REQUIRE(index_base >= 0 && index_base + n_rows <= (1ll << shift), "gl_l2_knn_i8: global index does not fit the %d index bits of a key at d=%lld", shift,
               (long long)d);
    if (n_rows == 0 || nq == 0) return GL_OK;
    GL_REQUIRE(bank_i8_dev && bank_norm_dev && query_i8_dev && query_norm_dev && keys_dev, "gl_l2_knn_i8: NULL device pointer");
    GL_REQUIRE(((reinterpret_cast<uintptr_t>(bank_i8_dev) | reinterpret_cast<uintptr_t>(query_i8_dev)) & 15) == 0,
               "gl_l2_knn_i8: prepared rows must be 16-byte aligned");
    const int64_t stride = gl_l2_row_stride(d);
    const int64_t q_tiles = gl_ceil_div(nq, TILE_Q), n_tiles = gl_ceil_div(n_rows, TILE_N);
    GL_REQUIRE(q_tiles * n_tiles < (1ll << 31), "gl_l2_knn_i8: grid too large");
    const int lds = 4 * OPER_BYTES;
    GL_ONCE_PER_DEVICE(ctx, \
        GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(l2_knn_i8_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(l2_knn_i8_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)););
    gl_prof_scope prof_(ctx, GL_PROF_L2_KNN);
    // the large tile needs enough tiles to fill 256 CUs; GL_L2_TILE=128|256 forces one (tuning / tests)
    const int force_tile = gl_tuning_int("GL_L2_TILE", 0);
    const int64_t q256 = gl_ceil_div(nq, BT), n256 = gl_ceil_div(n_rows, BT);
    if (d <= 66051 && force_tile != 128 && (force_tile == 256 || q256 * n256 >= 1024)) {
#ifdef GL_TUNING
        GL_ONCE_PER_DEVICE(ctx, \
            GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(l2_knn_i8_256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * BOPER)););
#endif
#ifdef GL_TUNING
        // tuning build only: 0 = the round-1 kernel, 2 / 4 = other DMA spreads, 11..14 = timing experiments whose RESULTS ARE WRONG (gl_pair256.h DIAG)
        const int variant = gl_tuning_int("GL_PAIR_VARIANT", 1);
        GL_ONCE_PER_DEVICE(ctx, \
            GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(l2_knn_i8_256p_kernel<0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, gl_pair256::LDS_BYTES)); \
            GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(l2_knn_i8_256p_kernel<0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, gl_pair256::LDS_BYTES)); \
            GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(l2_knn_i8_256p_kernel<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, gl_pair256::LDS_BYTES)); \
            GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(l2_knn_i8_256p_kernel<1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, gl_pair256::LDS_BYTES)); \
            GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(l2_knn_i8_256p_kernel<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, gl_pair256::LDS_BYTES)); \
            GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(l2_knn_i8_256p_kernel<3, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, gl_pair256::LDS_BYTES)); \
            GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(l2_knn_i8_256p_kernel<4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, gl_pair256::LDS_BYTES)););
        auto k256 = variant == 0 ? l2_knn_i8_256_kernel : variant == 11 ? l2_knn_i8_256p_kernel<1, 8> : variant == 12 ? l2_knn_i8_256p_kernel<2, 8>
                    : variant == 13 ? l2_knn_i8_256p_kernel<3, 8> : variant == 14 ? l2_knn_i8_256p_kernel<4, 8> : variant == 4 ? l2_knn_i8_256p_kernel<0, 4> : variant == 2 ? l2_knn_i8_256p_kernel<0, 1>
                    : l2_knn_i8_256p_kernel<0, 8>;
#else
        GL_ONCE_PER_DEVICE(ctx, \
            GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(l2_knn_i8_256p_kernel<0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, gl_pair256::LDS_BYTES)););
        auto k256 = l2_knn_i8_256p_kernel<0, 8>;
#endif
        hipLaunchKernelGGL(k256, dim3((unsigned)(q256 * n256)), dim3(512), 4 * BOPER, ctx->stream, bank_i8_dev, bank_norm_dev, n_rows, index_base,
                           query_i8_dev, query_norm_dev, nq, stride, reinterpret_cast<unsigned long long *>(keys_dev), (int)q256, (int)n256, shift);
        GL_LAUNCH_CHECK();
        return GL_OK;
    }
    auto kern = d > 66051 ? l2_knn_i8_kernel<true> : l2_knn_i8_kernel<false>;      // 65025 * 66051 < 2^32
    hipLaunchKernelGGL(kern, dim3((unsigned)(q_tiles * n_tiles)), dim3(THREADS), lds, ctx->stream, bank_i8_dev, bank_norm_dev, n_rows,
                       index_base, query_i8_dev, query_norm_dev, nq, stride, reinterpret_cast<unsigned long long *>(keys_dev), (int)q_tiles,
                       (int)n_tiles, shift);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int gl_fbb_knn_l2_host(gl_ctx *ctx, const uint8_t *bank_u8_host, int64_t n_bank, const uint8_t *queries_u8_host, int64_t nq, int64_t d,
                       int64_t batch_size, float *dist_host, int64_t *idx_host)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && n_bank >= 0 && nq >= 0 && d > 0 && batch_size > 0, "gl_fbb_knn_l2_host: bad sizes");
    const int64_t n_eff = (n_bank / batch_size) * batch_size;   // attack_models/fbb.py:77
    if (n_eff == 0) {
        gl_set_error("gl_fbb_knn_l2_host: bank of %lld rows holds no full batch of %lld (reference: ValueError at fbb.py:83)", (long long)n_bank,
                     (long long)batch_size);
        return GL_ERR_EMPTY_BANK;
    }
    if (nq == 0) return GL_OK;
    GL_REQUIRE(bank_u8_host && queries_u8_host && dist_host && idx_host, "gl_fbb_knn_l2_host: NULL host pointer");
    const int64_t stride = gl_l2_row_stride(d);
    uint8_t *raw = nullptr;
    int8_t *bank_i8 = nullptr, *q_i8 = nullptr;
    int32_t *bank_nrm = nullptr, *q_nrm = nullptr;
    uint64_t *keys = nullptr;
    float *dist = nullptr;
    int64_t *idx = nullptr;
    int rc = GL_OK;
    const size_t raw_bytes = (size_t)(n_eff > nq ? n_eff : nq) * d;
#define GL_TRY(e) do { rc = (e); if (rc != GL_OK) goto done; } while (0)
    GL_TRY(gl_malloc(ctx, raw_bytes, (void **)&raw));
    GL_TRY(gl_malloc(ctx, (size_t)n_eff * stride, (void **)&bank_i8));
    GL_TRY(gl_malloc(ctx, (size_t)nq * stride, (void **)&q_i8));
    GL_TRY(gl_malloc(ctx, (size_t)n_eff * 4, (void **)&bank_nrm));
    GL_TRY(gl_malloc(ctx, (size_t)nq * 4, (void **)&q_nrm));
    GL_TRY(gl_malloc(ctx, (size_t)nq * 8, (void **)&keys));
    GL_TRY(gl_malloc(ctx, (size_t)nq * 4, (void **)&dist));
    GL_TRY(gl_malloc(ctx, (size_t)nq * 8, (void **)&idx));
    GL_TRY(gl_memcpy_h2d(ctx, raw, bank_u8_host, (size_t)n_eff * d));
    GL_TRY(gl_l2_prepare(ctx, raw, n_eff, d, bank_i8, bank_nrm));
    GL_TRY(gl_ctx_sync(ctx));
    GL_TRY(gl_memcpy_h2d(ctx, raw, queries_u8_host, (size_t)nq * d));
    GL_TRY(gl_l2_prepare(ctx, raw, nq, d, q_i8, q_nrm));
    GL_TRY(gl_keys_init(ctx, keys, nq));
    GL_TRY(gl_l2_knn_i8(ctx, bank_i8, bank_nrm, n_eff, 0, q_i8, q_nrm, nq, d, keys));
    GL_TRY(gl_keys_unpack(ctx, keys, nq, d, dist, idx));
    GL_TRY(gl_memcpy_d2h(ctx, dist_host, dist, (size_t)nq * 4));
    GL_TRY(gl_memcpy_d2h(ctx, idx_host, idx, (size_t)nq * 8));
#undef GL_TRY
done:
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(raw); (void)hipFree(bank_i8); (void)hipFree(q_i8); (void)hipFree(bank_nrm);
    (void)hipFree(q_nrm); (void)hipFree(keys); (void)hipFree(dist); (void)hipFree(idx);
    return rc;
}

}  // extern "C"
