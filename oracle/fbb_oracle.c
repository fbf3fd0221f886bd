/* CPU ORACLE (C restatement) -- TEST INFRASTRUCTURE ONLY, never linked into the product.
 *
 * Restates the per-query nearest-neighbour search of the reference for images on the 8-bit
 * lattice, in exact integer arithmetic:
 *   attack_models/fbb.py:73-88    custom_knn  (loop over full batches, concatenate, torch.min,
 *                                 first index on ties)
 *   attack_models/utils.py:163    L2 term  mean((y - x)^2, dim=[1,2,3])
 *   attack_models/utils.py:82     pixels are 2*(u/255)-1 with u in 0..255, so
 *                                 mean((y-x)^2) = (4/255^2/D) * sum (uy-ux)^2  exactly.
 * The caller passes the already truncated bank length n_eff = (N / BATCH_SIZE) * BATCH_SIZE
 * (fbb.py:77).  Checked against tests/golden/knn_*.npz (made by the reference's custom_knn) in
 * tests/test_oracle.py.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -shared).
 */
#include <stdint.h>
#include <stddef.h>

#if defined(__x86_64__)
#define CLONES __attribute__((target_clones("default", "avx2", "arch=skylake-avx512")))
#else
#define CLONES
#endif

CLONES
int64_t gl_oracle_ssd_pair(const uint8_t *a, const uint8_t *b, int64_t d)
{
    int64_t total = 0;
    int64_t k = 0;
    /* int32 partial sums over chunks of <= 32768 elements cannot overflow (255^2*32768 < 2^31) */
    while (k < d) {
        int64_t end = k + 32768 < d ? k + 32768 : d;
        int32_t s = 0;
        for (int64_t j = k; j < end; ++j) {
            int32_t t = (int32_t)a[j] - (int32_t)b[j];
            s += t * t;
        }
        total += s;
        k = end;
    }
    return total;
}

/* out_idx[q], out_ssd[q]: first index of the minimum SSD over bank rows [0, n_eff). returns 0. */
int gl_oracle_knn_l2_u8(const uint8_t *bank, int64_t n_eff, const uint8_t *queries, int64_t nq,
                        int64_t d, int64_t *out_idx, int64_t *out_ssd)
{
    if (n_eff <= 0 || d <= 0) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t q = 0; q < nq; ++q) {
        const uint8_t *qa = queries + q * d;
        int64_t best = INT64_MAX, besti = 0;
        for (int64_t n = 0; n < n_eff; ++n) {
            int64_t s = gl_oracle_ssd_pair(qa, bank + n * d, d);
            if (s < best) { best = s; besti = n; }   /* strict <: first occurrence wins */
        }
        out_idx[q] = besti;
        out_ssd[q] = best;
    }
    return 0;
}

/* all SSDs of one query (for spot checks of full rows) */
int gl_oracle_ssd_row_u8(const uint8_t *bank, int64_t n, const uint8_t *query, int64_t d, int64_t *out)
{
#pragma omp parallel for
    for (int64_t i = 0; i < n; ++i) out[i] = gl_oracle_ssd_pair(query, bank + i * d, d);
    return 0;
}

/* sum of squares of (u - 128) per row: the norm the device path precomputes */
int gl_oracle_row_norms_u8(const uint8_t *x, int64_t n, int64_t d, int32_t *out)
{
#pragma omp parallel for
    for (int64_t i = 0; i < n; ++i) {
        int64_t s = 0;
        for (int64_t k = 0; k < d; ++k) { int32_t t = (int32_t)x[i * d + k] - 128; s += t * t; }
        out[i] = (int32_t)s;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * General fp32 images (not on the 8-bit lattice).  attack_models/utils.py:163 computes
 * mean((y - x)**2) in fp32 with a reduction order that depends on the torch build; this oracle and
 * the device path (gan-leaks_amd/csrc/gl_l2f32.hip) fix one order and share it bit for bit:
 *   d_k = fl32(y_k - x_k);  c_j = fmaf chain over k = j, j+4, ... (j = 0..3);
 *   dist = fl32(fl32(fl32(c0 + c1) + fl32(c2 + c3)) / D)
 * argmin over fl32 dist, first index on ties (fbb.py:86).  Within ~1e-6 of any other fp32 order.
 * ------------------------------------------------------------------------------------------------ */
#include <math.h>

#if defined(__x86_64__)
#define CLONES_FMA __attribute__((target_clones("default", "arch=haswell", "arch=skylake-avx512")))
#else
#define CLONES_FMA
#endif

CLONES_FMA
float gl_oracle_l2_pair_f32(const float *y, const float *x, int64_t d)
{
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t k = 0; k < d; ++k) {
        const float t = y[k] - x[k];
        c[k & 3] = fmaf(t, t, c[k & 3]);
    }
    const float s01 = c[0] + c[1], s23 = c[2] + c[3];
    const float s = s01 + s23;
    return s / (float)d;
}

int gl_oracle_knn_l2_f32(const float *bank, int64_t n_eff, const float *queries, int64_t nq, int64_t d, int64_t *out_idx, float *out_dist)
{
    if (n_eff <= 0 || d <= 0) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t q = 0; q < nq; ++q) {
        const float *qa = queries + q * d;
        float best = INFINITY;
        int64_t besti = 0;
        for (int64_t n = 0; n < n_eff; ++n) {
            const float s = gl_oracle_l2_pair_f32(qa, bank + n * d, d);
            if (s < best) { best = s; besti = n; }
        }
        out_idx[q] = besti;
        out_dist[q] = best;
    }
    return 0;
}
