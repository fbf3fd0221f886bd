// TUNING BUILD ONLY (linked into libganleaks_hip_tuning.so, never into libganleaks_hip.so; not part of include/ganleaks.h).
//
// gl_tune_winograd_bound: the measured UPPER bound of what Winograd F(2x2,3x3) could buy on the split-fp16 convolution path (VERDICT r2
// "Next 1").  F(2x2,3x3) replaces a 3x3 convolution over P positions by 16 independent GEMMs  M_j = U_j (C_out x C_in) . V_j (C_in x P/4)
// plus an input transform (4 C bytes / position read, 16 C written) and an output transform (16 C_out read, 4 C_out written).  The GEMM stage
// is exactly what the existing tap-gather kernel runs for a ONE-tap convolution, so it can be timed today with the tuned kernel:
//     ms[0] = the direct 3x3 split-fp16 convolution (gl_launch_gather_conv_h3, 9 taps) on n images of H x W, C_in -> C_out, split output
//     ms[1] = the 16 Winograd GEMMs: 16 launches of the same kernel with 1 tap, P/4 rows each, own weights, fp32 output (the accumulators M_j)
// on random operands (the clock the chip holds depends on the data).  Everything a real Winograd kernel adds -- the two transforms, and for a
// fused kernel the small per-component tile (the 16 accumulator sets of a tile share one register file: 64 x 64 per component and CU instead
// of 256 x 256, i.e. 4 x the staging bytes per MFMA) -- comes on top of ms[1].
#include "gl_conv.h"
#include <cmath>
#include <vector>

namespace {

__global__ void __launch_bounds__(256) tune_fill_split_kernel(char *__restrict__ out, int64_t chunks, unsigned seed)
{
    // one thread = one 128-byte chunk: 32 hi halves + 32 lo halves of ReLU-like values (half of them zero), |lo| <= 2^-11 |hi|
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned s = seed ^ (unsigned)(i * 2654435761u);
        _Float16 *hi = reinterpret_cast<_Float16 *>(out + i * 128), *lo = hi + 32;
        for (int k = 0; k < 32; ++k) {
            s = s * 1664525u + 1013904223u;
            const float u = (float)((s >> 9) & 0x7FFF) / 32768.0f;          // [0, 1)
            const float v = (s & 0x100u) ? 0.0f : u * 8.0f;
            s = s * 1664525u + 1013904223u;
            const float l = v * ((float)((s >> 9) & 0x7FFF) / 32768.0f - 0.5f) * (1.0f / 2048.0f);
            hi[k] = (_Float16)v;
            lo[k] = (_Float16)l;
        }
    }
}

}  // namespace

extern "C" int gl_tune_winograd_bound(gl_ctx *ctx, int Cin, int Cout, int H, int W, int64_t n_img, int reps, float *ms_out)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && ms_out && Cin % 32 == 0 && Cout % 32 == 0 && H % 2 == 0 && W % 2 == 0 && n_img > 0 && reps > 0, "gl_tune_winograd_bound: bad argument");
    const int64_t P = n_img * H * W, T = P / 4;
    const int cols_pad = (int)gl_ceil_div(Cout, 128) * 128;
    char *in = nullptr, *V = nullptr;
    float *out = nullptr, *M = nullptr, *w9 = nullptr, *w1 = nullptr, *ones = nullptr, *zeros = nullptr;
    int rc = GL_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(in); (void)hipFree(V); (void)hipFree(out); (void)hipFree(M); (void)hipFree(w9); (void)hipFree(w1); (void)hipFree(ones); (void)hipFree(zeros);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
#define TUNE_HIP(e) do { if ((e) != hipSuccess) { gl_set_error("gl_tune_winograd_bound: %s failed", #e); cleanup(); return GL_ERR_HIP; } } while (0)
    TUNE_HIP(hipMalloc((void **)&in, (size_t)P * Cin * 4));
    TUNE_HIP(hipMalloc((void **)&V, (size_t)16 * T * Cin * 4));
    TUNE_HIP(hipMalloc((void **)&out, (size_t)P * Cout * 4));
    TUNE_HIP(hipMalloc((void **)&M, (size_t)16 * T * Cout * 4));
    TUNE_HIP(hipMalloc((void **)&w9, (size_t)cols_pad * 9 * Cin * 4));
    TUNE_HIP(hipMalloc((void **)&w1, (size_t)16 * cols_pad * Cin * 4));
    TUNE_HIP(hipMalloc((void **)&ones, (size_t)Cout * 4));
    TUNE_HIP(hipMalloc((void **)&zeros, (size_t)Cout * 4));
    TUNE_HIP(hipEventCreate(&e0));
    TUNE_HIP(hipEventCreate(&e1));
    {
        std::vector<float> w((size_t)cols_pad * 9 * Cin), sp(w.size()), one(Cout, 1.0f), zero(Cout, 0.0f);
        unsigned s = 12345u;
        auto gauss = [&]() { float a = 0; for (int k = 0; k < 4; ++k) { s = s * 1664525u + 1013904223u; a += (float)(s >> 8) / 16777216.0f; } return (a - 2.0f) * 1.7f; };
        for (float &v : w) v = gauss() * 4096.0f;                       // already at the packed scale (max ~ 2^13)
        for (size_t r = (size_t)Cout * 9 * Cin; r < w.size(); ++r) w[r] = 0.0f;
        gl_split_weights_host(w.data(), cols_pad, (size_t)9 * Cin, 1.0f, sp.data());
        TUNE_HIP(hipMemcpy(w9, sp.data(), sp.size() * 4, hipMemcpyHostToDevice));
        for (int j = 0; j < 16; ++j) {
            std::vector<float> u((size_t)cols_pad * Cin, 0.0f), us(u.size());
            for (size_t r = 0; r < (size_t)Cout * Cin; ++r) u[r] = gauss() * 4096.0f;
            gl_split_weights_host(u.data(), cols_pad, (size_t)Cin, 1.0f, us.data());
            TUNE_HIP(hipMemcpy(w1 + (size_t)j * cols_pad * Cin, us.data(), us.size() * 4, hipMemcpyHostToDevice));
        }
        TUNE_HIP(hipMemcpy(ones, one.data(), Cout * 4, hipMemcpyHostToDevice));
        TUNE_HIP(hipMemcpy(zeros, zero.data(), Cout * 4, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(tune_fill_split_kernel, dim3(4096), dim3(256), 0, ctx->stream, in, P * Cin / 32, 1u);
    hipLaunchKernelGGL(tune_fill_split_kernel, dim3(4096), dim3(256), 0, ctx->stream, V, 16 * T * Cin / 32, 2u);
    TUNE_HIP(hipGetLastError());

    GlGatherConv d = {};
    d.in = reinterpret_cast<const float *>(in); d.positions = P; d.H = H; d.W = W; d.Cin = Cin; d.ntaps = 9;
    {
        uint32_t dy = 0, dx = 0;
        for (int t = 0; t < 9; ++t) { dy |= (uint32_t)(t / 3) << (2 * t); dx |= (uint32_t)(t % 3) << (2 * t); }
        d.tap_dy[0] = dy; d.tap_dx[0] = dx;
    }
    d.wpack = w9; d.cols = Cout; d.cols_pad = cols_pad;
    d.out = out; d.Ho = H; d.Wo = W; d.omul = 1; d.out_mode = 2;
    d.scale = ones; d.shift = zeros; d.cmod = Cout; d.act = 1; d.zero = ctx->zero_page;
    GlGatherConv g = d;
    g.positions = T; g.H = 1; g.W = 1; g.Ho = 1; g.Wo = 1; g.ntaps = 1; g.tap_dy[0] = 1; g.tap_dx[0] = 1;       // one tap at (0, 0)
    g.out_mode = 0; g.act = 0;

    auto run_direct = [&]() { return gl_launch_gather_conv_h3(ctx, d, 1); };
    auto run_gemms = [&]() {
        for (int j = 0; j < 16; ++j) {
            GlGatherConv q = g;
            q.in = reinterpret_cast<const float *>(V + (size_t)j * T * Cin * 4);
            q.wpack = w1 + (size_t)j * cols_pad * Cin;
            q.out = M + (size_t)j * T * Cout;
            const int r = gl_launch_gather_conv_h3(ctx, q, 1);
            if (r != GL_OK) return r;
        }
        return (int)GL_OK;
    };
    for (int which = 0; which < 2 && rc == GL_OK; ++which) {
        rc = which ? run_gemms() : run_direct();                     // warm-up
        if (rc != GL_OK) break;
        TUNE_HIP(hipEventRecord(e0, ctx->stream));
        for (int r = 0; r < reps && rc == GL_OK; ++r) rc = which ? run_gemms() : run_direct();
        TUNE_HIP(hipEventRecord(e1, ctx->stream));
        TUNE_HIP(hipEventSynchronize(e1));
        float ms = 0.0f;
        TUNE_HIP(hipEventElapsedTime(&ms, e0, e1));
        ms_out[which] = ms / reps;
    }
#undef TUNE_HIP
    cleanup();
    return rc;
}
