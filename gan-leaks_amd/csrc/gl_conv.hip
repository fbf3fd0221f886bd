// fp32 matrix-core convolution kernels of the generator stack
// (gan_models/dcgan/model_torch.py:78-86 == gan_models/wgangp/model.py:40-48).
//
// gather_conv_kernel: C[position][channel] = sum_{tap,ci} in[pos + tap][ci] * W[channel][tap*Cin+ci]
//   - v_mfma_f32_32x32x2_f32: exact fp32 products and fp32 accumulation, i.e. the same arithmetic
//     class as the reference's fp32 convolution (only the summation order differs).
//   - workgroup tile 128 positions x 128 channels, 4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles
//     (64 accumulator VGPRs); K in slices of 32 floats (128 B), double buffered in LDS (64 KiB).
//   - both operands are staged with global_load_lds (16 B/lane).  The A operand is a GATHER: each
//     lane's source address is its position's pixel shifted by the slice's tap, or the context's
//     zero page when the tap falls outside the image (padding costs no branch and no LDS write).
//   - LDS rows are 128 B; chunk c of row r sits at slot c ^ ((r >> 1) & 7) (applied to the source
//     address) so that the 32-row ds_read_b128 operand fetch is bank-conflict free.
//   - one ds_read_b128 per lane feeds FOUR MFMAs: lane (row, h) holds k = 4h..4h+3 of an 8-float
//     group and MFMA step s multiplies the k-pair {s, 4+s}; A and B use the same permutation, so the
//     result is the plain dot product summed in a fixed order.
//   - epilogue fuses BatchNorm(eval) as scale/shift and ReLU, and scatters each sub-pixel phase to
//     its interleaved output position, NHWC.
#include "gl_conv.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int THREADS = 256;
constexpr int OPER_BYTES = BM * BK * 4;   // 16 KiB

__device__ __forceinline__ int swz(int r) { return (r >> 1) & 7; }

__global__ void __launch_bounds__(THREADS, 2) gather_conv_kernel(const GlGatherConv p, int m_tiles, int n_tiles)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A 16 KiB | B 16 KiB]; reused for output rows

    const int phase = blockIdx.y;
    const unsigned nwg = (unsigned)m_tiles * (unsigned)n_tiles;
    const unsigned id = gl_xcd_remap(blockIdx.x, nwg);
    const int nt = (int)(id % (unsigned)n_tiles);   // neighbours share the activation panel
    const int mt = (int)(id / (unsigned)n_tiles);
    const int64_t m0 = (int64_t)mt * BM;
    const int c0 = nt * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int K = p.ntaps * p.Cin;
    const int nk = K / BK;
    const int HW = p.H * p.W;
    const uint32_t tdy = p.tap_dy[phase], tdx = p.tap_dx[phase];
    const float *__restrict__ wp = p.wpack + (int64_t)phase * p.cols_pad * K;

    // ---- per-thread staging bookkeeping: 4 A rows + 4 B rows, one 16-B chunk each
    const int rsub = lane >> 3, slot = lane & 7;
    int64_t a_off[4];     // element offset of (img, y, x, 0)
    int a_y[4], a_x[4];   // -1000000 marks a row beyond `positions`
    int a_chunk[4];
    const float *b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + rsub;
        const int chunk = slot ^ swz(r);
        a_chunk[i] = chunk * 4;
        const int64_t pos = m0 + r;
        if (pos < p.positions) {
            const int64_t img = pos / HW;
            const int rem = (int)(pos - img * HW);
            a_y[i] = rem / p.W;
            a_x[i] = rem - a_y[i] * p.W;
            a_off[i] = pos * p.Cin;
        } else {
            a_y[i] = -1000000;
            a_x[i] = 0;
            a_off[i] = 0;
        }
        b_src[i] = wp + (int64_t)(c0 + r) * K + chunk * 4;
    }

    auto stage = [&](int kt, char *buf) {
        const int k0 = kt * BK;
        const int tap = k0 / p.Cin;
        const int ci0 = k0 - tap * p.Cin;
        const int dy = (int)((tdy >> (2 * tap)) & 3u) - 1;
        const int dx = (int)((tdx >> (2 * tap)) & 3u) - 1;
        const int64_t shift = ((int64_t)dy * p.W + dx) * p.Cin + ci0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = a_y[i] + dy, xx = a_x[i] + dx;
            const bool ok = (yy >= 0) & (yy < p.H) & (xx >= 0) & (xx < p.W);
            const float *src = ok ? p.in + a_off[i] + shift + a_chunk[i] : p.zero;
            gl_glds16(src, buf + (wave * 4 + i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) gl_glds16(b_src[i] + k0, buf + OPER_BYTES + (wave * 4 + i) * 1024);
    };

    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    stage(0, smem);
    const int frow = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();   // slice kt has landed (vmcnt(0)) and nobody still reads the other buffer
        char *cur = smem + (kt & 1) * 2 * OPER_BYTES;
        if (kt + 1 < nk) stage(kt + 1, smem + ((kt + 1) & 1) * 2 * OPER_BYTES);
        const char *la = cur + (wm * 64) * (BK * 4);
        const char *lb = cur + OPER_BYTES + (wn * 64) * (BK * 4);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int chunk = 2 * g + fh;
            v4f a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = i * 32 + frow;
                a[i] = *reinterpret_cast<const v4f *>(la + r * (BK * 4) + ((chunk ^ swz(r)) << 4));
                b[i] = *reinterpret_cast<const v4f *>(lb + r * (BK * 4) + ((chunk ^ swz(r)) << 4));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue.  Output row index of every tile position, computed once into LDS.
    __syncthreads();
    int64_t *orow = reinterpret_cast<int64_t *>(smem);
    if (tid < BM) {
        const int64_t pos = m0 + tid;
        int64_t o = -1;
        if (pos < p.positions) {
            const int64_t img = pos / HW;
            const int rem = (int)(pos - img * HW);
            const int y = rem / p.W, x = rem - y * p.W;
            o = (img * p.Ho + (y * p.omul + p.oy[phase])) * p.Wo + (x * p.omul + p.ox[phase]);
        }
        orow[tid] = o;
    }
    __syncthreads();
    // C layout (32x32): column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = c0 + wn * 64 + j * 32 + frow;
        if (c >= p.cols) continue;
        const int cm = c % p.cmod;
        const float sc = p.scale[cm], sh = p.shift[cm];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int64_t o = orow[row];
                if (o < 0) continue;
                float v = fmaf(acc[i][j][r], sc, sh);
                if (p.act == 1) v = fmaxf(v, 0.0f);
                p.out[o * p.cols + c] = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// ConvTranspose2d(Cin -> 3, k4, s2, p1) + bias + tanh (+ quantise).  Three output channels cannot
// fill a matrix-core tile, so this layer runs on the vector ALUs: one thread owns the 2 x 2 output
// pixels of one input position (all four sub-pixel phases, 12 accumulators) and walks the 3 x 3
// input neighbourhood; the 48 weights of each input channel are wave-uniform (scalar loads).
//   out(2y+py, 2x+px) = sum over neighbours (dy,dx) with ky = KY[dy][py], kx = KY[dx][px]
//   KY: (dy=-1,py=0)->3  (0,0)->1  (0,1)->2  (+1,1)->0
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t quantize_byte(float x)
{
    // gan_models/dcgan/train_torch.py:154-158,172: (x+1)/2, *255, truncate
    float t = __fmul_rn(__fdiv_rn(__fadd_rn(x, 1.0f), 2.0f), 255.0f);
    t = fminf(fmaxf(truncf(t), 0.0f), 255.0f);
    return (uint32_t)(int)t;
}

__global__ void __launch_bounds__(THREADS) convt_rgb_tanh_kernel(const float *__restrict__ in, int64_t n_img, int H, int W, int Cin,
                                                                  const float *__restrict__ w, const float *__restrict__ bias,
                                                                  float *__restrict__ out_f32, uint8_t *__restrict__ out_u8)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = n_img * H * W;
    if (gid >= total) return;
    const int64_t img = gid / (H * W);
    const int rem = (int)(gid - img * (H * W));
    const int y = rem / W, x = rem - y * W;

    float acc[2][2][3];   // [py][px][co]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[a][b][c] = 0.0f;

    const float *nb[3][3];
    bool ok[3][3];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            ok[dy + 1][dx + 1] = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W);
            nb[dy + 1][dx + 1] = in + ((img * H + yy) * W + xx) * (int64_t)Cin;
        }

    for (int ci = 0; ci < Cin; ci += 4) {
        float4 v[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                v[a][b] = ok[a][b] ? *reinterpret_cast<const float4 *>(nb[a][b] + ci) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float *wc = w + (int64_t)(ci + u) * 48;   // [co][ky][kx], wave-uniform
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const float val = u == 0 ? v[a][b].x : u == 1 ? v[a][b].y : u == 2 ? v[a][b].z : v[a][b].w;
#pragma unroll
                    for (int py = 0; py < 2; ++py) {
                        // dy = a-1 contributes to phase py iff (dy,py) in {(-1,0),(0,0),(0,1),(1,1)}
                        if ((a == 0 && py == 1) || (a == 2 && py == 0)) continue;
                        const int ky = a == 0 ? 3 : (a == 2 ? 0 : (py == 0 ? 1 : 2));
#pragma unroll
                        for (int px = 0; px < 2; ++px) {
                            if ((b == 0 && px == 1) || (b == 2 && px == 0)) continue;
                            const int kx = b == 0 ? 3 : (b == 2 ? 0 : (px == 0 ? 1 : 2));
#pragma unroll
                            for (int co = 0; co < 3; ++co) acc[py][px][co] = fmaf(val, wc[co * 16 + ky * 4 + kx], acc[py][px][co]);
                        }
                    }
                }
        }
    }

    const int Ho = 2 * H, Wo = 2 * W;
#pragma unroll
    for (int co = 0; co < 3; ++co) {
        const float bco = bias[co];
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            const float t0 = tanhf(acc[py][0][co] + bco), t1 = tanhf(acc[py][1][co] + bco);
            const int64_t o = ((img * 3 + co) * Ho + (2 * y + py)) * (int64_t)Wo + 2 * x;
            if (out_f32) *reinterpret_cast<float2 *>(out_f32 + o) = make_float2(t0, t1);
            if (out_u8) *reinterpret_cast<uint16_t *>(out_u8 + o) = (uint16_t)(quantize_byte(t0) | (quantize_byte(t1) << 8));
        }
    }
}

}  // namespace

int gl_launch_gather_conv(gl_ctx *ctx, const GlGatherConv &p, int phases)
{
    GL_REQUIRE(p.Cin % BK == 0, "gather_conv: Cin=%d must be a multiple of %d", p.Cin, BK);
    GL_REQUIRE(p.cols_pad % BN == 0 && p.cols <= p.cols_pad, "gather_conv: cols_pad=%d must be a multiple of %d", p.cols_pad, BN);
    GL_REQUIRE(phases >= 1 && phases <= 4 && p.ntaps >= 1 && p.ntaps <= 16, "gather_conv: bad phases/taps");
    GL_REQUIRE((reinterpret_cast<uintptr_t>(p.in) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.wpack) & 15) == 0, "gather_conv: unaligned operand");
    if (p.positions == 0) return GL_OK;
    const int64_t m_tiles = gl_ceil_div(p.positions, BM);
    const int n_tiles = p.cols_pad / BN;
    GL_REQUIRE(m_tiles * n_tiles < (1ll << 31), "gather_conv: grid too large");
    static bool attr_set = false;
    const int lds = 4 * OPER_BYTES;
    if (!attr_set) {
        GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gather_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    gl_prof_scope prof_(ctx, GL_PROF_GATHER_CONV);
    hipLaunchKernelGGL(gather_conv_kernel, dim3((unsigned)(m_tiles * n_tiles), phases), dim3(THREADS), lds, ctx->stream, p, (int)m_tiles, n_tiles);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int gl_launch_convt_rgb_tanh(gl_ctx *ctx, const float *in, int64_t n_img, int H, int W, int Cin, const float *w, const float *bias,
                             float *out_f32, uint8_t *out_u8)
{
    GL_REQUIRE(Cin % 4 == 0, "convt_rgb: Cin must be a multiple of 4");
    if (n_img == 0) return GL_OK;
    const int64_t total = n_img * H * W;
    gl_prof_scope prof_(ctx, GL_PROF_CONVT_RGB);
    hipLaunchKernelGGL(convt_rgb_tanh_kernel, dim3((unsigned)gl_ceil_div(total, THREADS)), dim3(THREADS), 0, ctx->stream, in, n_img, H, W, Cin, w,
                       bias, out_f32, out_u8);
    GL_LAUNCH_CHECK();
    return GL_OK;
}
