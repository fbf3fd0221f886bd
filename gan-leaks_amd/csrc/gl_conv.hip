// fp32 matrix-core convolution kernels of the generator stack
// (gan_models/dcgan/model_torch.py:78-86 == gan_models/wgangp/model.py:40-48).
//
// gather_conv_kernel: C[position][channel] = sum_{tap,ci} in[pos + tap][ci] * W[channel][k(tap,ci)]  (k: gl_conv.h)
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
#include <cstdlib>
#include <type_traits>

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int BK = 32;        // floats of K per slice (128 B rows in LDS)
constexpr int THREADS = 256;  // 4 waves

__device__ __forceinline__ int swz(int r) { return (r >> 1) & 7; }

// WAVES_M x WAVES_N waves (product 4), each owning TM x TN MFMA tiles of 32 x 32.
//   <2,2,2,2>: 128 positions x 128 channels per workgroup (generator layers 0-3)
//   <4,1,1,2>: 128 positions x  64 channels (narrow outputs: the 48-column RGB tail)
template <int WAVES_M, int WAVES_N, int TM, int TN>
__global__ void __launch_bounds__(THREADS, 2) gather_conv_kernel(const GlGatherConv p, int m_tiles, int n_tiles, int phases)
{
#if __HIP_DEVICE_COMPILE__   // the body uses device-only types (__amdgpu_buffer_rsrc_t); the host pass only needs the launch stub
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 4, BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int A_PER = BM / 32, B_PER = BN / 32;   // 16-B chunks (= 8-row pieces) each thread / wave stages per slice
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A | B]; reused for output rows

    // logical block order: (phase, channel tile) fastest, position tile slowest -> the blocks that read the same
    // activation panel (all phases x channel tiles of one position tile) are adjacent and, after the XCD remap,
    // run together on one XCD: the panel is fetched into that L2 once.
    const unsigned inner = (unsigned)phases * (unsigned)n_tiles;
    const unsigned id = gl_xcd_remap(blockIdx.x, (unsigned)m_tiles * inner);
    const int mt = (int)(id / inner);
    const int phase = (int)((id % inner) / (unsigned)n_tiles);
    const int nt = (int)(id % (unsigned)n_tiles);
    const int64_t m0 = (int64_t)mt * BM;
    const int c0 = nt * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int K = p.ntaps * p.Cin;
    const int nk = K / BK;
    const int HW = p.H * p.W;
    const uint32_t tdy = p.tap_dy[phase], tdx = p.tap_dx[phase];
    const float *__restrict__ wp = p.wpack + (int64_t)phase * p.cols_pad * K;

    // ---- staging through buffer_load ... lds (LDS-DMA with a buffer descriptor):
    //   * the per-slice part of every address (tap shift, channel chunk, K offset) is wave-uniform and goes into
    //     the instruction's scalar offset: no vector ALU work per slice for the weights, 3 ops per activation piece;
    //   * a lane whose total offset is >= num_records gets ZEROS written to LDS by the hardware (checked on MI355X):
    //     that is the convolution's zero padding and the ragged last position tile, with no branch and no zero page.
    // The activation descriptor's base sits (W+1) pixels BEFORE the tensor so that the most negative tap shift still
    // gives a non-negative scalar offset; in-image taps always land inside the real tensor.
    const int rsub = lane >> 3, slot = lane & 7;
    const int up = p.up;          // 1: the input grid is the nearest-neighbour 2x upsampling of the stored tensor
    const int Hs = p.H >> up, Ws = p.W >> up;
    constexpr unsigned kOOB = 0xC0000000u;     // voffset of a lane that must read zeros (host checks sizes < 0xC0000000)
    const unsigned lead_bytes = (unsigned)(p.W + 1) * (unsigned)p.Cin * 4u;
    const __amdgpu_buffer_rsrc_t a_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.in)) - (up ? 0 : (int64_t)lead_bytes), 0,
                                          (int)(p.in_bytes + (up ? 0u : 2u * lead_bytes)), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wp), 0, (int)((unsigned)p.cols_pad * (unsigned)K * 4u), 0x00020000);
    unsigned a_voff[A_PER];       // up == 0: byte offset of (img, y, x, chunk) from the shifted base
    unsigned a_mask[A_PER];       // bit t: tap t of this row falls inside the image
    int a_img[A_PER], a_y[A_PER], a_x[A_PER], a_chunk4[A_PER];   // up == 1 only: the source pixel depends on the tap
    unsigned b_voff[B_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int r = (wave * A_PER + i) * 8 + rsub;
        const int chunk = slot ^ swz(r);
        const int64_t pos = m0 + r;
        a_mask[i] = 0;
        a_voff[i] = kOOB;
        a_img[i] = a_y[i] = a_x[i] = 0;
        a_chunk4[i] = chunk * 16;
        if (pos < p.positions) {
            const int64_t img = pos / HW;
            const int rem = (int)(pos - img * HW);
            const int y = rem / p.W, x = rem - y * p.W;
            for (int t = 0; t < p.ntaps; ++t) {
                const int yy = y + (int)((tdy >> (2 * t)) & 3u) - 1, xx = x + (int)((tdx >> (2 * t)) & 3u) - 1;
                if ((yy >= 0) & (yy < p.H) & (xx >= 0) & (xx < p.W)) a_mask[i] |= 1u << t;
            }
            a_voff[i] = (unsigned)pos * (unsigned)p.Cin * 4u + (unsigned)chunk * 16u;
            a_img[i] = (int)(img * Hs * Ws);
            a_y[i] = y;
            a_x[i] = x;
        }
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int r = (wave * B_PER + i) * 8 + rsub;
        b_voff[i] = ((unsigned)(c0 + r) * (unsigned)K + (unsigned)((slot ^ swz(r)) * 4)) * 4u;
    }

    auto stage = [&](int kt, char *buf) {
        // K order: taps innermost -- 32-channel chunk s32 / ntaps, tap s32 % ntaps (gl_conv_k_index) -- so consecutive
        // slices re-read the same pixels shifted by one tap while they are still in L2
        const int k0 = kt * BK;
        const int tap = kt % p.ntaps;
        const int ci0 = (kt / p.ntaps) * BK;
        const int dy = (int)((tdy >> (2 * tap)) & 3u) - 1;
        const int dx = (int)((tdx >> (2 * tap)) & 3u) - 1;
        const unsigned tapbit = 1u << tap;
        if (!up) {
            const unsigned soff = (unsigned)(((dy + 1) * p.W + (dx + 1)) * p.Cin + ci0) * 4u;   // >= 0 thanks to the shifted base
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const unsigned voff = (a_mask[i] & tapbit) ? a_voff[i] : kOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (gl_lptr)(buf + (wave * A_PER + i) * 1024), 16, voff, soff, 0, 0);
            }
        } else {
            const unsigned soff = (unsigned)ci0 * 4u;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const int yy = a_y[i] + dy, xx = a_x[i] + dx;
                const unsigned pix = (unsigned)(a_img[i] + (yy >> 1) * Ws + (xx >> 1));
                const unsigned voff = (a_mask[i] & tapbit) ? pix * (unsigned)p.Cin * 4u + (unsigned)a_chunk4[i] : kOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (gl_lptr)(buf + (wave * A_PER + i) * 1024), 16, voff, soff, 0, 0);
            }
        }
        const unsigned soff_b = (unsigned)k0 * 4u;
#pragma unroll
        for (int i = 0; i < B_PER; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (gl_lptr)(buf + A_BYTES + (wave * B_PER + i) * 1024), 16, b_voff[i], soff_b, 0, 0);
    };

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    stage(0, smem);
    const int frow = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();   // slice kt has landed (vmcnt(0)) and nobody still reads the other buffer
        char *cur = smem + (kt & 1) * BUF_BYTES;
        if (kt + 1 < nk) stage(kt + 1, smem + ((kt + 1) & 1) * BUF_BYTES);
        const char *la = cur + (wm * TM * 32) * (BK * 4);
        const char *lb = cur + A_BYTES + (wn * TN * 32) * (BK * 4);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int chunk = 2 * g + fh;
            v4f a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = i * 32 + frow;
                a[i] = *reinterpret_cast<const v4f *>(la + r * (BK * 4) + ((chunk ^ swz(r)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = j * 32 + frow;
                b[j] = *reinterpret_cast<const v4f *>(lb + r * (BK * 4) + ((chunk ^ swz(r)) << 4));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue.  Output row index of every tile position, computed once into LDS, then pulled into registers
    // BEFORE the first store: with stores in flight hipcc drains vmcnt(0) ahead of every LDS read (it cannot tell
    // them from pending LDS-DMA), which would serialise the whole store tail.
    __syncthreads();
    int *orow = reinterpret_cast<int *>(smem);
    if (tid < BM) {
        const int64_t pos = m0 + tid;
        int o = -1;
        if (pos < p.positions) {
            const int64_t img = pos / HW;
            const int rem = (int)(pos - img * HW);
            const int y = rem / p.W, x = rem - y * p.W;
            o = (int)((img * p.Ho + (y * p.omul + p.oy[phase])) * p.Wo + (x * p.omul + p.ox[phase]));
        }
        orow[tid] = o;
    }
    __syncthreads();
    // C layout (32x32): column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    int o32[TM][16];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int4 v = *reinterpret_cast<const int4 *>(&orow[(wm * TM + i) * 32 + 8 * g + 4 * fh]);
            o32[i][4 * g + 0] = v.x; o32[i][4 * g + 1] = v.y; o32[i][4 * g + 2] = v.z; o32[i][4 * g + 3] = v.w;
        }
    // branch-free store tail for full tiles (per-store branches make hipcc wait vmcnt(0) between stores);
    // only the last, ragged position tile takes the checked variant.
    const float relu_floor = p.act == 1 ? 0.0f : -__builtin_inff();
    const float neg_slope = p.act == 2 ? 0.2f : 1.0f;     // act 2: LeakyReLU(0.2) = max(v, 0.2 v)
    auto store_tile = [&](auto checked, auto slow_act) {
        constexpr bool CHECK = decltype(checked)::value;
        constexpr bool SLOW = decltype(slow_act)::value;     // act 3 (tanh) / 4 (sigmoid) and/or a residual input
        auto finish = [&](float v, int64_t idx) {
            v = fmaxf(fmaxf(v, v * neg_slope), relu_floor);
            if (SLOW) {
                if (p.act == 3) v = tanhf(v);
                else if (p.act == 4) v = 1.0f / (1.0f + __expf(-v));
                if (p.residual) v += p.residual[idx];
            }
            return v;
        };
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = c0 + (wn * TN + j) * 32 + frow;
            const bool c_ok = c < p.cols;
            const int cm = (c_ok ? c : 0) % p.cmod;
            const float sc = p.scale[cm], sh = p.shift[cm];
            if (p.planar) {
                // column-major output [cols][ld_planar]: registers 4g..4g+3 are four consecutive positions -> one 16-B store
                float *colp = p.out + (int64_t)c * p.ld_planar;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int o = o32[i][4 * g];
                        const int64_t base = (int64_t)c * p.ld_planar + o;
                        float4 v;
                        v.x = finish(fmaf(acc[i][j][4 * g + 0], sc, sh), base + 0);
                        v.y = finish(fmaf(acc[i][j][4 * g + 1], sc, sh), base + 1);
                        v.z = finish(fmaf(acc[i][j][4 * g + 2], sc, sh), base + 2);
                        v.w = finish(fmaf(acc[i][j][4 * g + 3], sc, sh), base + 3);
                        if (c_ok && (!CHECK || o >= 0)) *reinterpret_cast<float4 *>(colp + o) = v;
                    }
            } else {
                float *colp = p.out + c;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = o32[i][r];
                        if (c_ok && (!CHECK || o >= 0)) {
                            const int64_t idx = (int64_t)o * p.cols + c;
                            colp[(int64_t)o * p.cols] = finish(fmaf(acc[i][j][r], sc, sh), idx);
                        }
                    }
            }
        }
    };
    const bool slow = p.act >= 3 || p.residual != nullptr;
    const bool full = m0 + BM <= p.positions;
    if (slow) {
        if (full) store_tile(std::false_type{}, std::true_type{});
        else store_tile(std::true_type{}, std::true_type{});
    } else {
        if (full) store_tile(std::false_type{}, std::false_type{});
        else store_tile(std::true_type{}, std::false_type{});
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Tail of the generator: ConvTranspose2d(Cin -> 3, k4, s2, p1) + bias + tanh (+ 8-bit code).
// Three output channels cannot fill a matrix-core tile per sub-pixel phase, so the layer is split:
//   1. scatter form on the matrix cores: P[(ky*4+kx)*3+co][pos] (column-major) = sum_ci in[pos][ci] * W[ci][co][ky][kx]
//      -- one dense GEMM [positions x Cin] x [Cin x 48] (gather_conv_kernel<4,1,1,2>, single tap);
//   2. this kernel gathers the four contributions of every output pixel (col2im), adds the bias,
//      applies tanh and writes NCHW fp32 and/or the 8-bit code.  HBM-bound.
//   out(2y+py, 2x+px) = sum over (dy,ky) in T(py), (dx,kx) in T(px) of P[y+dy][x+dx][ky][kx]
//   T(0) = {(0,1), (-1,3)},  T(1) = {(+1,0), (0,2)}      (oy = 2*iy - 1 + ky)
// One thread per input-grid position = 2 x 2 output pixels x 3 channels.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t quantize_byte(float x)
{
    // gan_models/dcgan/train_torch.py:154-158,172: (x+1)/2, *255, truncate
    float t = __fmul_rn(__fdiv_rn(__fadd_rn(x, 1.0f), 2.0f), 255.0f);
    t = fminf(fmaxf(truncf(t), 0.0f), 255.0f);
    return (uint32_t)(int)t;
}

__global__ void __launch_bounds__(THREADS) col2im_rgb_tanh_kernel(const float *__restrict__ P, int64_t ldp, int64_t n_img, int H, int W,
                                                                   const float *__restrict__ bias, float *__restrict__ out_f32,
                                                                   uint8_t *__restrict__ out_u8)
{
    // P is column-major: P[col * ldp + position], so a wave (consecutive x) reads 256 contiguous bytes per column
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = n_img * H * W;
    if (gid >= total) return;
    const int64_t img = gid / (H * W);
    const int rem = (int)(gid - img * (H * W));
    const int y = rem / W, x = rem - y * W;
    const int Ho = 2 * H, Wo = 2 * W;
    constexpr int DY[2][2] = {{0, -1}, {1, 0}};
    constexpr int KY[2][2] = {{1, 3}, {0, 2}};
    float acc[2][2][3];
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
#pragma unroll
            for (int co = 0; co < 3; ++co) acc[py][px][co] = 0.0f;
#pragma unroll
            for (int ty = 0; ty < 2; ++ty)
#pragma unroll
                for (int tx = 0; tx < 2; ++tx) {
                    const int yy = y + DY[py][ty], xx = x + DY[px][tx];
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    const float *src = P + (int64_t)((KY[py][ty] * 4 + KY[px][tx]) * 3) * ldp + gid + (int64_t)DY[py][ty] * W + DY[px][tx];
#pragma unroll
                    for (int co = 0; co < 3; ++co) acc[py][px][co] += src[co * ldp];
                }
        }
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

template <int WAVES_M, int WAVES_N, int TM, int TN>
int launch_gather(gl_ctx *ctx, const GlGatherConv &p, int phases)
{
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int lds = 2 * (BM + BN) * BK * 4;
    const int64_t m_tiles = gl_ceil_div(p.positions, BM);
    const int n_tiles = p.cols_pad / BN;
    GL_REQUIRE(p.cols_pad % BN == 0 && p.cols <= p.cols_pad, "gather_conv: cols_pad=%d must be a multiple of %d", p.cols_pad, BN);
    GL_REQUIRE(m_tiles * n_tiles * phases < (1ll << 31), "gather_conv: grid too large");
    const int lds_req = lds;
    auto kern = gather_conv_kernel<WAVES_M, WAVES_N, TM, TN>;
    GL_ONCE_PER_DEVICE(ctx, \
        GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_req)););
    gl_prof_scope prof_(ctx, GL_PROF_GATHER_CONV);
    hipLaunchKernelGGL(kern, dim3((unsigned)(m_tiles * n_tiles * phases)), dim3(THREADS), lds_req, ctx->stream, p, (int)m_tiles, n_tiles, phases);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

}  // namespace

int gl_launch_gather_conv(gl_ctx *ctx, const GlGatherConv &p_in, int phases)
{
    gl_make_current(ctx);
    GlGatherConv p = p_in;
    {
        const uint64_t imgs = (uint64_t)(p.positions / ((int64_t)p.H * p.W));
        const uint64_t bytes = imgs * (uint64_t)(p.H >> p.up) * (uint64_t)(p.W >> p.up) * (uint64_t)p.Cin * 4ull;
        GL_REQUIRE(bytes < 0xC0000000ull, "gather_conv: input tensor of %llu bytes exceeds 3 GiB; use a smaller pass", (unsigned long long)bytes);
        p.in_bytes = (unsigned)bytes;
    }
    GL_REQUIRE(p.Cin % BK == 0, "gather_conv: Cin=%d must be a multiple of %d", p.Cin, BK);
    GL_REQUIRE(phases >= 1 && phases <= 4 && p.ntaps >= 1 && p.ntaps <= 16, "gather_conv: bad phases/taps");
    GL_REQUIRE((reinterpret_cast<uintptr_t>(p.in) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.wpack) & 15) == 0, "gather_conv: unaligned operand");
    GL_REQUIRE((p.positions / ((int64_t)p.H * p.W)) * p.Ho * p.Wo < (1ll << 31) && p.positions < (1ll << 31),
               "gather_conv: more than 2^31 positions in one launch");
    GL_REQUIRE(p.act >= 0 && p.act <= 4 && !(p.planar && p.residual), "gather_conv: bad activation code / residual with planar output");
    GL_REQUIRE(p.in_bytes > 0 && (uint64_t)p.in_bytes + 2ull * (uint64_t)(p.W + 1) * p.Cin * 4ull < 0xC0000000ull,
               "gather_conv: input tensor of %llu bytes exceeds the 3 GiB a buffer descriptor may span here; use a smaller pass", (unsigned long long)p.in_bytes);
    GL_REQUIRE((uint64_t)p.cols_pad * p.ntaps * p.Cin * 4ull < 0xC0000000ull, "gather_conv: packed weights too large for one descriptor");
    GL_REQUIRE(p.up == 0 || (p.up == 1 && p.H % 2 == 0 && p.W % 2 == 0), "gather_conv: up must be 0 or 1 (even H, W)");
    if (p.planar)
        GL_REQUIRE(phases == 1 && p.omul == 1 && p.Ho == p.H && p.Wo == p.W && p.positions % 4 == 0 && p.ld_planar % 4 == 0 &&
                       p.ld_planar >= p.positions && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0,
                   "gather_conv: planar output needs an identity position map and 4-aligned sizes");
    if (p.positions == 0) return GL_OK;
    if (p.cols_pad % 128 == 0) return launch_gather<2, 2, 2, 2>(ctx, p, phases);
    return launch_gather<4, 1, 1, 2>(ctx, p, phases);
}

int gl_launch_col2im_rgb_tanh(gl_ctx *ctx, const float *P, int64_t ldp, int64_t n_img, int H, int W, const float *bias, float *out_f32,
                              uint8_t *out_u8)
{
    gl_make_current(ctx);
    if (n_img == 0) return GL_OK;
    const int64_t total = n_img * H * W;
    gl_prof_scope prof_(ctx, GL_PROF_CONVT_RGB);
    hipLaunchKernelGGL(col2im_rgb_tanh_kernel, dim3((unsigned)gl_ceil_div(total, THREADS)), dim3(THREADS), 0, ctx->stream, P, ldp, n_img, H, W,
                       bias, out_f32, out_u8);
    GL_LAUNCH_CHECK();
    return GL_OK;
}
