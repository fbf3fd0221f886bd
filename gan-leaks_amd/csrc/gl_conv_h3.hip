// Split-fp16 ("h3") variant of the tap-gather implicit GEMM: fp32-class accuracy on the fp16 matrix cores.
//
//   every operand x is stored as a pair of halves  x*S ~= hi + lo,  hi = fp16(x*S), lo = fp16(x*S - hi)   (~22 mantissa bits)
//   a*b ~= hi_a*hi_b + hi_a*lo_b + lo_a*hi_b     (the dropped lo*lo term is 2^-22 relative)
//   three v_mfma_f32_16x16x32_f16 per 16x16 tile and 32-deep K slice, fp32 accumulation:
//   16x the fp32-MFMA rate / 3 = 5.3x, with errors of the same class as an fp32 convolution (measured in tests).
//
// Storage ("split layout"): a 32-channel chunk of one position is 128 bytes = 32 hi halves then 32 lo halves, i.e. the
// SAME byte geometry as 32 fp32 values: activations [pos][C/32][128 B], packed weights [col][K/32][128 B] with the K
// order of gl_conv_k_index.  The gather addressing (buffer_load ... lds with scalar per-slice offsets and hardware
// zero-fill for padding) is therefore byte-identical to gather_conv_kernel's.
//
// Tiling: WC x WP waves per workgroup, each TC x TP tiles of 16 x 16 (shapes below).  Weights are the MFMA A operand, positions the
// B operand, so a lane ends up with 4 consecutive channels of one position: one 8-byte store for the hi halves, one for the lo
// halves.  LDS rows are 128 B, chunk c of row r at slot c ^ (r & 7): conflict-free ds_read_b128 (same scheme as l2_knn_i8_kernel).
// FUSE_TAIL: the next layer -- a per-position GEMM with 48 columns over this layer's 64 or 128 channels, i.e. the generator's
// ConvTranspose2d(C -> 3, k4 s2 p1) in scatter form -- is evaluated in the epilogue on the activations still in registers.
#include "gl_conv.h"
#include "gl_conv_h3_epi.h"
#include <cstdlib>
#include <type_traits>

namespace {

using gl_h3::v8h;
using gl_h3::v4h;
using gl_h3::v4f;

constexpr int HBK_BYTES = 128;          // one K slice of 32 channels: 64 B hi + 64 B lo

// WC x WP waves (channel direction x position direction), each owning TC x TP tiles of 16 x 16:
//   <2,2,4,4>: 128 channels x 128 positions, 4 waves of 64 x 64, 64 KiB LDS (2 workgroups per CU)
//   <1,4,4,4>: 64 channels x 256 positions, 4 waves, 80 KiB LDS (2 workgroups per CU): layers with <= 64 output columns
//   <2,4,8,4>: 256 channels x 256 positions, 8 waves of 128 x 64, 128 KiB LDS (1 workgroup per CU): wide layers of large passes
//   <1,8,8,4>: 128 channels x 512 positions, 8 waves of 128 x 64, 160 KiB LDS: 65..128-column layers of large passes (PGGAN at 128 x 128,
//              VGG16 conv2_x): +4.5 % on PGGAN-256 over <2,2,4,4> (A/B alternating on one device)
// K slices are double buffered with one __syncthreads() per slice.  Tried and dropped (no gain): the software pipeline
// of gl_pair256.h on this kernel (inline-asm fragment reads with counted waits, a ring of two weight tiles, the barrier before the last
// weight tile, DMA pieces spread over the weight tiles: bit-identical, 2-3 % SLOWER on every generator -- round 2), a 128 x 256
// tile with 8 waves of 64 x 64, 256 x 256 with 4 waves of 128 x 128 (1 wave per SIMD), 128 x 512 / 128 x 256 tiles for 128-column
// layers (and, round 2, the 128 x 512 tile WITH the fused RGB tail for DCGAN's last hidden layer, K loop free of spills: 193 -> 211-220 ms per step; that layer's K is only 32 slices long, and one 8-wave workgroup per CU cannot hide its long epilogue behind another workgroup's main loop the way two 128 x 128 ones do), a ring of three slices with counted s_waitcnt vmcnt(N), 4 workgroups per CU, staggered staging of the two wave halves,
// halo staging (one staged pixel range per channel chunk, taps as shifted rows).  Also tried (round 2): GEMM rows in border-sorted order (the 9
// classes first / inner / last row x column), so that tiles inside one class SKIP the K slices of their outside taps -- 23 % of the MFMAs of
// DCGAN's 4 x 4 -> 8 x 8 layer, 12 % of the next one, bit-identical outputs.  Measured with clean K loops (tools/check_loop_spills.py): the
// headline step went from 192 to 210-222 ms.  A class-pure tile of 256 rows spans 64 to 256 images instead of 16, and the re-use of an input
// pixel by its 4 taps x 4 phases, which a raster tile gets from its own L2 footprint, is gone.  Not kept.
// UP: the input is read through nearest-neighbour x2 upsampling (p.up; a template parameter so that the common form does not carry its registers)
// T4 (round 3; the 256 x 256 tile on a 4 x 4 input grid: DCGAN's 4 x 4 -> 8 x 8 layer, VGG16 conv5_x and PGGAN's first block at 64 x 64):
//   a tile holds the 16 positions of 16 images either way, but its rows are ordered so that every 16-row MFMA fragment is ONE position of
//   the 16 images, and the fragments are dealt to the 4 position-waves as a Latin square -- wave wp, fragment j holds position
//   (y, x) = (j, (j + wp) mod 4): one of every row and one of every column per wave.  A tap that falls outside the image for a position
//   does so for the whole fragment (all-zero operand), so its 3 x TC MFMAs and its two fragment reads are skipped: 18.75 % of the
//   layer's MFMAs for ConvTranspose k4 s2 p1 (the ideal, 23.4 %, would need the corner fragment on every wave), ~27 % for a 3 x 3 p1
//   convolution; every wave skips (nearly) the same number per slice, so the SIMDs stay in step.  Same images per tile as the raster
//   order, hence the same L2 footprint (the border-sorted order above lost exactly that).  Adding a zero product changes no bit: outputs are
//   identical to the raster form's (tests/test_gpu_dcgan.py).
template <int WC, int WP, int TC = 4, int TP = 4, bool FUSE_TAIL = false, bool UP = false, bool T4 = false>
__global__ void __launch_bounds__(64 * WC * WP, 2) gather_conv_h3_kernel(const GlGatherConv p, int m_tiles, int n_tiles, int phases)
{
#if __HIP_DEVICE_COMPILE__
    constexpr int HTC = 16 * TC * WC, HTP = 16 * TP * WP;        // tile: channels x positions (a wave owns TC x TP tiles of 16 x 16)
    constexpr int W_BYTES = HTC * HBK_BYTES, X_BYTES = HTP * HBK_BYTES, BUF = W_BYTES + X_BYTES;
    constexpr int NW = WC * WP;
    constexpr int PW = (HTC / 8) / NW, PX = (HTP / 8) / NW;   // 1-KiB staging pieces per wave and slice
    static_assert(!T4 || (HTP == 256 && TP == 4 && WP == 4 && PX == 4 && !UP && !FUSE_TAIL), "T4: the 256-position tile with 4 position-waves of 4 fragments");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][W | X]

    const unsigned inner = (unsigned)phases * (unsigned)n_tiles;
    const unsigned id = gl_xcd_remap(blockIdx.x, (unsigned)m_tiles * inner);
    const int mt = (int)(id / inner);
    const int phase = (int)((id % inner) / (unsigned)n_tiles);
    const int nt = (int)(id % (unsigned)n_tiles);
    const int64_t m0 = (int64_t)mt * HTP;     // first position of the tile
    const int c0 = nt * HTC;                  // first output channel of the tile

    const int tid0 = threadIdx.x, lane0 = tid0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int wc = wave / WP, wp_ = wave % WP;         // wave position: channel block, position block

    const int K = p.ntaps * p.Cin;                     // in elements; a slice is 32 elements = 128 bytes
    const int nk = K / 32;
    const int HW = p.H * p.W;
    const uint32_t tdy = p.tap_dy[phase], tdx = p.tap_dx[phase];
    const char *wbase = reinterpret_cast<const char *>(p.wpack) + (int64_t)phase * p.cols_pad * K * 4;

    const int rsub = lane0 >> 3, slot = lane0 & 7;
    constexpr int up = UP ? 1 : 0;
    const int Ws = p.W >> up, Hs = p.H >> up;
    constexpr unsigned kOOB = 0xC0000000u;
    const unsigned lead_bytes = (unsigned)(p.W + 1) * (unsigned)p.Cin * 4u;
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.in)) - (up ? 0 : (int64_t)lead_bytes), 0,
                                          (int)(p.in_bytes + (up ? 0u : 2u * lead_bytes)), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wbase), 0, (int)((unsigned)p.cols_pad * (unsigned)K * 4u), 0x00020000);
    // per-lane offsets of piece 0; piece i of a wave lies 8 rows further: a wave-uniform distance that goes into the scalar offset
    unsigned x_voff0 = kOOB, x_mask2[(PX + 1) / 2], w_voff0 = 0;      // tap masks of two pieces per word (ntaps <= 16)
#pragma unroll
    for (int i = 0; i < (PX + 1) / 2; ++i) x_mask2[i] = 0;
    auto x_mask_of = [&](int i) -> unsigned { return (x_mask2[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu; };
    int x_img[UP ? PX : 1], x_yx[UP ? PX : 1];            // UP only: first pixel of the image (low resolution), y | x << 16
    {
        const int r = wave * PW * 8 + rsub;                 // (r & 7) == rsub for every piece
        w_voff0 = ((unsigned)(c0 + r) * (unsigned)K) * 4u + (unsigned)(slot ^ rsub) * 16u;
    }
    const unsigned w_step = 8u * (unsigned)K * 4u, x_step = 8u * (unsigned)p.Cin * 4u;
    // T4: tile row r = fragment (r >> 4) x image (r & 15); fragment f = 4 wp + j holds position 4 j + ((j + wp) & 3) of the 4 x 4 grid
    auto t4_pos = [](int r) -> int { const int f = r >> 4, wpp = f >> 2, j = f & 3; return (r & 15) * 16 + 4 * j + ((j + wpp) & 3); };
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const int r = (wave * PX + i) * 8 + rsub;
        const int chunk = slot ^ (r & 7);
        const int64_t pos = m0 + (T4 ? t4_pos(r) : r);
        if constexpr (UP) { x_img[i] = 0; x_yx[i] = 0; }
        if (pos < p.positions) {
            const int64_t img = pos / HW;
            const int rem = (int)(pos - img * HW);
            const int y = rem / p.W, x = rem - y * p.W;
            for (int t = 0; t < p.ntaps; ++t) {
                const int yy = y + (int)((tdy >> (2 * t)) & 3u) - 1, xx = x + (int)((tdx >> (2 * t)) & 3u) - 1;
                if ((yy >= 0) & (yy < p.H) & (xx >= 0) & (xx < p.W)) x_mask2[i >> 1] |= 1u << (t + (i & 1) * 16);
            }
            // raster: pieces are 8 positions apart; T4: the same lane of piece i is (t4_pos(8 i + r0) - t4_pos(r0)) positions further, a
            // wave-uniform distance as well (the row inside the piece is the image, r & 7 = rsub for every piece)
            if (i == 0 || T4) { if (i == 0) x_voff0 = (unsigned)pos * (unsigned)p.Cin * 4u + (unsigned)chunk * 16u; }
            if constexpr (UP) { x_img[i] = (int)(img * Hs * Ws); x_yx[i] = y | (x << 16); }
        }
    }

    auto stage = [&](int kt, char *buf) {
        const int tap = kt % p.ntaps;
        const int cc = kt / p.ntaps;                   // 32-channel chunk of the input
        const int dy = (int)((tdy >> (2 * tap)) & 3u) - 1;
        const int dx = (int)((tdx >> (2 * tap)) & 3u) - 1;
        const unsigned tapbit = 1u << tap;
#pragma unroll
        for (int i = 0; i < PW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (gl_lptr)(buf + (wave * PW + i) * 1024), 16, w_voff0, (unsigned)kt * 128u + (unsigned)i * w_step, 0, 0);
        if (!up) {
            const unsigned soff = (unsigned)(((dy + 1) * p.W + (dx + 1)) * p.Cin) * 4u + (unsigned)cc * 128u;
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                const unsigned voff = (x_mask_of(i) & tapbit) ? x_voff0 : kOOB;
                unsigned piece = (unsigned)i * x_step;
                if constexpr (T4) piece = (unsigned)(t4_pos((wave * PX + i) * 8) - t4_pos(wave * PX * 8)) * (unsigned)p.Cin * 4u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (gl_lptr)(buf + W_BYTES + (wave * PX + i) * 1024), 16, voff, soff + piece, 0, 0);
            }
        } else {
            const unsigned soff = (unsigned)cc * 128u;
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                const int r = (wave * PX + i) * 8 + rsub;
                const int yy = (x_yx[UP ? i : 0] & 0xFFFF) + dy, xx = (x_yx[UP ? i : 0] >> 16) + dx;
                const unsigned pix = (unsigned)(x_img[UP ? i : 0] + (yy >> 1) * Ws + (xx >> 1));
                const unsigned voff = (x_mask_of(i) & tapbit) ? pix * (unsigned)p.Cin * 4u + (unsigned)((slot ^ (r & 7)) * 16) : kOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (gl_lptr)(buf + W_BYTES + (wave * PX + i) * 1024), 16, voff, soff, 0, 0);
            }
        }
    };

    v4f acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

    const int frow0 = lane0 & 15, fk0 = lane0 >> 4;
    auto compute = [&](const char *cur, int kt) {
        const char *lw = cur + (wc * 16 * TC) * HBK_BYTES;
        const char *lx = cur + W_BYTES + (wp_ * 16 * TP) * HBK_BYTES;
        v8h w_hi[TC], w_lo[TC], x_hi[TP], x_lo[TP];
        unsigned skip = 0;                                 // T4: bit j = fragment j of this wave lies outside the image for this slice's tap
        if constexpr (T4) {
            const int tap = kt % p.ntaps;
            const int dy = (int)((tdy >> (2 * tap)) & 3u) - 1, dx = (int)((tdx >> (2 * tap)) & 3u) - 1;
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int yy = j + dy, xx = ((j + wp_) & 3) + dx;
                skip |= ((yy < 0) | (yy > 3) | (xx < 0) | (xx > 3)) ? 1u << j : 0u;
            }
        }
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int r = i * 16 + frow0;
            w_hi[i] = *reinterpret_cast<const v8h *>(lw + r * HBK_BYTES + ((fk0 ^ (r & 7)) << 4));
            w_lo[i] = *reinterpret_cast<const v8h *>(lw + r * HBK_BYTES + (((4 + fk0) ^ (r & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int r = j * 16 + frow0;
            x_hi[j] = *reinterpret_cast<const v8h *>(lx + r * HBK_BYTES + ((fk0 ^ (r & 7)) << 4));
            x_lo[j] = *reinterpret_cast<const v8h *>(lx + r * HBK_BYTES + (((4 + fk0) ^ (r & 7)) << 4));
        }
        if constexpr (T4) {
            // fragment-major, so that a skipped fragment is one wave-uniform branch around 3 TC MFMAs (the zero products it drops change no bit)
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                if ((skip >> j) & 1u) continue;
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_lo[i], x_hi[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi[i], x_lo[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi[i], x_hi[j], acc[i][j], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_lo[i], x_hi[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi[i], x_lo[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi[i], x_hi[j], acc[i][j], 0, 0, 0);
                }
        }
    };
    stage(0, smem);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        if (kt + 1 < nk) stage(kt + 1, smem + ((kt + 1) & 1) * BUF);
        compute(smem + (kt & 1) * BUF, kt);
    }

    // ---- epilogue.  C tile (16 x 16): column (position) = lane & 15, row (channel) = 4 * (lane >> 4) + reg.
    __syncthreads();
    // the lane-derived indices are derived AGAIN here, behind an opaque copy of threadIdx.x, so that the ones above are dead during the main
    // loop: the loop needs every register it can get (one spilled accumulator tile costs a scratch reload per slice -- and the
    // s_waitcnt vmcnt(0) that comes with it drains the LDS-DMA queue)
    int tid_again = threadIdx.x;
    asm volatile("" : "+v"(tid_again));
    const int tid = tid_again, lane = tid_again & 63, frow = tid_again & 15, fk = (tid_again & 63) >> 4;
    int *orow = reinterpret_cast<int *>(smem);
    if (tid < HTP) {
        const int64_t pos = m0 + (T4 ? t4_pos(tid) : tid);
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
    int o4[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) o4[j] = orow[wp_ * 16 * TP + j * 16 + frow];
    const float relu_floor = p.act == 1 ? 0.0f : -__builtin_inff();
    const float neg_slope = p.act == 2 ? 0.2f : 1.0f;
    bool saturated = false;
    if constexpr (FUSE_TAIL) {
        // The next layer's GEMM (48 columns over this layer's 128 channels) on the activations while they are still in registers.
        // A wave holds 64 of the 128 channels (tiles i = 0..3) of its 64 positions: a lane's 4 + 4 values of tiles 2s and 2s + 1 form the
        // 8 k-values of its k group in k-step s (the weights are stored in that order), so the B operands need no data movement.
        static_assert((WC == 2 || WC == 1) && TC == 4 && TP == 4, "fused tail: waves of 64 channels x 64 positions, 64 or 128 channels in all");
        v8h b_hi[2][TP], b_lo[2][TP];
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int ch = wc * 64 + i * 16 + 4 * fk;
            float sc[4], sh[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc[r] = p.scale[ch + r]; sh[r] = p.shift[ch + r]; }
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = fmaf(acc[i][j][r], sc[r], sh[r]);
                    t = fmaxf(fmaxf(t, t * neg_slope), relu_floor);
                    const float c = fminf(fmaxf(t, -65504.0f), 65504.0f);
                    saturated |= (c != t) && (o4[j] >= 0);
                    const _Float16 h = (_Float16)c;
                    b_hi[i >> 1][j][(i & 1) * 4 + r] = h;
                    b_lo[i >> 1][j][(i & 1) * 4 + r] = (_Float16)(c - (float)h);
                }
        }
        v4f pacc[3][TP];
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
#pragma unroll
            for (int j = 0; j < TP; ++j) pacc[pt][j] = (v4f){0.f, 0.f, 0.f, 0.f};
        const v8h *tw = reinterpret_cast<const v8h *>(p.tail_w) + (size_t)wc * 3 * 2 * 2 * 64;
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const v8h a_hi = tw[((pt * 2 + s2) * 2 + 0) * 64 + frow * 4 + fk];
                const v8h a_lo = tw[((pt * 2 + s2) * 2 + 1) * 64 + frow * 4 + fk];
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    pacc[pt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, b_hi[s2][j], pacc[pt][j], 0, 0, 0);
                    pacc[pt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_lo[s2][j], pacc[pt][j], 0, 0, 0);
                    pacc[pt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_hi[s2][j], pacc[pt][j], 0, 0, 0);
                }
            }
        // 128 channels: the two channel halves (waves wc = 0, 1 of the same positions) are summed through LDS, then the wc = 0 wave stores
        float *xch = reinterpret_cast<float *>(smem + 4096) + (size_t)wp_ * 3 * TP * 4 * 64;
        if (WC == 2 && wc == 1) {
#pragma unroll
            for (int pt = 0; pt < 3; ++pt)
#pragma unroll
                for (int j = 0; j < TP; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xch[((pt * TP + j) * 4 + r) * 64 + lane] = pacc[pt][j][r];
        }
        if (WC == 2) __syncthreads();
        if (wc == 0) {
#pragma unroll
            for (int pt = 0; pt < 3; ++pt)
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    const int o = o4[j];
                    if (o < 0) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = pacc[pt][j][r] + (WC == 2 ? xch[((pt * TP + j) * 4 + r) * 64 + lane] : 0.0f);
                        p.tail_out[(int64_t)(pt * 16 + 4 * fk + r) * p.tail_ld + o] = v * p.tail_scale;
                    }
                }
        }
    } else {
        saturated |= gl_h3::epilogue<WC, WP, TC, TP>(p, acc, c0, wc, wp_, lane, o4, smem, p.tap_V ? (p.Wo >> 4) : 0);
    }
    if (__any(saturated) && lane == 0) atomicAdd(p.sat_flag, 1);
#endif
}

// fp32 rows [n][d] -> split layout [n][dpad/32][hi 32 | lo 32], value * scale, zero padded to dpad
__global__ void __launch_bounds__(256) split_rows_kernel(const float *__restrict__ in, int64_t n, int d, int dpad, float scale, char *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * dpad) return;
    const int64_t r = i / dpad;
    const int c = (int)(i - r * dpad);
    const float v = c < d ? fminf(fmaxf(in[r * d + c] * scale, -65504.0f), 65504.0f) : 0.0f;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    char *dst = out + (r * dpad + (c >> 5) * 32) * 4 + (c & 31) * 2;
    *reinterpret_cast<_Float16 *>(dst) = hi;
    *reinterpret_cast<_Float16 *>(dst + 64) = lo;
}

}  // namespace

template <int WC, int WP, int TC = 4, int TP = 4, bool FUSE_TAIL = false, bool UP = false, bool T4 = false>
static int launch_h3(gl_ctx *ctx, const GlGatherConv &p, int phases)
{
    if constexpr (!UP && !FUSE_TAIL && !T4) {
        if (p.up) return launch_h3<WC, WP, TC, TP, FUSE_TAIL, true>(ctx, p, phases);
    }
    if constexpr (WC == 2 && WP == 4 && TC == 8 && TP == 4 && !UP && !FUSE_TAIL && !T4) {
        // a 4 x 4 input grid: the fragment-per-position row order, which skips the MFMAs of taps that fall outside the image
        const bool plain = !p.up && !p.tail_w && !p.tap_V && !p.rgb_out && p.pixnorm_act == 0.0f && !p.planar;
        if (plain && p.H == 4 && p.W == 4 && gl_tuning_int("GL_H3_T4", 1)) return launch_h3<WC, WP, TC, TP, false, false, true>(ctx, p, phases);
    }
    constexpr int HTC = 16 * TC * WC, HTP = 16 * TP * WP;
    const int64_t m_tiles = gl_ceil_div(p.positions, HTP);
    const int n_tiles = (int)gl_ceil_div(p.cols, HTC);       // weight rows are padded to cols_pad >= n_tiles * HTC
    GL_REQUIRE(m_tiles * n_tiles * phases < (1ll << 31), "gather_conv_h3: grid too large");
    constexpr int lds = 2 * (HTC + HTP) * HBK_BYTES;
    auto kern = gather_conv_h3_kernel<WC, WP, TC, TP, FUSE_TAIL, UP, T4>;
    GL_ONCE_PER_DEVICE(ctx, \
        GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds)););
    gl_prof_scope prof_(ctx, GL_PROF_GATHER_CONV);
    hipLaunchKernelGGL(kern, dim3((unsigned)(m_tiles * n_tiles * phases)), dim3(64 * WC * WP), lds, ctx->stream, p, (int)m_tiles, n_tiles, phases);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// which tile gl_launch_gather_conv_h3 picks: 0 = <2,2> 128 x 128, 1 = <1,4> 64 channels x 256 positions, 2 = <2,4,8,4> 256 x 256,
// 3 / 4 = the fused-tail forms of 0 / 1
static int h3_tile_choice(const GlGatherConv &p, int phases)
{
    if (p.tail_w) return p.cols == 128 ? 3 : 4;
    // narrow outputs (the generator's 48-column RGB tail, toRGB): 64 channels x 256 positions, half the padded MFMAs of the square tile
    // (64 channels x 512 positions with 8 waves measured 2 % slower on PGGAN-256 than this tile at two workgroups per CU)
    if (p.cols <= 64) return 1;
    // wide layers with enough work to fill the chip: 256 channels x 256 positions, 8 waves of 128 x 64 (24 LDS fragment reads per 96
    // MFMAs instead of 16 per 48, half the staging per MFMA): +7 % on the DCGAN stack (A/B on one device, profiles/r01/README.md)
    if (p.cols % 256 == 0 && gl_ceil_div(p.positions, 256) * (p.cols / 256) * phases >= 256) return 2;
    // 65..128 columns with many positions (PGGAN's 128-channel block at 128 x 128): 128 channels x 512 positions, 8 waves of 128 x 64 like the
    // wide tile's, all 160 KiB of LDS
    const int wide128 = gl_tuning_int("GL_H3_TILE128", 1);
    if (wide128 && p.cols <= 128 && gl_ceil_div(p.positions, 512) * phases >= 512) return 5;
    return 0;
}

int gl_conv_h3_tile_channels(const GlGatherConv &p, int phases)
{
    if (gl_conv_halo_applies(p, phases)) return p.cols <= 64 ? 64 : 128;
    const int t = h3_tile_choice(p, phases);
    // the 8 x 4 wave tiles (2: 256 x 256, 5: 128 x 512) have no fused epilogue (gl_conv_h3_epi.h): 0
    return (t == 2 || t == 5) ? 0 : (t == 1 || t == 4) ? 64 : 128;       // tiles 0 and 3 hold 128 channels
}

bool gl_conv_h3_tap_fusable(const GlGatherConv &p, int phases)
{
    const int enabled = gl_tuning_int("GL_TAP_FUSE", 1);
    if (!enabled || phases != 1 || p.tail_w || p.out_mode != 2 || p.omul != 1 || p.Ho != p.H || p.Wo != p.W || p.planar) return false;
    if (p.cols > gl_conv_h3_tile_channels(p, phases) || p.H % 2 != 0 || p.W % 16 != 0) return false;
    if (gl_conv_halo_applies(p, phases)) return true;           // a wave owns 4 rows of its 16 x 16 block
    // tap-gather tiles: a wave owns 64 consecutive positions, which must hold whole pairs of rows
    return (p.W == 16 || p.W == 32) && p.positions % 64 == 0;
}

int gl_launch_gather_conv_h3(gl_ctx *ctx, const GlGatherConv &p_in, int phases)
{
    gl_make_current(ctx);
    GlGatherConv p = p_in;
    p.sat_flag = ctx->h3_sat;
    GL_REQUIRE(p.Cin % 32 == 0, "gather_conv_h3: Cin=%d must be a multiple of 32", p.Cin);
    GL_REQUIRE(p.cmod > 0 && p.cmod % 4 == 0, "gather_conv_h3: cmod=%d must be a positive multiple of 4", p.cmod);
    GL_REQUIRE(p.cols_pad % 128 == 0 && p.cols <= p.cols_pad && p.cols % 4 == 0, "gather_conv_h3: cols=%d / cols_pad=%d (multiple of 128)", p.cols, p.cols_pad);
    GL_REQUIRE(phases >= 1 && phases <= 4 && p.ntaps >= 1 && p.ntaps <= 16, "gather_conv_h3: bad phases/taps");
    GL_REQUIRE(p.act >= 0 && p.act <= 2 && !p.residual, "gather_conv_h3: activation %d / residual not supported", p.act);
    GL_REQUIRE(p.out_mode != 2 || p.cols % 32 == 0, "gather_conv_h3: split output needs cols %% 32 == 0");
    GL_REQUIRE(p.up == 0 || (p.up == 1 && p.H % 2 == 0 && p.W % 2 == 0), "gather_conv_h3: up must be 0 or 1");
    {
        const uint64_t imgs = (uint64_t)(p.positions / ((int64_t)p.H * p.W));
        const uint64_t bytes = imgs * (uint64_t)(p.H >> p.up) * (uint64_t)(p.W >> p.up) * (uint64_t)p.Cin * 4ull;
        GL_REQUIRE(bytes + 2ull * (uint64_t)(p.W + 1) * p.Cin * 4ull < 0xC0000000ull, "gather_conv_h3: input of %llu bytes exceeds 3 GiB; use a smaller pass",
                   (unsigned long long)bytes);
        p.in_bytes = (unsigned)bytes;
    }
    GL_REQUIRE((uint64_t)p.cols_pad * p.ntaps * p.Cin * 4ull < 0xC0000000ull, "gather_conv_h3: packed weights too large");
    if (p.positions == 0) return GL_OK;
    GL_REQUIRE(p.positions < (1ll << 31) && (p.positions / ((int64_t)p.H * p.W)) * p.Ho * p.Wo < (1ll << 31), "gather_conv_h3: too many positions");
    GL_REQUIRE(p.pixnorm_act == 0.0f || (p.out_mode == 2 && !p.tail_w && phases == 1 && p.cols <= gl_conv_h3_tile_channels(p, phases)),
               "gather_conv_h3: the fused PixelNorm needs split output and all %d channels in one tile", p.cols);
    GL_REQUIRE(!p.tap_V || (p.pixnorm_act == 0.0f && p.tap_coef && gl_conv_h3_tap_fusable(p, phases)), "gather_conv_h3: this layer's tap cannot be fused");
    if (p.tail_w)
        GL_REQUIRE(p.up == 0 && (p.cols == 128 || p.cols == 64) && p.cmod == p.cols && p.tail_out && p.tail_ld > 0, "gather_conv_h3: the fused tail needs a 64- or 128-channel layer");
    if (gl_conv_halo_applies(p, phases)) return gl_launch_conv_halo_h3(ctx, p);
    switch (h3_tile_choice(p, phases)) {
    case 3: return launch_h3<2, 2, 4, 4, true>(ctx, p, phases);
    case 4: return launch_h3<1, 4, 4, 4, true>(ctx, p, phases);
    case 1: return launch_h3<1, 4>(ctx, p, phases);
    case 2: return launch_h3<2, 4, 8, 4>(ctx, p, phases);
    case 5: return launch_h3<1, 8, 8, 4>(ctx, p, phases);
    default: return launch_h3<2, 2>(ctx, p, phases);
    }
}

int gl_launch_split_rows(gl_ctx *ctx, const float *in, int64_t n, int d, int dpad, float scale, void *out)
{
    gl_make_current(ctx);
    if (n == 0) return GL_OK;
    GL_REQUIRE(dpad % 32 == 0 && d <= dpad, "split_rows: bad padding");
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)gl_ceil_div(n * dpad, 256)), dim3(256), 0, ctx->stream, in, n, d, dpad, scale, reinterpret_cast<char *>(out));
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// host: fp32 packed weight rows [rows][K] (K order of gl_conv_k_index, K % 32 == 0) -> split layout, value * scale
void gl_split_weights_host(const float *w, size_t rows, size_t K, float scale, void *out)
{
    char *o = reinterpret_cast<char *>(out);
    for (size_t r = 0; r < rows; ++r)
        for (size_t k = 0; k < K; ++k) {
            float v = w[r * K + k] * scale;
            if (v > 65504.0f) v = 65504.0f;
            if (v < -65504.0f) v = -65504.0f;
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            char *dst = o + (r * K + (k >> 5) * 32) * 4 + (k & 31) * 2;
            *reinterpret_cast<_Float16 *>(dst) = hi;
            *reinterpret_cast<_Float16 *>(dst + 64) = lo;
        }
}

void gl_pack_tail_weights_host(const float *w, int channels, float scale, void *out)
{
    _Float16 *o = reinterpret_cast<_Float16 *>(out);
    for (int half = 0; half < channels / 64; ++half)
        for (int pt = 0; pt < 3; ++pt)
            for (int st = 0; st < 2; ++st)
                for (int row = 0; row < 16; ++row)
                    for (int g = 0; g < 4; ++g)
                        for (int e = 0; e < 8; ++e) {
                            const int ch = half * 64 + (2 * st + e / 4) * 16 + 4 * g + e % 4;
                            float v = w[(size_t)(pt * 16 + row) * channels + ch] * scale;
                            v = v > 65504.0f ? 65504.0f : (v < -65504.0f ? -65504.0f : v);
                            const _Float16 hi = (_Float16)v;
                            const _Float16 lo = (_Float16)(v - (float)hi);
                            const size_t base = ((((size_t)half * 3 + pt) * 2 + st) * 2) * 512 + (size_t)(row * 4 + g) * 8 + e;
                            o[base] = hi;
                            o[base + 512] = lo;
                        }
}
