// Halo form of the split-fp16 3 x 3 (pad 1) convolution for NARROW layers (<= 128 output channels) at high resolution:
// VGG16 conv1_2 / conv2_x (pretrained_networks.py:106-109) and PGGAN's 128- and 64-channel blocks at 128 x 128 / 256 x 256
// (gan_models/pggan/model_torch.py:52-69), including the blocks' first convolution that reads its input through nearest x2 upsampling.
//
// Why another kernel: the tap-gather form (gl_conv_h3.hip) stages, for every one of the 9 taps and every 32-channel chunk, the 256
// input pixels of its tile again -- 9 x 32 KiB of LDS-DMA per chunk for a tile that produces only 64 (or 128) output channels, i.e.
// 4 (or 8) MFMA tile rows per staged pixel row.  Those layers run at half the rate of the wide ones, and the DMA stream is what
// bounds them.  Here a workgroup owns a 16 x 16 PIXEL BLOCK of one image and stages its 18 x 18 halo ONCE per channel chunk
// (324 pixel rows of 128 B = 40.5 KiB instead of 288 KiB); the 9 taps are 9 shifted views of the same LDS image.  Only the
// weight slice (8 or 16 KiB) is staged per tap.
//
//   LDS image of a chunk: pixel slot q = hr * 18 + hx (hr, hx in 0..17 = image row / column y0 - 1 + hr, x0 - 1 + hx) holds the
//   128 bytes (32 hi halves | 32 lo halves) of that pixel's channel chunk, the 16-byte piece c at slot c ^ (q & 7): the 16 lanes of a
//   fragment read are 16 consecutive q (one image row segment), so every ds_read_b128 is conflict-free for ANY tap shift.
//   Pixels outside the image are hardware zero fills (buffer loads beyond the descriptor's range).
//
// Tried for the WIDE layers too (round 2: 256-channel tile, 8 waves of 128 channels x 4 rows, column tiles).  First form: the halo offsets and the
// wider fragment sets did not fit the 256 registers of a 2-waves-per-SIMD kernel (spills in the K loop, PGGAN-256 376 -> 300 TFLOP/s).  Lean form
// (offsets recomputed per chunk, taps as a real loop; K loop free of spills, bit-identical): PGGAN-256 393.4 -> 393.2 / 396.9 TFLOP/s, configs[2]
// 1603.5 -> 1601.7 ms -- nothing.  The wide layers are not bound by the staging the halo form saves; they stay on the tap-gather kernel.
//
// K order and MFMA order are those of gather_conv_h3_kernel (chunk-major, tap inside; lo*hi, hi*lo, hi*hi), so every output element
// is the same fp32 sum: the two kernels are interchangeable bit for bit (a pass of another size may be dispatched to the other one).
#include "gl_conv.h"
#include "gl_conv_h3_epi.h"
#include <cstdlib>

namespace {

using gl_h3::v4f;
using gl_h3::v8h;

constexpr int BLK = 16;                       // pixel block edge
constexpr int HALO = BLK + 2;                 // 18
constexpr int HPIX = HALO * HALO;             // 324 pixel slots
constexpr int HPIECES = (HPIX + 7) / 8;       // 41 LDS-DMA pieces of 8 pixels (1 KiB)
constexpr int HBYTES = HPIECES * 1024;        // 41 KiB per halo buffer
// RING form (round 3): every wave issues the same number of halo pieces (the waits below are counted), so the buffer is padded to a whole
// number of pieces per wave: 44 KiB for 4 waves, 48 KiB for 8; the surplus pieces are out-of-range loads (zero fill) into the padding
constexpr int halo_bytes(int nw, bool ring) { return ring ? ((HPIECES + nw - 1) / nw) * nw * 1024 : HBYTES; }

// WC waves along the channels (TC = 4 tiles of 16 each: 64 channels per wave), 4 waves along the block rows (4 rows each)
//   <1>: 64 channels, 4 waves, one halo buffer (57 KiB of LDS, two workgroups per CU overlap each other's staging)
//   <2>: 128 channels, 8 waves, two halo buffers (the next chunk is staged during the current one; 114 KiB, one workgroup per CU)
// RING: the weight slices go through a ring of THREE buffers with counted waits (s_waitcnt vmcnt(N) + a raw s_barrier) instead of two buffers
// behind __syncthreads(): the slice of tap t + 2 is requested while tap t is computed, so a tap's 48 MFMAs per wave (768 cycles at the full rate,
// about one L2 round trip) no longer have to cover the whole latency of the next slice.  Same K order, same MFMA order: same bits.
template <int WC, bool RING = false>
__global__ void __launch_bounds__(256 * WC, 2) halo_conv_h3_kernel(const GlGatherConv p, int blocks_x, int blocks_per_img, unsigned total_blocks, int diag)
{
    // diag (tuning build only, timing experiments, RESULTS ARE WRONG when set): 1 = no weight DMA inside the loop, 2 = no barrier inside the loop, 4 = no epilogue
#if __HIP_DEVICE_COMPILE__
    constexpr int TC = 4, TP = 4, WP = 4, NW = WC * WP;
    constexpr int HTC = 64 * WC;
    constexpr int W_BYTES = HTC * 128;                       // one weight slice (tap, chunk)
    constexpr int NHBUF = WC;                                // halo buffers
    constexpr int PH = (HPIECES + NW - 1) / NW;              // halo pieces per wave
    constexpr int PW = (HTC / 8) / NW;                       // weight pieces per wave and slice (2)
    constexpr int HB = halo_bytes(NW, RING);
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [halo x NHBUF][W x 2 (RING: x 3)]
    char *const wbuf = smem + NHBUF * HB;

    const unsigned id = gl_xcd_remap(blockIdx.x, total_blocks);
    const int img = (int)(id / (unsigned)blocks_per_img);
    const int brem = (int)(id % (unsigned)blocks_per_img);
    const int y0 = (brem / blocks_x) * BLK, x0 = (brem % blocks_x) * BLK;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp_ = wave % WP;
    const int rsub = lane >> 3, slot = lane & 7;
    const int frow = lane & 15, fk = lane >> 4;

    const int K = 9 * p.Cin, nchunks = p.Cin / 32;
    const int up = p.up;
    const int Ws = p.W >> up, Hs = p.H >> up;
    constexpr unsigned kOOB = 0xC0000000u;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.in)), 0, (int)p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.wpack)), 0, (int)((unsigned)p.cols_pad * (unsigned)K * 4u), 0x00020000);

    // halo pieces of this wave: piece index ph = wave + NW * i; lane -> pixel slot q = 8 ph + rsub, 16-byte piece (slot ^ (q & 7))
    unsigned h_voff[PH];
#pragma unroll
    for (int i = 0; i < PH; ++i) {
        const int ph = wave + NW * i;
        const int q = ph * 8 + rsub;
        h_voff[i] = kOOB;
        if (ph < HPIECES && q < HPIX) {
            const int hr = q / HALO, hx = q - hr * HALO;
            const int yy = y0 + hr - 1, xx = x0 + hx - 1;
            if ((yy >= 0) & (yy < p.H) & (xx >= 0) & (xx < p.W))
                h_voff[i] = (unsigned)((img * Hs + (yy >> up)) * Ws + (xx >> up)) * (unsigned)p.Cin * 4u + (unsigned)((slot ^ (q & 7)) * 16);
        }
    }
    unsigned w_voff[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int r = (wave * PW + i) * 8 + rsub;
        w_voff[i] = ((unsigned)r * (unsigned)K) * 4u + (unsigned)(slot ^ (r & 7)) * 16u;
    }
    auto stage_halo = [&](int cc, char *hb) {
#pragma unroll
        for (int i = 0; i < PH; ++i) {
            const int ph = wave + NW * i;
            // RING: all PH pieces are issued by every wave (the ones beyond the image are zero fills into the buffer's padding)
            if (RING || ph < HPIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (gl_lptr)(hb + ph * 1024), 16, h_voff[i], (unsigned)cc * 128u, 0, 0);
        }
    };
    auto stage_w = [&](int kt, char *wb) {
#pragma unroll
        for (int i = 0; i < PW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (gl_lptr)(wb + (wave * PW + i) * 1024), 16, w_voff[i], (unsigned)kt * 128u, 0, 0);
    };

    v4f acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

    // pixel slot of this lane's column in position tile j at tap (0, 0): block row 4 wp + j, column frow
    int qbase[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) qbase[j] = (wp_ * 4 + j + 1) * HALO + frow + 1;
    const int w_row0 = (wc * 64 + frow) * 128;               // weight fragment of channel tile i: row wc * 64 + i * 16 + frow

    auto compute = [&](const char *hb, const char *wb, int delta) {
        v8h w_hi[TC], w_lo[TC], x_hi[TP], x_lo[TP];
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const char *r = wb + w_row0 + i * 16 * 128;
            w_hi[i] = *reinterpret_cast<const v8h *>(r + ((fk ^ (frow & 7)) << 4));             // (row & 7) == (frow & 7): tiles are 16 rows apart
            w_lo[i] = *reinterpret_cast<const v8h *>(r + (((4 + fk) ^ (frow & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int q = qbase[j] + delta;
            const char *r = hb + q * 128;
            x_hi[j] = *reinterpret_cast<const v8h *>(r + ((fk ^ (q & 7)) << 4));
            x_lo[j] = *reinterpret_cast<const v8h *>(r + (((4 + fk) ^ (q & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_lo[i], x_hi[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi[i], x_lo[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi[i], x_hi[j], acc[i][j], 0, 0, 0);
            }
    };

    // ---- main loop: chunks of 32 input channels, 9 taps each; one barrier per tap (the weight slice of the tap has landed)
    if constexpr (!RING) {
        stage_halo(0, smem);
        stage_w(0, wbuf);
        for (int cc = 0; cc < nchunks; ++cc) {
            const char *hb = smem + (NHBUF == 2 ? (cc & 1) * HB : 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kt = cc * 9 + tap;
                if (!(diag & 2)) __syncthreads();                 // slice kt (and, at tap 0, the halo of this chunk) landed; the other weight buffer is free
                if (tap < 8 && !(diag & 1)) stage_w(kt + 1, wbuf + ((kt + 1) & 1) * W_BYTES);
                if (NHBUF == 2 && tap == 0 && cc + 1 < nchunks) stage_halo(cc + 1, smem + ((cc + 1) & 1) * HB);
                compute(hb, wbuf + (kt & 1) * W_BYTES, (tap / 3 - 1) * HALO + (tap % 3 - 1));
            }
            if (cc + 1 < nchunks) {
                if (NHBUF == 1) {
                    __syncthreads();                              // everyone is done with the halo image
                    stage_halo(cc + 1, smem);
                }
                stage_w((cc + 1) * 9, wbuf + (((cc + 1) * 9) & 1) * W_BYTES);
            }
        }
    } else {
        // Per wave the LDS-DMA instructions retire in issue order, and s_waitcnt vmcnt(N) returns when all but the N youngest have: before the
        // barrier of slice kt a wave waits for its pieces of slice kt (issued two iterations ago) and lets exactly the younger ones fly --
        // slice kt + 1 (PW pieces) and, when one was requested after slice kt, the next chunk's halo (PH pieces).
        //   issue order:  H(0) W(0) W(1) | iteration kt: W(kt + 2) [two halo buffers, tap 0: H(cc + 1)]  [one halo buffer, after tap 8: barrier, H(cc + 1)]
        // The barrier that follows the wait tells every wave that all waves have the slice (and have finished computing slice kt - 1, whose ring
        // slot W(kt + 2) may now overwrite).
        const int nk = 9 * nchunks;
        stage_halo(0, smem);
        stage_w(0, wbuf);
        if (nk > 1) stage_w(1, wbuf + W_BYTES);
        int slot = 0;                                            // ring slot of slice kt
        for (int cc = 0; cc < nchunks; ++cc) {
            const char *hb = smem + (NHBUF == 2 ? (cc & 1) * HB : 0);
            const bool more = cc + 1 < nchunks;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kt = cc * 9 + tap;
                if (kt + 1 >= nk) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else if (NHBUF == 1 && tap == 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the halo of this chunk was the last request (no-op cost at cc = 0: W(1) is all that flies)
                } else if (NHBUF == 2 && (tap == 1 || tap == 2) && more) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW + PH) : "memory");  // H(cc + 1), requested at tap 0 behind W(kt) or W(kt + 1)'s predecessor, may fly
                } else {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
                }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                const int s2 = slot == 0 ? 2 : slot - 1;                            // slot of slice kt + 2 (= slice kt - 1's)
                if (kt + 2 < nk) stage_w(kt + 2, wbuf + s2 * W_BYTES);
                if (NHBUF == 2 && tap == 0 && more) stage_halo(cc + 1, smem + ((cc + 1) & 1) * HB);
                compute(hb, wbuf + slot * W_BYTES, (tap / 3 - 1) * HALO + (tap % 3 - 1));
                slot = slot == 2 ? 0 : slot + 1;
            }
            if (NHBUF == 1 && more) {
                __builtin_amdgcn_s_barrier();                    // everyone is done with the halo image (LDS reads are consumed before a wave gets here)
                stage_halo(cc + 1, smem);
            }
        }
    }

    // ---- epilogue (shared with the tap-gather kernel)
    __syncthreads();
    int o4[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) o4[j] = (img * p.Ho + y0 + wp_ * 4 + j) * p.Wo + x0 + frow;
    if (diag & 4) {
        float keep = 0.0f;
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (keep == 123.456f) atomicAdd(p.sat_flag, 1);      // keeps the accumulators alive
        return;
    }
    const bool saturated = gl_h3::epilogue<WC, WP, TC, TP>(p, acc, 0, wc, wp_, lane, o4, smem, 1);
    if (__any(saturated) && lane == 0) atomicAdd(p.sat_flag, 1);
#endif
}

template <int WC, bool RING = false>
int launch_halo(gl_ctx *ctx, const GlGatherConv &p)
{
    const int bx = p.W / BLK, by = p.H / BLK;
    const int64_t imgs = p.positions / ((int64_t)p.H * p.W);
    const int64_t total = imgs * bx * by;
    GL_REQUIRE(total < (1ll << 31), "halo_conv_h3: grid too large");
    constexpr int lds = WC * halo_bytes(4 * WC, RING) + (RING ? 3 : 2) * 64 * WC * 128;
    auto kern = halo_conv_h3_kernel<WC, RING>;
    GL_ONCE_PER_DEVICE(ctx, \
        GL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds)););
    gl_prof_scope prof_(ctx, GL_PROF_GATHER_CONV);
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256 * WC), lds, ctx->stream, p, bx, bx * by, (unsigned)total, gl_tuning_int("GL_HALO_DIAG", 0));
    GL_LAUNCH_CHECK();
    return GL_OK;
}

}  // namespace

bool gl_conv_halo_applies(const GlGatherConv &p, int phases)
{
    const int enabled = gl_tuning_int("GL_H3_HALO", 1);
    if (!enabled || phases != 1 || p.ntaps != 9 || p.tail_w || p.cols > 128 || p.Cin % 32 != 0) return false;
    uint32_t dy = 0, dx = 0;
    for (int t = 0; t < 9; ++t) { dy |= (uint32_t)(t / 3) << (2 * t); dx |= (uint32_t)(t % 3) << (2 * t); }
    if (p.tap_dy[0] != dy || p.tap_dx[0] != dx) return false;                      // the standard 3 x 3 tap order
    if (p.omul != 1 || p.Ho != p.H || p.Wo != p.W || p.oy[0] != 0 || p.ox[0] != 0 || p.planar || p.residual) return false;
    if (p.H % BLK != 0 || p.W % BLK != 0 || p.H < 32 || p.W < 32) return false;
    // enough 16 x 16 blocks to fill the chip several times over; small passes stay on the tap-gather kernel (same bits either way)
    return p.positions / 256 >= 1024;
}

int gl_launch_conv_halo_h3(gl_ctx *ctx, const GlGatherConv &p)
{
#ifdef GL_TUNING      // measured (round 3, tools/ab_halo_ring.py, alternating in one process): VGG16 features 1.000x, PGGAN-256 0.991x -- not the limiter; tuning build only
    if (gl_tuning_int("GL_HALO_RING", 0)) return p.cols <= 64 ? launch_halo<1, true>(ctx, p) : launch_halo<2, true>(ctx, p);
#endif
    return p.cols <= 64 ? launch_halo<1>(ctx, p) : launch_halo<2>(ctx, p);
}
