// Shared main loop of the two 256 x 256 pairwise kernels (l2_knn_i8_256_kernel: int8 codes, feat_knn_h1_kernel: fp16 search rows):
//     C[n][q] += sum_k A[n][k] * B[q][k]      A = 256 bank rows, B = 256 query rows, both K-contiguous ("NT" GEMM)
// 8 waves as 2 (bank) x 4 (query); a wave owns 128 bank rows x 64 queries = 8 x 4 MFMA tiles of 16 x 16.
//
// K is streamed in slices of 128 bytes per row (two MFMA k-steps of 64 bytes), 64 KiB per slice for both operands, two slice
// buffers in 128 KiB of LDS filled by LDS-DMA (global_load_lds, 16 B per lane).  LDS rows are 128 B; the 16-byte chunk c of row r
// sits at slot c ^ (r & 7) -- applied on the per-lane SOURCE address, the LDS image of a wave-instruction is lane-linear -- which
// makes every ds_read_b128 of an MFMA operand conflict-free.
//
// Schedule (one workgroup barrier per slice, at the slice's midpoint):
//     top of slice t : set0 = fragments of k-step 0 of slice t (read during the previous slice)
//       read k-step 1 of slice t            -> set1        (LDS reads overlap the MFMAs below)
//       32 MFMAs on set0
//       wait: own reads returned, own LDS-DMA of slice t+1 landed;  s_barrier
//       start LDS-DMA of slice t+2 into the buffer of slice t (everyone is done reading it)
//       read k-step 0 of slice t+1          -> set0
//       32 MFMAs on set1
// with the reads and the DMA pieces spread over the groups of 4 MFMAs (see mainloop), so the fragment reads of the next MFMA block
// are always issued during the current block and no wave sits behind a barrier with an empty matrix pipe.
#pragma once
#include "gl_common.h"

namespace gl_pair256 {

constexpr int TILE = 256;                 // rows per operand per workgroup
constexpr int ROW = 128;                  // bytes of K per row and slice
constexpr int OPER = TILE * ROW;          // 32 KiB per operand per slice
constexpr int SLICE = 2 * OPER;           // 64 KiB
constexpr int LDS_BYTES = 2 * SLICE;      // 128 KiB
constexpr int THREADS = 512;

// Source addressing of one operand for one wave: 4 pieces of 8 rows (1 KiB each) per slice.  Piece i of wave w covers tile rows
// (4w+i)*8 .. +7; lane -> (row rsub = lane >> 3, LDS slot = lane & 7), the slot holds global chunk slot ^ rsub.
// Rows at or beyond `valid` (ragged last tile) are clamped to the last valid row; the epilogue masks them.
struct Source {
    const char *base[4];      // wave-uniform: first row of the piece (clamped), K offset 0
    uint32_t off[4];          // per lane: (row within piece, clamped) * row_bytes + chunk * 16
};

__device__ __forceinline__ Source make_source(const char *rows, int64_t row0, int64_t valid, int64_t row_bytes, int wave, int lane)
{
    Source s;
    const int rsub = lane >> 3, slot = lane & 7;
    const uint32_t chunk = (uint32_t)((slot ^ rsub) << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t first = row0 + (wave * 4 + i) * 8;
        if (first >= valid) first = valid - 1;
        int64_t r = first + rsub;
        if (r >= valid) r = valid - 1;
        s.base[i] = rows + first * row_bytes;
        s.off[i] = (uint32_t)((r - first) * row_bytes) + chunk;
    }
    return s;
}

// The same for rows in the K-blocked layout (gl_conv.h gl_vrow_elem: [row / 256][K / 64][row % 256][128 B]; row0 is a multiple of 256): the
// K slice of a tile is one contiguous 32 KiB piece, a wave's piece i its rows (4w+i)*8 .. +7, i.e. 1 KiB of it.  Consecutive slices of a tile lie
// 32 KiB apart (the caller passes kstep = 32768 to mainloop instead of ROW); a block of 256 rows is `cells` * 32 KiB long.  The buffer holds whole blocks, so no row is clamped; rows
// beyond `valid` are whatever the buffer holds and the epilogue masks them.
__device__ __forceinline__ Source make_source_blocked(const char *rows, int64_t row0, int64_t cells, int wave, int lane)
{
    Source s;
    const int rsub = lane >> 3, slot = lane & 7;
    const char *tile = rows + (row0 >> 8) * cells * 32768;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s.base[i] = tile + (wave * 4 + i) * 1024;
        s.off[i] = (uint32_t)(rsub * 128 + ((slot ^ rsub) << 4));
    }
    return s;
}

__device__ __forceinline__ void stage(const Source &a, const Source &b, int64_t kbyte, char *buf, int wave)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) gl_glds16(a.base[i] + kbyte + a.off[i], buf + (wave * 4 + i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) gl_glds16(b.base[i] + kbyte + b.off[i], buf + OPER + (wave * 4 + i) * 1024);
}

// One DMA piece of a slice: piece 0..3 = operand A, 4..7 = operand B.
__device__ __forceinline__ void stage_piece(const Source &a, const Source &b, int64_t kbyte, char *buf, int wave, int piece)
{
    const Source &s = piece < 4 ? a : b;
    const int i = piece & 3;
    gl_glds16(s.base[i] + kbyte + s.off[i], buf + (piece < 4 ? 0 : OPER) + (wave * 4 + i) * 1024);
}

// Frag: 16-byte MFMA operand type (v4i for int8, v8h for fp16).  mfma(a, b, c) -> c.  kbyte0: K offset (bytes) of the first slice.
// DIAG (timing experiments only, results are wrong): 1 = no LDS-DMA inside the loop, 2 = also no barrier, 3 = also no fragment reads,
// 4 = LDS-DMA of the same two slices over and over (always cache hits)
// SPREAD: the 8 DMA pieces of a slice are issued one (8) or two (4) per group of 4 MFMAs after the barrier, or all at once (1)
template <typename Frag, int DIAG = 0, int SPREAD = 8, typename Acc, typename Mfma>
__device__ __forceinline__ void mainloop(const Source &sa, const Source &sb, int64_t nk, char *smem, Acc (&acc)[8][4], int wave, int lane, Mfma mfma,
                                         int64_t kbyte0 = 0, int64_t kstep = ROW)
{
    const int wn = wave >> 2, wq = wave & 3;
    const int frow = lane & 15, fk = lane >> 4;
    // fragment of tile row i*16 + frow, k-step ks: chunk 4*ks + fk, stored at slot chunk ^ (row & 7) = chunk ^ (frow & 7)
    const int oa = (wn * 128 + frow) * ROW, ob = OPER + (wq * 64 + frow) * ROW;
    const int c0 = ((fk ^ (frow & 7)) << 4), c1 = (((4 + fk) ^ (frow & 7)) << 4);

    // Fragment reads are inline asm so that their place in the instruction stream and their waits are set by hand: hipcc's own
    // scoreboard falls back to lgkmcnt(0) at the loop head and its scheduler clusters the reads and the DMAs into bursts, during
    // which both waves of a SIMD issue no MFMA.  Here every group of 4 MFMAs carries at most one DMA piece and two fragment reads,
    // so the two waves of a SIMD dovetail.  LDS reads return in issue order; every use of a fragment set is preceded by an explicit
    // s_waitcnt + sched_barrier (hipcc moves register-only MFMAs across an inline-asm wait otherwise).
    typedef __attribute__((address_space(3))) char *lds_ptr;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)smem;
    const uint32_t pa0 = lds0 + oa + c0, pa1 = lds0 + oa + c1, pb0 = lds0 + ob + c0, pb1 = lds0 + ob + c1;
    Frag a0[8], b0[4], a1[8], b1[4];
#define GL_P256_READ(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
    // read number r (0..11) of a fragment set: the 4 B fragments first (every MFMA group needs all of them), then A in group order
    auto read1 = [&](Frag (&a)[8], Frag (&b)[4], uint32_t pa, uint32_t pb, int r) {
        switch (r) {
        case 0: GL_P256_READ(b[0], pb, 0 * 2048); break;
        case 1: GL_P256_READ(b[1], pb, 1 * 2048); break;
        case 2: GL_P256_READ(b[2], pb, 2 * 2048); break;
        case 3: GL_P256_READ(b[3], pb, 3 * 2048); break;
        case 4: GL_P256_READ(a[0], pa, 0 * 2048); break;
        case 5: GL_P256_READ(a[1], pa, 1 * 2048); break;
        case 6: GL_P256_READ(a[2], pa, 2 * 2048); break;
        case 7: GL_P256_READ(a[3], pa, 3 * 2048); break;
        case 8: GL_P256_READ(a[4], pa, 4 * 2048); break;
        case 9: GL_P256_READ(a[5], pa, 5 * 2048); break;
        case 10: GL_P256_READ(a[6], pa, 6 * 2048); break;
        default: GL_P256_READ(a[7], pa, 7 * 2048); break;
        }
    };
    auto read_all = [&](Frag (&a)[8], Frag (&b)[4], uint32_t pa, uint32_t pb) {
#pragma unroll
        for (int r = 0; r < 12; ++r) read1(a, b, pa, pb, r);
    };
#define GL_P256_MMA4(A, B, g)                                                                                     \
    do {                                                                                                          \
        acc[g][0] = mfma(A[g], B[0], acc[g][0]); acc[g][1] = mfma(A[g], B[1], acc[g][1]);                         \
        acc[g][2] = mfma(A[g], B[2], acc[g][2]); acc[g][3] = mfma(A[g], B[3], acc[g][3]);                         \
    } while (0)

    if (nk <= 0) return;
    stage(sa, sb, kbyte0, smem, wave);
    if (nk > 1) {
        stage(sa, sb, kbyte0 + kstep, smem + SLICE, wave);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // slice 0 landed, the 8 DMAs of slice 1 stay in flight
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_all(a0, b0, pa0, pb0);
    for (int64_t kt = 0; kt < nk; ++kt) {
        const uint32_t cur = (uint32_t)(kt & 1) * SLICE, nxt = SLICE - cur;
        const bool more = kt + 2 < nk;
        const int64_t kb = kbyte0 + (DIAG == 4 ? (kt & 1) * kstep : (kt + 2) * kstep);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // set0 (read during the previous half slice) is in
        __builtin_amdgcn_sched_barrier(0);
        // ---- first half: 32 MFMAs on set0 (k-step 0 of slice kt); k-step 1 is read into set1 two fragments per group
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if constexpr (DIAG < 3 || DIAG == 4)
                if (g < 6) { read1(a1, b1, pa1 + cur, pb1 + cur, 2 * g); read1(a1, b1, pa1 + cur, pb1 + cur, 2 * g + 1); }
            __builtin_amdgcn_sched_barrier(0);
            GL_P256_MMA4(a0, b0, g);
            __builtin_amdgcn_sched_barrier(0);
        }
        // own fragment reads of this slice's buffer have returned; own DMAs of slice kt+1 (issued during the previous slice) have landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if constexpr (DIAG < 2 || DIAG == 4) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- second half: 32 MFMAs on set1; the DMA of slice kt+2 into this slice's buffer and the reads of k-step 0 of slice kt+1
        //      (stale bytes after the last slice, never used) ride along
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if constexpr (DIAG < 1 || DIAG == 4) {
                if (more) {
                    if constexpr (SPREAD == 8) stage_piece(sa, sb, kb, smem + cur, wave, g);
                    if constexpr (SPREAD == 4) if (g < 4) { stage_piece(sa, sb, kb, smem + cur, wave, 2 * g); stage_piece(sa, sb, kb, smem + cur, wave, 2 * g + 1); }
                    if constexpr (SPREAD == 1) if (g == 0) stage(sa, sb, kb, smem + cur, wave);
                }
            }
            if constexpr (DIAG < 3 || DIAG == 4)
                if (g < 6) { read1(a0, b0, pa0 + nxt, pb0 + nxt, 2 * g); read1(a0, b0, pa0 + nxt, pb0 + nxt, 2 * g + 1); }
            __builtin_amdgcn_sched_barrier(0);
            GL_P256_MMA4(a1, b1, g);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef GL_P256_READ
#undef GL_P256_MMA4
}

}  // namespace gl_pair256
