// Internal definitions shared by the HIP translation units of libganleaks_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../include/ganleaks.h"

struct gl_prof_span {
    int tag;
    hipEvent_t start, stop;
};

struct gl_ctx {
    int device;
    hipStream_t own_stream;
    hipStream_t stream;      // the stream work is enqueued on (own_stream or the caller's)
    float *zero_page;        // 4 KiB of zeros on the device: source for out-of-image taps
    int *h3_sat;             // device counter: split-fp16 stores that had to clamp to the fp16 range (gl_ctx_h3_saturations)
    int num_cu;              // compute units of the device
    char *pair_scratch;      // lazily allocated workspace of the persistent pairwise kernel: per-workgroup fp32 totals + cluster counters
    size_t pair_scratch_bytes;
    // arena: device blocks of >= 256 MiB that gl_free keeps for the next gl_malloc of (almost) the same size instead of returning them to the
    // driver (allocating and freeing the 153 GiB of query rows of a 256 x 256 attack costs 2-7 s per call); gl_ctx_trim releases them
    std::mutex *arena_mu;
    std::vector<std::pair<size_t, void *>> arena_free;                 // (bytes, block) not in use
    std::vector<std::pair<void *, size_t>> arena_live;                 // blocks handed out by gl_malloc that gl_free may keep
    bool prof_on;            // gl_prof_enable: bracket tagged kernel launches with HIP events
    std::vector<gl_prof_span> prof_spans;
    std::vector<hipEvent_t> prof_pool;
};

// RAII bracket around one kernel launch; a no-op unless profiling is enabled on the context.
struct gl_prof_scope {
    gl_ctx *ctx;
    gl_prof_span span;
    bool active;
    gl_prof_scope(gl_ctx *c, int tag);
    ~gl_prof_scope();
};

void gl_set_error(const char *fmt, ...);
// hipMalloc for the library's own workspaces: when the device is out of memory, the context's cached arena blocks are released and the
// allocation is tried once more
hipError_t gl_device_alloc(gl_ctx *ctx, void **out, size_t bytes);

#define GL_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            gl_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return GL_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

#define GL_REQUIRE(cond, ...)                                                                \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            gl_set_error(__VA_ARGS__);                                                       \
            return GL_ERR_INVALID;                                                           \
        }                                                                                    \
    } while (0)

#define GL_LAUNCH_CHECK()                                                                    \
    do {                                                                                     \
        hipError_t _e = hipGetLastError();                                                   \
        if (_e != hipSuccess) {                                                              \
            gl_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return GL_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

static inline int64_t gl_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Tuning switches.  The shipped library has ONE code path per shape: gl_tuning_int() is the constant `dflt` there and the experiment
// kernels (e.g. the DIAG instantiations of gl_pair256.h, whose results are wrong on purpose) are not compiled in.  `make tuning` builds
// libganleaks_hip_tuning.so with -DGL_TUNING, where the named environment variable is read on every call (tools/bench_pairwise.py and the
// other A/B tools alternate variants inside one process and load that library through $GANLEAKS_LIB).
#ifdef GL_TUNING
#include <cstdlib>
static inline int gl_tuning_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
#else
static constexpr int gl_tuning_int(const char *, int dflt) { return dflt; }
#endif

// every entry point runs on its context's device, whatever device the calling thread had current (a host thread may drive several contexts,
// and PyTorch may have switched devices in between)
static inline void gl_make_current(const gl_ctx *ctx)
{
    if (!ctx) return;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != ctx->device) (void)hipSetDevice(ctx->device);
}

// run `body` once per device ordinal (function attributes such as the dynamic-LDS limit are per device, and a process may hold
// contexts on several GPUs); `body` may use GL_HIP (returns on error; the lock is released by the guard)
#define GL_ONCE_PER_DEVICE(ctx, body)                                                        \
    do {                                                                                     \
        static std::mutex _mu;                                                               \
        static unsigned _mask = 0;                                                           \
        std::lock_guard<std::mutex> _lk(_mu);                                                \
        const unsigned _bit = 1u << ((ctx)->device & 31);                                    \
        if (!(_mask & _bit)) {                                                               \
            body;                                                                            \
            _mask |= _bit;                                                                   \
        }                                                                                    \
    } while (0)

// exact-integer L2 path: S = sum_k (u_q - u_n)^2 <= 65025 d, packed as key = S << shift | global index.  The key stays below 2^63
// because the cross-GPU minimum runs on the int64 view of the keys; shift is 32 for d <= 33025 (everything up to 3 x 104 x 104)
// and shrinks by one bit per doubling of d beyond that (29 at 3 x 256 x 256, leaving 2^29 bank rows).
constexpr int64_t GL_L2_MAX_D = 262143;          // row norms sum (u-128)^2 <= 16384 d must fit 32 unsigned bits
static inline int gl_l2_key_shift(int64_t d)
{
    const unsigned long long smax = 65025ull * (unsigned long long)d;
    const int bits = 64 - __builtin_clzll(smax);
    return 63 - bits > 32 ? 32 : 63 - bits;
}

// XCD-aware block id remap (8 XCDs, blocks are dealt round-robin): gives every XCD a contiguous
// chunk of the logical grid so blocks sharing an operand panel share an L2.  Bijective for any nwg.
__device__ __forceinline__ unsigned gl_xcd_remap(unsigned bid, unsigned nwg)
{
    const unsigned xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
    const unsigned start = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return start + (bid >> 3);
}

// pairwise kernels: logical block id -> (bank tile nt, query tile qt) in strips of STRIP bank tiles with the bank
// tile varying fastest.  The ~64 blocks an XCD runs at once then cover ~8 bank tiles x 8 query tiles, so each
// K slice needs 16 operand panels from beyond L2 instead of 65 (1 bank + 64 query panels) -- measured with
// FETCH_SIZE: 107 GB -> see profiles/.
__device__ __forceinline__ void gl_strip_order(unsigned id, int q_tiles, int n_tiles, int &qt, int &nt)
{
    constexpr int STRIP = 8;
    const unsigned per_strip = (unsigned)STRIP * (unsigned)q_tiles;
    const int strip = (int)(id / per_strip);
    const unsigned r = id % per_strip;
    const int width = n_tiles - strip * STRIP < STRIP ? n_tiles - strip * STRIP : STRIP;   // last strip may be narrower
    nt = strip * STRIP + (int)(r % (unsigned)width);
    qt = (int)(r / (unsigned)width);
}

typedef __attribute__((address_space(1))) const void *gl_gptr;
typedef __attribute__((address_space(3))) void *gl_lptr;

// async global -> LDS, 16 B per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void gl_glds16(const void *src, void *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((gl_gptr)src, (gl_lptr)lds_wave_base, 16, 0, 0);
}
