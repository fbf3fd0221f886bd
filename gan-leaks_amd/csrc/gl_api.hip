// Context, memory and event entry points of the C ABI (include/ganleaks.h).
#include "gl_common.h"
#include <cstring>

static thread_local char g_err[512] = "";

void gl_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static hipEvent_t prof_event(gl_ctx *c)
{
    if (!c->prof_pool.empty()) {
        hipEvent_t e = c->prof_pool.back();
        c->prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

gl_prof_scope::gl_prof_scope(gl_ctx *c, int tag) : ctx(c), active(c && c->prof_on)
{
    if (!active) return;
    span.tag = tag;
    span.start = prof_event(c);
    span.stop = prof_event(c);
    (void)hipEventRecord(span.start, c->stream);
}

gl_prof_scope::~gl_prof_scope()
{
    if (!active) return;
    (void)hipEventRecord(span.stop, ctx->stream);
    ctx->prof_spans.push_back(span);
}

hipError_t gl_device_alloc(gl_ctx *ctx, void **out, size_t bytes)
{
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
        (void)hipGetLastError();
        bool any;
        {
            std::lock_guard<std::mutex> lk(*ctx->arena_mu);
            any = !ctx->arena_free.empty();
        }
        if (any) {
            (void)gl_ctx_trim(ctx);
            e = hipMalloc(out, bytes);
        }
    }
    return e;
}

extern "C" {

int gl_prof_enable(gl_ctx *ctx, int on)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx, "gl_prof_enable: NULL ctx");
    ctx->prof_on = on != 0;
    return GL_OK;
}

int gl_prof_read(gl_ctx *ctx, int tag, double *out_total_ms, int64_t *out_launches)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && out_total_ms && out_launches, "gl_prof_read: NULL argument");
    GL_HIP(hipStreamSynchronize(ctx->stream));
    double total = 0.0;
    int64_t n = 0;
    for (const gl_prof_span &s : ctx->prof_spans)
        if (s.tag == tag) {
            float ms = 0.f;
            GL_HIP(hipEventElapsedTime(&ms, s.start, s.stop));
            total += ms;
            ++n;
        }
    *out_total_ms = total;
    *out_launches = n;
    return GL_OK;
}

int gl_prof_reset(gl_ctx *ctx)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx, "gl_prof_reset: NULL ctx");
    GL_HIP(hipStreamSynchronize(ctx->stream));
    for (const gl_prof_span &s : ctx->prof_spans) {
        ctx->prof_pool.push_back(s.start);
        ctx->prof_pool.push_back(s.stop);
    }
    ctx->prof_spans.clear();
    return GL_OK;
}

int gl_abi_version(void) { return GL_ABI_VERSION; }
const char *gl_last_error(void) { return g_err; }

int gl_device_count(int *out_count)
{
    GL_REQUIRE(out_count, "gl_device_count: NULL out_count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { n = 0; (void)hipGetLastError(); }
    *out_count = n;
    return GL_OK;
}

int gl_ctx_create(int device, gl_ctx **out_ctx)
{
    GL_REQUIRE(out_ctx, "gl_ctx_create: NULL out_ctx");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        gl_set_error("gl_ctx_create: no HIP device visible (this library has no CPU fallback)");
        return GL_ERR_NO_DEVICE;
    }
    GL_REQUIRE(device >= 0 && device < n, "gl_ctx_create: device %d out of range [0,%d)", device, n);
    GL_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    GL_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        gl_set_error("gl_ctx_create: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return GL_ERR_NO_DEVICE;
    }
    gl_ctx *c = new gl_ctx();
    c->device = device;
    c->prof_on = false;
    c->num_cu = prop.multiProcessorCount;
    c->pair_scratch = nullptr;
    c->pair_scratch_bytes = 0;
    c->arena_mu = new std::mutex();
    GL_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    GL_HIP(hipMalloc((void **)&c->zero_page, 4096));
    GL_HIP(hipMemsetAsync(c->zero_page, 0, 4096, c->stream));
    GL_HIP(hipMalloc((void **)&c->h3_sat, 64));
    GL_HIP(hipMemsetAsync(c->h3_sat, 0, 64, c->stream));
    GL_HIP(hipStreamSynchronize(c->stream));
    *out_ctx = c;
    return GL_OK;
}

int gl_ctx_destroy(gl_ctx *ctx)
{
    gl_make_current(ctx);
    if (!ctx) return GL_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)gl_prof_reset(ctx);
    for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
    (void)hipFree(ctx->zero_page);
    (void)hipFree(ctx->h3_sat);
    (void)hipFree(ctx->pair_scratch);
    (void)gl_ctx_trim(ctx);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx->arena_mu;
    delete ctx;
    return GL_OK;
}

int gl_ctx_set_stream(gl_ctx *ctx, void *hip_stream)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx, "gl_ctx_set_stream: NULL ctx");
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return GL_OK;
}

int gl_ctx_get_stream(gl_ctx *ctx, void **out_hip_stream)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && out_hip_stream, "gl_ctx_get_stream: NULL argument");
    *out_hip_stream = (void *)ctx->stream;
    return GL_OK;
}

int gl_ctx_sync(gl_ctx *ctx)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx, "gl_ctx_sync: NULL ctx");
    GL_HIP(hipStreamSynchronize(ctx->stream));
    return GL_OK;
}

int gl_ctx_h3_saturations(gl_ctx *ctx, int64_t *out_count)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && out_count, "gl_ctx_h3_saturations: NULL argument");
    int host = 0;
    GL_HIP(hipMemcpyAsync(&host, ctx->h3_sat, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    GL_HIP(hipMemsetAsync(ctx->h3_sat, 0, sizeof(int), ctx->stream));
    GL_HIP(hipStreamSynchronize(ctx->stream));
    *out_count = host;
    return GL_OK;
}

constexpr size_t kArenaMin = 256ull << 20;      // blocks below this go straight back to the driver

int gl_ctx_trim(gl_ctx *ctx)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx, "gl_ctx_trim: NULL ctx");
    std::vector<std::pair<size_t, void *>> blocks;
    {
        std::lock_guard<std::mutex> lk(*ctx->arena_mu);
        blocks.swap(ctx->arena_free);
    }
    if (!blocks.empty()) (void)hipStreamSynchronize(ctx->stream);
    for (auto &b : blocks) (void)hipFree(b.second);
    return GL_OK;
}

int gl_mem_info(gl_ctx *ctx, size_t *out_available, size_t *out_total)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && out_available && out_total, "gl_mem_info: NULL argument");
    GL_HIP(hipSetDevice(ctx->device));
    size_t free_b = 0, total_b = 0;
    GL_HIP(hipMemGetInfo(&free_b, &total_b));
    {
        std::lock_guard<std::mutex> lk(*ctx->arena_mu);
        for (auto &b : ctx->arena_free) free_b += b.first;
    }
    *out_available = free_b;
    *out_total = total_b;
    return GL_OK;
}

int gl_malloc(gl_ctx *ctx, size_t bytes, void **out_dev)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && out_dev, "gl_malloc: NULL argument");
    GL_HIP(hipSetDevice(ctx->device));
    *out_dev = nullptr;
    if (bytes == 0) return GL_OK;
    if (bytes >= kArenaMin) {
        std::lock_guard<std::mutex> lk(*ctx->arena_mu);
        // the smallest kept block that holds the request without wasting more than an eighth of it
        size_t best = ctx->arena_free.size();
        for (size_t i = 0; i < ctx->arena_free.size(); ++i) {
            const size_t have = ctx->arena_free[i].first;
            if (have >= bytes && have - bytes <= bytes / 8 && (best == ctx->arena_free.size() || have < ctx->arena_free[best].first)) best = i;
        }
        if (best != ctx->arena_free.size()) {
            *out_dev = ctx->arena_free[best].second;
            ctx->arena_live.emplace_back(*out_dev, ctx->arena_free[best].first);
            ctx->arena_free.erase(ctx->arena_free.begin() + (long)best);
            return GL_OK;
        }
    }
    GL_HIP(gl_device_alloc(ctx, out_dev, bytes));
    if (bytes >= kArenaMin) {
        std::lock_guard<std::mutex> lk(*ctx->arena_mu);
        ctx->arena_live.emplace_back(*out_dev, bytes);
    }
    return GL_OK;
}

int gl_free(gl_ctx *ctx, void *dev)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx, "gl_free: NULL ctx");
    if (!dev) return GL_OK;
    GL_HIP(hipStreamSynchronize(ctx->stream));
    {
        std::lock_guard<std::mutex> lk(*ctx->arena_mu);
        for (size_t i = 0; i < ctx->arena_live.size(); ++i)
            if (ctx->arena_live[i].first == dev) {
                ctx->arena_free.emplace_back(ctx->arena_live[i].second, dev);      // kept for the next request of this size (gl_ctx_trim releases)
                ctx->arena_live.erase(ctx->arena_live.begin() + (long)i);
                return GL_OK;
            }
    }
    GL_HIP(hipFree(dev));
    return GL_OK;
}

int gl_memcpy_h2d(gl_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && (bytes == 0 || (dst_dev && src_host)), "gl_memcpy_h2d: NULL argument");
    if (bytes == 0) return GL_OK;
    GL_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    GL_HIP(hipStreamSynchronize(ctx->stream));
    return GL_OK;
}

int gl_memcpy_d2h(gl_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && (bytes == 0 || (dst_host && src_dev)), "gl_memcpy_d2h: NULL argument");
    if (bytes == 0) return GL_OK;
    GL_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    GL_HIP(hipStreamSynchronize(ctx->stream));
    return GL_OK;
}

int gl_memset(gl_ctx *ctx, void *dev, int value, size_t bytes)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && (bytes == 0 || dev), "gl_memset: NULL argument");
    if (bytes == 0) return GL_OK;
    GL_HIP(hipMemsetAsync(dev, value, bytes, ctx->stream));
    return GL_OK;
}

int gl_event_create(void **out_event)
{
    GL_REQUIRE(out_event, "gl_event_create: NULL out_event");
    hipEvent_t e;
    GL_HIP(hipEventCreate(&e));
    *out_event = (void *)e;
    return GL_OK;
}

int gl_event_destroy(void *event)
{
    if (!event) return GL_OK;
    GL_HIP(hipEventDestroy((hipEvent_t)event));
    return GL_OK;
}

int gl_event_record(gl_ctx *ctx, void *event)
{
    gl_make_current(ctx);
    GL_REQUIRE(ctx && event, "gl_event_record: NULL argument");
    GL_HIP(hipEventRecord((hipEvent_t)event, ctx->stream));
    return GL_OK;
}

int gl_event_elapsed_ms(void *start, void *stop, float *out_ms)
{
    GL_REQUIRE(start && stop && out_ms, "gl_event_elapsed_ms: NULL argument");
    GL_HIP(hipEventSynchronize((hipEvent_t)stop));
    GL_HIP(hipEventElapsedTime(out_ms, (hipEvent_t)start, (hipEvent_t)stop));
    return GL_OK;
}

}  // extern "C"
