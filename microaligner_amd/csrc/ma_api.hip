// Context, memory, events and error plumbing of libmicroaligner_hip.so.
#include "ma_internal.h"

#include <cstring>
#include <map>
#include <mutex>

static thread_local std::string g_last_error;

void ma_set_error(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

extern "C" {

const char* ma_version(void) { return "microaligner_hip 0.1 (gfx950)"; }
const char* ma_last_error(void) { return g_last_error.c_str(); }

int ma_device_count(int* count)
{
    MA_REQUIRE(count != nullptr, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        ma_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return MA_ENODEV;
    }
    *count = n;
    return MA_OK;
}

int ma_ctx_create(int device, ma_ctx** out)
{
    MA_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        ma_set_error("no HIP device available");
        return MA_ENODEV;
    }
    MA_REQUIRE(device >= 0 && device < n, "device index out of range");
    MA_HIP(hipSetDevice(device));
    ma_ctx* ctx = new (std::nothrow) ma_ctx();
    if (!ctx) { ma_set_error("out of host memory"); return MA_ENOMEM; }
    ctx->device = device;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        ma_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        return MA_EHIP;
    }
    *out = ctx;
    return MA_OK;
}

void ma_ctx_destroy(ma_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& r : ctx->pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : ctx->free_events) (void)hipEventDestroy(e);
    for (auto& kv : ctx->pool_free) (void)hipFree(kv.second);
    for (auto& kv : ctx->pool_live) (void)hipFree(kv.first);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->dconst) (void)hipFree(ctx->dconst);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int ma_sync(ma_ctx* ctx)
{
    MA_REQUIRE(ctx, "ctx is NULL");
    MA_HIP(hipStreamSynchronize(ctx->stream));
    return MA_OK;
}

int ma_ctx_set_workspace_limit(ma_ctx* ctx, size_t bytes)
{
    MA_REQUIRE(ctx, "ctx is NULL");
    MA_REQUIRE(bytes >= ((size_t)1 << 20), "workspace limit must be >= 1 MiB");
    ctx->ws_limit = bytes;
    return MA_OK;
}

void* ma_ctx_stream(ma_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int ma_ctx_trim(ma_ctx* ctx)
{
    MA_REQUIRE(ctx, "ctx is NULL");
    MA_HIP(hipSetDevice(ctx->device));
    MA_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& kv : ctx->pool_free) (void)hipFree(kv.second);
    ctx->pool_free.clear();
    return MA_OK;
}

int ma_ctx_transfer_stats(ma_ctx* ctx, unsigned long long* h2d_bytes, unsigned long long* d2h_bytes, int reset)
{
    MA_REQUIRE(ctx, "ctx is NULL");
    if (h2d_bytes) *h2d_bytes = ctx->h2d_bytes;
    if (d2h_bytes) *d2h_bytes = ctx->d2h_bytes;
    if (reset) ctx->h2d_bytes = ctx->d2h_bytes = 0;
    return MA_OK;
}

int ma_malloc(ma_ctx* ctx, size_t bytes, void** dptr)
{
    MA_REQUIRE(ctx && dptr, "NULL argument");
    *dptr = nullptr;
    if (bytes == 0) bytes = 1;
    MA_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) {
        ma_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? MA_ENOMEM : MA_EHIP;
    }
    return MA_OK;
}

int ma_free(ma_ctx* ctx, void* dptr)
{
    MA_REQUIRE(ctx, "ctx is NULL");
    if (!dptr) return MA_OK;
    MA_HIP(hipSetDevice(ctx->device));
    MA_HIP(hipStreamSynchronize(ctx->stream));
    MA_HIP(hipFree(dptr));
    return MA_OK;
}

int ma_memcpy_h2d(ma_ctx* ctx, void* dst, const void* src_host, size_t bytes)
{
    MA_REQUIRE(ctx && (bytes == 0 || (dst && src_host)), "NULL argument");
    if (!bytes) return MA_OK;
    ctx->h2d_bytes += bytes;
    MA_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    // pageable host memory: the async copy is staged, return only when the source may be reused
    MA_HIP(hipStreamSynchronize(ctx->stream));
    return MA_OK;
}

int ma_memcpy_d2h(ma_ctx* ctx, void* dst_host, const void* src, size_t bytes)
{
    MA_REQUIRE(ctx && (bytes == 0 || (dst_host && src)), "NULL argument");
    if (!bytes) return MA_OK;
    ctx->d2h_bytes += bytes;
    MA_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MA_HIP(hipStreamSynchronize(ctx->stream));
    return MA_OK;
}

int ma_memcpy_d2h_async(ma_ctx* ctx, void* dst_host, const void* src, size_t bytes)
{
    MA_REQUIRE(ctx && (bytes == 0 || (dst_host && src)), "NULL argument");
    if (!bytes) return MA_OK;
    ctx->d2h_bytes += bytes;
    MA_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return MA_OK;
}

int ma_host_alloc(size_t bytes, void** hptr)
{
    MA_REQUIRE(hptr != nullptr, "hptr is NULL");
    *hptr = nullptr;
    if (bytes == 0) bytes = 1;
    hipError_t e = hipHostMalloc(hptr, bytes, hipHostMallocPortable);
    if (e != hipSuccess) {
        ma_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? MA_ENOMEM : MA_EHIP;
    }
    return MA_OK;
}

int ma_host_free(void* hptr)
{
    if (hptr) MA_HIP(hipHostFree(hptr));
    return MA_OK;
}

int ma_memcpy_d2d(ma_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    MA_REQUIRE(ctx && (bytes == 0 || (dst && src)), "NULL argument");
    if (!bytes) return MA_OK;
    MA_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return MA_OK;
}

int ma_memset(ma_ctx* ctx, void* dst, int value, size_t bytes)
{
    MA_REQUIRE(ctx && (bytes == 0 || dst), "NULL argument");
    if (!bytes) return MA_OK;
    MA_HIP(hipMemsetAsync(dst, value, bytes, ctx->stream));
    return MA_OK;
}

int ma_event_create(ma_ctx* ctx, void** ev)
{
    MA_REQUIRE(ctx && ev, "NULL argument");
    hipEvent_t e;
    MA_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return MA_OK;
}
int ma_event_destroy(ma_ctx* ctx, void* ev)
{
    MA_REQUIRE(ctx, "ctx is NULL");
    if (ev) MA_HIP(hipEventDestroy((hipEvent_t)ev));
    return MA_OK;
}
int ma_event_record(ma_ctx* ctx, void* ev)
{
    MA_REQUIRE(ctx && ev, "NULL argument");
    MA_HIP(hipEventRecord((hipEvent_t)ev, ctx->stream));
    return MA_OK;
}
int ma_event_elapsed_ms(ma_ctx* ctx, void* ev_start, void* ev_stop, float* ms)
{
    MA_REQUIRE(ctx && ev_start && ev_stop && ms, "NULL argument");
    MA_HIP(hipEventSynchronize((hipEvent_t)ev_stop));
    MA_HIP(hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return MA_OK;
}

int ma_profile_enable(ma_ctx* ctx, int on)
{
    MA_REQUIRE(ctx, "ctx is NULL");
    MA_TRY(ma_profile_flush(ctx));
    ctx->profile = on != 0;
    return MA_OK;
}
int ma_profile_reset(ma_ctx* ctx)
{
    MA_REQUIRE(ctx, "ctx is NULL");
    MA_TRY(ma_profile_flush(ctx));
    for (int i = 0; i < MA_K_COUNT; i++) { ctx->prof_ms[i] = 0; ctx->prof_n[i] = 0; ctx->prof_px[i] = 0; }
    return MA_OK;
}
int ma_profile_get(ma_ctx* ctx, int kernel_id, double* total_ms, long long* launches, double* px)
{
    MA_REQUIRE(ctx && kernel_id >= 0 && kernel_id < MA_K_COUNT, "bad kernel id");
    MA_TRY(ma_profile_flush(ctx));
    if (total_ms) *total_ms = ctx->prof_ms[kernel_id];
    if (launches) *launches = ctx->prof_n[kernel_id];
    if (px) *px = ctx->prof_px[kernel_id];
    return MA_OK;
}

} // extern "C"

int ma_profile_flush(ma_ctx* ctx)
{
    if (ctx->pending.empty()) return MA_OK;
    MA_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& r : ctx->pending) {
        float ms = 0;
        MA_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        ctx->prof_ms[r.id] += ms;
        ctx->free_events.push_back(r.a);
        ctx->free_events.push_back(r.b);
    }
    ctx->pending.clear();
    return MA_OK;
}

static hipEvent_t take_event(ma_ctx* ctx)
{
    if (!ctx->free_events.empty()) {
        hipEvent_t e = ctx->free_events.back();
        ctx->free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

MaProfScope::MaProfScope(ma_ctx* c, int kid, double px) : ctx(c), id(kid), on(c->profile)
{
    if (!on) return;
    ctx->prof_n[id] += 1;
    ctx->prof_px[id] += px;
    a = take_event(ctx);
    b = take_event(ctx);
    (void)hipEventRecord(a, ctx->stream);
}
MaProfScope::~MaProfScope()
{
    if (!on) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->pending.push_back({a, b, id});
    if (ctx->pending.size() > 4096) (void)ma_profile_flush(ctx);
}

static int reserve(void** p, size_t* have, size_t want, bool pinned, ma_ctx* ctx)
{
    if (*have >= want) return MA_OK;
    MA_HIP(hipStreamSynchronize(ctx->stream));
    if (*p) {
        if (pinned) MA_HIP(hipHostFree(*p)); else MA_HIP(hipFree(*p));
        *p = nullptr; *have = 0;
    }
    hipError_t e = pinned ? hipHostMalloc(p, want, hipHostMallocDefault) : hipMalloc(p, want);
    if (e != hipSuccess) {
        ma_set_error("%s(%zu) failed: %s", pinned ? "hipHostMalloc" : "hipMalloc", want, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? MA_ENOMEM : MA_EHIP;
    }
    *have = want;
    return MA_OK;
}

int ma_ws_reserve(ma_ctx* ctx, size_t bytes) { return reserve(&ctx->ws, &ctx->ws_bytes, bytes, false, ctx); }
int ma_pinned_reserve(ma_ctx* ctx, size_t bytes)
{
    return reserve(&ctx->pinned, &ctx->pinned_bytes, ma_align_up(bytes, 4096), true, ctx);
}
int ma_dconst_reserve(ma_ctx* ctx, size_t bytes)
{
    return reserve(&ctx->dconst, &ctx->dconst_bytes, ma_align_up(bytes, 4096), false, ctx);
}

void* ma_pool_alloc(ma_ctx* ctx, size_t bytes)
{
    const size_t bucket = ma_align_up(bytes ? bytes : 1, (size_t)1 << 16);
    auto it = ctx->pool_free.find(bucket);
    void* p = nullptr;
    if (it != ctx->pool_free.end()) {
        p = it->second;
        ctx->pool_free.erase(it);
    } else {
        hipError_t e = hipMalloc(&p, bucket);
        if (e == hipErrorOutOfMemory && !ctx->pool_free.empty()) {
            // hand the cached buffers of other sizes back and try once more
            (void)hipStreamSynchronize(ctx->stream);
            for (auto& kv : ctx->pool_free) (void)hipFree(kv.second);
            ctx->pool_free.clear();
            e = hipMalloc(&p, bucket);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            ma_set_error("hipMalloc(%zu) failed: %s", bucket, hipGetErrorString(e));
            return nullptr;
        }
    }
    ctx->pool_live[p] = bucket;
    return p;
}

void ma_pool_free(ma_ctx* ctx, void* p)
{
    if (!p) return;
    auto it = ctx->pool_live.find(p);
    if (it == ctx->pool_live.end()) return;
    ctx->pool_free.emplace(it->second, p);
    ctx->pool_live.erase(it);
}

static std::mutex g_table_mutex;
static std::map<std::pair<int, uint64_t>, float*> g_tables;

int ma_const_table(ma_ctx* ctx, uint64_t key, const float* host, size_t n, const float** dev)
{
    std::lock_guard<std::mutex> lock(g_table_mutex);
    auto k = std::make_pair(ctx->device, key);
    auto it = g_tables.find(k);
    if (it != g_tables.end()) { *dev = it->second; return MA_OK; }
    float* d = nullptr;
    MA_HIP(hipMalloc((void**)&d, n * sizeof(float)));
    MA_HIP(hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice));
    g_tables[k] = d;
    *dev = d;
    return MA_OK;
}
