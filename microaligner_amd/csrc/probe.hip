// Measurement helpers of bench.py (not on the data path): identity of the device a rank drives, and the shader clock
// the chip sustains under a packed-FP32 load of the kind the window-blur kernels issue.
#include "ma_internal.h"

#include <cstring>

namespace {

// Every lane keeps 8 independent chains of v_pk_mul_f32 / v_pk_add_f32 busy (the instruction mix of the symmetric FIR:
// one multiply per two additions, and like the FIR two 8-byte LDS reads per 24 packed operations feed two of the
// chains), 4 waves per SIMD on every CU.  Lane 0 of every block stamps the shader clock (s_memtime) and the 100 MHz
// constant clock (s_memrealtime) around its loop: their ratio is the clock the chip held.
__global__ __launch_bounds__(256) void clock_probe_kernel(int iters, float seed, float* __restrict__ sink,
                                                          unsigned long long* __restrict__ stamps)
{
    __shared__ float tile[64 * 65];
    for (int i = threadIdx.x; i < 64 * 65; i += 256) tile[i] = seed * 1e-4f * (float)(i & 63);
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(tile + (threadIdx.x & 63) * 65);
    ma_f2 a[8];
    const ma_f2 k = {seed * 0.999f, seed * 1.001f};
    ma_f2 c = {seed * 1e-3f, -seed * 1e-3f};
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = (ma_f2){seed + i + threadIdx.x, seed - i};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        ma_f2 p, q;
        asm volatile("ds_read2_b32 %0, %1 offset0:3 offset1:10" : "=v"(p) : "v"(addr));
        asm volatile("ds_read2_b32 %0, %1 offset0:17 offset1:24" : "=v"(q) : "v"(addr));
#pragma unroll
        for (int i = 0; i < 8; i++) {
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p), "+v"(q));
        c = c + (p - q) * (ma_f2){1e-30f, 1e-30f};
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

}  // namespace

extern "C" {

int ma_device_info(int device, char* name, size_t name_len, char* pci_bus_id, size_t pci_len, size_t* mem_free,
                   size_t* mem_total, int* compute_units)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { ma_set_error("no HIP device available"); return MA_ENODEV; }
    MA_REQUIRE(device >= 0 && device < n, "device index out of range");
    hipDeviceProp_t prop;
    MA_HIP(hipGetDeviceProperties(&prop, device));
    if (name && name_len) {
        // some driver stacks leave the marketing name empty: fall back to the ISA name so that a rank row is never anonymous
        if (prop.name[0]) std::snprintf(name, name_len, "%s", prop.name);
        else std::snprintf(name, name_len, "AMD GPU (%s, %d CUs)", prop.gcnArchName, prop.multiProcessorCount);
    }
    if (pci_bus_id && pci_len) MA_HIP(hipDeviceGetPCIBusId(pci_bus_id, (int)pci_len, device));
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (mem_free || mem_total) {
        int cur = 0;
        MA_HIP(hipGetDevice(&cur));
        MA_HIP(hipSetDevice(device));
        size_t f = 0, t = 0;
        MA_HIP(hipMemGetInfo(&f, &t));
        MA_HIP(hipSetDevice(cur));
        if (mem_free) *mem_free = f;
        if (mem_total) *mem_total = t;
    }
    return MA_OK;
}

int ma_clock_probe(ma_ctx* ctx, double milliseconds, double* sustained_ghz)
{
    MA_REQUIRE(ctx && sustained_ghz, "NULL argument");
    MA_REQUIRE(milliseconds > 0 && milliseconds <= 200, "probe length must be in (0, 200] ms");
    MA_HIP(hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    MA_HIP(hipGetDeviceProperties(&prop, ctx->device));
    const int blocks = prop.multiProcessorCount * 4;       // 4 blocks x 4 waves per CU = 4 waves per SIMD
    MA_TRY(ma_ws_reserve(ctx, (size_t)blocks * 256 * sizeof(float) + (size_t)blocks * 2 * sizeof(unsigned long long) + 64));
    float* sink = (float*)ctx->ws;
    unsigned long long* stamps = (unsigned long long*)ma_align_up((size_t)(sink + (size_t)blocks * 256), 16);
    // 24 packed ops per iteration and wave, 4 waves per SIMD, ~4 cycles each at ~2 GHz: ~0.2 us per iteration
    const int iters = (int)(milliseconds * 1e-3 / 0.19e-6) + 1;
    MA_TRY(ma_pinned_reserve(ctx, (size_t)blocks * 2 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, iters, 1.0f, sink, stamps);
    MA_HIP(hipGetLastError());
    MA_HIP(hipMemcpyAsync(ctx->pinned, stamps, (size_t)blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    MA_HIP(hipStreamSynchronize(ctx->stream));
    const unsigned long long* h = (const unsigned long long*)ctx->pinned;
    std::vector<double> ghz;
    for (int b = 0; b < blocks; b++)
        if (h[2 * b + 1] > 0) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);   // s_memrealtime: 100 MHz
    MA_REQUIRE(!ghz.empty(), "clock probe returned no stamps");
    std::sort(ghz.begin(), ghz.end());
    *sustained_ghz = ghz[ghz.size() / 2];                   // median over the blocks
    return MA_OK;
}

}  // extern "C"
