// Device-to-host rate right after shader-heavy work: the DMA engine (hipMemcpyAsync) against a copy kernel that writes
// page-locked host memory itself.  hipcc --offload-arch=gfx950 -O3 -o /tmp/ub tools/ubench_d2h_after_work.hip && /tmp/ub
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void burn(float* out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; i++) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; }
    if (a == 12345.f) out[0] = a + b;
}

__global__ void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint4 v = src[i];
        __builtin_nontemporal_store(v.x, &dst[i].x);
        __builtin_nontemporal_store(v.y, &dst[i].y);
        __builtin_nontemporal_store(v.z, &dst[i].z);
        __builtin_nontemporal_store(v.w, &dst[i].w);
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const size_t N = (size_t)1 << 30;
    const int blocks = argc > 1 ? atoi(argv[1]) : 64;
    void *dev, *dev2, *host;
    float* scratch;
    CK(hipMalloc(&dev, N)); CK(hipMalloc(&dev2, N)); CK(hipMalloc(&scratch, 4096));
    CK(hipHostMalloc(&host, N, hipHostMallocDefault));
    CK(hipMemset(dev, 1, N));
    hipStream_t s, c;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
    auto heavy = [&]() {   // ~0.5 s of every CU busy
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(burn, dim3(256 * 32), dim3(256), 0, c, scratch, 400000);
        CK(hipStreamSynchronize(c));
    };
    auto dma = [&](void* h, void* d, bool d2h) {
        double t0 = now();
        CK(hipMemcpyAsync(d2h ? h : d, d2h ? d : h, N, d2h ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        return N / (now() - t0) / 1e9;
    };
    auto kern = [&](void* h, void* d, bool d2h) {
        double t0 = now();
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s, (const uint4*)(d2h ? d : h), (uint4*)(d2h ? h : d), N / 16);
        CK(hipStreamSynchronize(s));
        return N / (now() - t0) / 1e9;
    };
    dma(host, dev, true); kern(host, dev, true); dma(host, dev2, false); kern(host, dev2, false);
    printf("idle      : D2H dma %.1f GB/s, D2H kernel (%d blocks) %.1f GB/s | H2D dma %.1f, H2D kernel %.1f\n", dma(host, dev, true), blocks,
           kern(host, dev, true), dma(host, dev2, false), kern(host, dev2, false));
    for (int rep = 0; rep < 2; rep++) {
        heavy(); double a = dma(host, dev, true);
        heavy(); double b = kern(host, dev, true);
        heavy(); double c2 = dma(host, dev2, false);
        heavy(); double d = kern(host, dev2, false);
        printf("after work: D2H dma %.1f GB/s, D2H kernel %.1f GB/s | H2D dma %.1f, H2D kernel %.1f\n", a, b, c2, d);
    }
    // freeing device memory: the driver clears released VRAM in the background
    for (size_t gb : {(size_t)4, (size_t)16, (size_t)64}) {
        void* big;
        CK(hipMalloc(&big, gb << 30));
        CK(hipMemset(big, 1, gb << 30));
        CK(hipDeviceSynchronize());
        double t0 = now();
        CK(hipFree(big));
        double tf = now() - t0;
        printf("hipFree of %zu GiB took %.3f s; then:\n", gb, tf);
        for (int i = 0; i < 8; i++) {
            double t = now();
            double r = dma(host, dev, true);
            double k = kern(host, dev, true);
            double u = dma(host, dev2, false);
            printf("  t=%.2f s: D2H dma %.1f GB/s, D2H kernel %.1f, H2D dma %.1f\n", t - t0 - tf, r, k, u);
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    }
    heavy();
    for (int i = 0; i < 2; i++) {
        double t = now();
        double r = dma(host, dev, true);
        printf("  D2H dma call %d after work (t=%.2f s): %.1f GB/s\n", i, now() - t, r);
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    return 0;
}
