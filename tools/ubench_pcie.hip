// Host<->device transfer rates that decide how the page-warp driver moves pages: pageable vs pinned staging vs
// hipHostRegister of the caller's buffer.  hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = (size_t)512 << 20;
    char* pageable = (char*)malloc(n);
    memset(pageable, 1, n);
    char* pageable2 = (char*)malloc(n);
    memset(pageable2, 2, n);
    void *d, *pinned;
    hipMalloc(&d, n);
    double t = now();
    hipHostMalloc(&pinned, n, hipHostMallocDefault);
    printf("hipHostMalloc 512 MiB: %.1f ms\n", (now() - t) * 1e3);
    memset(pinned, 3, n);
    for (int rep = 0; rep < 2; rep++) {
        t = now(); hipMemcpy(d, pageable, n, hipMemcpyHostToDevice); double a = now() - t;
        t = now(); hipMemcpy(pageable2, d, n, hipMemcpyDeviceToHost); double b = now() - t;
        printf("pageable  H2D %.1f GB/s  D2H %.1f GB/s\n", n / a / 1e9, n / b / 1e9);
        t = now(); hipMemcpy(d, pinned, n, hipMemcpyHostToDevice); a = now() - t;
        t = now(); hipMemcpy(pinned, d, n, hipMemcpyDeviceToHost); b = now() - t;
        printf("pinned    H2D %.1f GB/s  D2H %.1f GB/s\n", n / a / 1e9, n / b / 1e9);
        t = now(); memcpy(pinned, pageable, n); a = now() - t;
        printf("memcpy pageable->pinned (1 thread) %.1f GB/s\n", n / a / 1e9);
    }
    t = now();
    hipError_t e = hipHostRegister(pageable, n, hipHostRegisterDefault);
    double reg = now() - t;
    printf("hipHostRegister 512 MiB: %.1f ms (%s)\n", reg * 1e3, hipGetErrorString(e));
    if (e == hipSuccess) {
        t = now(); hipMemcpy(d, pageable, n, hipMemcpyHostToDevice); double a = now() - t;
        t = now(); hipMemcpy(pageable, d, n, hipMemcpyDeviceToHost); double b = now() - t;
        printf("registered H2D %.1f GB/s  D2H %.1f GB/s\n", n / a / 1e9, n / b / 1e9);
        t = now(); hipHostUnregister(pageable); printf("hipHostUnregister: %.1f ms\n", (now() - t) * 1e3);
    }
    return 0;
}
