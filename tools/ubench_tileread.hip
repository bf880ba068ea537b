// Micro-benchmark: HBM read rate of a [planes][P][P] float array fetched in tiles of ROWS x COLS elements (one block of 256
// threads per tile, every row segment of COLS floats contiguous), against a plain linear sweep.  Answers: how long must a row
// segment be for the DAISY smoothing passes (csrc/daisy.hip) to get streaming bandwidth?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int COLS, int VEC>
__global__ __launch_bounds__(256) void tile_read(const float* __restrict__ src, int P, int rows, float* __restrict__ out)
{
    // threads of a row: COLS / VEC; rows per pass: 256 / (COLS / VEC)
    constexpr int TPR = COLS / VEC, RPP = 256 / TPR;
    const int tx = threadIdx.x % TPR, ty = threadIdx.x / TPR;
    const int x0 = blockIdx.x * COLS + tx * VEC, y0 = blockIdx.y * rows;
    const float* s = src + (size_t)blockIdx.z * P * P;
    float acc = 0.f;
    if (x0 + VEC <= P) {
        for (int r = ty; r < rows; r += RPP) {
            const int y = y0 + r;
            if (y >= P) break;
            const float* p = s + (size_t)y * P + x0;
            if (VEC == 1) acc += p[0];
            if (VEC == 2) { acc += p[0] + p[1]; }
            if (VEC == 4) { acc += p[0] + p[1] + p[2] + p[3]; }
        }
    }
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void linear_read(const float4* __restrict__ src, size_t n4, float* __restrict__ out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}

template <int COLS, int VEC>
int run(const float* d, int P, int planes, int rows, float* out)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const dim3 grid((P + COLS - 1) / COLS, (P + rows - 1) / rows, planes);
    tile_read<COLS, VEC><<<grid, 256>>>(d, P, rows, out);
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; i++) tile_read<COLS, VEC><<<grid, 256>>>(d, P, rows, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("tile %3d cols (x%d per thread) x %3d rows: %.3f ms per sweep, %.0f GB/s\n", COLS, VEC, rows, ms / 5,
           (double)planes * P * P * 4 / (ms / 5 * 1e-3) / 1e9);
    return 0;
}

int main()
{
    const int P = 1102, planes = 72;
    const size_t n = (size_t)planes * P * P;
    float *d, *out;
    CK(hipMalloc(&d, n * 4 + 64)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(d, 0, n * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    linear_read<<<4096, 256>>>((const float4*)d, n / 4, out);
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; i++) linear_read<<<4096, 256>>>((const float4*)d, n / 4, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("linear float4 sweep: %.3f ms, %.0f GB/s\n", ms / 5, (double)n * 4 / (ms / 5 * 1e-3) / 1e9);
    for (int rows : {44, 88, 176, 352}) {
        if (run<64, 1>(d, P, planes, rows, out)) return 1;
        if (run<128, 1>(d, P, planes, rows, out)) return 1;
        if (run<128, 2>(d, P, planes, rows, out)) return 1;
        if (run<256, 1>(d, P, planes, rows, out)) return 1;
        if (run<256, 2>(d, P, planes, rows, out)) return 1;
    }
    return 0;
}
