// Micro-benchmark: VALU issue rates on gfx950 (scalar f32 add/mul/fma vs packed v_pk_* f32).
// Decides whether the window-blur kernels should use packed math.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters, float seed)
{
    float a[8];
    float2v p[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; p[i] = (float2v){a[i], a[i] + 1.f}; }
    double d[8];
    for (int i = 0; i < 8; i++) d[i] = (double)a[i];
    const double db = 1.0 + 1e-9 * seed;
    float b = seed * 0.5f + 1.0f;
    float2v pb = {b, b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[u]) : "v"(b));
            if (MODE == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[u]) : "v"(b));
            if (MODE == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[u]) : "v"(b));
            if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[u]) : "v"(pb));
            if (MODE == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[u]) : "v"(pb));
            if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[u]) : "v"(pb));
            if (MODE == 6) asm volatile("v_mov_b32 %0, %1" : "+v"(a[u]) : "v"(b));
            if (MODE == 7) asm volatile("v_mov_b64 %0, %1" : "+v"(p[u]) : "v"(pb));
            if (MODE == 8) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[u]) : "v"(db));
            if (MODE == 9) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[u]) : "v"(db));
            if (MODE == 10) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[u]) : "v"(db));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd, float lanes_per_instr)
{
    int cus = 256;
    int threads = 256;                 // 4 waves per block = 1 per SIMD
    int blocks = cus * waves_per_simd; // -> waves_per_simd waves on each SIMD
    int iters = 20000;
    float* out;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, 100, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * (threads / 64) * iters * 8;  // wave-instructions
    double per_simd = instr / (cus * 4);
    double ns_per_instr = ms * 1e6 / per_simd;
    printf("%-14s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)  %.1f T lane-ops/s\n",
           name, waves_per_simd, ms, ns_per_instr, ns_per_instr * 2.4, instr * 64 * lanes_per_instr / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 3, 4, 8}) {
        run<0>("v_fma_f32", w, 1);
        run<1>("v_add_f32", w, 1);
        run<2>("v_mul_f32", w, 1);
        run<3>("v_pk_fma_f32", w, 2);
        run<4>("v_pk_add_f32", w, 2);
        run<5>("v_pk_mul_f32", w, 2);
        run<6>("v_mov_b32", w, 1);
        run<7>("v_mov_b64", w, 2);
        run<8>("v_add_f64", w, 1);
        run<9>("v_mul_f64", w, 1);
        run<10>("v_fma_f64", w, 1);
    }
    return 0;
}
