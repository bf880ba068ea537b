// Micro-benchmark: does v_pk_add_f32 slow down when its two 64-bit sources sit in the same VGPR banks (reg % 4)?
// hipcc --offload-arch=gfx950 -O3 tools/ubench_bank.hip -o tools/ubench_bank
#include <hip/hip_runtime.h>
#include <cstdio>

#define INIT                                                                                                     \
    "v_mov_b32 v8, %2\n v_mov_b32 v9, %2\n v_mov_b32 v12, %2\n v_mov_b32 v13, %2\n v_mov_b32 v16, %2\n v_mov_b32 v17, %2\n" \
    "v_mov_b32 v20, %2\n v_mov_b32 v21, %2\n v_mov_b32 v24, %2\n v_mov_b32 v25, %2\n v_mov_b32 v26, %2\n v_mov_b32 v27, %2\n" \
    "s_mov_b32 s20, %1\n s_mov_b32 s22, 0x3f7fff00\n s_mov_b32 s23, 0x3f7fff00\n"                                          \
    "1:\n"
#define TAIL "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n v_add_f32 %0, v8, v16\n"
#define CLOB "v8", "v9", "v12", "v13", "v16", "v17", "v20", "v21", "v24", "v25", "v26", "v27", "s20", "s22", "s23", "scc"
#define BODY4(OP, A, B)                                                                                   \
    OP " v[8:9], v[12:13], " A "\n " OP " v[16:17], v[20:21], " B "\n" OP " v[12:13], v[8:9], " A "\n " OP " v[20:21], v[16:17], " B "\n" \
    OP " v[8:9], v[12:13], " A "\n " OP " v[16:17], v[20:21], " B "\n" OP " v[12:13], v[8:9], " A "\n " OP " v[20:21], v[16:17], " B "\n"

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float r;
    if (MODE == 0)        // both sources in banks (0,1): v[12:13] / v[8:9] with v[24:25]
        asm volatile(INIT BODY4("v_pk_add_f32", "v[24:25]", "v[24:25]") TAIL : "=v"(r) : "s"(iters), "v"(seed) : CLOB);
    else if (MODE == 1)   // sources in banks (0,1) and (2,3)
        asm volatile(INIT BODY4("v_pk_add_f32", "v[26:27]", "v[26:27]") TAIL : "=v"(r) : "s"(iters), "v"(seed) : CLOB);
    else if (MODE == 2)   // one source an SGPR pair (the tap multiply of the FIR)
        asm volatile(INIT BODY4("v_pk_mul_f32", "s[22:23]", "s[22:23]") TAIL : "=v"(r) : "s"(iters), "v"(seed) : CLOB);
    else if (MODE == 3)   // scalar f32, both sources in bank 0
        asm volatile(INIT "v_add_f32 v8, v12, v24\n v_add_f32 v16, v20, v24\n v_add_f32 v12, v8, v24\n v_add_f32 v20, v16, v24\n"
                          "v_add_f32 v8, v12, v24\n v_add_f32 v16, v20, v24\n v_add_f32 v12, v8, v24\n v_add_f32 v20, v16, v24\n" TAIL
                     : "=v"(r) : "s"(iters), "v"(seed) : CLOB);
    else                  // scalar f32, sources in banks 0 and 1
        asm volatile(INIT "v_add_f32 v8, v12, v25\n v_add_f32 v16, v20, v25\n v_add_f32 v12, v8, v25\n v_add_f32 v20, v16, v25\n"
                          "v_add_f32 v8, v12, v25\n v_add_f32 v16, v20, v25\n v_add_f32 v12, v8, v25\n v_add_f32 v20, v16, v25\n" TAIL
                     : "=v"(r) : "s"(iters), "v"(seed) : CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
void run(const char* name, int wps)
{
    const int cus = 256, blocks = cus * wps, iters = 200000;
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 1000, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)wps * iters * 8;
    printf("%-44s waves/SIMD=%d  %.3f ms  %.2f ns per instr per SIMD (%.2f cycles @2.4)\n", name, wps, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    hipFree(out);
}

int main()
{
    for (int w : {2, 4, 8}) {
        run<0>("v_pk_add_f32 both sources banks (0,1)", w);
        run<1>("v_pk_add_f32 sources banks (0,1) and (2,3)", w);
        run<2>("v_pk_mul_f32 VGPR pair x SGPR pair", w);
        run<3>("v_add_f32 sources in the same bank", w);
        run<4>("v_add_f32 sources in different banks", w);
    }
    return 0;
}
