// Exact two-nearest-neighbour search (L2) between two descriptor sets: the matching step of FeatureRegistrator.
// The reference calls cv2.FlannBasedMatcher().knnMatch(des2, des1, k=2) (microaligner/feature_reg/
// feature_detection.py:137-141) on up to 45 000 x 45 000 DAISY descriptors of 200 floats per image pair and
// iteration -- an approximate kd-tree search in OpenCV C++; here it is the exact search, one pass over the distance
// matrix that is never stored: a block owns 64 queries (resident in LDS), streams the train set through LDS in
// 64-row tiles and 32-dimension chunks, every thread accumulates a 4 x 4 patch of squared distances
// (d2 = d2 + (q - t)^2 in float32, ascending dimension, product and sum rounded separately) and keeps the two smallest per query.
// Ties go to the lower train index (what numpy's argmin does in feature_reg/sparse_cpu.py).
#include "ma_internal.h"

#include <algorithm>
#include <cfloat>

namespace {

constexpr int KN_TQ = 64, KN_TT = 64, KN_KC = 32, KN_TP = KN_KC + 4;   // T chunk pitch 36 floats: 9 x 16 B, odd

struct Top2 { float d0, d1; int i0, i1; };

__device__ __forceinline__ void top2_push(Top2& b, float d, int i)
{
    // candidates arrive in ascending index order inside a thread: strict comparisons keep the lower index on ties
    if (d < b.d0) { b.d1 = b.d0; b.i1 = b.i0; b.d0 = d; b.i0 = i; }
    else if (d < b.d1) { b.d1 = d; b.i1 = i; }
}
__device__ __forceinline__ bool before(float da, int ia, float db, int ib) { return da < db || (da == db && ia < ib); }

__global__ __launch_bounds__(256) void knn2_kernel(const float* __restrict__ q, const float* __restrict__ t, int nq, int nt,
                                                   int dim, int* __restrict__ idx, float* __restrict__ dist)
{
    extern __shared__ float lds[];
    const int qp = dim + 4;                  // query pitch; dim % 4 == 0 and (qp / 4) odd when dim % 8 == 0
    float* Qs = lds;                         // [KN_TQ][qp]
    float* Ts = lds + KN_TQ * qp;            // [KN_TT][KN_TP]
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int q0 = blockIdx.x * KN_TQ;
    for (int e = tid; e < KN_TQ * (dim / 4); e += 256) {
        const int row = e / (dim / 4), k4 = e - row * (dim / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + row < nq) v = reinterpret_cast<const float4*>(q + (size_t)(q0 + row) * dim)[k4];
        *reinterpret_cast<float4*>(Qs + row * qp + 4 * k4) = v;
    }
    Top2 best[4];
#pragma unroll
    for (int r = 0; r < 4; r++) { best[r].d0 = best[r].d1 = FLT_MAX; best[r].i0 = best[r].i1 = 0x7fffffff; }

    for (int t0 = 0; t0 < nt; t0 += KN_TT) {
        float acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[r][c] = 0.f;
        for (int k0 = 0; k0 < dim; k0 += KN_KC) {
            __syncthreads();                 // previous chunk consumed (first pass: Qs complete)
#pragma unroll
            for (int u = 0; u < 2; u++) {    // 64 rows x 8 float4 = 512 float4, two per thread
                const int e = tid + 256 * u, row = e >> 3, k4 = e & 7;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t0 + row < nt && k0 + 4 * k4 < dim) v = reinterpret_cast<const float4*>(t + (size_t)(t0 + row) * dim + k0)[k4];
                *reinterpret_cast<float4*>(Ts + row * KN_TP + 4 * k4) = v;
            }
            __syncthreads();
            const int kc = min(KN_KC, dim - k0) / 4;
            for (int k4 = 0; k4 < kc; k4++) {
                float4 qv[4], tv[4];
#pragma unroll
                for (int r = 0; r < 4; r++) qv[r] = *reinterpret_cast<const float4*>(Qs + (ty * 4 + r) * qp + k0 + 4 * k4);
#pragma unroll
                for (int c = 0; c < 4; c++) tv[c] = *reinterpret_cast<const float4*>(Ts + (tx + 16 * c) * KN_TP + 4 * k4);
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        // d2 = d2 + d*d, product and sum rounded separately (the file is built with -ffp-contract=off):
                        // the definition sparse_cpu.knn2_sequential restates with numpy, bit for bit
                        float d;
                        d = qv[r].x - tv[c].x; acc[r][c] = acc[r][c] + d * d;
                        d = qv[r].y - tv[c].y; acc[r][c] = acc[r][c] + d * d;
                        d = qv[r].z - tv[c].z; acc[r][c] = acc[r][c] + d * d;
                        d = qv[r].w - tv[c].w; acc[r][c] = acc[r][c] + d * d;
                    }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int j = t0 + tx + 16 * c;
            if (j < nt) {
#pragma unroll
                for (int r = 0; r < 4; r++) top2_push(best[r], acc[r][c], j);
            }
        }
    }
    // merge the 16 partial results of every query row (Qs is free now)
    __syncthreads();
    float* cd = lds;                                         // [64][16][2] distances
    int* ci = reinterpret_cast<int*>(lds + KN_TQ * 16 * 2);  // [64][16][2] indices
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int o = ((ty * 4 + r) * 16 + tx) * 2;
        cd[o] = best[r].d0; cd[o + 1] = best[r].d1;
        ci[o] = best[r].i0; ci[o + 1] = best[r].i1;
    }
    __syncthreads();
    if (tid < KN_TQ && q0 + tid < nq) {
        float d0 = FLT_MAX, d1 = FLT_MAX;
        int i0 = 0x7fffffff, i1 = 0x7fffffff;
        for (int e = 0; e < 32; e++) {
            const float d = cd[tid * 32 + e];
            const int i = ci[tid * 32 + e];
            if (i == 0x7fffffff) continue;
            if (before(d, i, d0, i0)) { d1 = d0; i1 = i0; d0 = d; i0 = i; }
            else if (before(d, i, d1, i1)) { d1 = d; i1 = i; }
        }
        idx[(size_t)(q0 + tid) * 2] = i0;
        idx[(size_t)(q0 + tid) * 2 + 1] = i1;
        dist[(size_t)(q0 + tid) * 2] = d0;      // squared; the caller takes the root
        dist[(size_t)(q0 + tid) * 2 + 1] = d1;
    }
}

} // namespace

extern "C" int ma_knn2_l2(ma_ctx* ctx, const float* query, int nq, const float* train, int nt, int dim, int* idx_out,
                          float* dist_out)
{
    MA_REQUIRE(ctx && query && train && idx_out && dist_out, "NULL argument");
    MA_REQUIRE(nq >= 1 && nt >= 2, "need at least one query and two train descriptors");
    MA_REQUIRE(dim >= 4 && dim % 4 == 0, "the descriptor length must be a multiple of 4 (pad with zeros)");
    size_t lds = (size_t)(KN_TQ * (dim + 4) + KN_TT * KN_TP) * sizeof(float);
    lds = std::max(lds, (size_t)KN_TQ * 16 * 4 * sizeof(float));   // the final merge reuses the buffer
    MA_REQUIRE(lds <= 160 * 1024, "descriptor length out of range");
    MA_HIP(hipSetDevice(ctx->device));
    MaProfScope ps(ctx, MA_K_OTHER, (double)nq);
    hipLaunchKernelGGL(knn2_kernel, dim3((nq + KN_TQ - 1) / KN_TQ), dim3(256), lds, ctx->stream, query, train, nq, nt, dim,
                       idx_out, dist_out);
    MA_HIP(hipGetLastError());
    return MA_OK;
}
