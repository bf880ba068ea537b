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
#include <cmath>
#include <cstdlib>
#include <string>

namespace {

constexpr int KN_TQ = 64, KN_TT = 64, KN_KC = 32, KN_TP = KN_KC + 4;   // LDS budget of a T chunk (64 rows x 36 floats)
typedef float kn_f2 __attribute__((ext_vector_type(2)));

struct Top2 { float d0, d1; int i0, i1; };

__device__ __forceinline__ void top2_push(Top2& b, float d, int i)
{
    // candidates arrive in ascending index order inside a thread: strict comparisons keep the lower index on ties
    if (d < b.d0) { b.d1 = b.d0; b.i1 = b.i0; b.d0 = d; b.i0 = i; }
    else if (d < b.d1) { b.d1 = d; b.i1 = i; }
}
__device__ __forceinline__ bool before(float da, int ia, float db, int ib) { return da < db || (da == db && ia < ib); }

// qlist / qcount: when given, the kernel serves the query rows qlist[0 .. *qcount) instead of 0 .. nq (the fallback of
// the filtered search below: the grid is sized for nq and the blocks beyond the list leave at once)
template <bool PK>
__global__ __launch_bounds__(256) void knn2_kernel(const float* __restrict__ q, const float* __restrict__ t, int nq, int nt,
                                                   int dim, int* __restrict__ idx, float* __restrict__ dist,
                                                   const int* __restrict__ qlist, const int* __restrict__ qcount,
                                                   int split_rows, float4* __restrict__ part)
{
    extern __shared__ float lds[];
    const int cap = nq;                      // list positions the partial results are laid out for
    if (qlist) nq = *qcount;
    // a list is served by a FEW blocks per train range, each walking the list in steps of the grid (the grid cannot be sized
    // for a count that lives on the device: one block per 64 of ALL queries made 22 000 blocks of which six had work, and
    // their launch cost more than the work -- 150 us per match)
  // all queries: grid (query tiles, train ranges).  A list: grid (train ranges, blocks per range) -- the hardware hands out
  // blocks x-fastest, so the blocks that HAVE a tile of the list (the first few of the 16 per range) all start at once and the
  // idle ones follow; with the tiles along x the working blocks sat one in eight among idle ones that each hold half a CU's LDS
  // for their ~15 us of launch, one load and exit: the last working block started 60 us into a kernel whose blocks run 30 us
  const int bq = qlist ? blockIdx.y : blockIdx.x, nbq = qlist ? gridDim.y : gridDim.x;
  const int bs = qlist ? blockIdx.x : blockIdx.y, nbs = qlist ? gridDim.x : gridDim.y;
  for (int qtile = bq; qtile * KN_TQ < nq; qtile += nbq) {
    __syncthreads();                         // the previous tile's merge has left the LDS buffer
    // nbs > 1: this block serves the train rows [tbeg, tend) only and leaves its pair per query in `part`
    // (knn2_merge_parts folds them): a handful of queries then occupies the whole chip instead of one CU
    const int tbeg = nbs > 1 ? bs * split_rows : 0;
    const int tend = nbs > 1 ? min(nt, tbeg + split_rows) : nt;
    const int qp = dim + 4;                  // query pitch; dim % 4 == 0 and (qp / 4) odd when dim % 8 == 0
    float* Qs = lds;                         // [KN_TQ][qp]
    float* Ts = lds + KN_TQ * qp;            // PK: [2][KN_KC][16 tx][4 cc], else [2][KN_TT][KN_TP]
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int q0 = qtile * KN_TQ;
    // the tile's query rows -> LDS: the list's row numbers first (one load per row, not one in front of every 16-byte load of
    // the row), then four loads in flight per thread -- a block of the list lives ~100 us, a load after a load after a load
    // was a fifth of that
    __shared__ int qrow[KN_TQ];
    if (tid < KN_TQ) qrow[tid] = q0 + tid < nq ? (qlist ? qlist[q0 + tid] : q0 + tid) : -1;
    __syncthreads();
    for (int e0 = tid; e0 < KN_TQ * (dim / 4); e0 += 256 * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int e = e0 + 256 * u, row = e / (dim / 4), k4 = e - row * (dim / 4);
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < KN_TQ * (dim / 4) && qrow[row] >= 0) v[u] = reinterpret_cast<const float4*>(q + (size_t)qrow[row] * dim)[k4];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int e = e0 + 256 * u, row = e / (dim / 4), k4 = e - row * (dim / 4);
            if (e < KN_TQ * (dim / 4)) *reinterpret_cast<float4*>(Qs + row * qp + 4 * k4) = v[u];
        }
    }
    Top2 best[4];
#pragma unroll
    for (int r = 0; r < 4; r++) { best[r].d0 = best[r].d1 = FLT_MAX; best[r].i0 = best[r].i1 = 0x7fffffff; }

    // the train rows pass through two LDS buffers: the next chunk (64 rows x 32 dimensions) is loaded into registers before the
    // current one is consumed and written behind it -- one barrier per chunk, no load latency between chunks (the fallback of
    // the filtered search spent two thirds of its time there: 35 chunks of 4 us for a few dozen queries).
    // A chunk lies in LDS as [dimension][tx][cc] -- the four rows tx, tx + 16, tx + 32, tx + 48 a thread serves, adjacent -- so
    // that ONE 16-byte read gives a thread its four train values of a dimension as two register pairs, and the arithmetic runs
    // on PACKED float32 instructions (v_pk_add_f32 / v_pk_mul_f32: two accumulators per instruction, each component rounded as
    // the scalar instruction rounds it; the query value is broadcast into both halves).  The kernel is bound by instruction
    // issue: scalar float32 instructions issue every 7.8 cycles from one wave per SIMD (tools/ubench_valu.hip), which is what
    // the few blocks of the fallback get -- 192 of them per four dimensions before, 96 packed ones now.
    const int cpt = (dim + KN_KC - 1) / KN_KC;                       // chunks per train tile
    const int ntile = tend > tbeg ? (tend - tbeg + KN_TT - 1) / KN_TT : 0, nchunk = ntile * cpt;
    float4 stage[2];
    auto fetch = [&](int c) {
        const int t0 = tbeg + (c / cpt) * KN_TT, k0 = (c % cpt) * KN_KC;
#pragma unroll
        for (int u = 0; u < 2; u++) {        // 64 rows x 8 float4 = 512 float4, two per thread
            const int e = tid + 256 * u, row = e >> 3, k4 = e & 7;
            stage[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t0 + row < tend && k0 + 4 * k4 < dim) stage[u] = reinterpret_cast<const float4*>(t + (size_t)(t0 + row) * dim + k0)[k4];
        }
    };
  if constexpr (PK) {
    auto commit = [&](int c) {
        float* dst = Ts + (c & 1) * KN_TT * KN_KC;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = tid + 256 * u, row = e >> 3, k4 = e & 7;
            float* o = dst + (4 * k4) * KN_TT + (row & 15) * 4 + (row >> 4);
            o[0] = stage[u].x; o[KN_TT] = stage[u].y; o[2 * KN_TT] = stage[u].z; o[3 * KN_TT] = stage[u].w;
        }
    };
    if (nchunk > 0) { fetch(0); commit(0); }
    __syncthreads();                         // Qs and the first chunk complete
    kn_f2 acc[4][2];                         // acc[r][p]: rows tx + 32 p (.x) and tx + 32 p + 16 (.y) against query ty * 4 + r
    for (int c = 0; c < nchunk; c++) {
        const int t0 = tbeg + (c / cpt) * KN_TT, kcix = c % cpt, k0 = kcix * KN_KC;
        if (c + 1 < nchunk) fetch(c + 1);
        if (kcix == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) { acc[r][0] = (kn_f2)(0.f); acc[r][1] = (kn_f2)(0.f); }
        }
        const float* Tc = Ts + (c & 1) * KN_TT * KN_KC;
        const int kc = min(KN_KC, dim - k0) / 4;
        for (int k4 = 0; k4 < kc; k4++) {
            float4 qv[4], tv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) qv[r] = *reinterpret_cast<const float4*>(Qs + (ty * 4 + r) * qp + k0 + 4 * k4);
#pragma unroll
            for (int kk = 0; kk < 4; kk++) tv[kk] = *reinterpret_cast<const float4*>(Tc + (4 * k4 + kk) * KN_TT + tx * 4);
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const kn_f2 ta = {tv[kk].x, tv[kk].y}, tb = {tv[kk].z, tv[kk].w};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    // d2 = d2 + d*d, product and sum rounded separately (the file is built with -ffp-contract=off):
                    // the definition sparse_cpu.knn2_sequential restates with numpy, bit for bit
                    const float qs = kk == 0 ? qv[r].x : kk == 1 ? qv[r].y : kk == 2 ? qv[r].z : qv[r].w;
                    const kn_f2 q2 = {qs, qs};
                    kn_f2 d;
                    d = q2 - ta; acc[r][0] = acc[r][0] + d * d;
                    d = q2 - tb; acc[r][1] = acc[r][1] + d * d;
                }
            }
        }
        if (kcix == cpt - 1) {
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                const int j = t0 + tx + 16 * cc;
                if (j < tend) {
#pragma unroll
                    for (int r = 0; r < 4; r++) top2_push(best[r], (cc & 1) ? acc[r][cc >> 1].y : acc[r][cc >> 1].x, j);
                }
            }
        }
        if (c + 1 < nchunk) commit(c + 1);
        __syncthreads();
    }
  } else {
    auto commit = [&](int c) {
        float* dst = Ts + (c & 1) * KN_TT * KN_TP;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = tid + 256 * u, row = e >> 3, k4 = e & 7;
            *reinterpret_cast<float4*>(dst + row * KN_TP + 4 * k4) = stage[u];
        }
    };
    if (nchunk > 0) { fetch(0); commit(0); }
    __syncthreads();                         // Qs and the first chunk complete
    float acc[4][4];
    for (int c = 0; c < nchunk; c++) {
        const int t0 = tbeg + (c / cpt) * KN_TT, kcix = c % cpt, k0 = kcix * KN_KC;
        if (c + 1 < nchunk) fetch(c + 1);
        if (kcix == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int cc = 0; cc < 4; cc++) acc[r][cc] = 0.f;
        }
        const float* Tc = Ts + (c & 1) * KN_TT * KN_TP;
        const int kc = min(KN_KC, dim - k0) / 4;
        for (int k4 = 0; k4 < kc; k4++) {
            float4 qv[4], tv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) qv[r] = *reinterpret_cast<const float4*>(Qs + (ty * 4 + r) * qp + k0 + 4 * k4);
#pragma unroll
            for (int cc = 0; cc < 4; cc++) tv[cc] = *reinterpret_cast<const float4*>(Tc + (tx + 16 * cc) * KN_TP + 4 * k4);
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int cc = 0; cc < 4; cc++) {
                    // d2 = d2 + d*d, product and sum rounded separately (the file is built with -ffp-contract=off):
                    // the definition sparse_cpu.knn2_sequential restates with numpy, bit for bit
                    float d;
                    d = qv[r].x - tv[cc].x; acc[r][cc] = acc[r][cc] + d * d;
                    d = qv[r].y - tv[cc].y; acc[r][cc] = acc[r][cc] + d * d;
                    d = qv[r].z - tv[cc].z; acc[r][cc] = acc[r][cc] + d * d;
                    d = qv[r].w - tv[cc].w; acc[r][cc] = acc[r][cc] + d * d;
                }
        }
        if (kcix == cpt - 1) {
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                const int j = t0 + tx + 16 * cc;
                if (j < tend) {
#pragma unroll
                    for (int r = 0; r < 4; r++) top2_push(best[r], acc[r][cc], j);
                }
            }
        }
        if (c + 1 < nchunk) commit(c + 1);
        __syncthreads();
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
        if (nbs > 1) {
            part[(size_t)bs * cap + q0 + tid] = make_float4(d0, d1, __int_as_float(i0), __int_as_float(i1));
        } else {
            const size_t o = (size_t)(qlist ? qlist[q0 + tid] : q0 + tid) * 2;
            idx[o] = i0;
            idx[o + 1] = i1;
            dist[o] = d0;      // squared; the caller takes the root
            dist[o + 1] = d1;
        }
    }
  }
}

// the pairs the train splits of knn2_kernel left for list position p -> the pair of query qlist[p].  One WAVE per position:
// the lanes take the splits 64 apart and the partial pairs are folded across the wave (a strict total order on (distance,
// index) and disjoint index sets: the two smallest do not depend on the order of the folding).  A thread per position read
// its 179 partial results one after the other -- 96 us for the top level's 350 queries, latency all of it.
__global__ __launch_bounds__(256) void knn2_merge_parts(const float4* __restrict__ part, int cap, int nsplit,
                                                        const int* __restrict__ qlist, const int* __restrict__ qcount,
                                                        int* __restrict__ idx, float* __restrict__ dist)
{
    const int lane = threadIdx.x & 63, n = *qcount;
    for (int p = blockIdx.x * 4 + (threadIdx.x >> 6); p < n; p += gridDim.x * 4) {      // wave-uniform
        float d0 = FLT_MAX, d1 = FLT_MAX;
        int i0 = 0x7fffffff, i1 = 0x7fffffff;
        auto push = [&](float d, int i) {
            if (i == 0x7fffffff) return;
            if (before(d, i, d0, i0)) { d1 = d0; i1 = i0; d0 = d; i0 = i; }
            else if (before(d, i, d1, i1)) { d1 = d; i1 = i; }
        };
        for (int s = lane; s < nsplit; s += 64) {
            const float4 v = part[(size_t)s * cap + p];
            push(v.x, __float_as_int(v.z));
            push(v.y, __float_as_int(v.w));
        }
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            const float e0 = __shfl_xor(d0, off), e1 = __shfl_xor(d1, off);
            const int j0 = __shfl_xor(i0, off), j1 = __shfl_xor(i1, off);
            push(e0, j0);
            push(e1, j1);
        }
        if (lane == 0) {
            const size_t o = (size_t)qlist[p] * 2;
            idx[o] = i0;
            idx[o + 1] = i1;
            dist[o] = d0;
            dist[o + 1] = d1;
        }
    }
}

// ---- filtered search: FP32 MFMA shortlist, exact re-evaluation, certificate, exact fallback --------------------------
// The result of ma_knn2_l2 is DEFINED by knn2_kernel above (sequential float32 sums of (q - t)^2, ties to the lower
// index).  For big sets the same result comes cheaper: (1) km_shortlist ranks every train row of a split by
// a_j = |t_j|^2 - 2 q.t_j, the dot product on v_mfma_f32_32x32x2_f32 (an exact k-ordered fmaf chain), and keeps the four
// smallest per query and split; (2) km_refine evaluates the defining sum for those 4 S candidates and takes the two
// smallest; (3) a rigorous rounding bound turns the fourth-smallest a_j of every split into a lower bound on the defining
// sum of every row that is NOT a candidate -- if the exact second distance is strictly below it, no such row can enter or
// tie and the pair is final; (4) the few queries without a certificate (near-ties, duplicated descriptors) go through
// knn2_kernel.  Bit-identical to the exact search by construction, not by tolerance.
typedef float km_f16 __attribute__((ext_vector_type(16)));
constexpr int KM_NQ = 128, KM_NT = 128, KM_KC = 40, KM_TP = KM_KC + 4, KM_K = 4, KM_THREADS = 512;
constexpr int KM_MAX_DIM = 216;                 // Q tile (128 x (dim + 4) floats) + two T chunks + norms within 160 KB
constexpr int KM_CPR = KM_KC / 4;               // float4 per staged train row and chunk

__global__ __launch_bounds__(256) void km_norms(const float* __restrict__ t, int nt, int dim, float* __restrict__ nt2,
                                                unsigned* __restrict__ tmax_bits)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    if (j < nt) {
        const float4* r = reinterpret_cast<const float4*>(t + (size_t)j * dim);
        for (int k = 0; k < dim / 4; k++) {
            const float4 v = r[k];
            s = __builtin_fmaf(v.x, v.x, s); s = __builtin_fmaf(v.y, v.y, s);
            s = __builtin_fmaf(v.z, v.z, s); s = __builtin_fmaf(v.w, v.w, s);
        }
        nt2[j] = s;
    }
    // non-negative floats order like their bit patterns
    float m = s;
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(tmax_bits, __float_as_uint(m));
}

struct Top4 { float d[KM_K]; int i[KM_K]; };

__device__ __forceinline__ void top4_push(Top4& b, float v, int j)
{
    // sorted ascending; the order among equal approximate values does not matter (candidates are re-evaluated)
    const bool c0 = v < b.d[0], c1 = v < b.d[1], c2 = v < b.d[2];
    b.d[3] = c2 ? b.d[2] : v;          b.i[3] = c2 ? b.i[2] : j;
    b.d[2] = c1 ? b.d[1] : (c2 ? v : b.d[2]); b.i[2] = c1 ? b.i[1] : (c2 ? j : b.i[2]);
    b.d[1] = c0 ? b.d[0] : (c1 ? v : b.d[1]); b.i[1] = c0 ? b.i[0] : (c1 ? j : b.i[1]);
    b.d[0] = c0 ? v : b.d[0];          b.i[0] = c0 ? j : b.i[0];
}

// grid (query tiles, splits); 8 waves: wave w ranks the train rows 32 (w & 3) .. + 32 of every 128-row tile against the
// queries 64 (w >> 2) .. + 64 (two 32 x 32 accumulator tiles: train rows down the registers, one query per lane).
// The k index is permuted inside groups of 8 (lane half h takes k = 8 g + 4 h + s in step s) identically for both
// operands: one 16-byte LDS read per operand tile feeds four MFMAs.
__global__ __launch_bounds__(KM_THREADS) void km_shortlist(const float* __restrict__ q, const float* __restrict__ t,
                                                           const float* __restrict__ nt2, int nq, int nt, int dim,
                                                           int tiles_per_split, int* __restrict__ cand_idx,
                                                           float* __restrict__ cand_a4)
{
    extern __shared__ float lds[];
    const int dimp = (dim + 7) & ~7, qp = dimp + 4;
    float* Qs = lds;                                   // [KM_NQ][qp]
    float* Ts = Qs + KM_NQ * qp;                       // [2][KM_NT][KM_TP]
    float* Ns = Ts + 2 * KM_NT * KM_TP;                // [2][KM_NT]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w & 3, wn = w >> 2, lr = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.x * KM_NQ, split = blockIdx.y;
    const int tile0 = split * tiles_per_split, ntiles_all = (nt + KM_NT - 1) / KM_NT;
    const int ntiles = min(tiles_per_split, ntiles_all - tile0);
    const int cpt = (dimp + KM_KC - 1) / KM_KC;        // chunks per train tile
    const int nchunks = ntiles * cpt;

    for (int e = tid; e < KM_NQ * (qp / 4); e += KM_THREADS) {
        const int row = e / (qp / 4), k4 = e - row * (qp / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + row < nq && 4 * k4 < dim) v = reinterpret_cast<const float4*>(q + (size_t)(q0 + row) * dim)[k4];
        *reinterpret_cast<float4*>(Qs + row * qp + 4 * k4) = v;
    }

    float4 stage[3];
    float nstage = 0.f;
    auto fetch = [&](int c) {          // chunk c of the block's sequence -> registers
        const int tile = tile0 + c / cpt, k0 = (c % cpt) * KM_KC, t0 = tile * KM_NT;
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int e = tid + KM_THREADS * u, row = e / KM_CPR, c4 = e - row * KM_CPR;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < KM_NT * KM_CPR && t0 + row < nt && k0 + 4 * c4 < dim)
                v = reinterpret_cast<const float4*>(t + (size_t)(t0 + row) * dim + k0)[c4];
            stage[u] = v;
        }
        if (c % cpt == 0 && tid < KM_NT) nstage = t0 + tid < nt ? nt2[t0 + tid] : INFINITY;   // rows past the end rank last
    };
    auto commit = [&](int c) {
        float* dst = Ts + (c & 1) * KM_NT * KM_TP;
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int e = tid + KM_THREADS * u, row = e / KM_CPR, c4 = e - row * KM_CPR;
            if (e < KM_NT * KM_CPR) *reinterpret_cast<float4*>(dst + row * KM_TP + 4 * c4) = stage[u];
        }
        if (c % cpt == 0 && tid < KM_NT) Ns[((c / cpt) & 1) * KM_NT + tid] = nstage;
    };

    Top4 best[2];
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
        for (int e = 0; e < KM_K; e++) { best[n].d[e] = INFINITY; best[n].i[e] = 0x7fffffff; }

    if (nchunks > 0) { fetch(0); commit(0); }
    __syncthreads();
    km_f16 acc[2];
    for (int c = 0; c < nchunks; c++) {
        const int kc = c % cpt, k0 = kc * KM_KC;
        if (c + 1 < nchunks) fetch(c + 1);
        if (kc == 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        }
        const float* Tc = Ts + (c & 1) * KM_NT * KM_TP + (32 * wm + lr) * KM_TP + 4 * lh;
        const float* Qc = Qs + (64 * wn + lr) * qp + k0 + 4 * lh;
        const int groups = min(KM_KC, dimp - k0) / 8;
        auto group = [&](int g) {
            const float4 a = *reinterpret_cast<const float4*>(Tc + 8 * g);
            const float4 b0 = *reinterpret_cast<const float4*>(Qc + 8 * g);
            const float4 b1 = *reinterpret_cast<const float4*>(Qc + 32 * qp + 8 * g);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc[1], 0, 0, 0);
        };
        if (groups == KM_KC / 8) {      // a whole chunk (always, when dim is a multiple of 40): straight-line code
#pragma unroll
            for (int g = 0; g < KM_KC / 8; g++) group(g);
        } else {
            for (int g = 0; g < groups; g++) group(g);
        }
        if (kc == cpt - 1) {
            // a_j = |t_j|^2 - 2 q.t_j for the 16 train rows this lane holds of either query
            const int tile = c / cpt, t0 = (tile0 + tile) * KM_NT + 32 * wm + 4 * lh;
            const float* nrm = Ns + (tile & 1) * KM_NT + 32 * wm + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2);
                const float n2 = nrm[row];
#pragma unroll
                for (int n = 0; n < 2; n++) {
                    const float v = __builtin_fmaf(-2.f, acc[n][r], n2);
                    if (v < best[n].d[KM_K - 1]) top4_push(best[n], v, t0 + row);
                }
            }
        }
        if (c + 1 < nchunks) commit(c + 1);
        __syncthreads();
    }

    // the eight partial lists of every query (four row groups x two lane halves) -> its four smallest
    float* cd = Ts;                                            // [KM_NQ][8][KM_K]
    int* ci = reinterpret_cast<int*>(Ts + KM_NQ * 8 * KM_K);
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
        for (int e = 0; e < KM_K; e++) {
            const int o = ((64 * wn + 32 * n + lr) * 8 + wm * 2 + lh) * KM_K + e;
            cd[o] = best[n].d[e];
            ci[o] = best[n].i[e];
        }
    __syncthreads();
    if (tid < KM_NQ && q0 + tid < nq) {
        Top4 m;
#pragma unroll
        for (int e = 0; e < KM_K; e++) { m.d[e] = INFINITY; m.i[e] = 0x7fffffff; }
        for (int e = 0; e < 8 * KM_K; e++) {
            const float v = cd[tid * 8 * KM_K + e];
            if (v < m.d[KM_K - 1]) top4_push(m, v, ci[tid * 8 * KM_K + e]);
        }
        const size_t o = (size_t)split * nq + q0 + tid;
#pragma unroll
        for (int e = 0; e < KM_K; e++) cand_idx[o * KM_K + e] = m.i[e];
        cand_a4[o] = m.d[KM_K - 1];     // every row of the split that is not listed ranks at or above this
    }
}

// One thread per (query, candidate): the defining sum; then per query the two smallest by (distance, index) and the
// certificate.  sc = candidates per query rounded up to a power of two (<= 32), 256 / sc queries per block.
// half16: the shortlist came from kh_shortlist (split-float16 products on the FP16 matrix cores); its ranking values carry a
// larger error than the FP32 chain's (bound below); amax: bit patterns of the largest squared norm of the queries and of the
// train rows (km_norms), which set the operands' scales.
__global__ __launch_bounds__(256) void km_refine(const float* __restrict__ q, const float* __restrict__ t, int nq, int dim,
                                                 int nsplit, int sc, const int* __restrict__ cand_idx,
                                                 const float* __restrict__ cand_a4, const unsigned* __restrict__ tmax_bits,
                                                 int* __restrict__ idx, float* __restrict__ dist, int* __restrict__ qlist,
                                                 int* __restrict__ qcount, int half16, const unsigned* __restrict__ amax)
{
    __shared__ float sd[256];
    __shared__ int si[256];
    const int tid = threadIdx.x, qpb = 256 / sc, ql = tid / sc, c = tid - ql * sc;
    const int qi = blockIdx.x * qpb + ql;
    float d2 = INFINITY, qn = 0.f;
    int j = 0x7fffffff;
    if (qi < nq && c < nsplit * KM_K) {
        const int s = c / KM_K, e = c - s * KM_K;
        j = cand_idx[((size_t)s * nq + qi) * KM_K + e];
    }
    if (qi < nq && (j != 0x7fffffff || c == 0)) {
        // the loads of a row ten at a time, then the sums in their order: one load in front of every four terms was one
        // memory latency per 16 bytes, fifty in a row (205 us for the top level's 730 000 candidates).  Thread 0 of a query
        // also takes the query's squared norm for the certificate from the same registers (the fmaf chain of km_norms).
        const bool has = j != 0x7fffffff;
        const float4* qr = reinterpret_cast<const float4*>(q + (size_t)qi * dim);
        const float4* tr = reinterpret_cast<const float4*>(t + (size_t)(has ? j : 0) * dim);
        constexpr int KB = 10;
        const int n4 = dim / 4;
        float a = 0.f;
        for (int k0 = 0; k0 < n4; k0 += KB) {
            float4 x[KB], y[KB];
#pragma unroll
            for (int u = 0; u < KB; u++) {
                const int k = min(k0 + u, n4 - 1);
                x[u] = qr[k];
                y[u] = tr[k];
            }
#pragma unroll
            for (int u = 0; u < KB; u++) {
                if (k0 + u < n4) {
                    float d;
                    d = x[u].x - y[u].x; a = a + d * d;
                    d = x[u].y - y[u].y; a = a + d * d;
                    d = x[u].z - y[u].z; a = a + d * d;
                    d = x[u].w - y[u].w; a = a + d * d;
                    qn = __builtin_fmaf(x[u].x, x[u].x, qn); qn = __builtin_fmaf(x[u].y, x[u].y, qn);
                    qn = __builtin_fmaf(x[u].z, x[u].z, qn); qn = __builtin_fmaf(x[u].w, x[u].w, qn);
                }
            }
        }
        if (has) d2 = a;
    }
    sd[tid] = d2;
    si[tid] = j;
    __syncthreads();
    if (c != 0 || qi >= nq) return;
    float d0 = INFINITY, d1 = INFINITY;
    int i0 = 0x7fffffff, i1 = 0x7fffffff;
    for (int e = 0; e < sc; e++) {
        const float d = sd[tid + e];
        const int i = si[tid + e];
        if (i == 0x7fffffff) continue;
        if (before(d, i, d0, i0)) { d1 = d0; i1 = i0; d0 = d; i0 = i; }
        else if (before(d, i, d1, i1)) { d1 = d; i1 = i; }
    }
    // certificate, in double.  u = 2^-24.  a_j differs from T_j = |t_j|^2 - 2 q.t_j by at most
    // g (|t_j|^2 + 2 |q| |t_j|), g = (dim + 4) u (two fmaf chains of dim terms and one more rounding); the defining sum
    // s_j is at least (T_j + |q|^2) (1 - (dim + 3) u) (a sum of non-negative terms, three roundings per term and one
    // per addition).  Norms enter through their computed values, inflated by their own bound.
    float a4 = INFINITY;
    for (int s = 0; s < nsplit; s++) a4 = fminf(a4, cand_a4[(size_t)s * nq + qi]);
    const double u = 5.9604644775390625e-8, g = (dim + 4) * u * 1.01, dl = (dim + 3) * u * 1.01;
    const double tm = (double)__uint_as_float(*tmax_bits) * (1.0 + 2.0 * g), qq = (double)qn;
    double err = g * (tm + 2.0 * sqrt(qq * (1.0 + 2.0 * g) * tm)) * 1.001;
    if (half16) {
        // a_j = n2_j - 2 D with n2_j the FP32 chain of km_norms (error <= g |t_j|^2) and D the sum of the three split products
        // xh yh + xh yl + xl yh of the scaled operands on v_mfma_f32_32x32x16_f16.  With x = xh + xl + e:
        //   representation  |e| <= 2^-22 |x| (two roundings to 11 bits) where the low part is a normal float16, else the low
        //                   part's magnitude <= 2^-14 in scaled units (a denormal the unit may flush) = e_abs per element;
        //   dropped term    |xl yl| <= 2^-22 |x||y|;
        //   accumulation    products of two 11-bit numbers are exact in FP32; the instruction aligns its sixteen products and
        //                   the accumulator to the largest of them and drops what falls below the last place (measured,
        //                   tools/ubench_mfma_f16_err.hip: up to 0.89 K u sum|terms| on operands spanning nine decades; plain
        //                   truncation to 24 bits would allow 2 u per term): every addition is charged FOUR units u = 2^-24 of
        //                   the running sum of magnitudes; float16 DENORMAL operands are flushed (same measurement): e_abs.
        // |D - q.t| <= eps |q||t| + e_abs sqrt(dim) (|q| + |t|),  eps = 4 * 2^-22 + 4 * (3 dimp + 2) * 2^-24, all times 1.01.
        const int dimp = (dim + 15) & ~15;
        const double eps = (4.0 * 2.384185791015625e-7 + 4.0 * (3.0 * dimp + 2.0) * u) * 1.01;
        // the scale of a set is at least 2^11.5 / (its largest norm): a scaled 2^-14 is at most 2^-25.5 of that norm
        const double aq = sqrt((double)__uint_as_float(amax[0])), at = sqrt((double)__uint_as_float(amax[1]));
        const double eabs = 2.1073424255447017e-8 * (aq > at ? aq : at) * 1.01;
        const double nq2 = sqrt(qq * (1.0 + 2.0 * g)), nt2 = sqrt(tm);
        err = (g * tm + 2.0 * (eps * nq2 * nt2 + eabs * sqrt((double)dim) * (nq2 + nt2)) + u * (tm + 2.0 * nq2 * nt2)) * 1.001;
    }
    const double lower = ((double)a4 - err + qq * (1.0 - 2.0 * g)) * (1.0 - dl);
    // The bounds are RELATIVE ones: they hold while squared norms and products stay in float32's normal range.  Sets whose
    // magnitudes sit near its ends (|x| ~ 1e-20: squared norms are denormals, the split operands' unscaling underflows) get
    // no certificate at all -- every query then takes the exact pass.
    bool in_range = tm > 1e-30 && tm < 1e30;
    if (half16) {
        const int eq = (int)((amax[0] >> 23) & 255u) - 127, et = (int)((amax[1] >> 23) & 255u) - 127;
        in_range = in_range && amax[0] != 0u && amax[1] != 0u && eq > -80 && eq < 80 && et > -80 && et < 80;
    }
    const bool certified = in_range && (double)d1 < lower;      // strictly: a tie would have to be broken by index
    idx[(size_t)qi * 2] = i0;
    idx[(size_t)qi * 2 + 1] = i1;
    dist[(size_t)qi * 2] = d0;
    dist[(size_t)qi * 2 + 1] = d1;
    if (!certified) qlist[atomicAdd(qcount, 1)] = qi;
}

// ---- split-float16 shortlist (round 6) ---------------------------------------------------------------------------------
// The same ranking a_j = |t_j|^2 - 2 q.t_j with the dot product on the FP16 matrix cores (v_mfma_f32_32x32x16_f16: sixteen
// times the FP32 rate).  Every operand is scaled by a power of two (its set's largest magnitude goes to [2^12, 2^13)) and cut
// into two float16 numbers, x = xh + xl (22 bits of the 24); q.t ~ sum qh th + qh tl + ql th, three matrix instructions per
// sixteen dimensions, accumulated in FP32.  What that costs in accuracy is in km_refine's bound; the certificate decides, as
// before, which queries need the exact pass -- the RESULT stays that of knn2_kernel, bit for bit.
typedef _Float16 kh_h8 __attribute__((ext_vector_type(8)));
constexpr int KH_NQ = 128, KH_NT = 128, KH_KC = 32, KH_TP = KH_KC + 8, KH_THREADS = 512;
constexpr int KH_MAX_DIM = 208;       // Q tile 2 x 128 x (208 + 8) halves + two T chunks + norms within 160 KB

// exponent e of the scale 2^e for a set whose largest squared row norm has these bits: no element exceeds the largest norm,
// which the scale takes into [2^11.5, 2^12.5) (km_norms has the norms anyway; a pass for the largest element would cost more
// than the four binades of float16 range it would gain)
__device__ __forceinline__ int kh_scale_exp(unsigned maxn2_bits)
{
    const int e2 = (int)((maxn2_bits >> 23) & 255u) - 127;       // largest squared norm in [2^e2, 2^(e2 + 1))
    if (maxn2_bits == 0u || e2 <= -80 || e2 >= 80) return 0;     // empty / tiny / huge: unscaled (km_refine certifies nothing then)
    return 12 - ((e2 + 1) >> 1);                                 // largest norm < 2^((e2 + 1) / 2) <= 2^(ex + 1/2)
}

__global__ __launch_bounds__(256) void kh_split(const float* __restrict__ x, int n, int dim, int dimp,
                                                const unsigned* __restrict__ maxbits, _Float16* __restrict__ xh,
                                                _Float16* __restrict__ xl)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)n * dimp) return;
    const int row = (int)(e / dimp), k = (int)(e - (size_t)row * dimp);
    float v = k < dim ? ldexpf(x[(size_t)row * dim + k], kh_scale_exp(*maxbits)) : 0.f;
    const _Float16 h = (_Float16)v;
    xh[e] = h;
    xl[e] = (_Float16)(v - (float)h);
}

// grid (query tiles, splits); 8 waves as in km_shortlist: wave w ranks the train rows 32 (w & 3) .. + 32 of every 128-row tile
// against the queries 64 (w >> 2) .. + 64 (two 32 x 32 accumulator tiles).  Lane half h holds the dimensions 16 s + 8 h .. + 8
// of step s for both operands (one 16-byte LDS read each).
__global__ __launch_bounds__(KH_THREADS) void kh_shortlist(const _Float16* __restrict__ qh, const _Float16* __restrict__ ql,
                                                           const _Float16* __restrict__ th, const _Float16* __restrict__ tl,
                                                           const float* __restrict__ nt2, int nq, int nt, int dimp,
                                                           int tiles_per_split, const unsigned* __restrict__ amax,
                                                           int* __restrict__ cand_idx, float* __restrict__ cand_a4)
{
    extern __shared__ _Float16 hl[];
    const int qp = dimp + 8;
    _Float16* Qh = hl;                                 // [KH_NQ][qp]
    _Float16* Ql = Qh + KH_NQ * qp;
    _Float16* Th = Ql + KH_NQ * qp;                    // [2][KH_NT][KH_TP]
    _Float16* Tl = Th + 2 * KH_NT * KH_TP;
    float* Ns = reinterpret_cast<float*>(Tl + 2 * KH_NT * KH_TP);   // [2][KH_NT]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w & 3, wn = w >> 2, lr = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.x * KH_NQ, split = blockIdx.y;
    const int tile0 = split * tiles_per_split, ntiles_all = (nt + KH_NT - 1) / KH_NT;
    const int ntiles = min(tiles_per_split, ntiles_all - tile0);
    const int cpt = (dimp + KH_KC - 1) / KH_KC;        // chunks per train tile
    const int nchunks = ntiles * cpt;
    // -2 / (scale of q x scale of t): exact (a power of two)
    const float unscale = ldexpf(-2.f, -(kh_scale_exp(amax[0]) + kh_scale_exp(amax[1])));

    {
        const int segs = dimp / 8;                     // 16-byte pieces per row
        for (int e = tid; e < KH_NQ * segs; e += KH_THREADS) {
            const int row = e / segs, sg = e - row * segs;
            uint4 vh = make_uint4(0, 0, 0, 0), vl = vh;
            if (q0 + row < nq) {
                vh = reinterpret_cast<const uint4*>(qh + (size_t)(q0 + row) * dimp)[sg];
                vl = reinterpret_cast<const uint4*>(ql + (size_t)(q0 + row) * dimp)[sg];
            }
            *reinterpret_cast<uint4*>(Qh + row * qp + 8 * sg) = vh;
            *reinterpret_cast<uint4*>(Ql + row * qp + 8 * sg) = vl;
        }
    }

    uint4 sh = make_uint4(0, 0, 0, 0), sl = sh;
    float nstage = 0.f;
    const int srow = tid >> 2, sseg = tid & 3;         // this thread's piece of a staged chunk: row, 8-half segment
    auto fetch = [&](int c) {
        const int tile = tile0 + c / cpt, k0 = (c % cpt) * KH_KC, t0 = tile * KH_NT;
        sh = make_uint4(0, 0, 0, 0); sl = sh;
        if (t0 + srow < nt && k0 + 8 * sseg < dimp) {
            sh = *reinterpret_cast<const uint4*>(th + (size_t)(t0 + srow) * dimp + k0 + 8 * sseg);
            sl = *reinterpret_cast<const uint4*>(tl + (size_t)(t0 + srow) * dimp + k0 + 8 * sseg);
        }
        if (c % cpt == 0 && tid < KH_NT) nstage = t0 + tid < nt ? nt2[t0 + tid] : INFINITY;   // rows past the end rank last
    };
    auto commit = [&](int c) {
        *reinterpret_cast<uint4*>(Th + ((c & 1) * KH_NT + srow) * KH_TP + 8 * sseg) = sh;
        *reinterpret_cast<uint4*>(Tl + ((c & 1) * KH_NT + srow) * KH_TP + 8 * sseg) = sl;
        if (c % cpt == 0 && tid < KH_NT) Ns[((c / cpt) & 1) * KH_NT + tid] = nstage;
    };

    Top4 best[2];
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
        for (int e = 0; e < KM_K; e++) { best[n].d[e] = INFINITY; best[n].i[e] = 0x7fffffff; }

    if (nchunks > 0) { fetch(0); commit(0); }
    __syncthreads();
    km_f16 acc[2];
    for (int c = 0; c < nchunks; c++) {
        const int kc = c % cpt, k0 = kc * KH_KC;
        if (c + 1 < nchunks) fetch(c + 1);
        if (kc == 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        }
        const int toff = ((c & 1) * KH_NT + 32 * wm + lr) * KH_TP + 8 * lh;
        const int qoff = (64 * wn + lr) * qp + k0 + 8 * lh;
        const int steps = min(KH_KC, dimp - k0) / 16;
        for (int st = 0; st < steps; st++) {
            const kh_h8 ah = *reinterpret_cast<const kh_h8*>(Th + toff + 16 * st);
            const kh_h8 al = *reinterpret_cast<const kh_h8*>(Tl + toff + 16 * st);
            const kh_h8 b0h = *reinterpret_cast<const kh_h8*>(Qh + qoff + 16 * st);
            const kh_h8 b0l = *reinterpret_cast<const kh_h8*>(Ql + qoff + 16 * st);
            const kh_h8 b1h = *reinterpret_cast<const kh_h8*>(Qh + qoff + 32 * qp + 16 * st);
            const kh_h8 b1l = *reinterpret_cast<const kh_h8*>(Ql + qoff + 32 * qp + 16 * st);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0h, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1h, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0l, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1l, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b0h, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b1h, acc[1], 0, 0, 0);
        }
        if (kc == cpt - 1) {
            // a_j = |t_j|^2 - 2 q.t_j for the 16 train rows this lane holds of either query
            const int tile = c / cpt, t0 = (tile0 + tile) * KH_NT + 32 * wm + 4 * lh;
            const float* nrm = Ns + (tile & 1) * KH_NT + 32 * wm + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2);
                const float n2 = nrm[row];
#pragma unroll
                for (int n = 0; n < 2; n++) {
                    const float v = __builtin_fmaf(unscale, acc[n][r], n2);
                    if (v < best[n].d[KM_K - 1]) top4_push(best[n], v, t0 + row);
                }
            }
        }
        if (c + 1 < nchunks) commit(c + 1);
        __syncthreads();
    }

    // the eight partial lists of every query (four row groups x two lane halves) -> its four smallest
    float* cd = reinterpret_cast<float*>(Th);                  // [KH_NQ][8][KM_K]: 16 KB of the T buffers' 40
    int* ci = reinterpret_cast<int*>(cd + KH_NQ * 8 * KM_K);
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
        for (int e = 0; e < KM_K; e++) {
            const int o = ((64 * wn + 32 * n + lr) * 8 + wm * 2 + lh) * KM_K + e;
            cd[o] = best[n].d[e];
            ci[o] = best[n].i[e];
        }
    __syncthreads();
    if (tid < KH_NQ && q0 + tid < nq) {
        Top4 m;
#pragma unroll
        for (int e = 0; e < KM_K; e++) { m.d[e] = INFINITY; m.i[e] = 0x7fffffff; }
        for (int e = 0; e < 8 * KM_K; e++) {
            const float v = cd[tid * 8 * KM_K + e];
            if (v < m.d[KM_K - 1]) top4_push(m, v, ci[tid * 8 * KM_K + e]);
        }
        const size_t o = (size_t)split * nq + q0 + tid;
#pragma unroll
        for (int e = 0; e < KM_K; e++) cand_idx[o * KM_K + e] = m.i[e];
        cand_a4[o] = m.d[KM_K - 1];     // every row of the split that is not listed ranks at or above this
    }
}

// The same ranking with the QUERY operands in registers (descriptors of STEPS * 16 dimensions; DAISY's 200 -> 13 steps).
// kh_shortlist reads both operands of every matrix instruction from LDS -- 6 KB per wave and 16 dimensions for 6 instructions --
// and ran at a fifth of the FP16 matrix rate (1.41 ms for 22 800 x 22 900 descriptors).  Here a wave owns 32 QT queries for the
// life of the block, their split operands in QT x STEPS x 2 vector registers of 8 halves each (104 VGPRs per query tile at 13
// steps), and only the train rows pass through LDS: 64-row tiles, whole rows, double buffered.
//   QT = 2: 4 waves (one per SIMD), 2 KB of LDS reads (ah, al) per 6 matrix instructions;
//   QT = 1: 8 waves (two per SIMD), 2 KB per 3 -- but the top-4 insertion of one wave's candidates (VALU: about as many cycles
//           as the sub-tile's matrix instructions) runs under the other wave's matrix instructions.
// grid (tiles of 256 queries, splits of the train set).
constexpr int KR_NQ = 256, KR_NT = 64;

template <int STEPS, int QT>
__global__ __launch_bounds__(KR_NQ / (32 * QT) * 64) void kh_shortlist_regq(const _Float16* __restrict__ qh, const _Float16* __restrict__ ql,
                                                                        const _Float16* __restrict__ th, const _Float16* __restrict__ tl,
                                                                        const float* __restrict__ nt2, int nq, int nt,
                                                                        int tiles_per_split, const unsigned* __restrict__ amax,
                                                                        int* __restrict__ cand_idx, float* __restrict__ cand_a4)
{
    constexpr int THREADS = KR_NQ / (32 * QT) * 64;
    constexpr int DIMP = STEPS * 16, TP = DIMP + 8;     // row pitch in halves: 16-byte reads of 8 rows fall into distinct banks
    constexpr int SEGS = DIMP / 8;                      // 16-byte pieces per row
    constexpr int PER = (KR_NT * SEGS + THREADS - 1) / THREADS;   // pieces a thread stages per operand and tile
    extern __shared__ _Float16 hl[];
    _Float16* Th = hl;                                  // [2][KR_NT][TP]
    _Float16* Tl = Th + 2 * KR_NT * TP;
    float* Ns = reinterpret_cast<float*>(Tl + 2 * KR_NT * TP);   // [2][KR_NT]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.x * KR_NQ + 32 * QT * w, split = blockIdx.y;
    const int tile0 = split * tiles_per_split, ntiles_all = (nt + KR_NT - 1) / KR_NT;
    const int ntiles = min(tiles_per_split, ntiles_all - tile0);
    const float unscale = ldexpf(-2.f, -(kh_scale_exp(amax[0]) + kh_scale_exp(amax[1])));

    // this lane's query operands: query q0 + 32 n + lr, dimensions 16 s + 8 lh .. + 8 (rows past the end: zeros)
    kh_h8 bh[QT][STEPS], bl[QT][STEPS];
#pragma unroll
    for (int n = 0; n < QT; n++) {
        const int qi = q0 + 32 * n + lr;
#pragma unroll
        for (int st = 0; st < STEPS; st++) {
            uint4 vh = make_uint4(0, 0, 0, 0), vl = vh;
            if (qi < nq) {
                vh = *reinterpret_cast<const uint4*>(qh + (size_t)qi * DIMP + 16 * st + 8 * lh);
                vl = *reinterpret_cast<const uint4*>(ql + (size_t)qi * DIMP + 16 * st + 8 * lh);
            }
            bh[n][st] = *reinterpret_cast<kh_h8*>(&vh);
            bl[n][st] = *reinterpret_cast<kh_h8*>(&vl);
        }
    }

    uint4 sh[PER], sl[PER];
    float nstage = 0.f;
    auto fetch = [&](int c) {                           // train tile c of the block's sequence -> registers
        const int t0 = (tile0 + c) * KR_NT;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int e = tid + THREADS * u, row = e / SEGS, sg = e - row * SEGS;
            sh[u] = make_uint4(0, 0, 0, 0); sl[u] = sh[u];
            if (e < KR_NT * SEGS && t0 + row < nt) {
                sh[u] = *reinterpret_cast<const uint4*>(th + (size_t)(t0 + row) * DIMP + 8 * sg);
                sl[u] = *reinterpret_cast<const uint4*>(tl + (size_t)(t0 + row) * DIMP + 8 * sg);
            }
        }
        if (tid < KR_NT) nstage = t0 + tid < nt ? nt2[t0 + tid] : INFINITY;      // rows past the end rank last
    };
    auto commit = [&](int c) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int e = tid + THREADS * u, row = e / SEGS, sg = e - row * SEGS;
            if (e < KR_NT * SEGS) {
                *reinterpret_cast<uint4*>(Th + ((c & 1) * KR_NT + row) * TP + 8 * sg) = sh[u];
                *reinterpret_cast<uint4*>(Tl + ((c & 1) * KR_NT + row) * TP + 8 * sg) = sl[u];
            }
        }
        if (tid < KR_NT) Ns[(c & 1) * KR_NT + tid] = nstage;
    };

    Top4 best[QT];
#pragma unroll
    for (int n = 0; n < QT; n++)
#pragma unroll
        for (int e = 0; e < KM_K; e++) { best[n].d[e] = INFINITY; best[n].i[e] = 0x7fffffff; }

    if (ntiles > 0) { fetch(0); commit(0); }
    __syncthreads();
    for (int c = 0; c < ntiles; c++) {
        if (c + 1 < ntiles) fetch(c + 1);               // in flight during this tile's arithmetic
#pragma unroll
        for (int sub = 0; sub < KR_NT / 32; sub++) {    // 32 train rows at a time against the wave's query tiles
            km_f16 acc[QT];
#pragma unroll
            for (int n = 0; n < QT; n++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
            const _Float16* ta = Th + ((c & 1) * KR_NT + 32 * sub + lr) * TP + 8 * lh;
            const _Float16* tb = Tl + ((c & 1) * KR_NT + 32 * sub + lr) * TP + 8 * lh;
#pragma unroll
            for (int st = 0; st < STEPS; st++) {
                const kh_h8 ah = *reinterpret_cast<const kh_h8*>(ta + 16 * st);
                const kh_h8 al = *reinterpret_cast<const kh_h8*>(tb + 16 * st);
#pragma unroll
                for (int n = 0; n < QT; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[n][st], acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < QT; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[n][st], acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < QT; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[n][st], acc[n], 0, 0, 0);
            }
            // a_j = |t_j|^2 - 2 q.t_j for the 16 train rows this lane holds of either query
            const int t0 = (tile0 + c) * KR_NT + 32 * sub + 4 * lh;
            const float* nrm = Ns + (c & 1) * KR_NT + 32 * sub + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2);
                const float n2 = nrm[row];
#pragma unroll
                for (int n = 0; n < QT; n++) {
                    const float v = __builtin_fmaf(unscale, acc[n][r], n2);
                    if (v < best[n].d[KM_K - 1]) top4_push(best[n], v, t0 + row);
                }
            }
        }
        if (c + 1 < ntiles) commit(c + 1);
        __syncthreads();
    }

    // the two partial lists of every query (the lane halves hold different train rows) -> its four smallest
    float* cd = reinterpret_cast<float*>(Th);                  // [KR_NQ][2][KM_K]
    int* ci = reinterpret_cast<int*>(cd + KR_NQ * 2 * KM_K);
#pragma unroll
    for (int n = 0; n < QT; n++)
#pragma unroll
        for (int e = 0; e < KM_K; e++) {
            const int o = ((32 * QT * w + 32 * n + lr) * 2 + lh) * KM_K + e;
            cd[o] = best[n].d[e];
            ci[o] = best[n].i[e];
        }
    __syncthreads();
    for (int ql_ = tid; ql_ < KR_NQ; ql_ += THREADS) {
        const int qq = blockIdx.x * KR_NQ + ql_;
        if (qq >= nq) continue;
        Top4 m;
#pragma unroll
        for (int e = 0; e < KM_K; e++) { m.d[e] = INFINITY; m.i[e] = 0x7fffffff; }
        for (int e = 0; e < 2 * KM_K; e++) {
            const float v = cd[ql_ * 2 * KM_K + e];
            if (v < m.d[KM_K - 1]) top4_push(m, v, ci[ql_ * 2 * KM_K + e]);
        }
        const size_t o = (size_t)split * nq + qq;
#pragma unroll
        for (int e = 0; e < KM_K; e++) cand_idx[o * KM_K + e] = m.i[e];
        cand_a4[o] = m.d[KM_K - 1];     // every row of the split that is not listed ranks at or above this
    }
}

} // namespace

// qlist == nullptr: every query, one block per 64 of them over the whole train set.  With a list (the uncertified
// queries of the filtered search, few) the train set is cut into `nsplit` ranges served by separate blocks, so that the
// list costs microseconds instead of one block's pass over every train row; part: nsplit x nq float4 of scratch.
static int knn2_exact(ma_ctx* ctx, const float* query, int nq, const float* train, int nt, int dim, int* idx_out,
                      float* dist_out, const int* qlist, const int* qcount, int nsplit, float4* part)
{
    size_t lds = (size_t)(KN_TQ * (dim + 4) + 2 * KN_TT * KN_TP) * sizeof(float);
    lds = std::max(lds, (size_t)KN_TQ * 16 * 4 * sizeof(float));   // the final merge reuses the buffer
    MA_REQUIRE(lds <= 160 * 1024, "descriptor length out of range");
    MaProfScope ps(ctx, MA_K_OTHER, (double)nq);
    const int split_rows = nsplit > 1 ? ((nt + nsplit - 1) / nsplit + KN_TT - 1) / KN_TT * KN_TT : nt;
    if (nsplit > 1) nsplit = (nt + split_rows - 1) / split_rows;
    // all queries: a block per 64 of them; a list (count on the device): at most 16 blocks per train range walk it
    const int qblocks = (nq + KN_TQ - 1) / KN_TQ;
    const dim3 grid = qlist ? dim3(nsplit, std::min(qblocks, 16)) : dim3(qblocks, nsplit);
    // the same sums on packed float32 instructions for the LIST of the filtered search -- a few blocks, a wave per SIMD: half
    // the instructions, 120 -> 99 us for the coarse levels' lists; with more blocks the scalar form is the faster one (8.5 against
    // 9.2 ms for all 22 800 x 22 900 descriptors of the top level, tools/bench_match.py, three alternations on one box)
    const bool packed = qlist != nullptr;
    const void* fn = packed ? reinterpret_cast<const void*>(knn2_kernel<true>) : reinterpret_cast<const void*>(knn2_kernel<false>);
    if (lds > 64 * 1024) MA_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (packed)
        hipLaunchKernelGGL(knn2_kernel<true>, grid, dim3(256), lds, ctx->stream, query, train, nq, nt, dim, idx_out, dist_out, qlist,
                           qcount, split_rows, part);
    else
        hipLaunchKernelGGL(knn2_kernel<false>, grid, dim3(256), lds, ctx->stream, query, train, nq, nt, dim, idx_out, dist_out, qlist,
                           qcount, split_rows, part);
    if (nsplit > 1)
        hipLaunchKernelGGL(knn2_merge_parts, dim3(std::min((nq + 3) / 4, 1024)), dim3(256), 0, ctx->stream, part, nq, nsplit, qlist,
                           qcount, idx_out, dist_out);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

extern "C" int ma_knn2_l2_ex(ma_ctx* ctx, const float* query, int nq, const float* train, int nt, int dim, int* idx_out,
                             float* dist_out, int mode, int* uncertified_host)
{
    MA_REQUIRE(ctx && query && train && idx_out && dist_out, "NULL argument");
    MA_REQUIRE(nq >= 1 && nt >= 2, "need at least one query and two train descriptors");
    MA_REQUIRE(dim >= 4 && dim % 4 == 0, "the descriptor length must be a multiple of 4 (pad with zeros)");
    MA_REQUIRE(mode == MA_KNN_AUTO || mode == MA_KNN_EXACT || mode == MA_KNN_FILTERED || mode == MA_KNN_FILTERED_F32,
               "unknown search mode");
    MA_REQUIRE((mode != MA_KNN_FILTERED && mode != MA_KNN_FILTERED_F32) || dim <= KM_MAX_DIM,
               "the filtered search holds descriptors of up to 216 floats");
    MA_HIP(hipSetDevice(ctx->device));
    if (uncertified_host) *uncertified_host = 0;
    // the shortlist pays once the distance matrix is big; below that the exact kernel is already launch-bound
    const bool filtered = mode == MA_KNN_FILTERED || mode == MA_KNN_FILTERED_F32 ||
                          (mode == MA_KNN_AUTO && dim <= KM_MAX_DIM && nt >= 4 * KM_NT && (double)nq * nt >= 4e6);
    // the shortlist on the FP16 matrix cores (split-float16 operands) unless the caller asks for round 3's FP32 one
    static const bool env_f32 = [] { const char* e = getenv("MICROALIGNER_KNN_SHORTLIST"); return e && std::string(e) == "f32"; }();
    const bool half16 = filtered && mode != MA_KNN_FILTERED_F32 && !env_f32 && dim <= KH_MAX_DIM;
    if (!filtered) return knn2_exact(ctx, query, nq, train, nt, dim, idx_out, dist_out, nullptr, nullptr, 1, nullptr);

    // splits of the train set: enough blocks to fill 256 CUs in whole rounds, every split at least one 128-row tile
    const int nqt = (nq + KM_NQ - 1) / KM_NQ, ntiles = (nt + KM_NT - 1) / KM_NT;
    int nsplit = 1;
    double best = 0.0;
    for (int s = 1; s <= std::min(8, ntiles); s++) {
        const double blocks = (double)nqt * s, eff = blocks / (256.0 * std::ceil(blocks / 256.0));
        if (eff > best + 0.02) { best = eff; nsplit = s; }
    }
    int tiles_per_split = (ntiles + nsplit - 1) / nsplit;
    nsplit = (ntiles + tiles_per_split - 1) / tiles_per_split;
    // descriptors of 208 (padded) dimensions -- DAISY's 200 -- take the shortlist with the queries in registers: tiles of 256
    // queries and 64 train rows
    const bool regq = half16 && ((dim + 15) & ~15) == 208;
    int nqt_r = 0;
    if (regq) {
        nqt_r = (nq + KR_NQ - 1) / KR_NQ;
        const int nt64 = (nt + KR_NT - 1) / KR_NT;
        nsplit = 1;
        best = 0.0;
        for (int s = 1; s <= std::min(8, nt64); s++) {
            const double blocks = (double)nqt_r * s, eff = blocks / (256.0 * std::ceil(blocks / 256.0));
            if (eff > best + 0.02) { best = eff; nsplit = s; }
        }
        tiles_per_split = (nt64 + nsplit - 1) / nsplit;
        nsplit = (nt64 + tiles_per_split - 1) / tiles_per_split;
    }
    int sc = 4;
    while (sc < nsplit * KM_K) sc *= 2;

    // train ranges of the exact fallback.  A block works through a 64-query tile whatever the number of real queries in it, so
    // the few uncertified queries (one to six tiles) are spread over the chip by cutting the train set finely: one 64-row tile
    // per range where the partial results (16 bytes per query and range) stay within 256 MB
    const int fsplit = (int)std::max<size_t>(1, std::min<size_t>({(size_t)256, ((size_t)nt + 63) / 64, ((size_t)256 << 20) / ((size_t)nq * 16)}));
    const size_t b_nt2 = ma_align_up((size_t)nt * 4, 256), b_ci = ma_align_up((size_t)nsplit * nq * KM_K * 4, 256),
                 b_a4 = ma_align_up((size_t)nsplit * nq * 4, 256), b_ql = ma_align_up((size_t)nq * 4, 256),
                 b_part = fsplit > 1 ? (size_t)fsplit * nq * sizeof(float4) : 0;
    const int dimh = (dim + 15) & ~15;                           // split operands: rows of dimh float16
    const size_t b_qs = half16 ? ma_align_up((size_t)nq * dimh * 2, 256) : 0, b_ts = half16 ? ma_align_up((size_t)nt * dimh * 2, 256) : 0;
    const size_t b_qn = half16 ? ma_align_up((size_t)nq * 4, 256) : 0;
    char* ws = static_cast<char*>(ma_pool_alloc(ctx, b_nt2 + b_ci + b_a4 + b_ql + 256 + b_part + 2 * b_qs + 2 * b_ts + b_qn));
    if (!ws) return MA_ENOMEM;
    float* nt2 = reinterpret_cast<float*>(ws);
    int* cand_idx = reinterpret_cast<int*>(ws + b_nt2);
    float* cand_a4 = reinterpret_cast<float*>(ws + b_nt2 + b_ci);
    int* qlist = reinterpret_cast<int*>(ws + b_nt2 + b_ci + b_a4);
    unsigned* tmax = reinterpret_cast<unsigned*>(ws + b_nt2 + b_ci + b_a4 + b_ql);
    int* qcount = reinterpret_cast<int*>(tmax + 1);
    float4* part = b_part ? reinterpret_cast<float4*>(ws + b_nt2 + b_ci + b_a4 + b_ql + 256) : nullptr;
    unsigned* amax = tmax + 2;                                   // max |q|, max |t| (bit patterns)
    char* split0 = ws + b_nt2 + b_ci + b_a4 + b_ql + 256 + b_part;
    _Float16 *qh = (_Float16*)split0, *ql = (_Float16*)(split0 + b_qs), *th = (_Float16*)(split0 + 2 * b_qs),
             *tl = (_Float16*)(split0 + 2 * b_qs + b_ts);
    float* qn2 = reinterpret_cast<float*>(split0 + 2 * b_qs + 2 * b_ts);
    int rc = MA_OK;
    do {
        if (hipMemsetAsync(tmax, 0, 16, ctx->stream) != hipSuccess) { rc = MA_EHIP; break; }
        if (half16) {
            MaProfScope ps(ctx, MA_K_OTHER, (double)nq);
            hipLaunchKernelGGL(km_norms, dim3((nt + 255) / 256), dim3(256), 0, ctx->stream, train, nt, dim, nt2, tmax);
            // the queries' norms set their scale (amax[0]); the train rows' are km_norms' tmax, copied beside it (amax[1])
            hipLaunchKernelGGL(km_norms, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, query, nq, dim, qn2, amax);
            if (hipMemcpyAsync(amax + 1, tmax, sizeof(unsigned), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { rc = MA_EHIP; break; }
            hipLaunchKernelGGL(kh_split, dim3((unsigned)(((size_t)nq * dimh + 255) / 256)), dim3(256), 0, ctx->stream, query, nq, dim, dimh,
                               (const unsigned*)amax, qh, ql);
            hipLaunchKernelGGL(kh_split, dim3((unsigned)(((size_t)nt * dimh + 255) / 256)), dim3(256), 0, ctx->stream, train, nt, dim, dimh,
                               (const unsigned*)(amax + 1), th, tl);
            if (regq) {
                const size_t lds = (size_t)(4 * KR_NT * (208 + 8)) * sizeof(_Float16) + 2 * KR_NT * sizeof(float);
                static const int qt = [] { const char* e = getenv("MICROALIGNER_KNN_QT"); return e && e[0] == '2' ? 2 : 1; }();
                hipError_t he;
                if (qt == 2) {
                    he = hipFuncSetAttribute(reinterpret_cast<const void*>(kh_shortlist_regq<13, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    if (he == hipSuccess)
                        hipLaunchKernelGGL((kh_shortlist_regq<13, 2>), dim3(nqt_r, nsplit), dim3(256), lds, ctx->stream,
                                           (const _Float16*)qh, (const _Float16*)ql, (const _Float16*)th, (const _Float16*)tl,
                                           (const float*)nt2, nq, nt, tiles_per_split, (const unsigned*)amax, cand_idx, cand_a4);
                } else {
                    he = hipFuncSetAttribute(reinterpret_cast<const void*>(kh_shortlist_regq<13, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    if (he == hipSuccess)
                        hipLaunchKernelGGL((kh_shortlist_regq<13, 1>), dim3(nqt_r, nsplit), dim3(512), lds, ctx->stream,
                                           (const _Float16*)qh, (const _Float16*)ql, (const _Float16*)th, (const _Float16*)tl,
                                           (const float*)nt2, nq, nt, tiles_per_split, (const unsigned*)amax, cand_idx, cand_a4);
                }
                if (he != hipSuccess) { rc = MA_EHIP; break; }
            } else {
                const size_t lds = (size_t)(2 * KH_NQ * (dimh + 8) + 4 * KH_NT * KH_TP) * sizeof(_Float16) + 2 * KH_NT * sizeof(float);
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(kh_shortlist), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds) != hipSuccess) { rc = MA_EHIP; break; }
                hipLaunchKernelGGL(kh_shortlist, dim3(nqt, nsplit), dim3(KH_THREADS), lds, ctx->stream, (const _Float16*)qh,
                                   (const _Float16*)ql, (const _Float16*)th, (const _Float16*)tl, (const float*)nt2, nq, nt, dimh,
                                   tiles_per_split, (const unsigned*)amax, cand_idx, cand_a4);
            }
            hipLaunchKernelGGL(km_refine, dim3((nq + 256 / sc - 1) / (256 / sc)), dim3(256), 0, ctx->stream, query, train, nq,
                               dim, nsplit, sc, cand_idx, cand_a4, tmax, idx_out, dist_out, qlist, qcount, 1, (const unsigned*)amax);
            if (hipGetLastError() != hipSuccess) { rc = MA_EHIP; break; }
        } else {
            MaProfScope ps(ctx, MA_K_OTHER, (double)nq);
            hipLaunchKernelGGL(km_norms, dim3((nt + 255) / 256), dim3(256), 0, ctx->stream, train, nt, dim, nt2, tmax);
            const int dimp = (dim + 7) & ~7;
            const size_t lds = (size_t)(KM_NQ * (dimp + 4) + 2 * KM_NT * KM_TP + 2 * KM_NT) * sizeof(float);
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(km_shortlist), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) { rc = MA_EHIP; break; }
            hipLaunchKernelGGL(km_shortlist, dim3(nqt, nsplit), dim3(KM_THREADS), lds, ctx->stream, query, train, nt2, nq, nt,
                               dim, tiles_per_split, cand_idx, cand_a4);
            hipLaunchKernelGGL(km_refine, dim3((nq + 256 / sc - 1) / (256 / sc)), dim3(256), 0, ctx->stream, query, train, nq,
                               dim, nsplit, sc, cand_idx, cand_a4, tmax, idx_out, dist_out, qlist, qcount, 0, (const unsigned*)amax);
            if (hipGetLastError() != hipSuccess) { rc = MA_EHIP; break; }
        }
        rc = knn2_exact(ctx, query, nq, train, nt, dim, idx_out, dist_out, qlist, qcount, fsplit, part);
        if (rc != MA_OK) break;
        if (uncertified_host) {
            if (hipMemcpyAsync(uncertified_host, qcount, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = MA_EHIP; break; }
        }
    } while (0);
    if (rc == MA_EHIP) ma_set_error("ma_knn2_l2: %s", hipGetErrorString(hipGetLastError()));
    ma_pool_free(ctx, ws);
    return rc;
}

extern "C" int ma_knn2_l2(ma_ctx* ctx, const float* query, int nq, const float* train, int nt, int dim, int* idx_out,
                          float* dist_out)
{
    return ma_knn2_l2_ex(ctx, query, nq, train, nt, dim, idx_out, dist_out, MA_KNN_AUTO, nullptr);
}
