// Ratio test and RANSAC similarity fit of FeatureRegistrator's matching step on the device
// (reference: microaligner/feature_reg/feature_detection.py:142-158 -- Lowe's ratio 0.5 over the 2-NN pairs, then
// cv.estimateAffinePartial2D(dst_pts, src_pts, method=RANSAC, confidence=0.99)).  opencv-contrib is not available to this
// build; the DEFINITION these kernels reproduce bit for bit is the host statement
// microaligner_amd/feature_reg/sparse_cpu.py:estimate_affine_partial_2d (PARITY with OpenCV's estimator UNPINNED, see there):
//   * samples: numpy's Generator(PCG64(seed)).choice(n, 2, replace=False), in sequence -- generated HERE on the host by a
//     restatement of numpy's PCG64 / Lemire bounded integers / Floyd sampling + two-element shuffle
//     (tests/test_feature_reg.py compares ma_host_pcg64_choice2 with numpy for populations from 2 to 2^31);
//   * a sample's model: the similarity through its two point pairs in closed form, float64, one rounding per operation
//     (the file is built with -ffp-contract=off);
//   * inliers: err = ex ex + ey ey < thr^2 with ex = ((a x - b y) + tx) - u, ey = ((b x + a y) + ty) - v;
//   * the adaptive iteration count uses the C library's log() on the host (sparse_cpu.ransac_iterations uses math.log: the
//     same function), so the scan over the samples' inlier counts runs on the host between two launches;
//   * the final model: closed-form least squares on the inliers from CENTRED INTEGER sums (keypoints are integer pixel
//     positions: every partial sum is an integer below 2^53 and exact in float64 whatever the order of the reduction),
//     re-selected and refitted until the inlier set is stable, at most 10 times.
// Coordinates that are not integer-valued (never the case for FAST keypoints) are reported with status 3 and left to the
// host statement: the sums would no longer be exact and the bits could differ.
#include "ma_internal.h"

#include <cmath>

// ---- numpy's random stream, restated (host) -----------------------------------------------------------------------------
namespace {

typedef unsigned __int128 u128;
struct Pcg64 {
    u128 state, inc;
    bool has32 = false;
    uint32_t u32 = 0;
    static uint64_t rotr(uint64_t v, unsigned r) { return (v >> r) | (v << ((64 - r) & 63)); }
    uint64_t next64()
    {
        // PCG XSL-RR 128/64 with the default 128-bit multiplier (numpy/random/src/pcg64/pcg64.h)
        const u128 mult = ((u128)2549297995355413924ULL << 64) | 4865540595714422341ULL;
        state = state * mult + inc;
        const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
        return rotr(hi ^ lo, (unsigned)(hi >> 58));
    }
    uint32_t next32()
    {
        // a 64-bit draw serves two 32-bit requests: low half first, high half kept for the next one
        if (has32) { has32 = false; return u32; }
        const uint64_t n = next64();
        has32 = true;
        u32 = (uint32_t)(n >> 32);
        return (uint32_t)n;
    }
    uint32_t bounded(uint32_t rng)
    {
        // uniform integer in [0, rng]: Lemire's multiply-and-reject on 32-bit draws (rng < 2^32 - 1); rng == 0 draws nothing
        if (rng == 0) return 0;
        const uint32_t excl = rng + 1;
        uint64_t m = (uint64_t)next32() * excl;
        uint32_t left = (uint32_t)m;
        if (left < excl) {
            const uint32_t thr = (0xffffffffu - rng) % excl;
            while (left < thr) { m = (uint64_t)next32() * excl; left = (uint32_t)m; }
        }
        return (uint32_t)(m >> 32);
    }
    void choice2(int n, int& a, int& b)
    {
        // Generator.choice(n, 2, replace=False): Floyd's algorithm for j = n - 2, n - 1, then a shuffle of the two
        uint32_t x = bounded((uint32_t)(n - 2));
        uint32_t y = bounded((uint32_t)(n - 1));
        if (y == x) y = (uint32_t)(n - 1);
        if (bounded(1) == 0) { const uint32_t t = x; x = y; y = t; }
        a = (int)x; b = (int)y;
    }
};

constexpr int RS_T = 256;           // threads of a scoring block
constexpr int RS_F = 1024;          // threads of the single-block kernels
constexpr int RS_FIRST = 256;       // samples scored before the host looks at the counts for the first time

struct RsModel { double a, b, tx, ty; };

__device__ __forceinline__ bool rs_inlier(const RsModel& m, double x, double y, double u, double v, double thr2)
{
    const double ex = ((m.a * x - m.b * y) + m.tx) - u;
    const double ey = ((m.b * x + m.a * y) + m.ty) - v;
    return ex * ex + ey * ey < thr2;
}

// the similarity through the point pairs i and j; false: degenerate sample (the loop `continue`s)
__device__ __forceinline__ bool rs_two_point_model(const double* __restrict__ sx, const double* __restrict__ sy,
                                                   const double* __restrict__ dx, const double* __restrict__ dy, int i, int j,
                                                   RsModel& m)
{
    const double xi = sx[i], yi = sy[i], xj = sx[j], yj = sy[j];
    if (fabs(xi - xj) <= 1e-8 + 1e-5 * fabs(xj) && fabs(yi - yj) <= 1e-8 + 1e-5 * fabs(yj)) return false;
    const double ux = xj - xi, uy = yj - yi, vx = dx[j] - dx[i], vy = dy[j] - dy[i];
    const double den = ux * ux + uy * uy;
    if (den == 0.0) return false;
    m.a = (vx * ux + vy * uy) / den;
    m.b = (vy * ux - vx * uy) / den;
    m.tx = dx[i] - (m.a * xi - m.b * yi);
    m.ty = dy[i] - (m.b * xi + m.a * yi);
    return true;
}

template <int T>
__device__ __forceinline__ int rs_block_sum_int(int v, int* wsum)
{
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int k = 0; k < T / 64; k++) t += wsum[k];
    __syncthreads();
    return t;
}

// exact for integer-valued terms below 2^53 (any order); all threads get the total
template <int T>
__device__ __forceinline__ double rs_block_sum_f64(double v, double* wsum)
{
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < T / 64; k++) t += wsum[k];
    __syncthreads();
    return t;
}

// Ratio test + ordered compaction, one block: good[k] = k-th query q (ascending) with sqrt(d0) < ratio * sqrt(d1) in float32
// (np.sqrt of the float32 squared distances, the product with the float32 ratio, feature_detection.py:143-146); its point
// pair goes to (sx, sy) = the query's keypoint and (dx, dy) = the keypoint of its nearest train descriptor, both through
// float32 as the reference's np.float32 point arrays do.  info[0] = n_good, info[1] = 1 if a coordinate is not an integer
// below 2^24, info[2] = the largest |coordinate| (rounded up).
__global__ __launch_bounds__(RS_F) void rs_ratio_compact(const int* __restrict__ idx, const float* __restrict__ d2, int nq,
                                                         const double* __restrict__ qpts, const double* __restrict__ tpts,
                                                         int nt, float ratio, double* __restrict__ sx, double* __restrict__ sy,
                                                         double* __restrict__ dx, double* __restrict__ dy, int* __restrict__ info)
{
    __shared__ int wsum[RS_F / 64];
    __shared__ int s_bad, s_max;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { s_bad = 0; s_max = 0; }
    __syncthreads();
    int base = 0;
    for (int q0 = 0; q0 < nq; q0 += RS_F) {
        const int q = q0 + tid;
        bool good = false;
        int j = 0;
        if (q < nq) {
            const float r0 = __fsqrt_rn(d2[2 * q]), r1 = __fsqrt_rn(d2[2 * q + 1]);
            good = r0 < ratio * r1;
            j = idx[2 * q];
            if (good && (unsigned)j >= (unsigned)nt) good = false;     // cannot happen with indices from the search
        }
        int inc = good ? 1 : 0;
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_up(inc, o);
            if (lane >= o) inc += n;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < RS_F / 64; k++) {
            const int s = wsum[k];
            if (k < wv) before += s;
            total += s;
        }
        if (good) {
            const int k = base + before + inc - 1;
            const double x = (double)(float)qpts[2 * q], y = (double)(float)qpts[2 * q + 1];
            const double u = (double)(float)tpts[2 * j], v = (double)(float)tpts[2 * j + 1];
            sx[k] = x; sy[k] = y; dx[k] = u; dy[k] = v;
            const bool ok = x == floor(x) && y == floor(y) && u == floor(u) && v == floor(v) && fabs(x) < 16777216.0 &&
                            fabs(y) < 16777216.0 && fabs(u) < 16777216.0 && fabs(v) < 16777216.0;
            if (!ok) s_bad = 1;
            else atomicMax(&s_max, (int)fmax(fmax(fabs(x), fabs(y)), fmax(fabs(u), fabs(v))));
        }
        base += total;
        __syncthreads();
    }
    if (tid == 0) { info[0] = base; info[1] = s_bad; info[2] = s_max; }
}

// The same selection with a block per 1024 queries, in two launches (the single block above walks the queries 1024 at a time,
// every step behind the loads of the one before: 82 us for the top level's 22 800 queries): rs_ratio_count leaves every block's
// number of good matches, rs_ratio_write places the blocks one behind the other -- the order of the queries, as above.
__device__ __forceinline__ bool rs_ratio_good(const int* __restrict__ idx, const float* __restrict__ d2, int q, int nq, int nt,
                                              float ratio, int& j)
{
    j = 0;
    if (q >= nq) return false;
    const float r0 = __fsqrt_rn(d2[2 * q]), r1 = __fsqrt_rn(d2[2 * q + 1]);
    j = idx[2 * q];
    return r0 < ratio * r1 && (unsigned)j < (unsigned)nt;
}

__global__ __launch_bounds__(RS_F) void rs_ratio_count(const int* __restrict__ idx, const float* __restrict__ d2, int nq, int nt,
                                                       float ratio, int* __restrict__ bcount, int* __restrict__ info)
{
    __shared__ int wsum[RS_F / 64];
    int j;
    const bool good = rs_ratio_good(idx, d2, blockIdx.x * RS_F + threadIdx.x, nq, nt, ratio, j);
    const int total = rs_block_sum_int<RS_F>(good ? 1 : 0, wsum);
    if (threadIdx.x == 0) {
        bcount[blockIdx.x] = total;
        if (blockIdx.x == 0) { info[1] = 0; info[2] = 0; }
    }
}

__global__ __launch_bounds__(RS_F) void rs_ratio_write(const int* __restrict__ idx, const float* __restrict__ d2, int nq,
                                                       const double* __restrict__ qpts, const double* __restrict__ tpts, int nt,
                                                       float ratio, const int* __restrict__ bcount, double* __restrict__ sx,
                                                       double* __restrict__ sy, double* __restrict__ dx, double* __restrict__ dy,
                                                       int* __restrict__ info)
{
    __shared__ int wsum[RS_F / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, q = blockIdx.x * RS_F + tid;
    int before_blocks = 0;
    for (int b = tid; b < (int)blockIdx.x; b += RS_F) before_blocks += bcount[b];
    const int base = rs_block_sum_int<RS_F>(before_blocks, wsum);
    int j;
    const bool good = rs_ratio_good(idx, d2, q, nq, nt, ratio, j);
    int inc = good ? 1 : 0;
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(inc, o);
        if (lane >= o) inc += n;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < RS_F / 64; k++) {
        const int s = wsum[k];
        if (k < wv) before += s;
        total += s;
    }
    if (good) {
        const int k = base + before + inc - 1;
        const double x = (double)(float)qpts[2 * q], y = (double)(float)qpts[2 * q + 1];
        const double u = (double)(float)tpts[2 * j], v = (double)(float)tpts[2 * j + 1];
        sx[k] = x; sy[k] = y; dx[k] = u; dy[k] = v;
        const bool ok = x == floor(x) && y == floor(y) && u == floor(u) && v == floor(v) && fabs(x) < 16777216.0 &&
                        fabs(y) < 16777216.0 && fabs(u) < 16777216.0 && fabs(v) < 16777216.0;
        if (!ok) atomicOr(&info[1], 1);
        else atomicMax(&info[2], (int)fmax(fmax(fabs(x), fabs(y)), fmax(fabs(u), fabs(v))));
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) info[0] = base + total;
}

// inlier count of one two-point sample per block; -1: degenerate sample
__global__ __launch_bounds__(RS_T) void rs_score_samples(const double* __restrict__ sx, const double* __restrict__ sy,
                                                         const double* __restrict__ dx, const double* __restrict__ dy, int n,
                                                         const int* __restrict__ pairs, int first, double thr2,
                                                         int* __restrict__ counts)
{
    __shared__ int wsum[RS_T / 64];
    const int s = first + blockIdx.x;
    RsModel m;
    if (!rs_two_point_model(sx, sy, dx, dy, pairs[2 * s], pairs[2 * s + 1], m)) {     // block-uniform
        if (threadIdx.x == 0) counts[s] = -1;
        return;
    }
    int c = 0;
    for (int p = threadIdx.x; p < n; p += RS_T) c += rs_inlier(m, sx[p], sy[p], dx[p], dy[p], thr2) ? 1 : 0;
    c = rs_block_sum_int<RS_T>(c, wsum);
    if (threadIdx.x == 0) counts[s] = c;
}

// N sums at once (two barriers instead of two per sum); exact for integer-valued terms below 2^53, as above
template <int T, int N>
__device__ __forceinline__ void rs_block_sum_f64n(double (&v)[N], double* wsum /* [N][T / 64] */)
{
#pragma unroll
    for (int k = 0; k < N; k++)
        for (int o = 32; o; o >>= 1) v[k] += __shfl_xor(v[k], o);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < N; k++) wsum[k * (T / 64) + (threadIdx.x >> 6)] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < T / 64; w++) t += wsum[k * (T / 64) + w];
        v[k] = t;
    }
    __syncthreads();
}

// closed-form least squares over the points with mask[p] != 0 (sparse_cpu._fit_similarity); false: rank deficient
__device__ bool rs_fit(const double* __restrict__ sx, const double* __restrict__ sy, const double* __restrict__ dx,
                       const double* __restrict__ dy, const unsigned char* __restrict__ mask, int n, int count, RsModel& m,
                       double* wsum /* [7][RS_F / 64] */)
{
    if (count < 2) return false;
    const double fn = (double)count;
    double c4[4] = {0, 0, 0, 0};
    for (int p = threadIdx.x; p < n; p += RS_F)
        if (mask[p]) { c4[0] += sx[p]; c4[1] += sy[p]; c4[2] += dx[p]; c4[3] += dy[p]; }
    rs_block_sum_f64n<RS_F, 4>(c4, wsum);
    const double cx = floor(c4[0] / fn), cy = floor(c4[1] / fn), cu = floor(c4[2] / fn), cv = floor(c4[3] / fn);
    double q[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int p = threadIdx.x; p < n; p += RS_F)
        if (mask[p]) {
            const double x = sx[p] - cx, y = sy[p] - cy, u = dx[p] - cu, v = dy[p] - cv;
            q[0] += x; q[1] += y; q[2] += u; q[3] += v;
            q[4] += x * x + y * y;
            q[5] += x * u + y * v;
            q[6] += x * v - y * u;
        }
    rs_block_sum_f64n<RS_F, 7>(q, wsum);
    const double Sx = q[0], Sy = q[1], Su = q[2], Sv = q[3], Sxx = q[4], Sxu = q[5], Sxv = q[6];
    const double D = fn * Sxx - (Sx * Sx + Sy * Sy);
    if (!(D > 0.0)) return false;
    m.a = (fn * Sxu - (Sx * Su + Sy * Sv)) / D;
    m.b = (fn * Sxv - (Sx * Sv - Sy * Su)) / D;
    double tx = (Su - (m.a * Sx - m.b * Sy)) / fn;
    double ty = (Sv - (m.b * Sx + m.a * Sy)) / fn;
    m.tx = (tx + cu) - (m.a * cx - m.b * cy);
    m.ty = (ty + cv) - (m.b * cx + m.a * cy);
    return true;
}

// the tail of estimate_affine_partial_2d for the winning sample, one block: its inlier mask, the least-squares model of the
// inliers, re-selection and refit until the set is stable (at most 10 times).  out[0..5] = the 2 x 3 matrix, out[6] = 1 if
// the first fit failed (the function returns None), out[7] = inliers of the final mask.
__global__ __launch_bounds__(RS_F) void rs_refine(const double* __restrict__ sx, const double* __restrict__ sy,
                                                  const double* __restrict__ dx, const double* __restrict__ dy, int n,
                                                  const int* __restrict__ pairs, int best, double thr2,
                                                  unsigned char* __restrict__ mask_a, unsigned char* __restrict__ mask_b,
                                                  double* __restrict__ out)
{
    __shared__ double wsum[7 * (RS_F / 64)];
    __shared__ int isum[RS_F / 64];
    RsModel m;
    rs_two_point_model(sx, sy, dx, dy, pairs[2 * best], pairs[2 * best + 1], m);   // not degenerate: it was counted
    unsigned char* best_mask = mask_a;
    unsigned char* mask = mask_b;
    int c = 0;
    for (int p = threadIdx.x; p < n; p += RS_F) {
        const bool in = rs_inlier(m, sx[p], sy[p], dx[p], dy[p], thr2);
        best_mask[p] = in ? 1 : 0;
        c += in ? 1 : 0;
    }
    int best_count = rs_block_sum_int<RS_F>(c, isum);
    __syncthreads();
    RsModel M;
    const bool ok = rs_fit(sx, sy, dx, dy, best_mask, n, best_count, M, wsum);
    if (!ok) {
        if (threadIdx.x == 0) { out[6] = 1.0; out[7] = (double)best_count; }
        return;
    }
    for (int round = 0; round < 10; round++) {
        int cnt = 0, diff = 0;
        for (int p = threadIdx.x; p < n; p += RS_F) {
            const unsigned char in = rs_inlier(M, sx[p], sy[p], dx[p], dy[p], thr2) ? 1 : 0;
            mask[p] = in;
            cnt += in;
            diff += in != best_mask[p] ? 1 : 0;
        }
        {   // both counts in one reduction (n < 2^26: a count fits 26 bits; the sum of the packed words does not overflow 63)
            long long both = ((long long)cnt << 32) | (long long)diff;
            for (int o = 32; o; o >>= 1) both += __shfl_xor(both, o);
            __shared__ long long lsum[RS_F / 64];
            if ((threadIdx.x & 63) == 0) lsum[threadIdx.x >> 6] = both;
            __syncthreads();
            long long t = 0;
            for (int w = 0; w < RS_F / 64; w++) t += lsum[w];
            __syncthreads();
            cnt = (int)(t >> 32); diff = (int)(t & 0xffffffffll);
        }
        if (cnt < 2 || diff == 0) break;
        unsigned char* t = best_mask; best_mask = mask; mask = t;
        best_count = cnt;
        __syncthreads();
        RsModel M2;
        if (!rs_fit(sx, sy, dx, dy, best_mask, n, best_count, M2, wsum)) break;
        M = M2;
    }
    if (threadIdx.x == 0) {
        out[0] = M.a; out[1] = -M.b; out[2] = M.tx;
        out[3] = M.b; out[4] = M.a; out[5] = M.ty;
        out[6] = 0.0; out[7] = (double)best_count;
    }
}

// sparse_cpu.ransac_iterations with the C library's log (math.log in the host statement)
int rs_iterations(int count, int n, double confidence, int max_iters, int it)
{
    const double w = (double)count / (double)n;
    const double one = 1.0 - w * w;
    const double denom = std::log(one > 1e-12 ? one : 1e-12);
    if (!(denom < 0)) return it;
    const double need = std::ceil(std::log(1.0 - confidence) / denom);
    return need < (double)max_iters ? (int)need : max_iters;
}

}  // namespace

extern "C" {

int ma_host_pcg64_choice2(const unsigned long long state[4], int n, int count, int* pairs_out)
{
    static const unsigned long long seed0[4] = {0x1aa1b5345996452dULL, 0x09585eb7a69561e3ULL, 0x418ddadb3af71a82ULL,
                                                0x588133bc447873a9ULL};   // numpy.random.PCG64(0)
    if (!state) state = seed0;
    MA_REQUIRE(pairs_out && n >= 2 && count >= 0, "bad arguments");
    Pcg64 g;
    g.state = ((u128)state[0] << 64) | state[1];
    g.inc = ((u128)state[2] << 64) | state[3];
    for (int c = 0; c < count; c++) g.choice2(n, pairs_out[2 * c], pairs_out[2 * c + 1]);
    return MA_OK;
}

int ma_host_ransac_iterations(int count, int n, double confidence, int max_iters, int it, int* iters)
{
    MA_REQUIRE(iters && n >= 1 && count >= 0 && count <= n && max_iters >= 0, "bad arguments");
    *iters = rs_iterations(count, n, confidence, max_iters, it);
    return MA_OK;
}

int ma_match_similarity(ma_ctx* ctx, const int* idx, const float* dist_sq, int nq, const double* query_pts,
                        const double* train_pts, int nt, float ratio, double confidence, double reproj_threshold,
                        int max_iters, const unsigned long long rng_state[4], double* m2x3_host, int* n_good_host,
                        int* status_host)
{
    MA_REQUIRE(ctx && idx && dist_sq && query_pts && train_pts && m2x3_host && n_good_host && status_host, "NULL argument");
    // rng_state == NULL: the state of numpy.random.PCG64(0) -- the seed the host statement defaults to -- so that a C host needs
    // no numpy to make the call (tests/test_feature_reg.py compares the constants with numpy's)
    static const unsigned long long seed0[4] = {0x1aa1b5345996452dULL, 0x09585eb7a69561e3ULL, 0x418ddadb3af71a82ULL,
                                                0x588133bc447873a9ULL};
    if (!rng_state) rng_state = seed0;
    MA_REQUIRE(nq >= 1 && nt >= 1 && max_iters >= 1 && max_iters <= 1000000, "bad sizes");
    MA_REQUIRE(confidence > 0.0 && confidence < 1.0 && reproj_threshold > 0.0, "bad RANSAC parameters");
    MA_HIP(hipSetDevice(ctx->device));
    const double thr2 = reproj_threshold * reproj_threshold;
    const size_t b_pts = ma_align_up((size_t)nq * sizeof(double), 256), b_pairs = ma_align_up((size_t)max_iters * 8, 256),
                 b_cnt = ma_align_up((size_t)max_iters * 4, 256), b_mask = ma_align_up((size_t)nq, 256);
    char* ws = static_cast<char*>(ma_pool_alloc(ctx, 4 * b_pts + b_pairs + b_cnt + 2 * b_mask + 512));
    if (!ws) return MA_ENOMEM;
    double* sx = reinterpret_cast<double*>(ws);
    double* sy = reinterpret_cast<double*>(ws + b_pts);
    double* dx = reinterpret_cast<double*>(ws + 2 * b_pts);
    double* dy = reinterpret_cast<double*>(ws + 3 * b_pts);
    int* pairs = reinterpret_cast<int*>(ws + 4 * b_pts);
    int* counts = reinterpret_cast<int*>(ws + 4 * b_pts + b_pairs);
    unsigned char* mask_a = reinterpret_cast<unsigned char*>(ws + 4 * b_pts + b_pairs + b_cnt);
    unsigned char* mask_b = mask_a + b_mask;
    double* out = reinterpret_cast<double*>(ws + 4 * b_pts + b_pairs + b_cnt + 2 * b_mask);
    int* info = reinterpret_cast<int*>(out + 8);     // 3 ints
    // page-locked scratch: [0, 64) results, then the sample pairs, then the counts
    const size_t pin_bytes = 64 + (size_t)max_iters * 8 + (size_t)max_iters * 4;
    int rc = ma_pinned_reserve(ctx, pin_bytes);
    if (rc != MA_OK) { ma_pool_free(ctx, ws); return rc; }
    char* pin = static_cast<char*>(ctx->pinned);
    int* h_info = reinterpret_cast<int*>(pin);
    double* h_out = reinterpret_cast<double*>(pin);
    int* h_pairs = reinterpret_cast<int*>(pin + 64);
    int* h_counts = reinterpret_cast<int*>(pin + 64 + (size_t)max_iters * 8);
    auto fail = [&](hipError_t e, const char* what) {
        ma_set_error("ma_match_similarity: %s failed: %s", what, hipGetErrorString(e));
        ma_pool_free(ctx, ws);
        return MA_EHIP;
    };
    hipError_t e;
    *status_host = 0;
    *n_good_host = 0;
    {
        MaProfScope ps(ctx, MA_K_OTHER, (double)nq);
        const int nb = (nq + RS_F - 1) / RS_F;
        if (nb > 1 && nb <= max_iters) {       // (the blocks' counts borrow the samples' count array)
            hipLaunchKernelGGL(rs_ratio_count, dim3(nb), dim3(RS_F), 0, ctx->stream, idx, dist_sq, nq, nt, ratio, counts, info);
            hipLaunchKernelGGL(rs_ratio_write, dim3(nb), dim3(RS_F), 0, ctx->stream, idx, dist_sq, nq, query_pts, train_pts, nt,
                               ratio, (const int*)counts, sx, sy, dx, dy, info);
        } else {
            hipLaunchKernelGGL(rs_ratio_compact, dim3(1), dim3(RS_F), 0, ctx->stream, idx, dist_sq, nq, query_pts, train_pts, nt,
                               ratio, sx, sy, dx, dy, info);
        }
    }
    if ((e = hipMemcpyAsync(h_info, info, 12, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) return fail(e, "copy");
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(e, "ratio test");
    const int n = h_info[0];
    // exact sums need integer terms below 2^53: a centred coordinate is at most 2 max in magnitude, a term of the sums at
    // most 8 max^2, a sum at most n times that
    const bool non_integer = h_info[1] != 0 || (double)h_info[0] * 8.0 * (double)h_info[2] * (double)h_info[2] >= 9007199254740992.0;
    *n_good_host = n;
    if (n < 3) { *status_host = 1; ma_pool_free(ctx, ws); return MA_OK; }       // feature_detection.py:147-149
    if (non_integer) { *status_host = 3; ma_pool_free(ctx, ws); return MA_OK; }

    // the whole sample sequence (cheap: three 32-bit draws per sample), scored in two instalments: most registrations stop
    // within the first few dozen samples
    Pcg64 g;
    g.state = ((u128)rng_state[0] << 64) | rng_state[1];
    g.inc = ((u128)rng_state[2] << 64) | rng_state[3];
    for (int s = 0; s < max_iters; s++) g.choice2(n, h_pairs[2 * s], h_pairs[2 * s + 1]);
    if ((e = hipMemcpyAsync(pairs, h_pairs, (size_t)max_iters * 8, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        return fail(e, "copy");
    int best = -1, best_count = 0, iters = max_iters, it = 0, scored = 0;
    while (it < iters) {
        if (it >= scored) {
            const int upto = scored == 0 ? std::min(RS_FIRST, max_iters) : max_iters;
            {
                MaProfScope ps(ctx, MA_K_OTHER, (double)n * (upto - scored));
                hipLaunchKernelGGL(rs_score_samples, dim3(upto - scored), dim3(RS_T), 0, ctx->stream, (const double*)sx,
                                   (const double*)sy, (const double*)dx, (const double*)dy, n, (const int*)pairs, scored, thr2,
                                   counts);
            }
            if ((e = hipMemcpyAsync(h_counts + scored, counts + scored, (size_t)(upto - scored) * 4, hipMemcpyDeviceToHost,
                                    ctx->stream)) != hipSuccess) return fail(e, "copy");
            if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(e, "sample scoring");
            scored = upto;
        }
        const int c = h_counts[it];
        it++;
        if (c < 0) continue;                       // degenerate sample
        if (c > best_count) {
            best_count = c;
            best = it - 1;
            iters = rs_iterations(c, n, confidence, max_iters, it);
        }
    }
    if (best < 0 || best_count < 2) { *status_host = 2; ma_pool_free(ctx, ws); return MA_OK; }
    {
        MaProfScope ps(ctx, MA_K_OTHER, (double)n);
        hipLaunchKernelGGL(rs_refine, dim3(1), dim3(RS_F), 0, ctx->stream, (const double*)sx, (const double*)sy,
                           (const double*)dx, (const double*)dy, n, (const int*)pairs, best, thr2, mask_a, mask_b, out);
    }
    if ((e = hipMemcpyAsync(h_out, out, 64, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) return fail(e, "copy");
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(e, "refinement");
    if (h_out[6] != 0.0) *status_host = 2;
    else for (int k = 0; k < 6; k++) m2x3_host[k] = h_out[k];
    ma_pool_free(ctx, ws);
    return MA_OK;
}

}  // extern "C"
