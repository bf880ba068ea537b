// The mutual-information accept/reject gate: sklearn.metrics.normalized_mutual_info_score on u8
// label images, chunked as mi_tiled does (microaligner/shared_modules/similarity_scoring.py:27-50).
// Labels are pixel values, so the contingency matrix is a 256x256 joint histogram (exact integer
// counts); the score follows sklearn's formula in f64.
#include "ma_internal.h"

#include <cfloat>
#include <cstdlib>

namespace {

// Joint histogram with LDS privatisation, one pass over the pixels.  grid: (pixel slices, chunks).  A block holds the
// whole 256 x 256 joint histogram of its slice in LDS as 16-bit counters packed two to a word (128 KiB): a slice has
// at most 65 520 pixels, so no counter can reach 2^16 and a plain 32-bit LDS atomic add of 1 << 16 * (bin & 1) never
// carries into its neighbour.  Every pixel is read and decoded once (2 B/px of HBM traffic = the algorithmic bytes;
// the round-1 kernel kept 32-bit counters for a quarter of the labels and read every slice four times); then the
// non-zero counters are added to the chunk's histogram in HBM.  Lanes read 16 consecutive pixels each (one 16-byte
// load per array) so that the 64 lanes of an LDS-atomic instruction hit pixels 16 apart, which decorrelates the bins
// on smooth DOG images.  Measured alternatives (profiles/r02_notes.md): two label bands of 32-bit counters 0.167 ms
// per launch, this kernel 0.171, the four-band kernel 0.281.
constexpr int HIST_SLICE16 = 65520;
// blockIdx.z selects the second label image (b0 or b1: the "after" and the "before" half of the gate share `a` and one
// launch); its histograms follow those of b0.
// NB > 1: the labels of `a` are cut into NB bands and a block counts one band of its slice (blockIdx.z = image * NB + band):
// 128 / NB KiB of LDS instead of 128 -- the form for the coarse pyramid levels, whose gate runs while the companion stream's
// dog() blocks hold 141 of every CU's 160 KiB of LDS: a 128 KiB block is not placed before BOTH resident dog() blocks of a
// CU have gone, and other dog() blocks keep taking the half that frees up first (13 us alone, 1140 us beside the companion's
// full-resolution dog(): profiles/r05_notes.md -- the step time does not change, the wait is the companion's progress, but the
// gate no longer queues behind it).  The slice is read NB times (from L2: these levels are a few MB).
template <int NT, int NB>
__global__ __launch_bounds__(NT) void joint_hist16_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b0,
                                                          const uint8_t* __restrict__ b1, size_t n, size_t chunk,
                                                          unsigned* __restrict__ hist)
{
    extern __shared__ unsigned h[];              // [256 * 256 / 2 / NB]
    constexpr int WORDS = 32768 / NB, BAND = 256 / NB;
    const unsigned img = blockIdx.z / NB, band = blockIdx.z % NB;
    const uint8_t* __restrict__ b = img ? b1 : b0;
    const size_t c0 = (size_t)blockIdx.y * chunk;
    const size_t c1 = c0 + chunk < n ? c0 + chunk : n;
    size_t s0 = c0 + (size_t)blockIdx.x * HIST_SLICE16;
    const size_t s1 = s0 + HIST_SLICE16 < c1 ? s0 + HIST_SLICE16 : c1;
    if (s0 >= s1) return;
    for (int i = threadIdx.x; i < WORDS; i += NT) h[i] = 0;
    __syncthreads();
    auto count = [&](unsigned ai, unsigned bi) {
        if (NB > 1) {
            ai -= band * BAND;
            if (ai >= (unsigned)BAND) return;
        }
        const unsigned bin = ai * 256u + bi;
        atomicAdd(&h[bin >> 1], 1u << ((bin & 1u) * 16u));
    };
    size_t al = (s0 + 15) & ~(size_t)15;
    if (al > s1) al = s1;
    for (size_t i = s0 + threadIdx.x; i < al; i += NT) count(a[i], b[i]);
    const bool vec_ok = (((size_t)a | (size_t)b) & 15) == 0;
    const size_t nvec = vec_ok ? (s1 - al) / 16 : 0;
    const uint4* a4 = (const uint4*)(a + al);
    const uint4* b4 = (const uint4*)(b + al);
    for (size_t v = threadIdx.x; v < nvec; v += NT) {
        const uint4 av = a4[v], bv = b4[v];
        const unsigned aw[4] = {av.x, av.y, av.z, av.w}, bw[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int k = 0; k < 16; k++)
            count((aw[k >> 2] >> ((k & 3) * 8)) & 255u, (bw[k >> 2] >> ((k & 3) * 8)) & 255u);
    }
    for (size_t i = al + nvec * 16 + threadIdx.x; i < s1; i += NT) count(a[i], b[i]);
    __syncthreads();
    unsigned* hh = hist + ((size_t)img * gridDim.y + blockIdx.y) * 65536 + (size_t)band * (2 * WORDS);
    for (int i = threadIdx.x; i < WORDS; i += NT) {
        const unsigned c = h[i];
        if (c & 0xffffu) atomicAdd(&hh[2 * i], c & 0xffffu);
        if (c >> 16) atomicAdd(&hh[2 * i + 1], c >> 16);
    }
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// Four blocks of 256 threads per chunk histogram (blockIdx.y = q): thread j of block q owns column j of the rows [64q, 64q+64),
// so the 2 x 65536 double-precision logarithms of a chunk are spread over 16 waves -- as in the single block of 1024 threads of
// rounds 1 - 5, whose placement needed sixteen free wave slots on ONE CU at once and waited 380 us on average (107 alone) for
// the companion stream's dog() blocks to leave (profiles/r06_kernel_stats_cfg3_companion_on.csv).  Each wave leaves its partial
// sums in `part`; nmi_final_kernel adds them in the order the single block used (wave 0 .. 15), so the scores keep their bits.
constexpr int NR_T = 256, NR_Q = 4;
// per histogram: [0..15] mutual-information partials (wave 4q + w), [16..19] / [20..23] entropy partials of a / b (the waves of
// block 0), [24..27] / [28..31] number of non-empty labels of a among the block's rows / of b (block 0 only counts them)
constexpr int NR_PART = 32;
// block (b, q): histogram b, chunk b % nchunks (the histograms of a second label image follow those of the first)
__global__ __launch_bounds__(NR_T) void nmi_reduce_kernel(const unsigned* __restrict__ hist, size_t n, size_t chunk,
                                                          unsigned nchunks, double* __restrict__ part)
{
    __shared__ unsigned pa[64], pb[256];
    __shared__ int cnt[2];
    const unsigned* hh = hist + (size_t)blockIdx.x * 65536;
    const int j = threadIdx.x, q = blockIdx.y, lane = j & 63, w = j >> 6;
    const size_t c0 = (size_t)(blockIdx.x % nchunks) * chunk;
    const double N = (double)((c0 + chunk < n ? c0 + chunk : n) - c0);
    double* out = part + (size_t)blockIdx.x * NR_PART;

    if (j < 2) cnt[j] = 0;
    // marginals (counts < 2^32 by the chunk limit): column j over ALL rows (coalesced across the wave), and -- threads 0 .. 63 --
    // row 64 q + j (64 independent 16-byte loads)
    {
        unsigned sb = 0;
#pragma unroll 16
        for (int r = 0; r < 256; r++) sb += hh[r * 256 + j];
        pb[j] = sb;
        if (j < 64) {
            const uint4* row = reinterpret_cast<const uint4*>(hh + (64 * q + j) * 256);
            unsigned sa = 0;
#pragma unroll 16
            for (int k = 0; k < 64; k++) { uint4 v = row[k]; sa += v.x + v.y + v.z + v.w; }
            pa[j] = sa;
        }
    }
    __syncthreads();
    if (j < 64 && pa[j] > 0) atomicAdd(&cnt[0], 1);
    if (q == 0 && pb[j] > 0) atomicAdd(&cnt[1], 1);
    __syncthreads();
    const double logN = log(N);
    const unsigned long long pbj = pb[j];
    double mi = 0.0;
    if (pbj > 0) {
#pragma unroll 8
        for (int rr = 0; rr < 64; rr++) {
            unsigned nij = hh[(64 * q + rr) * 256 + j];
            if (nij) {
                double log_nm = log((double)nij);
                double nm = (double)nij / N;
                double outer = (double)((long long)pa[rr] * (long long)pbj);
                double log_outer = -log(outer) + logN + logN;
                double term = nm * (log_nm - logN) + nm * log_outer;
                if (fabs(term) < DBL_EPSILON) term = 0.0;
                mi += term;
            }
        }
    }
    mi = wave_sum(mi);
    if (lane == 0) out[4 * q + w] = mi;
    // entropies: label j of a is row j of the histogram -- block j / 64 holds its marginal (thread j % 64); label j of b: block 0
    double ha = 0.0, hb = 0.0;
    if (j < 64 && pa[j] > 0) ha = ((double)pa[j] / N) * (log((double)pa[j]) - logN);
    if (q == 0 && pbj > 0) hb = ((double)pbj / N) * (log((double)pbj) - logN);
    // the single block summed ha over its waves 0 .. 3 = labels 0 .. 255 in runs of 64: here run q is wave 0 of block q
    ha = wave_sum(ha);
    hb = wave_sum(hb);
    if (lane == 0) {
        if (w == 0) out[16 + q] = ha;
        if (q == 0) out[20 + w] = hb;
    }
    if (j == 0) {
        out[24 + q] = (double)cnt[0];
        if (q == 0) out[28] = (double)cnt[1];
    }
}

__global__ __launch_bounds__(64) void nmi_final_kernel(const double* __restrict__ part, unsigned nhist, double* __restrict__ scores)
{
    const unsigned b = blockIdx.x * 64 + threadIdx.x;
    if (b >= nhist) return;
    const double* p = part + (size_t)b * NR_PART;
    const int ca = (int)(p[24] + p[25] + p[26] + p[27]), cb = (int)p[28];
    if (ca == 1 && cb == 1) { scores[b] = 1.0; return; }      // both label sets have a single value
    double tot[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < 16; i++) tot[0] += p[i];
    for (int i = 0; i < 4; i++) { tot[1] += p[16 + i]; tot[2] += p[20 + i]; }
    double m = tot[0] < 0 ? 0.0 : tot[0];
    double score;
    if (fabs(m) < DBL_EPSILON) score = 0.0;
    else {
        double h_a = ca == 1 ? 0.0 : -tot[1], h_b = cb == 1 ? 0.0 : -tot[2];
        double norm = 0.5 * (h_a + h_b);
        if (norm < DBL_EPSILON) norm = DBL_EPSILON;
        score = m / norm;
    }
    scores[b] = score;
}

} // namespace

// NMI(a, b0) and, when b1 is given, NMI(a, b1), chunk by chunk, in one pair of launches
int ma_nmi_u8_enqueue2(ma_ctx* ctx, const uint8_t* a, const uint8_t* b0, const uint8_t* b1, size_t n, size_t chunk,
                       double* scores0_pinned_host, double* scores1_pinned_host, int max_scores, int* n_scores)
{
    MA_REQUIRE(ctx && a && b0 && scores0_pinned_host && n_scores && (!b1 || scores1_pinned_host), "NULL argument");
    MA_REQUIRE(n > 0, "empty arrays");
    if (chunk == 0 || chunk > n) chunk = n;
    const size_t nchunks = (n + chunk - 1) / chunk;
    const unsigned nimg = b1 ? 2 : 1;
    MA_REQUIRE(nchunks <= 65535 && (size_t)max_scores >= nchunks, "scores buffer too small");
    MA_REQUIRE(chunk < ((size_t)1 << 32), "chunk must be < 2^32 elements");
    MA_HIP(hipSetDevice(ctx->device));
    const size_t hist_bytes = nimg * nchunks * 65536 * sizeof(unsigned);
    const size_t total = hist_bytes + nimg * nchunks * (1 + NR_PART) * sizeof(double);
    MA_TRY(ma_ws_reserve(ctx, total));
    unsigned* hist = (unsigned*)ctx->ws;
    double* scores = (double*)((char*)ctx->ws + hist_bytes);
    double* part = scores + nimg * nchunks;
    {
        MaProfScope ps(ctx, MA_K_NMI, (double)n * nimg);
        MA_HIP(hipMemsetAsync(hist, 0, hist_bytes, ctx->stream));
        const size_t slices = (chunk + HIST_SLICE16 - 1) / HIST_SLICE16;
        MA_REQUIRE(slices <= 0x7fffffff, "chunk too large");
        // small inputs (the coarse pyramid levels: up to 2048^2 per image) in eight label bands of 16 KiB of LDS each
        constexpr int NB = 8;
        static const size_t band_max = [] { const char* e = getenv("MICROALIGNER_NMI_BAND_MAX_PX"); return e ? (size_t)atoll(e) : (size_t)1 << 22; }();
        if (n <= band_max && (size_t)nimg * NB <= 65535)
            hipLaunchKernelGGL((joint_hist16_kernel<1024, NB>), dim3((unsigned)slices, (unsigned)nchunks, nimg * NB), dim3(1024),
                               32768 / NB * sizeof(unsigned), ctx->stream, a, b0, b1, n, chunk, hist);
        else
            hipLaunchKernelGGL((joint_hist16_kernel<1024, 1>), dim3((unsigned)slices, (unsigned)nchunks, nimg), dim3(1024),
                               32768 * sizeof(unsigned), ctx->stream, a, b0, b1, n, chunk, hist);
        hipLaunchKernelGGL(nmi_reduce_kernel, dim3((unsigned)(nimg * nchunks), NR_Q), dim3(NR_T), 0, ctx->stream, hist, n, chunk,
                           (unsigned)nchunks, part);
        hipLaunchKernelGGL(nmi_final_kernel, dim3((unsigned)((nimg * nchunks + 63) / 64)), dim3(64), 0, ctx->stream,
                           (const double*)part, (unsigned)(nimg * nchunks), scores);
        MA_HIP(hipGetLastError());
    }
    MA_HIP(hipMemcpyAsync(scores0_pinned_host, scores, nchunks * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (b1)
        MA_HIP(hipMemcpyAsync(scores1_pinned_host, scores + nchunks, nchunks * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    *n_scores = (int)nchunks;
    return MA_OK;
}

int ma_nmi_u8_enqueue(ma_ctx* ctx, const uint8_t* a, const uint8_t* b, size_t n, size_t chunk, double* scores_pinned_host,
                      int max_scores, int* n_scores)
{
    return ma_nmi_u8_enqueue2(ctx, a, b, nullptr, n, chunk, scores_pinned_host, nullptr, max_scores, n_scores);
}

extern "C" {

int ma_nmi_u8(ma_ctx* ctx, const uint8_t* a, const uint8_t* b, size_t n, size_t chunk, double* scores_host,
              int max_scores, int* n_scores)
{
    MA_REQUIRE(ctx && scores_host && n_scores, "NULL argument");
    const size_t nchunks = (chunk == 0 || chunk >= n || n == 0) ? 1 : (n + chunk - 1) / chunk;
    MA_REQUIRE(nchunks <= 65535 && (size_t)max_scores >= nchunks, "scores buffer too small");
    MA_TRY(ma_pinned_reserve(ctx, nchunks * sizeof(double)));
    MA_TRY(ma_nmi_u8_enqueue(ctx, a, b, n, chunk, (double*)ctx->pinned, max_scores, n_scores));
    MA_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->profile) MA_TRY(ma_profile_flush(ctx));   // the stream is idle: recycle the accounting events
    for (int i = 0; i < *n_scores; i++) scores_host[i] = ((double*)ctx->pinned)[i];
    return MA_OK;
}

int ma_nmi_u8_pair(ma_ctx* ctx, const uint8_t* a, const uint8_t* b0, const uint8_t* b1, size_t n, size_t chunk,
                   double* scores0_host, double* scores1_host, int max_scores, int* n_scores)
{
    MA_REQUIRE(ctx && b1 && scores0_host && scores1_host && n_scores, "NULL argument");
    const size_t nchunks = (chunk == 0 || chunk >= n || n == 0) ? 1 : (n + chunk - 1) / chunk;
    MA_REQUIRE(nchunks <= 65535 && (size_t)max_scores >= nchunks, "scores buffer too small");
    MA_TRY(ma_pinned_reserve(ctx, 2 * nchunks * sizeof(double)));
    double* p0 = (double*)ctx->pinned;
    double* p1 = p0 + nchunks;
    MA_TRY(ma_nmi_u8_enqueue2(ctx, a, b0, b1, n, chunk, p0, p1, max_scores, n_scores));
    MA_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->profile) MA_TRY(ma_profile_flush(ctx));
    for (int i = 0; i < *n_scores; i++) { scores0_host[i] = p0[i]; scores1_host[i] = p1[i]; }
    return MA_OK;
}

} // extern "C"
