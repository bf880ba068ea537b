// Dense halves of the feature stage of FeatureRegistrator, batched over the feature tiles of one
// pyramid level: the FAST-9/16 corner score with 3x3 non-maximum suppression and the DAISY descriptor
// (reference: cv.FastFeatureDetector_create(threshold=1, nonmaxSuppression=True, TYPE_9_16) and
// cv.xfeatures2d.DAISY_create(radius=21, q_radius=3, q_theta=8, q_hist=8, NRM_NONE, interpolation=True),
// microaligner/feature_reg/feature_detection.py:88-120).  opencv-contrib is not available to this build: the kernels
// restate microaligner_amd/feature_reg/sparse_cpu.py (numpy / scipy) operation by operation -- same float32 / float64
// placement, same order -- so that device and host features are interchangeable; PARITY with opencv-contrib stays
// UNPINNED exactly as for the host code (sparse_cpu.py header).  Keypoint selection (sort by response, per-tile
// limit) stays on the host: a few thousand points per tile.
#include "ma_internal.h"

#include <cmath>

namespace {

// Bresenham circle of radius 3 in OpenCV's order (dx, dy)
__constant__ int c_ring[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3}, {0, -3}, {-1, -3},
                                  {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// score of every pixel of the tile interior (margin cut off on all sides): max over the 16 arcs of 9 contiguous ring
// pixels of min(v - ring) (darker arc) or min(ring - v) (brighter arc), minus 1; 0 where that maximum is <= threshold
// or within 3 px of the interior's border
__global__ __launch_bounds__(256) void fast_score_kernel(const uint8_t* __restrict__ tiles, int P, int margin, int threshold,
                                                         int* __restrict__ score)
{
    const int Pi = P - 2 * margin;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, t = blockIdx.z;
    if (x >= Pi) return;
    int s = 0;
    if (x >= 3 && x < Pi - 3 && y >= 3 && y < Pi - 3) {
        const uint8_t* img = tiles + (size_t)t * P * P + (size_t)(y + margin) * P + (x + margin);
        const int v = img[0];
        int d[16];
#pragma unroll
        for (int k = 0; k < 16; k++) d[k] = v - (int)img[c_ring[k][1] * P + c_ring[k][0]];
        int best = -512;
#pragma unroll
        for (int a = 0; a < 16; a++) {
            int lo = d[a], hi = d[a];
#pragma unroll
            for (int j = 1; j < 9; j++) { lo = min(lo, d[(a + j) & 15]); hi = max(hi, d[(a + j) & 15]); }
            best = max(best, max(lo, -hi));
        }
        s = best > threshold ? best - 1 : 0;
    }
    score[((size_t)t * Pi + y) * Pi + x] = s;
}

// keeps a score only where it is strictly greater than its 8 neighbours (outside the interior counts as 0)
__global__ __launch_bounds__(256) void fast_nms_kernel(const int* __restrict__ score, int Pi, int* __restrict__ out)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, t = blockIdx.z;
    if (x >= Pi) return;
    const int* s = score + (size_t)t * Pi * Pi;
    const int c = s[(size_t)y * Pi + x];
    bool ok = c > 0;
    for (int dy = -1; dy <= 1 && ok; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            if (!dx && !dy) continue;
            const int yy = y + dy, xx = x + dx;
            const int nb = (yy >= 0 && yy < Pi && xx >= 0 && xx < Pi) ? s[(size_t)yy * Pi + xx] : 0;
            if (!(c > nb)) { ok = false; break; }
        }
    out[((size_t)t * Pi + y) * Pi + x] = ok ? c : 0;
}

// Daisy._cubes, first half: f = img / 255 (uint8) or img (float32); gy, gx = np.gradient(f) in float32 (central
// differences halved, one-sided at the border); layer o = float32(max(cos(th_o) * gx + sin(th_o) * gy, 0)) with the
// products and the sum in float64 (numpy promotes: the cosines are float64 scalars).
template <typename T>
__global__ __launch_bounds__(256) void daisy_layers_kernel(const T* __restrict__ tiles, int P, const double* __restrict__ cs,
                                                           float* __restrict__ layers)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, t = blockIdx.z;
    if (x >= P) return;
    const T* img = tiles + (size_t)t * P * P;
    auto f = [&](int yy, int xx) -> float {
        const float v = (float)img[(size_t)yy * P + xx];
        return sizeof(T) == 1 ? v / 255.0f : v;
    };
    float gx, gy;
    if (P == 1) { gx = gy = 0.f; }
    else {
        gx = x == 0 ? f(y, 1) - f(y, 0) : (x == P - 1 ? f(y, P - 1) - f(y, P - 2) : (f(y, x + 1) - f(y, x - 1)) / 2.0f);
        gy = y == 0 ? f(1, x) - f(0, x) : (y == P - 1 ? f(P - 1, x) - f(P - 2, x) : (f(y + 1, x) - f(y - 1, x)) / 2.0f);
    }
#pragma unroll
    for (int o = 0; o < 8; o++) {
        const double v = __dadd_rn(__dmul_rn(cs[2 * o], (double)gx), __dmul_rn(cs[2 * o + 1], (double)gy));
        layers[(((size_t)t * 8 + o) * P + y) * P + x] = (float)(v >= 0.0 ? v : 0.0);   // np.maximum(v, 0) keeps -0.0
    }
}

// scipy.ndimage.correlate1d(mode="nearest") with a symmetric kernel along one axis of a stack of planes: float32 in
// and out, float64 accumulation  acc = x[0]*w[0];  acc += (x[-j] + x[+j]) * w[j]  from the outermost tap inwards.
// w[0..r]: centre first.
template <bool ALONG_X>
__global__ __launch_bounds__(256) void smooth_axis_kernel(const float* __restrict__ src, int P, const double* __restrict__ w,
                                                          int r, float* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= P) return;
    const float* s = src + (size_t)blockIdx.z * P * P;
    auto at = [&](int d) -> double {
        if (ALONG_X) return (double)s[(size_t)y * P + d_clamp(x + d, 0, P - 1)];
        return (double)s[(size_t)d_clamp(y + d, 0, P - 1) * P + x];
    };
    double acc = __dmul_rn(at(0), w[0]);
    for (int j = r; j >= 1; j--) acc = __dadd_rn(acc, __dmul_rn(__dadd_rn(at(-j), at(j)), w[j]));
    dst[(size_t)blockIdx.z * P * P + (size_t)y * P + x] = (float)acc;
}

// The same filter for radii up to RB with the window of a thread in registers: the plane is read as [A][B] (B contiguous,
// one thread per b), filtered along A for NY consecutive outputs per thread -- every input row is loaded and converted
// once for up to NY outputs instead of once per tap -- and written TRANSPOSED, [B][A], through LDS.
// Two launches make scipy's two passes: y then x, the second one reading the transposed intermediate and transposing it
// back.  Per output the arithmetic is that of smooth_axis_kernel, tap for tap: acc = x[0] w[0]; acc += (x[-j] + x[+j]) w[j],
// j = r .. 1.  Taps beyond r are skipped by a uniform branch (their rows are loaded all the same: clamped addresses).
constexpr int ST_B = 64;      // one wave per block: 64 columns b

template <int RB, int NY, int NIT>
__global__ __launch_bounds__(ST_B) void smooth_transposing_kernel(const float* __restrict__ src, int A, int B,
                                                                  const double* __restrict__ w, int r, float* __restrict__ dst)
{
    // the wave's outputs, [b][a] with a odd pitch: written a column of NY per thread, read back a row of 64 per store,
    // so that the transposed plane is written in runs of 256 contiguous bytes and not in pieces of a thread's 64
    constexpr int NA = NY * NIT, TP = NA + 1;
    __shared__ float tile[ST_B * TP];
    const int lane = threadIdx.x, b0 = blockIdx.x * ST_B, a00 = blockIdx.y * NA;
    const int b = min(b0 + lane, B - 1);         // lanes past the edge compute a copy of the last column and store nothing
    const float* s = src + (size_t)blockIdx.z * A * B + b;
    double wv[RB + 1];
#pragma unroll
    for (int j = 0; j <= RB; j++) wv[j] = w[j];        // the table is padded: entries beyond r exist and are not used
    for (int it = 0; it < NIT; it++) {
        const int a0 = a00 + it * NY;
        if (a0 >= A) break;
        // every row of the window is loaded (clamped addresses are always valid) before anything is used: one batch of
        // loads in flight instead of a round trip per row
        float raw[NY + 2 * RB];
#pragma unroll
        for (int i = 0; i < NY + 2 * RB; i++) raw[i] = s[(size_t)d_clamp(a0 + i - RB, 0, A - 1) * B];
        double win[NY + 2 * RB];
#pragma unroll
        for (int i = 0; i < NY + 2 * RB; i++) win[i] = (double)raw[i];
        double acc[NY];
#pragma unroll
        for (int o = 0; o < NY; o++) acc[o] = __dmul_rn(win[o + RB], wv[0]);
#pragma unroll
        for (int j = RB; j >= 1; j--) {
            if (j <= r) {
#pragma unroll
                for (int o = 0; o < NY; o++)
                    acc[o] = __dadd_rn(acc[o], __dmul_rn(__dadd_rn(win[o + RB - j], win[o + RB + j]), wv[j]));
            }
        }
#pragma unroll
        for (int o = 0; o < NY; o++) tile[lane * TP + it * NY + o] = (float)acc[o];
    }
    __syncthreads();
    float* d = dst + (size_t)blockIdx.z * A * B + a00;
    const int nb = min(ST_B, B - b0);
    for (int a = lane; a < NA; a += 64) {
        if (a00 + a < A) {
            for (int bb = 0; bb < nb; bb++) d[(size_t)(b0 + bb) * A + a] = tile[bb * TP + a];
        }
    }
}

// one smoothing pass pair (y, then x) of `planes` planes of P x P: src -> mid (transposed) -> dst
template <int RB, int NY, int NIT>
static void smooth_pair(hipStream_t stream, const float* src, float* mid, float* dst, int P, int planes, const double* w, int r)
{
    const dim3 grid((P + ST_B - 1) / ST_B, (P + NY * NIT - 1) / (NY * NIT), planes);
    hipLaunchKernelGGL((smooth_transposing_kernel<RB, NY, NIT>), grid, dim3(ST_B), 0, stream, src, P, P, w, r, mid);
    hipLaunchKernelGGL((smooth_transposing_kernel<RB, NY, NIT>), grid, dim3(ST_B), 0, stream, (const float*)mid, P, P, w, r, dst);
}

static void smooth_planes(hipStream_t stream, const float* src, float* mid, float* dst, int P, int planes, const double* w, int r)
{
    if (r <= 12) smooth_pair<12, 16, 4>(stream, src, mid, dst, P, planes, w, r);
    else if (r <= 18) smooth_pair<18, 16, 4>(stream, src, mid, dst, P, planes, w, r);
    else if (r <= 24) smooth_pair<24, 16, 4>(stream, src, mid, dst, P, planes, w, r);
    else if (r <= 40) smooth_pair<40, 8, 8>(stream, src, mid, dst, P, planes, w, r);
    else {   // any radius: a thread per output, every tap from memory
        const dim3 pgrid((P + 255) / 256, P, planes);
        hipLaunchKernelGGL((smooth_axis_kernel<false>), pgrid, dim3(256), 0, stream, src, P, w, r, mid);
        hipLaunchKernelGGL((smooth_axis_kernel<true>), pgrid, dim3(256), 0, stream, (const float*)mid, P, w, r, dst);
    }
}

// Daisy.compute: one thread per (keypoint, histogram location); 25 locations x 8 orientation bins = 200 floats.
// Location 0 samples cube 0 at the keypoint, location 1 + 8 r + j samples cube r at the keypoint + offs[1 + 8 r + j]
// (float64 offsets computed by the host with numpy); bilinear weights and the blend in float32, left to right.
__global__ __launch_bounds__(256) void daisy_sample_kernel(const float* __restrict__ cubes, size_t cube_stride, int P,
                                                           const int* __restrict__ kp_tile, const double* __restrict__ kp_xy,
                                                           const double* __restrict__ offs, int nkp, float* __restrict__ desc)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nkp * 25) return;
    const int k = e / 25, loc = e - k * 25;
    const int cube = loc == 0 ? 0 : (loc - 1) / 8;
    double ys = kp_xy[2 * k + 1] + offs[2 * loc], xs = kp_xy[2 * k] + offs[2 * loc + 1];
    ys = fmin(fmax(ys, 0.0), (double)P - 1.0);
    xs = fmin(fmax(xs, 0.0), (double)P - 1.0);
    long long y0 = (long long)floor(ys), x0 = (long long)floor(xs);
    if (P > 1) { y0 = y0 < P - 2 ? y0 : P - 2; x0 = x0 < P - 2 ? x0 : P - 2; } else { y0 = x0 = 0; }
    const float fy = (float)(ys - (double)y0), fx = (float)(xs - (double)x0);
    const long long y1 = y0 + 1 < P - 1 ? y0 + 1 : P - 1, x1 = x0 + 1 < P - 1 ? x0 + 1 : P - 1;
    const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
    const float* base = cubes + cube * cube_stride + (size_t)kp_tile[k] * 8 * P * P;
    float* out = desc + (size_t)k * 200 + loc * 8;
#pragma unroll
    for (int o = 0; o < 8; o++) {
        const float* pl = base + (size_t)o * P * P;
        out[o] = pl[y0 * P + x0] * w00 + pl[y0 * P + x1] * w01 + pl[y1 * P + x0] * w10 + pl[y1 * P + x1] * w11;
    }
}


// ---- feature tiles cut on the device (tile_registration.py:27-34 / slicer.py:69-118 for a uint8 image) ----------------------
// tile t = (ty, tx) in row-major order; its window starts at (ty*T - ov, tx*T - ov) and is zero outside the image
__global__ __launch_bounds__(256) void cut_tiles_kernel(const uint8_t* __restrict__ img, int H, int W, int T, int ov, int ntx,
                                                        int P, int t0, uint8_t* __restrict__ tiles)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, t = blockIdx.z;
    if (x >= P) return;
    const int tile = t0 + t, ty = tile / ntx, tx = tile - ty * ntx;
    const int iy = ty * T - ov + y, ix = tx * T - ov + x;
    uint8_t v = 0;
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = img[(size_t)iy * W + ix];
    tiles[((size_t)t * P + y) * P + x] = v;
}

// ---- keypoint selection on the device ------------------------------------------------------------------------------
// feature_detection.py:105-106: the corners of a tile sorted by response, strongest first (Python's stable sort keeps
// the detector's row-major order among equal responses), cut to the per-tile limit.  Over the non-maximum-suppressed
// score map of every tile, in chunks of 4096 scores (one block each, so that a level of one or four tiles still fills
// the chip; a single block per tile spent a millisecond walking its map):
//   A. histogram of the scores (1 .. 254) per chunk and per tile -> the cut-off score s*, how many corners of exactly
//      that score still fit, and by a scan over the chunk histograms where every chunk's share of the selection starts;
//   B. every chunk collects its corners above s* and, by rank in row-major order, its share of the first `need_eq`
//      corners at s* as keys (65535 - score) << 32 | row-major index;
//   C. bitonic sort of the tile's <= 8192 keys in LDS: ascending key = descending score, row-major among equals.
constexpr int KS_T = 1024, KS_CAP = 8192;       // sorting block, most keys per tile
constexpr int KC_T = 256, KC_E = 16, KC_CH = KC_T * KC_E;   // a chunk of the score map: 256 threads x 16 consecutive scores

template <int T>
__device__ __forceinline__ int ks_block_exscan(int v, int* wsum, int& total)
{
    // exclusive scan of one int per thread over a block of T threads; total = block sum (all threads)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int n = __shfl_up(inc, off);
        if (lane >= off) inc += n;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < T / 64; k++) {
        const int sk = wsum[k];
        if (k < wv) base += sk;
        tot += sk;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// A. histograms: of every chunk (kept: the cut-off turns them into the chunk's share of the selection) and of the tile
__global__ __launch_bounds__(KC_T) void kp_chunk_hist_kernel(const int* __restrict__ score, int n, int nch,
                                                             int* __restrict__ chunk_hist, int* __restrict__ tile_hist)
{
    __shared__ int hist[256];
    const int t = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const int* s = score + (size_t)t * n;
    hist[tid] = 0;
    __syncthreads();
    const int i0 = c * KC_CH + tid * KC_E;
#pragma unroll
    for (int e = 0; e < KC_E; e++) {
        const int v = i0 + e < n ? s[i0 + e] : 0;
        if (v > 0) atomicAdd(&hist[v < 255 ? v : 255], 1);
    }
    __syncthreads();
    const int h = hist[tid];
    chunk_hist[((size_t)t * nch + c) * 256 + tid] = h;
    if (h && tid) atomicAdd(&tile_hist[t * 256 + tid], h);
}

// the cut-off score s*, how many corners of exactly that score still fit, and where every chunk's share starts
__global__ __launch_bounds__(KC_T) void kp_cut_kernel(const int* __restrict__ chunk_hist, const int* __restrict__ tile_hist,
                                                      int nch, int limit, int* __restrict__ cut, int* __restrict__ chunk_base)
{
    __shared__ int wsum[KC_T / 64];
    __shared__ int sc[3];
    const int t = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        const int* hist = tile_hist + t * 256;
        int total = 0;
        for (int k = 1; k < 256; k++) total += hist[k];
        int sstar = 0, need_eq = 0;
        if (total > limit) {
            int gt = 0;
            for (int k = 255; k >= 1; k--) {
                if (gt + hist[k] > limit) { sstar = k; need_eq = limit - gt; break; }
                gt += hist[k];
            }
        }
        sc[0] = sstar; sc[1] = need_eq; sc[2] = total < limit ? total : limit;
        cut[t * 3] = sstar; cut[t * 3 + 1] = need_eq; cut[t * 3 + 2] = sc[2];
    }
    __syncthreads();
    const int sstar = sc[0], need_eq = sc[1];
    int eq_run = 0, sel_run = 0;
    for (int c0 = 0; c0 < nch; c0 += KC_T) {
        const int c = c0 + tid;
        int gt = 0, eq = 0;
        if (c < nch) {
            const int* h = chunk_hist + ((size_t)t * nch + c) * 256;
            for (int k = sstar + 1; k < 256; k++) gt += h[k];
            eq = sstar > 0 ? h[sstar] : 0;
        }
        int eq_tot, sel_tot;
        const int eq_base = eq_run + ks_block_exscan<KC_T>(eq, wsum, eq_tot);
        const int take_eq = min(eq, max(need_eq - eq_base, 0));          // the first need_eq of them in row-major order
        const int sel_base = sel_run + ks_block_exscan<KC_T>(gt + take_eq, wsum, sel_tot);
        if (c < nch) {
            chunk_base[((size_t)t * nch + c) * 2] = eq_base;
            chunk_base[((size_t)t * nch + c) * 2 + 1] = sel_base;
        }
        eq_run += eq_tot;
        sel_run += sel_tot;
    }
}

// B. every chunk puts its corners above s* and its share of the corners at s* into the tile's key list as
//    (65535 - score) << 32 | row-major index (the order of the list does not matter: it is sorted next)
__global__ __launch_bounds__(KC_T) void kp_collect_kernel(const int* __restrict__ score, int n, int nch,
                                                          const int* __restrict__ cut, const int* __restrict__ chunk_base,
                                                          unsigned long long* __restrict__ keys)
{
    __shared__ int wsum[KC_T / 64];
    const int t = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const int* s = score + (size_t)t * n;
    const int sstar = cut[t * 3], need_eq = cut[t * 3 + 1];
    const int i0 = c * KC_CH + tid * KC_E;
    int v[KC_E], neq = 0;
#pragma unroll
    for (int e = 0; e < KC_E; e++) {
        v[e] = i0 + e < n ? s[i0 + e] : 0;
        neq += (sstar > 0 && v[e] == sstar) ? 1 : 0;
    }
    int tot;
    int eq_rank = chunk_base[((size_t)t * nch + c) * 2] + ks_block_exscan<KC_T>(neq, wsum, tot);
    int nsel = 0;
    bool take[KC_E];
#pragma unroll
    for (int e = 0; e < KC_E; e++) {
        bool tk = v[e] > sstar;                              // sstar == 0: every corner
        if (sstar > 0 && v[e] == sstar) { tk = eq_rank < need_eq; eq_rank++; }
        take[e] = tk;
        nsel += tk ? 1 : 0;
    }
    int pos = chunk_base[((size_t)t * nch + c) * 2 + 1] + ks_block_exscan<KC_T>(nsel, wsum, tot);
    unsigned long long* k = keys + (size_t)t * KS_CAP;
#pragma unroll
    for (int e = 0; e < KC_E; e++)
        if (take[e]) k[pos++] = ((unsigned long long)(65535 - v[e]) << 32) | (unsigned)(i0 + e);
}

// C. bitonic sort of the tile's <= 8192 keys in LDS: ascending key = descending score, row-major among equals
__global__ __launch_bounds__(KS_T) void kp_sort_kernel(const unsigned long long* __restrict__ keys_in, const int* __restrict__ cut,
                                                       int Pi, int limit, int* __restrict__ kp_out, int* __restrict__ counts)
{
    __shared__ unsigned long long keys[KS_CAP];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int n_sel = cut[t * 3 + 2];
    int m = 1;
    while (m < n_sel) m <<= 1;
    for (int i = tid; i < m; i += KS_T) keys[i] = i < n_sel ? keys_in[(size_t)t * KS_CAP + i] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < m; i += KS_T) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = keys[i], b = keys[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < n_sel; i += KS_T) {
        const unsigned long long key = keys[i];
        const int idx = (int)(key & 0xffffffffu), sc = 65535 - (int)(key >> 32);
        int* o = kp_out + ((size_t)t * limit + i) * 3;
        o[0] = idx % Pi; o[1] = idx / Pi; o[2] = sc;
    }
    if (tid == 0) counts[t] = n_sel;
}

} // namespace

extern "C" {

int ma_cut_tiles_u8(ma_ctx* ctx, const uint8_t* img, int H, int W, int tile, int overlap, int first_tile, int n_tiles,
                    uint8_t* tiles_out)
{
    MA_REQUIRE(ctx && img && tiles_out, "NULL argument");
    MA_REQUIRE(H > 0 && W > 0 && tile > 0 && overlap >= 0 && first_tile >= 0 && n_tiles >= 1 && n_tiles <= 65535, "bad tile geometry");
    const int ntx = (W + tile - 1) / tile, nty = (H + tile - 1) / tile, P = tile + 2 * overlap;
    MA_REQUIRE(first_tile + n_tiles <= ntx * nty && P <= 65535, "tile range out of bounds");
    MA_HIP(hipSetDevice(ctx->device));
    MaProfScope ps(ctx, MA_K_OTHER, (double)n_tiles * P * P);
    hipLaunchKernelGGL(cut_tiles_kernel, dim3((P + 255) / 256, P, n_tiles), dim3(256), 0, ctx->stream, img, H, W, tile, overlap,
                       ntx, P, first_tile, tiles_out);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

int ma_fast_keypoints(ma_ctx* ctx, const uint8_t* tiles, int nt, int P, int margin, int threshold, int limit, int* kp_out,
                      int* counts_host)
{
    MA_REQUIRE(ctx && tiles && kp_out && counts_host, "NULL argument");
    MA_REQUIRE(nt >= 1 && nt <= 65535 && margin >= 0 && P - 2 * margin >= 1 && P - 2 * margin <= 46340, "bad tile geometry");
    MA_REQUIRE(limit >= 1 && limit <= KS_CAP, "limit must be in [1, 8192]");
    MA_REQUIRE(threshold >= 0 && threshold < 254, "bad threshold");
    MA_HIP(hipSetDevice(ctx->device));
    const int Pi = P - 2 * margin;
    const size_t npx = (size_t)Pi * Pi, map = (size_t)nt * npx * sizeof(int);
    const int nch = (int)((npx + KC_CH - 1) / KC_CH);
    MA_REQUIRE(nch <= 65535 * 32, "tile too large");
    // workspace: raw scores, suppressed scores, per-chunk histograms, key lists, tile histograms, chunk bases, cuts, counts
    const size_t b_ch = (size_t)nt * nch * 256 * sizeof(int), b_keys = (size_t)nt * KS_CAP * sizeof(unsigned long long),
                 b_th = (size_t)nt * 256 * sizeof(int), b_cb = (size_t)nt * nch * 2 * sizeof(int), b_cut = (size_t)nt * 4 * sizeof(int);
    MA_TRY(ma_ws_reserve(ctx, 2 * map + b_keys + b_ch + b_th + b_cb + b_cut + (size_t)nt * sizeof(int) + 64));
    MA_TRY(ma_pinned_reserve(ctx, (size_t)nt * sizeof(int)));
    int* raw = (int*)ctx->ws;
    int* nms = raw + (size_t)nt * npx;
    unsigned long long* keys = (unsigned long long*)((char*)ctx->ws + ma_align_up(2 * map, 8));
    int* chunk_hist = (int*)(keys + (size_t)nt * KS_CAP);
    int* tile_hist = chunk_hist + (size_t)nt * nch * 256;
    int* chunk_base = tile_hist + (size_t)nt * 256;
    int* cut = chunk_base + (size_t)nt * nch * 2;
    int* counts = cut + (size_t)nt * 4;
    {
        MaProfScope ps(ctx, MA_K_OTHER, (double)nt * Pi * Pi);
        const dim3 grid((Pi + 255) / 256, Pi, nt);
        MA_HIP(hipMemsetAsync(tile_hist, 0, b_th, ctx->stream));
        hipLaunchKernelGGL(fast_score_kernel, grid, dim3(256), 0, ctx->stream, tiles, P, margin, threshold, raw);
        hipLaunchKernelGGL(fast_nms_kernel, grid, dim3(256), 0, ctx->stream, (const int*)raw, Pi, nms);
        // selection: chunk histograms -> cut-off and chunk bases -> keys -> sort (one block per tile only for the sort)
        hipLaunchKernelGGL(kp_chunk_hist_kernel, dim3(nch, nt), dim3(KC_T), 0, ctx->stream, (const int*)nms, (int)npx, nch,
                           chunk_hist, tile_hist);
        hipLaunchKernelGGL(kp_cut_kernel, dim3(nt), dim3(KC_T), 0, ctx->stream, (const int*)chunk_hist, (const int*)tile_hist, nch,
                           limit, cut, chunk_base);
        hipLaunchKernelGGL(kp_collect_kernel, dim3(nch, nt), dim3(KC_T), 0, ctx->stream, (const int*)nms, (int)npx, nch,
                           (const int*)cut, (const int*)chunk_base, keys);
        hipLaunchKernelGGL(kp_sort_kernel, dim3(nt), dim3(KS_T), 0, ctx->stream, (const unsigned long long*)keys, (const int*)cut,
                           Pi, limit, kp_out, counts);
        MA_HIP(hipGetLastError());
    }
    MA_HIP(hipMemcpyAsync(ctx->pinned, counts, (size_t)nt * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MA_HIP(hipStreamSynchronize(ctx->stream));
    for (int t = 0; t < nt; t++) counts_host[t] = ((const int*)ctx->pinned)[t];
    return MA_OK;
}

int ma_fast_nms(ma_ctx* ctx, const uint8_t* tiles, int nt, int P, int margin, int threshold, int* score_out)
{
    MA_REQUIRE(ctx && tiles && score_out, "NULL argument");
    MA_REQUIRE(nt >= 1 && nt <= 65535 && margin >= 0 && P - 2 * margin >= 1 && P - 2 * margin <= 65535, "bad tile geometry");
    MA_HIP(hipSetDevice(ctx->device));
    const int Pi = P - 2 * margin;
    MA_TRY(ma_ws_reserve(ctx, (size_t)nt * Pi * Pi * sizeof(int)));
    MaProfScope ps(ctx, MA_K_OTHER, (double)nt * Pi * Pi);
    const dim3 grid((Pi + 255) / 256, Pi, nt);
    hipLaunchKernelGGL(fast_score_kernel, grid, dim3(256), 0, ctx->stream, tiles, P, margin, threshold, (int*)ctx->ws);
    hipLaunchKernelGGL(fast_nms_kernel, grid, dim3(256), 0, ctx->stream, (const int*)ctx->ws, Pi, score_out);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

int ma_daisy_describe(ma_ctx* ctx, const void* tiles, int dtype, int nt, int P, const double* const* weights_host,
                      const int* radii, const double* cos_sin_host, const double* offs_host, const int* kp_tile,
                      const double* kp_xy, int nkp, float* desc_out)
{
    MA_REQUIRE(ctx && tiles && weights_host && radii && cos_sin_host && offs_host && kp_tile && kp_xy && desc_out, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_F32, "tiles must be uint8 or float32");
    MA_REQUIRE(nt >= 1 && nt * 8 <= 65535 && P >= 1 && P <= 65535 && nkp >= 1, "bad tile geometry");
    MA_REQUIRE(radii[0] >= 0 && radii[1] >= 0 && radii[2] >= 0 && radii[0] + radii[1] + radii[2] < 4096, "bad radii");
    MA_HIP(hipSetDevice(ctx->device));
    // small tables: 8 (cos, sin) pairs, 25 (dy, dx) offsets, three centre-first half kernels
    const size_t ntab = 16 + 50 + (size_t)(radii[0] + radii[1] + radii[2] + 3);
    MA_TRY(ma_dconst_reserve(ctx, (ntab + 64) * sizeof(double)));   // + 64: the smoothing kernels read a whole bucket of weights
    std::vector<double> tab(ntab);
    for (int i = 0; i < 16; i++) tab[i] = cos_sin_host[i];
    for (int i = 0; i < 50; i++) tab[16 + i] = offs_host[i];
    size_t woff[3], o = 66;
    for (int c = 0; c < 3; c++) {
        woff[c] = o;
        for (int j = 0; j <= radii[c]; j++) tab[o++] = weights_host[c][j];
    }
    double* dtab = (double*)ctx->dconst;
    MA_HIP(hipMemcpyAsync(dtab, tab.data(), ntab * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MA_HIP(hipStreamSynchronize(ctx->stream));   // `tab` goes out of scope

    // workspace: cubes 0..2 and one temporary, each nt x 8 planes of P x P float32
    const size_t cube = (size_t)nt * 8 * P * P;
    MA_TRY(ma_ws_reserve(ctx, 4 * cube * sizeof(float)));
    float* cubes = (float*)ctx->ws;
    float* tmp = cubes + 3 * cube;
    MaProfScope ps(ctx, MA_K_OTHER, (double)nt * P * P);
    {
        const dim3 grid((P + 255) / 256, P, nt);
        if (dtype == MA_U8) hipLaunchKernelGGL((daisy_layers_kernel<uint8_t>), grid, dim3(256), 0, ctx->stream, (const uint8_t*)tiles, P, dtab, tmp);
        else hipLaunchKernelGGL((daisy_layers_kernel<float>), grid, dim3(256), 0, ctx->stream, (const float*)tiles, P, dtab, tmp);
    }
    const float* src = tmp;   // the orientation layers; smoothed successively: cube c = G(inc_c) * cube c-1
    for (int c = 0; c < 3; c++) {
        float* dst = cubes + c * cube;
        // scipy filters axis 1 (y) first, then axis 2 (x), each pass rounding to float32.  The intermediate of the two
        // passes lives in the next cube's slot (not yet written) or, for the last cube, in the layer buffer (done with)
        float* mid = c < 2 ? cubes + (c + 1) * cube : tmp;
        smooth_planes(ctx->stream, src, mid, dst, P, nt * 8, dtab + woff[c], radii[c]);
        src = dst;
    }
    hipLaunchKernelGGL(daisy_sample_kernel, dim3((nkp * 25 + 255) / 256), dim3(256), 0, ctx->stream, (const float*)cubes, cube, P,
                       kp_tile, kp_xy, dtab + 16, nkp, desc_out);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

} // extern "C"
