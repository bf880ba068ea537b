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
                                                           float* __restrict__ layers, const int4* __restrict__ rects)
{
    // the layers are made where they can be non-zero: the content rectangle and one pixel around it (see DzRect below)
    const int t = blockIdx.z;
    const int4 c = rects[t];                               // y0, y1, x0, x1
    if (c.y <= c.x || c.w <= c.z) return;
    const int vy0 = max(c.x - 1, 0), vy1 = min(c.y + 1, P), vx0 = max(c.z - 1, 0), vx1 = min(c.w + 1, P);
    const int x = vx0 + blockIdx.x * 256 + threadIdx.x, y = vy0 + blockIdx.y;
    if (x >= vx1 || y >= vy1) return;
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

// ---- content rectangles -------------------------------------------------------------------------------------------------
// A feature tile is a window of the image, ZERO outside it (slicer.py:69-118): of the 3 x 3 windows of a 2048^2 level at
// tile size 1000, five hold a strip of 48 + 51 image pixels and 1000 columns of zeros.  Where the tile is zero the
// orientation layers are zero, and a smoothed cube is zero beyond the reach of its kernels, so every kernel below works on
// the tile's CONTENT rectangle dilated by the halo its stage has accumulated (1 px for the gradient, + the radius of every
// smoothing pass so far) and treats what lies outside as the zeros it holds -- without reading or writing them.  Values are
// those of the full computation (the sign of a zero can differ: -0.0 where the full computation multiplies a zero gradient
// by a negative cosine; IEEE comparisons and every later sum do not see it).  rect = {y0, y1, x0, x1} of the content in tile
// coordinates, y1 <= y0 for an empty tile; the plain entry point passes the whole tile.
struct DzRect { int y0, y1, x0, x1; };

__device__ __forceinline__ DzRect dz_dilate(const DzRect c, int hy, int hx, int P)
{
    DzRect v;
    v.y0 = max(c.y0 - hy, 0); v.y1 = min(c.y1 + hy, P);
    v.x0 = max(c.x0 - hx, 0); v.x1 = min(c.x1 + hx, P);
    if (c.y1 <= c.y0 || c.x1 <= c.x0) { v.y0 = v.y1 = v.x0 = v.x1 = 0; }
    return v;
}

// The smoothing filter with the window of a thread in registers, sliding: the plane is read as [A][B] (B contiguous, one
// thread per b), filtered along A and written TRANSPOSED, [B][A], through LDS.  A block of SW waves covers 64 columns and
// SW * NY * NIT consecutive outputs along A: its input rows -- the outputs' rows and RB more on either side, shared by the waves
// -- are staged in LDS once, by all threads, with every load of the block in flight at the same time (the earlier form, in
// which each thread loaded its own rows one by one, spent 70 % of its wave cycles waiting for them: profiles/r06_notes.md);
// then every thread slides its window of NY + 2 RB inputs down its column of the LDS tile, NY rows per step.  Two launches
// make scipy's two passes: y then x, the second one reading the transposed intermediate and transposing it back.  Per output
// the arithmetic is that of smooth_axis_kernel, tap for tap: acc = x[0] w[0]; acc += (x[-j] + x[+j]) w[j], j = r .. 1.  Taps
// beyond r are skipped by a uniform branch.
// Geometry per plane (tile = plane / 8): the input holds values for a in [ia0, ia1), b in [b0, b1) -- the content rectangle
// dilated by (h_a, h_b) -- and zeros elsewhere; outputs are made for a in [ia0 - r, ia1 + r) within the plane.  swap: 0 when
// a runs along y (first pass), 1 when a runs along x (second pass, transposed input).
constexpr int ST_B = 64;      // columns b of a block: one per lane

template <int RB, int NY, int NIT, int SW>
__global__ __launch_bounds__(64 * SW) void smooth_slide_kernel(const float* __restrict__ src, int A, int B,
                                                               const double* __restrict__ w, int r, float* __restrict__ dst,
                                                               const DzRect* __restrict__ rects, int h_a, int h_b, int swap)
{
    constexpr int NA = NY * NIT, TP = NA + 1, WN = NY + 2 * RB, ROWS = SW * NA + 2 * RB;
    static_assert(SW * ST_B * TP <= ROWS * ST_B, "the transposition tiles reuse the staging buffer");
    extern __shared__ float stage[];                       // [ROWS][64] inputs; later SW x [64][TP] outputs
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    DzRect c = rects[blockIdx.z >> 3];
    if (swap) { DzRect t = c; c.y0 = t.x0; c.y1 = t.x1; c.x0 = t.y0; c.x1 = t.y1; }
    const DzRect vin = dz_dilate(c, h_a, h_b, A);          // A == B == P
    const int ia0 = vin.y0, ia1 = vin.y1, b0 = vin.x0 + blockIdx.x * ST_B, b1 = vin.x1;
    const int oa0 = max(ia0 - r, 0), oa1 = min(ia1 + r, A);
    const int ablk = oa0 + blockIdx.y * (SW * NA);
    if (b0 >= b1 || ablk >= oa1 || ia1 <= ia0) return;     // uniform: nothing of this plane in the block
    {
        // stage rows ablk - RB .. ablk + SW NA + RB: clamped like scipy's mode="nearest", zero outside the rows that hold
        // values; lanes past the edge hold a copy of the last column (they compute and store nothing of their own)
        const int b = min(b0 + lane, b1 - 1);
        const float* s = src + (size_t)blockIdx.z * A * B + b;
        const int last = min(ROWS, oa1 - ablk + 2 * RB);   // rows beyond the last output's window are not needed
        // all loads of a thread first, then the LDS writes: as a plain loop the compiler waits for every load before it
        // issues the next one (one memory latency per row; profiles/r06_notes.md)
        constexpr int PER = (ROWS + SW - 1) / SW;
        float tmp[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = wv + k * SW;
            const int ac = d_clamp(ablk - RB + i, 0, A - 1);
            float v = 0.f;
            if (i < last && ac >= ia0 && ac < ia1) v = s[(size_t)ac * B];
            tmp[k] = v;
        }
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = wv + k * SW;
            if (i < ROWS) stage[i * ST_B + lane] = tmp[k];
        }
    }
    __syncthreads();
    const int a00 = ablk + wv * NA;
    const bool live = a00 < oa1;                           // wave-uniform: waves past the last output only keep the barriers
    const float* col = stage + (wv * NA) * ST_B + lane;    // this thread's column, element i = input row a00 - RB + i
    double wv_[RB + 1];
#pragma unroll
    for (int j = 0; j <= RB; j++) wv_[j] = w[j];           // the table is padded: entries beyond r exist and are not used
    double win[WN];
    float outv[NA];
    if (live) {
#pragma unroll
        for (int i = 0; i < 2 * RB; i++) win[i] = (double)col[i * ST_B];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            if (a00 + it * NY < oa1) {
#pragma unroll
                for (int i = 0; i < NY; i++) win[2 * RB + i] = (double)col[(2 * RB + it * NY + i) * ST_B];
                double acc[NY];
#pragma unroll
                for (int o = 0; o < NY; o++) acc[o] = __dmul_rn(win[o + RB], wv_[0]);
#pragma unroll
                for (int j = RB; j >= 1; j--) {
                    if (j <= r) {
                        // the NY chains of a tap in three phases (sums, products, accumulation): left to itself the compiler
                        // threads all of them through ONE temporary register pair and every instruction waits for the last
                        double t[NY];
#pragma unroll
                        for (int o = 0; o < NY; o++) t[o] = __dadd_rn(win[o + RB - j], win[o + RB + j]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int o = 0; o < NY; o++) t[o] = __dmul_rn(t[o], wv_[j]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int o = 0; o < NY; o++) acc[o] = __dadd_rn(acc[o], t[o]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int o = 0; o < NY; o++) outv[it * NY + o] = (float)acc[o];
#pragma unroll
                for (int i = 0; i < 2 * RB; i++) win[i] = win[i + NY];
            }
        }
    }
    __syncthreads();                                       // every window has been read: the buffer becomes the output tiles
    float* tile = stage + wv * (ST_B * TP);
    if (live) {
#pragma unroll
        for (int i = 0; i < NA; i++) tile[lane * TP + i] = outv[i];
    }
    __syncthreads();
    if (!live) return;
    float* d = dst + (size_t)blockIdx.z * A * B + a00;
    const int nb = min(ST_B, b1 - b0);
    // rows of NA floats per column b; with NA < 64 a store instruction carries 64 / NA of them
    constexpr int PER = NA >= 64 ? 1 : 64 / NA;
    for (int a = lane % (64 / PER); a < NA; a += 64 / PER) {
        if (a00 + a < oa1) {
            for (int bb = lane / (64 / PER); bb < nb; bb += PER) d[(size_t)(b0 + bb) * A + a] = tile[bb * TP + a];
        }
    }
}

// smooth_axis_kernel with the same geometry (any radius: a thread per output, every tap from memory)
template <bool ALONG_X>
__global__ __launch_bounds__(256) void smooth_axis_rect_kernel(const float* __restrict__ src, int P, const double* __restrict__ w,
                                                               int r, float* __restrict__ dst, const DzRect* __restrict__ rects,
                                                               int h_y, int h_x)
{
    const DzRect vin = dz_dilate(rects[blockIdx.z >> 3], h_y, h_x, P);
    DzRect vo = vin;
    if (ALONG_X) { vo.x0 = max(vin.x0 - r, 0); vo.x1 = min(vin.x1 + r, P); }
    else { vo.y0 = max(vin.y0 - r, 0); vo.y1 = min(vin.y1 + r, P); }
    const int x = vo.x0 + blockIdx.x * 256 + threadIdx.x, y = vo.y0 + blockIdx.y;
    if (x >= vo.x1 || y >= vo.y1 || vin.y1 <= vin.y0) return;
    const float* s = src + (size_t)blockIdx.z * P * P;
    auto at = [&](int d) -> double {
        const int yy = ALONG_X ? y : d_clamp(y + d, 0, P - 1), xx = ALONG_X ? d_clamp(x + d, 0, P - 1) : x;
        if (yy < vin.y0 || yy >= vin.y1 || xx < vin.x0 || xx >= vin.x1) return 0.0;
        return (double)s[(size_t)yy * P + xx];
    };
    double acc = __dmul_rn(at(0), w[0]);
    for (int j = r; j >= 1; j--) acc = __dadd_rn(acc, __dmul_rn(__dadd_rn(at(-j), at(j)), w[j]));
    dst[(size_t)blockIdx.z * P * P + (size_t)y * P + x] = (float)acc;
}

// extent of the rectangles the launches below have to cover (host side: the rects are known there)
struct DzSpan { int h, w; };      // the largest content height / width over the tiles of the batch

// one smoothing pass pair (y, then x) of `planes` planes of P x P: src -> mid (transposed) -> dst.  h: halo the input has
// accumulated; span: largest content extent.
template <int RB, int NY, int NIT, int SW>
static int smooth_pair_launch(hipStream_t stream, const float* src, float* mid, float* dst, int P, int planes, const double* w, int r,
                              const DzRect* rects, int h, int in_h, int in_w, int out_h, int out_w)
{
    constexpr int NA = NY * NIT * SW;
    constexpr size_t lds = (size_t)(NA + 2 * RB) * ST_B * sizeof(float);
    static bool raised = false;         // per kernel instantiation; the attribute is per function, idempotent
    if (!raised && lds > 64 * 1024) {
        MA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(smooth_slide_kernel<RB, NY, NIT, SW>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        raised = true;
    }
    // pass 1: a = y (outputs out_h), b = x (in_w columns); pass 2: a = x (outputs out_w), b = y (out_h columns)
    const dim3 g1((in_w + ST_B - 1) / ST_B, (out_h + NA - 1) / NA, planes), g2((out_h + ST_B - 1) / ST_B, (out_w + NA - 1) / NA, planes);
    hipLaunchKernelGGL((smooth_slide_kernel<RB, NY, NIT, SW>), g1, dim3(64 * SW), lds, stream, src, P, P, w, r, mid, rects, h, h, 0);
    hipLaunchKernelGGL((smooth_slide_kernel<RB, NY, NIT, SW>), g2, dim3(64 * SW), lds, stream, (const float*)mid, P, P, w, r, dst, rects, h, h + r, 1);
    return MA_OK;
}

// one smoothing pass pair (y, then x) of `planes` planes of P x P: src -> mid (transposed) -> dst.  h: halo the input has
// accumulated; span: largest content extent.  A wave issues one arithmetic instruction every ~8 cycles on its own and the SIMD
// one every ~5 with two or more (tools/ubench_valu): a launch that would not put two long-running waves on every SIMD
// is cut into four times as many short ones (NIT = 1: NY outputs per thread instead of 4 NY).
template <int RB, int NY, int NIT>
static int smooth_pair(hipStream_t stream, const float* src, float* mid, float* dst, int P, int planes, const double* w, int r,
                       const DzRect* rects, int h, DzSpan span)
{
    const int in_h = std::min(P, span.h + 2 * h), in_w = std::min(P, span.w + 2 * h);
    const int out_h = std::min(P, in_h + 2 * r), out_w = std::min(P, in_w + 2 * r);
    const long long waves = (long long)((in_w + ST_B - 1) / ST_B) * ((out_h + NY * NIT - 1) / (NY * NIT)) * planes;
    if (NIT > 1 && waves < 3 * 1024) return smooth_pair_launch<RB, NY, 1, 4>(stream, src, mid, dst, P, planes, w, r, rects, h, in_h, in_w, out_h, out_w);
    return smooth_pair_launch<RB, NY, NIT, 4>(stream, src, mid, dst, P, planes, w, r, rects, h, in_h, in_w, out_h, out_w);
}

static int smooth_planes(hipStream_t stream, const float* src, float* mid, float* dst, int P, int planes, const double* w, int r,
                         const DzRect* rects, int h, DzSpan span)
{
    // NY * NIT outputs per thread wait in registers for the transposition: 32 of them leave room for two waves per SIMD
    if (r <= 12) return smooth_pair<12, 16, 2>(stream, src, mid, dst, P, planes, w, r, rects, h, span);
    if (r <= 18) return smooth_pair<18, 16, 2>(stream, src, mid, dst, P, planes, w, r, rects, h, span);
    if (r <= 24) return smooth_pair<24, 16, 2>(stream, src, mid, dst, P, planes, w, r, rects, h, span);
    if (r <= 40) return smooth_pair<40, 8, 4>(stream, src, mid, dst, P, planes, w, r, rects, h, span);
    // any radius: a thread per output, every tap from memory
    const int in_h = std::min(P, span.h + 2 * h), in_w = std::min(P, span.w + 2 * h);
    const int out_h = std::min(P, in_h + 2 * r), out_w = std::min(P, in_w + 2 * r);
    hipLaunchKernelGGL((smooth_axis_rect_kernel<false>), dim3((in_w + 255) / 256, out_h, planes), dim3(256), 0, stream, src, P, w,
                       r, mid, rects, h, h);
    hipLaunchKernelGGL((smooth_axis_rect_kernel<true>), dim3((out_w + 255) / 256, out_h, planes), dim3(256), 0, stream,
                       (const float*)mid, P, w, r, dst, rects, h + r, h);
    return MA_OK;
}

// Daisy.compute: one thread per (keypoint, histogram location); 25 locations x 8 orientation bins = 200 floats.
// Location 0 samples cube 0 at the keypoint, location 1 + 8 r + j samples cube r at the keypoint + offs[1 + 8 r + j]
// (float64 offsets computed by the host with numpy); bilinear weights and the blend in float32, left to right.
__global__ __launch_bounds__(256) void daisy_sample_kernel(const float* __restrict__ cubes, size_t cube_stride, int P,
                                                           const int* __restrict__ kp_tile, const double* __restrict__ kp_xy,
                                                           const double* __restrict__ offs, int nkp, const int* __restrict__ nkp_dev,
                                                           float* __restrict__ desc, const DzRect* __restrict__ rects, int h0,
                                                           int h1, int h2)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    // ma_feature_extract: nkp_dev[0] = the count the selection left on the device, nkp_dev[1] = descriptors of the batches
    // before this one (the slot of keypoint 0 in desc)
    size_t slot0 = 0;
    if (nkp_dev) { nkp = min(nkp, nkp_dev[0]); slot0 = (size_t)nkp_dev[1]; }
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
    const int t = kp_tile[k];
    // the cube holds values inside the content rectangle dilated by its accumulated halo and is zero (unwritten) outside
    const int h = cube == 0 ? h0 : (cube == 1 ? h1 : h2);
    const DzRect v = dz_dilate(rects[t], h, h, P);
    const bool iy0 = y0 >= v.y0 && y0 < v.y1, iy1 = y1 >= v.y0 && y1 < v.y1, ix0 = x0 >= v.x0 && x0 < v.x1,
               ix1 = x1 >= v.x0 && x1 < v.x1;
    const float* base = cubes + cube * cube_stride + (size_t)t * 8 * P * P;
    float* out = desc + (slot0 + k) * 200 + loc * 8;
#pragma unroll
    for (int o = 0; o < 8; o++) {
        const float* pl = base + (size_t)o * P * P;
        const float v00 = iy0 && ix0 ? pl[y0 * P + x0] : 0.f, v01 = iy0 && ix1 ? pl[y0 * P + x1] : 0.f;
        const float v10 = iy1 && ix0 ? pl[y1 * P + x0] : 0.f, v11 = iy1 && ix1 ? pl[y1 * P + x1] : 0.f;
        out[o] = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
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
//   C. stable counting sort of the tile's <= 8192 keys by score: descending score, row-major among equals.
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
    __shared__ int hist[256];
    static_assert(KC_T == 256, "one histogram bin per thread");
    const int t = blockIdx.x, tid = threadIdx.x;
    hist[tid] = tile_hist[t * 256 + tid];          // (thread 0 walked the 255 bins in global memory, one load after the other: 15 us)
    __syncthreads();
    if (tid == 0) {
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
            const int4* h4 = reinterpret_cast<const int4*>(chunk_hist + ((size_t)t * nch + c) * 256);
#pragma unroll 16
            for (int k4 = 0; k4 < 64; k4++) {        // all 64 loads independent of sstar: in flight together
                const int4 v = h4[k4];
                const int k = 4 * k4;
                gt += (k > sstar ? v.x : 0) + (k + 1 > sstar ? v.y : 0) + (k + 2 > sstar ? v.z : 0) + (k + 3 > sstar ? v.w : 0);
                eq += (k == sstar ? v.x : 0) + (k + 1 == sstar ? v.y : 0) + (k + 2 == sstar ? v.z : 0) + (k + 3 == sstar ? v.w : 0);
            }
            if (sstar <= 0) eq = 0;
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

// C. the tile's <= 8192 keys in order of descending score, row-major among equals.  kp_collect leaves the list in row-major
//    order (chunk by chunk, thread by thread, element by element), so this is a STABLE sort of the list by 255 - score, a key of
//    254 values: a counting sort.  Wave w owns a contiguous stretch of the list; per (score, wave) counts -> an exclusive scan in
//    (score, wave) order gives every wave the slot where its first key of a score goes; the wave then walks its stretch 64 keys
//    at a time, lanes with equal scores ranked by eight ballots.  Five block barriers (the bitonic network this replaces had
//    91 stages: 109 us per launch whatever the tile held, 1.3 ms per register(); now ~10 us).
__global__ __launch_bounds__(KS_T) void kp_sort_kernel(const unsigned long long* __restrict__ keys_in, const int* __restrict__ cut,
                                                       int Pi, int limit, int* __restrict__ kp_out, int* __restrict__ counts)
{
    constexpr int NW = KS_T / 64;
    static_assert(NW * 256 == 4 * KS_T, "four counters per thread in the scan");
    __shared__ int slot[256 * NW];                     // [255 - score][wave]: count, then next free output slot
    __shared__ int wsum[NW];
    const int t = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n_sel = cut[t * 3 + 2];
    const unsigned long long* list = keys_in + (size_t)t * KS_CAP;
    for (int i = tid; i < 256 * NW; i += KS_T) slot[i] = 0;
    __syncthreads();
    const int per = ((n_sel + NW - 1) / NW + 63) & ~63;
    const int beg = wave * per, end = min(n_sel, beg + per);
    auto bucket = [](unsigned long long key) { return (int)(((unsigned)(key >> 32) - (65535u - 255u)) & 255u); };   // score in 1 .. 254
    for (int i = beg + lane; i < end; i += 64) atomicAdd(&slot[bucket(list[i]) * NW + wave], 1);
    __syncthreads();
    {
        int v[4], sum = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) { v[e] = slot[tid * 4 + e]; sum += v[e]; }
        int inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int ex = inc - sum;
        for (int w = 0; w < wave; w++) ex += wsum[w];
#pragma unroll
        for (int e = 0; e < 4; e++) { slot[tid * 4 + e] = ex; ex += v[e]; }
    }
    __syncthreads();
    for (int i0 = beg; i0 < end; i0 += 64) {               // wave-uniform trip count
        const int i = i0 + lane;
        const bool valid = i < end;
        const unsigned long long key = valid ? list[i] : 0ull;
        const int b = valid ? bucket(key) : 0;
        unsigned long long same = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const bool on = (b >> bit) & 1;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(on);
            same &= on ? bal : ~bal;
        }
        const int rank = __popcll(same & ((1ull << lane) - 1ull));
        const int base = slot[b * NW + wave];
        __builtin_amdgcn_wave_barrier();                    // every lane has read its slot before a leader moves it on
        if (valid) {
            if (rank == 0) slot[b * NW + wave] = base + __popcll(same);
            const int idx = (int)(key & 0xffffffffu), sc = 65535 - (int)(key >> 32);
            int* o = kp_out + ((size_t)t * limit + base + rank) * 3;
            o[0] = idx % Pi; o[1] = idx / Pi; o[2] = sc;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (tid == 0) counts[t] = n_sel;
}

// ---- the selection's keypoints, compacted on the device (combine_features' layout, tile_registration.py:37-74) --------------
// base[t] = number of keypoints of the batch's tiles before t that stay (tiles with fewer than three are dropped,
// feature_detection.py:112-115); info[0] = keypoints of the batch, info[1] = keypoints of the batches before it; *total grows
__global__ void kp_offsets_kernel(const int* __restrict__ counts, int nt, int* __restrict__ base, int* __restrict__ info,
                                  int* __restrict__ total)
{
    if (threadIdx.x || blockIdx.x) return;
    int run = 0;
    for (int t = 0; t < nt; t++) {
        base[t] = run;
        run += counts[t] >= 3 ? counts[t] : 0;
    }
    info[0] = run;
    info[1] = *total;
    *total += run;
}

// keypoint k of tile t -> slot base[t] + k of the batch: its tile and tile coordinates for the descriptor, and -- at the
// level-wide slot info[1] + base[t] + k -- its image coordinates (tile origin + interior coordinate) and response
__global__ __launch_bounds__(256) void kp_compact_kernel(const int* __restrict__ kp, const int* __restrict__ counts,
                                                         const int* __restrict__ base, const int* __restrict__ info, int limit,
                                                         int tile_size, int ntx, int first_tile, int* __restrict__ kp_tile,
                                                         double* __restrict__ kp_xy, double* __restrict__ pts_out,
                                                         int* __restrict__ resp_out)
{
    const int t = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x, c = counts[t];
    if (c < 3 || k >= c) return;
    const int* e = kp + ((size_t)t * limit + k) * 3;
    const int o = base[t] + k, tile = first_tile + t;
    kp_tile[o] = t;
    kp_xy[2 * o] = (double)e[0];
    kp_xy[2 * o + 1] = (double)e[1];
    const size_t g = (size_t)info[1] + o;
    pts_out[2 * g] = (double)e[0] + (double)((tile % ntx) * tile_size);
    pts_out[2 * g + 1] = (double)e[1] + (double)((tile / ntx) * tile_size);
    resp_out[g] = e[2];
}

} // namespace

// ---- host side ----------------------------------------------------------------------------------------------------------
namespace {

// scratch of the detector + selection for nt tiles of interior Pi, laid out in one buffer
struct FastScratch {
    size_t npx, map, bytes;
    int nch;
    int *raw, *nms, *chunk_hist, *tile_hist, *chunk_base, *cut;
    unsigned long long* keys;
    static size_t size(int nt, int Pi, int& nch_out)
    {
        const size_t npx = (size_t)Pi * Pi, map = (size_t)nt * npx * sizeof(int);
        nch_out = (int)((npx + KC_CH - 1) / KC_CH);
        const size_t b_ch = (size_t)nt * nch_out * 256 * sizeof(int), b_keys = (size_t)nt * KS_CAP * sizeof(unsigned long long),
                     b_th = (size_t)nt * 256 * sizeof(int), b_cb = (size_t)nt * nch_out * 2 * sizeof(int),
                     b_cut = (size_t)nt * 4 * sizeof(int);
        return ma_align_up(2 * map, 16) + b_keys + b_ch + b_th + b_cb + b_cut + 64;
    }
    void place(void* buf, int nt, int Pi)
    {
        npx = (size_t)Pi * Pi; map = (size_t)nt * npx * sizeof(int);
        bytes = size(nt, Pi, nch);
        raw = (int*)buf;
        nms = raw + (size_t)nt * npx;
        keys = (unsigned long long*)((char*)buf + ma_align_up(2 * map, 16));
        chunk_hist = (int*)(keys + (size_t)nt * KS_CAP);
        tile_hist = chunk_hist + (size_t)nt * nch * 256;
        chunk_base = tile_hist + (size_t)nt * 256;
        cut = chunk_base + (size_t)nt * nch * 2;
    }
};

// detector + selection of nt tiles, enqueue only: kp_out (nt x limit x 3) and counts (nt) on the device
int fast_select_enqueue(ma_ctx* ctx, const uint8_t* tiles, int nt, int P, int margin, int threshold, int limit, FastScratch& fs,
                        int* kp_out, int* counts)
{
    const int Pi = P - 2 * margin;
    MaProfScope ps(ctx, MA_K_OTHER, (double)nt * Pi * Pi);
    const dim3 grid((Pi + 255) / 256, Pi, nt);
    MA_HIP(hipMemsetAsync(fs.tile_hist, 0, (size_t)nt * 256 * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(fast_score_kernel, grid, dim3(256), 0, ctx->stream, tiles, P, margin, threshold, fs.raw);
    hipLaunchKernelGGL(fast_nms_kernel, grid, dim3(256), 0, ctx->stream, (const int*)fs.raw, Pi, fs.nms);
    // selection: chunk histograms -> cut-off and chunk bases -> keys -> sort (one block per tile only for the sort)
    hipLaunchKernelGGL(kp_chunk_hist_kernel, dim3(fs.nch, nt), dim3(KC_T), 0, ctx->stream, (const int*)fs.nms, (int)fs.npx, fs.nch,
                       fs.chunk_hist, fs.tile_hist);
    hipLaunchKernelGGL(kp_cut_kernel, dim3(nt), dim3(KC_T), 0, ctx->stream, (const int*)fs.chunk_hist, (const int*)fs.tile_hist,
                       fs.nch, limit, fs.cut, fs.chunk_base);
    hipLaunchKernelGGL(kp_collect_kernel, dim3(fs.nch, nt), dim3(KC_T), 0, ctx->stream, (const int*)fs.nms, (int)fs.npx, fs.nch,
                       (const int*)fs.cut, (const int*)fs.chunk_base, fs.keys);
    hipLaunchKernelGGL(kp_sort_kernel, dim3(nt), dim3(KS_T), 0, ctx->stream, (const unsigned long long*)fs.keys, (const int*)fs.cut,
                       Pi, limit, kp_out, counts);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

// The small tables of the descriptor -- 8 (cos, sin) pairs, 25 (dy, dx) offsets, three centre-first half kernels (+ 64
// doubles: the smoothing kernels read a whole bucket of weights) -- as an immutable device copy, cached per device and
// content for the life of the process (a register() asks for the same tables a dozen times).
struct DaisyTables { const double* dev; size_t woff[3]; };
std::mutex g_dt_mu;
std::map<std::pair<int, std::vector<double>>, const double*> g_dt;

int daisy_tables(ma_ctx* ctx, const double* const* weights_host, const int* radii, const double* cos_sin_host,
                 const double* offs_host, DaisyTables& out)
{
    const size_t ntab = 16 + 50 + (size_t)(radii[0] + radii[1] + radii[2] + 3);
    std::vector<double> tab(ntab + 64, 0.0);
    for (int i = 0; i < 16; i++) tab[i] = cos_sin_host[i];
    for (int i = 0; i < 50; i++) tab[16 + i] = offs_host[i];
    size_t o = 66;
    for (int c = 0; c < 3; c++) {
        out.woff[c] = o;
        for (int j = 0; j <= radii[c]; j++) tab[o++] = weights_host[c][j];
    }
    std::lock_guard<std::mutex> lk(g_dt_mu);
    auto key = std::make_pair(ctx->device, tab);
    auto it = g_dt.find(key);
    if (it == g_dt.end()) {
        double* d = nullptr;
        MA_HIP(hipMalloc((void**)&d, tab.size() * sizeof(double)));
        MA_HIP(hipMemcpy(d, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
        it = g_dt.emplace(std::move(key), d).first;
    }
    out.dev = it->second;
    return MA_OK;
}

// layers -> three cubes -> descriptors for the keypoints (kp_tile, kp_xy) of nt tiles, enqueue only.  cubes: 4 x (nt x 8
// planes of P x P float32); rects (device): content rectangle per tile; span: the largest content extent (host).
// nkp: keypoints (an upper bound when nkp_dev holds the count on the device).
int daisy_enqueue(ma_ctx* ctx, const void* tiles, int dtype, int nt, int P, const DaisyTables& tb, const int* radii,
                  float* cubes, const DzRect* rects, DzSpan span, const int* kp_tile, const double* kp_xy, int nkp,
                  const int* nkp_dev, float* desc_out)
{
    const size_t cube = (size_t)nt * 8 * P * P;
    float* tmp = cubes + 3 * cube;
    MaProfScope ps(ctx, MA_K_OTHER, (double)nt * P * P);
    {
        const dim3 grid((std::min(P, span.w + 2) + 255) / 256, std::min(P, span.h + 2), nt);
        if (dtype == MA_U8) hipLaunchKernelGGL((daisy_layers_kernel<uint8_t>), grid, dim3(256), 0, ctx->stream, (const uint8_t*)tiles, P, tb.dev, tmp, (const int4*)rects);
        else hipLaunchKernelGGL((daisy_layers_kernel<float>), grid, dim3(256), 0, ctx->stream, (const float*)tiles, P, tb.dev, tmp, (const int4*)rects);
    }
    const float* src = tmp;   // the orientation layers; smoothed successively: cube c = G(inc_c) * cube c-1
    int h = 1, halo[3];
    for (int c = 0; c < 3; c++) {
        float* dst = cubes + c * cube;
        // scipy filters axis 1 (y) first, then axis 2 (x), each pass rounding to float32.  The intermediate of the two
        // passes lives in the next cube's slot (not yet written) or, for the last cube, in the layer buffer (done with)
        float* mid = c < 2 ? cubes + (c + 1) * cube : tmp;
        MA_TRY(smooth_planes(ctx->stream, src, mid, dst, P, nt * 8, tb.dev + tb.woff[c], radii[c], rects, h, span));
        h += radii[c];
        halo[c] = h;
        src = dst;
    }
    hipLaunchKernelGGL(daisy_sample_kernel, dim3(((size_t)nkp * 25 + 255) / 256), dim3(256), 0, ctx->stream, (const float*)cubes, cube,
                       P, kp_tile, kp_xy, tb.dev + 16, nkp, nkp_dev, desc_out, rects, halo[0], halo[1], halo[2]);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

} // namespace

namespace {
// content of window t of the row-major tile grid: the image rows [ty T - ov, ty T - ov + P) cut to the image, in window
// coordinates -- geometry only, computed where it is used (a copy from a host buffer tied every call to the ctx's one
// page-locked scratch, and so to a synchronisation before the next call could refill it)
__global__ void dz_window_rects_kernel(DzRect* __restrict__ rects, int nb, int first, int ntx, int tile, int overlap, int P, int H,
                                       int W)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nb) return;
    const int t = first + k, ty = t / ntx, tx = t - ty * ntx, wy = ty * tile - overlap, wx = tx * tile - overlap;
    DzRect r{max(0, -wy), min(P, H - wy), max(0, -wx), min(P, W - wx)};
    if (r.y1 <= r.y0 || r.x1 <= r.x0) r = DzRect{0, 0, 0, 0};
    rects[k] = r;
}
}  // namespace

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
    int nch = 0;
    const size_t need = FastScratch::size(nt, Pi, nch);
    MA_REQUIRE(nch <= 65535 * 32, "tile too large");
    MA_TRY(ma_ws_reserve(ctx, need + (size_t)nt * sizeof(int) + 64));
    MA_TRY(ma_pinned_reserve(ctx, (size_t)nt * sizeof(int)));
    FastScratch fs;
    fs.place(ctx->ws, nt, Pi);
    int* counts = (int*)((char*)ctx->ws + ma_align_up(need, 64));
    MA_TRY(fast_select_enqueue(ctx, tiles, nt, P, margin, threshold, limit, fs, kp_out, counts));
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
    DaisyTables tb;
    MA_TRY(daisy_tables(ctx, weights_host, radii, cos_sin_host, offs_host, tb));
    // workspace: cubes 0..2 and one temporary, each nt x 8 planes of P x P float32, then the content rectangles: arbitrary
    // tiles -- the whole tile counts as content
    const size_t cube = (size_t)nt * 8 * P * P;
    MA_TRY(ma_ws_reserve(ctx, 4 * cube * sizeof(float) + (size_t)nt * sizeof(DzRect)));
    MA_TRY(ma_pinned_reserve(ctx, (size_t)nt * sizeof(DzRect)));
    DzRect* h_rects = (DzRect*)ctx->pinned;
    for (int t = 0; t < nt; t++) h_rects[t] = DzRect{0, P, 0, P};
    DzRect* rects = (DzRect*)((float*)ctx->ws + 4 * cube);
    MA_HIP(hipMemcpyAsync(rects, h_rects, (size_t)nt * sizeof(DzRect), hipMemcpyHostToDevice, ctx->stream));
    return daisy_enqueue(ctx, tiles, dtype, nt, P, tb, radii, (float*)ctx->ws, rects, DzSpan{P, P}, kp_tile, kp_xy, nkp, nullptr,
                         desc_out);
}

static int feature_extract_impl(ma_ctx* ctx, const uint8_t* img, int H, int W, int tile, int overlap, int threshold, int limit,
                                const double* const* weights_host, const int* radii, const double* cos_sin_host,
                                const double* offs_host, size_t workspace_bytes, int capacity, float* desc_out, double* pts_out,
                                int* resp_out, int* n_out_host, bool wait)
{
    MA_REQUIRE(ctx && img && weights_host && radii && cos_sin_host && offs_host && desc_out && pts_out && resp_out && n_out_host,
               "NULL argument");
    MA_REQUIRE(H > 0 && W > 0 && tile > 0 && overlap >= 0 && tile <= 46340 && tile + 2 * overlap <= 65535, "bad tile geometry");
    MA_REQUIRE(limit >= 1 && limit <= KS_CAP, "limit must be in [1, 8192]");
    MA_REQUIRE(threshold >= 0 && threshold < 254, "bad threshold");
    MA_REQUIRE(radii[0] >= 0 && radii[1] >= 0 && radii[2] >= 0 && radii[0] + radii[1] + radii[2] < 4096, "bad radii");
    const int ntx = (W + tile - 1) / tile, nty = (H + tile - 1) / tile, P = tile + 2 * overlap, Pi = tile;
    const long long n_tiles = (long long)ntx * nty;
    MA_REQUIRE(n_tiles * limit <= capacity, "capacity must hold limit keypoints per tile");
    MA_HIP(hipSetDevice(ctx->device));
    *n_out_host = 0;
    DaisyTables tb;
    MA_TRY(daisy_tables(ctx, weights_host, radii, cos_sin_host, offs_host, tb));
    // tiles per batch: the DAISY cubes (128 P^2 bytes per tile) within the budget, 8 planes per tile within a grid dimension
    int nch = 0;
    const size_t cube_tile = (size_t)4 * 8 * P * P * sizeof(float);
    const size_t budget = workspace_bytes ? workspace_bytes : (size_t)8 << 30;
    const int step = (int)std::max<size_t>(1, std::min<size_t>({budget / cube_tile, (size_t)(65535 / 8), (size_t)n_tiles}));
    const size_t fast_bytes = FastScratch::size(step, Pi, nch);
    MA_REQUIRE(nch <= 65535 * 32, "tile too large");
    // one buffer: [tiles][kp_out][kp_tile][kp_xy][counts, base][rects][info][ detector scratch | cubes ]
    const size_t b_tiles = ma_align_up((size_t)step * P * P, 256), b_kp = ma_align_up((size_t)step * limit * 3 * sizeof(int), 256),
                 b_kt = ma_align_up((size_t)step * limit * sizeof(int), 256), b_xy = ma_align_up((size_t)step * limit * 16, 256),
                 b_cnt = ma_align_up((size_t)step * 2 * sizeof(int), 256), b_rect = ma_align_up((size_t)step * sizeof(DzRect), 256);
    const int n_batches = (int)((n_tiles + step - 1) / step);
    const size_t b_info = ma_align_up((size_t)(2 * n_batches + 1) * sizeof(int), 256);
    const size_t fixed = b_tiles + b_kp + b_kt + b_xy + b_cnt + b_rect + b_info;
    MA_TRY(ma_ws_reserve(ctx, fixed + std::max(fast_bytes, (size_t)step * cube_tile)));
    MA_TRY(ma_pinned_reserve(ctx, (size_t)step * sizeof(DzRect) * (size_t)n_batches + 64));
    char* ws = (char*)ctx->ws;
    uint8_t* tiles = (uint8_t*)ws;
    int* kp = (int*)(ws + b_tiles);
    int* kp_tile = (int*)(ws + b_tiles + b_kp);
    double* kp_xy = (double*)(ws + b_tiles + b_kp + b_kt);
    int* counts = (int*)(ws + b_tiles + b_kp + b_kt + b_xy);
    int* base = counts + step;
    DzRect* rects = (DzRect*)(ws + b_tiles + b_kp + b_kt + b_xy + b_cnt);
    int* info = (int*)(ws + b_tiles + b_kp + b_kt + b_xy + b_cnt + b_rect);
    int* total = info + 2 * n_batches;
    char* big = ws + fixed;
    MA_HIP(hipMemsetAsync(total, 0, sizeof(int), ctx->stream));
    for (int bi = 0; bi < n_batches; bi++) {
        const int first = bi * step, nb = (int)std::min<long long>(step, n_tiles - first);
        // the largest content rectangle of the batch (the launches are sized from it); the rectangles themselves: on the device
        DzSpan span{0, 0};
        for (int k = 0; k < nb; k++) {
            const int t = first + k, ty = t / ntx, tx = t - ty * ntx, wy = ty * tile - overlap, wx = tx * tile - overlap;
            const int y0 = std::max(0, -wy), y1 = std::min(P, H - wy), x0 = std::max(0, -wx), x1 = std::min(P, W - wx);
            if (y1 > y0 && x1 > x0) { span.h = std::max(span.h, y1 - y0); span.w = std::max(span.w, x1 - x0); }
        }
        hipLaunchKernelGGL(dz_window_rects_kernel, dim3((nb + 255) / 256), dim3(256), 0, ctx->stream, rects, nb, first, ntx, tile,
                           overlap, P, H, W);
        {
            MaProfScope ps(ctx, MA_K_OTHER, (double)nb * P * P);
            hipLaunchKernelGGL(cut_tiles_kernel, dim3((P + 255) / 256, P, nb), dim3(256), 0, ctx->stream, img, H, W, tile, overlap, ntx,
                               P, first, tiles);
        }
        FastScratch fs;
        fs.place(big, nb, Pi);
        MA_TRY(fast_select_enqueue(ctx, tiles, nb, P, overlap, threshold, limit, fs, kp, counts));
        hipLaunchKernelGGL(kp_offsets_kernel, dim3(1), dim3(64), 0, ctx->stream, (const int*)counts, nb, base, info + 2 * bi, total);
        hipLaunchKernelGGL(kp_compact_kernel, dim3((limit + 255) / 256, nb), dim3(256), 0, ctx->stream, (const int*)kp,
                           (const int*)counts, (const int*)base, (const int*)(info + 2 * bi), limit, tile, ntx, first, kp_tile,
                           kp_xy, pts_out, resp_out);
        MA_HIP(hipGetLastError());
        // the descriptors of the batch land behind those of the batches before it: the sample kernel takes the count and the
        // offset from the device (info[0], info[1])
        MA_TRY(daisy_enqueue(ctx, tiles, MA_U8, nb, P, tb, radii, (float*)big, rects, span, kp_tile, kp_xy, nb * limit,
                             info + 2 * bi, desc_out));
    }
    if (!wait) {      // n_out_host is page-locked: the count lands there in stream order
        MA_HIP(hipMemcpyAsync(n_out_host, total, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        return MA_OK;
    }
    MA_HIP(hipMemcpyAsync(ctx->pinned, total, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MA_HIP(hipStreamSynchronize(ctx->stream));
    *n_out_host = *(const int*)ctx->pinned;
    return MA_OK;
}

int ma_feature_extract(ma_ctx* ctx, const uint8_t* img, int H, int W, int tile, int overlap, int threshold, int limit,
                       const double* const* weights_host, const int* radii, const double* cos_sin_host, const double* offs_host,
                       size_t workspace_bytes, int capacity, float* desc_out, double* pts_out, int* resp_out, int* n_out_host)
{
    return feature_extract_impl(ctx, img, H, W, tile, overlap, threshold, limit, weights_host, radii, cos_sin_host, offs_host,
                                workspace_bytes, capacity, desc_out, pts_out, resp_out, n_out_host, true);
}

int ma_feature_extract_enqueue(ma_ctx* ctx, const uint8_t* img, int H, int W, int tile, int overlap, int threshold, int limit,
                               const double* const* weights_host, const int* radii, const double* cos_sin_host,
                               const double* offs_host, size_t workspace_bytes, int capacity, float* desc_out, double* pts_out,
                               int* resp_out, int* n_out_pinned)
{
    return feature_extract_impl(ctx, img, H, W, tile, overlap, threshold, limit, weights_host, radii, cos_sin_host, offs_host,
                                workspace_bytes, capacity, desc_out, pts_out, resp_out, n_out_pinned, false);
}

} // extern "C"
