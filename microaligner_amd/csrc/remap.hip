// Backward bilinear remap (cv2.remap INTER_LINEAR, BORDER_CONSTANT 0) and the tiled callers:
//   Warper.warp()                       microaligner/optflow_reg/warper.py:37-76
//   merge_two_flows / _merge_flow_in_tiles   optflow_reg/optflow_registrator.py:37-47,217-233
// Semantics: SURVEY.md Appendix A.2 -- coordinates quantised to 1/32 px with round-half-even,
// u8 uses the 15-bit fixed-point table, u16/f32 use float weights summed left to right.
// HBM-bound gathers, rows coalesced along x; the tiled kernels give a thread 8 rows and issue all their loads first.
#include "ma_internal.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Tap {
    int sx, sy;   // integer source coordinate of the top-left tap
    int fx, fy;   // 5-bit fractions
};

__device__ __forceinline__ short d_sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

__device__ __forceinline__ Tap quantise(float mx, float my)
{
    int sxq = d_cvround(mx * 32.f), syq = d_cvround(my * 32.f);
    Tap t;
    t.fx = sxq & 31; t.fy = syq & 31;
    t.sx = d_sat_short(sxq >> 5); t.sy = d_sat_short(syq >> 5);
    return t;
}

// 15-bit fixed-point bilinear weights of OpenCV's BilinearTab_i (A.2), including the
// [32767,0,0,1] entry that the table's sum fix-up produces at zero fraction.
__device__ __forceinline__ void weights_i(int fx, int fy, int w[4])
{
    if ((fx | fy) == 0) { w[0] = 32767; w[1] = 0; w[2] = 0; w[3] = 1; return; }
    w[0] = (32 - fy) * (32 - fx) * 32; w[1] = (32 - fy) * fx * 32;
    w[2] = fy * (32 - fx) * 32;        w[3] = fy * fx * 32;
}
__device__ __forceinline__ void weights_f(int fx, int fy, float w[4])
{
    // products of the exact 1-D weights (1 - f/32, f/32): exact in float
    const float s = 1.f / 32.f;
    float x1 = fx * s, x0 = 1.f - x1, y1 = fy * s, y0 = 1.f - y1;
    w[0] = y0 * x0; w[1] = y0 * x1; w[2] = y1 * x0; w[3] = y1 * x1;
}

template <typename T> struct Interp;
template <> struct Interp<uint8_t> {
    __device__ static uint8_t run(uint8_t v0, uint8_t v1, uint8_t v2, uint8_t v3, int fx, int fy)
    {
        int w[4];
        weights_i(fx, fy, w);
        int acc = v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
        return (uint8_t)d_clamp((acc + (1 << 14)) >> 15, 0, 255);
    }
};
template <> struct Interp<uint16_t> {
    __device__ static uint16_t run(uint16_t v0, uint16_t v1, uint16_t v2, uint16_t v3, int fx, int fy)
    {
        float w[4];
        weights_f(fx, fy, w);
        float acc = (float)v0 * w[0] + (float)v1 * w[1] + (float)v2 * w[2] + (float)v3 * w[3];
        return (uint16_t)d_clamp(d_cvround(acc), 0, 65535);
    }
};
template <> struct Interp<float> {
    __device__ static float run(float v0, float v1, float v2, float v3, int fx, int fy)
    {
        float w[4];
        weights_f(fx, fy, w);
        return v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
    }
};

// ---- generic cv2.remap ------------------------------------------------------------------------
template <typename T, int CN>
__global__ __launch_bounds__(256) void remap_kernel(const T* __restrict__ src, int sh, int sw,
                                                    const float2* __restrict__ map, int dh, int dw,
                                                    T* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    float2 m = map[(size_t)y * dw + x];
    Tap t = quantise(m.x, m.y);
    T out[CN];
    if (t.sx >= sw || t.sx + 1 < 0 || t.sy >= sh || t.sy + 1 < 0) {
#pragma unroll
        for (int k = 0; k < CN; k++) out[k] = 0;
    } else {
        const bool x0 = t.sx >= 0, x1 = t.sx + 1 < sw, y0 = t.sy >= 0, y1 = t.sy + 1 < sh;
#pragma unroll
        for (int k = 0; k < CN; k++) {
            T v0 = (x0 && y0) ? src[((size_t)t.sy * sw + t.sx) * CN + k] : (T)0;
            T v1 = (x1 && y0) ? src[((size_t)t.sy * sw + t.sx + 1) * CN + k] : (T)0;
            T v2 = (x0 && y1) ? src[((size_t)(t.sy + 1) * sw + t.sx) * CN + k] : (T)0;
            T v3 = (x1 && y1) ? src[((size_t)(t.sy + 1) * sw + t.sx + 1) * CN + k] : (T)0;
            out[k] = Interp<T>::run(v0, v1, v2, v3, t.fx, t.fy);
        }
    }
#pragma unroll
    for (int k = 0; k < CN; k++) dst[((size_t)y * dw + x) * CN + k] = out[k];
}

// ---- cv2.warpAffine(src, M, dsize), INTER_LINEAR, BORDER_CONSTANT 0 (feature_registrator.py:130) ----------
// WarpAffineInvoker's coordinates: 10-bit fixed point, adelta/bdelta per column, X0/Y0 per row (+16 = half of
// 1/32 px), then >> 5 leaves 5 fractional bits for the bilinear tables shared with remap.  Mi: the INVERTED matrix.
struct AffineCv { double m[6]; };
__device__ __forceinline__ int d_sat_int(double v)
{
    if (!(v > -2147483648.0)) return INT_MIN;
    if (!(v < 2147483647.0)) return INT_MAX;
    return __double2int_rn(v);
}
template <typename T>
__global__ __launch_bounds__(256) void warp_affine_cv_kernel(const T* __restrict__ src, int sh, int sw, AffineCv Mi,
                                                             int dh, int dw, T* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const int X0 = d_sat_int((Mi.m[1] * y + Mi.m[2]) * 1024.0) + 16, Y0 = d_sat_int((Mi.m[4] * y + Mi.m[5]) * 1024.0) + 16;
    const int X = (X0 + d_sat_int(Mi.m[0] * x * 1024.0)) >> 5, Y = (Y0 + d_sat_int(Mi.m[3] * x * 1024.0)) >> 5;
    Tap t;
    t.fx = X & 31; t.fy = Y & 31;
    t.sx = d_sat_short(X >> 5); t.sy = d_sat_short(Y >> 5);
    T out = 0;
    if (!(t.sx >= sw || t.sx + 1 < 0 || t.sy >= sh || t.sy + 1 < 0)) {
        const bool x0 = t.sx >= 0, x1 = t.sx + 1 < sw, y0 = t.sy >= 0, y1 = t.sy + 1 < sh;
        T v0 = (x0 && y0) ? src[(size_t)t.sy * sw + t.sx] : (T)0;
        T v1 = (x1 && y0) ? src[(size_t)t.sy * sw + t.sx + 1] : (T)0;
        T v2 = (x0 && y1) ? src[(size_t)(t.sy + 1) * sw + t.sx] : (T)0;
        T v3 = (x1 && y1) ? src[(size_t)(t.sy + 1) * sw + t.sx + 1] : (T)0;
        out = Interp<T>::run(v0, v1, v2, v3, t.fx, t.fy);
    }
    dst[(size_t)y * dw + x] = out;
}

// ---- Warper.warp(): window-local map = float(x_local) - flow, window-local constant border ----
// A tap contributes iff it lies inside the window [0,P) AND inside the image (the window is the
// zero-padded crop slicer.py builds); both cases read as 0, exactly what cv2.remap sees.
// Nearly every pixel has all four taps inside both: one compound test, one 32-bit element index and three constant
// offsets from it (IDX32: the image has fewer than 2^31 elements, uniform per launch); the pixels at window and image
// borders take the tap-by-tap form.
template <typename T>
__device__ __forceinline__ T warp_tiled_px(const T* __restrict__ img, const MaTiling& g, float2 f, int x, int y, int oy,
                                           int ox)
{
    const int lx = x - ox, ly = y - oy;
    // warper.py:57-59: float32(float64(-flow) + arange) == the correctly rounded lx - flow
    Tap t = quantise((float)lx - f.x, (float)ly - f.y);
    T res = 0;
    if (!(t.sx >= g.Pw || t.sx + 1 < 0 || t.sy >= g.Ph || t.sy + 1 < 0)) {
        T v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int sx = t.sx + (k & 1), sy = t.sy + (k >> 1);
            int jx = ox + sx, jy = oy + sy;
            bool ok = sx >= 0 && sx < g.Pw && sy >= 0 && sy < g.Ph && (unsigned)jx < (unsigned)g.W &&
                      (unsigned)jy < (unsigned)g.H;
            v[k] = ok ? img[(size_t)jy * g.W + jx] : (T)0;
        }
        res = Interp<T>::run(v[0], v[1], v[2], v[3], t.fx, t.fy);
    }
    return res;
}

// R rows of one column.  Nearly every thread has all four taps of all its rows inside both the window and the image: then
// (a wave-wide vote, so that the code stays straight-line and the loads of all rows are issued before the first result is
// computed) a tap is one 32-bit element index and three constant offsets from it, with no per-tap test (IDX32: the image
// has fewer than 2^31 elements, uniform per launch); waves that touch a window or image border take the tap-by-tap form.
template <typename T, bool IDX32, int R>
__device__ __forceinline__ void warp_tiled_rows(const T* __restrict__ img, const MaTiling& g, const float2 (&f)[R], int x,
                                                const int (&y)[R], const int (&oy)[R], int ox, T (&res)[R])
{
    Tap t[R];
    bool inside = true;
#pragma unroll
    for (int r = 0; r < R; r++) {
        t[r] = quantise((float)(x - ox) - f[r].x, (float)(y[r] - oy[r]) - f[r].y);
        inside = inside && (unsigned)t[r].sx < (unsigned)(g.Pw - 1) && (unsigned)t[r].sy < (unsigned)(g.Ph - 1) &&
                 (unsigned)(ox + t[r].sx) < (unsigned)(g.W - 1) && (unsigned)(oy[r] + t[r].sy) < (unsigned)(g.H - 1);
    }
    if (__all(inside)) {
        T v[R][4];
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (IDX32) {
                const unsigned p = (unsigned)(oy[r] + t[r].sy) * (unsigned)g.W + (unsigned)(ox + t[r].sx);
                v[r][0] = img[p]; v[r][1] = img[p + 1]; v[r][2] = img[p + (unsigned)g.W]; v[r][3] = img[p + (unsigned)g.W + 1];
            } else {
                const T* q = img + (size_t)(oy[r] + t[r].sy) * g.W + (ox + t[r].sx);
                v[r][0] = q[0]; v[r][1] = q[1]; v[r][2] = q[g.W]; v[r][3] = q[g.W + 1];
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) res[r] = Interp<T>::run(v[r][0], v[r][1], v[r][2], v[r][3], t[r].fx, t[r].fy);
    } else {
#pragma unroll
        for (int r = 0; r < R; r++) res[r] = warp_tiled_px<T>(img, g, f[r], x, y[r], oy[r], ox);
    }
}

// Window origin along x of column `x` (ox = (x / T) T - ov) without a division per lane: the 64 columns of a wave lie in at
// most two windows when T >= 64 -- the wave's first column decides (a scalar division), the columns at or beyond the next
// window's first take that one.
__device__ __forceinline__ int warp_window_origin_x(int x, const MaTiling& g)
{
    if (g.T <= 0) return 0;
    if (g.T < 64) return (x / g.T) * g.T - g.ov;
    const int x_first = __builtin_amdgcn_readfirstlane(x - (int)(threadIdx.x & 63));
    const int t0 = x_first / g.T, next = (t0 + 1) * g.T;
    return (x >= next ? next : t0 * g.T) - g.ov;
}
// Window origins along y of the rows y0 .. y0 + R of a block: one division, the rows at or beyond the next window's first
// row take that one (T >= R; smaller tiles divide per row)
struct WarpRowsY {
    int t0T, next, T, ov;
    __device__ __forceinline__ WarpRowsY(int y0, const MaTiling& g) : T(g.T), ov(g.ov)
    {
        const int t0 = g.T > 0 ? y0 / g.T : 0;
        t0T = t0 * g.T; next = t0T + g.T;
    }
    __device__ __forceinline__ int origin(int y) const
    {
        if (T <= 0) return 0;
        if (T < 16) return (y / T) * T - ov;
        return (y >= next ? next : t0T) - ov;
    }
};

// floats <-> unsigned keys whose integer order is the float order (so atomicMax works for any sign); every NaN maps
// to the largest key, so it survives the atomic reduction (numpy's .max() propagates NaN)
__device__ __forceinline__ unsigned f2key(float f)
{
    if (f != f) return 0xffffffffu;
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// NaN-propagating maximum (fmaxf drops NaNs)
__device__ __forceinline__ float d_max_nan(float m, float v) { return (m != m || v != v) ? NAN : fmaxf(m, v); }
// Segment of coordinate v along an axis cut at the window borders k*T - ov and k*T + ov (T > 2*ov):
// zone_k = [k*T - ov, k*T + ov) has index 2k, core_k = [k*T + ov, (k+1)*T - ov) index 2k+1; window t is the union of
// segments 2t, 2t+1, 2t+2 (see cell_max_kernel).
__device__ __forceinline__ int d_segment(int v, int T, int ov)
{
    const int k = (v + ov) / T;
    return 2 * k + ((v + ov - k * T) >= 2 * ov ? 1 : 0);
}

// WARP_ROWS rows per thread: the flow loads of all rows are issued before the first gather, the gathers of all
// rows before the first store -- 8 rows in flight per thread run 1.4x faster than one (measured, profiles/r01_notes.md)
constexpr int WARP_ROWS = 8;
constexpr int CELL_REPLICAS = MA_FLOW_CELL_REPLICAS;    // the public header sizes the caller's buffer from the same constant; copies of the flow cell-maxima array (ma_warp_tiled_flowcells); 8 / 32 / 128 copies: warp 0.39 ms each, merge 0.26 / 0.29 / 0.36
// MM: also reduce (min, max) of the block's output pixels into part[2 * block] (input conditioning of a following
// dog(): the consumer then skips its own pass over the image)
// cellkeys (may be NULL; needs T > 2*ov > 0): also fold the maximum of both FLOW components over the cells the window
// borders cut the image into, cellkeys[segy * nsegx + segx] (zero-initialised keys, f2key order).  Every flow that
// the registration merges passes through this kernel first (the pre-warp of a level reads the accumulated flow, the
// gate's warp reads the level's flow), so _merge_flow_in_tiles' per-window .max() tests (optflow_registrator.py:
// 38-42) need no pass of their own over the two flows.
template <typename T, bool MM, bool IDX32>
__global__ __launch_bounds__(256) void warp_tiled_kernel(const T* __restrict__ img, MaTiling g,
                                                         const float2* __restrict__ flow, T* __restrict__ out,
                                                         float* __restrict__ part, unsigned* __restrict__ cellkeys,
                                                         int nsegx, int nsegy)
{
    constexpr int WR = WARP_ROWS;
    const int x = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * WR;
    const bool xin = x < g.W;
    if (!MM && !cellkeys && !xin) return;
    float lo = INFINITY, hi = -INFINITY;
    unsigned ck_row[WARP_ROWS];   // keys of this thread's flow values, row by row (0 = nothing)
    int ck_sx = -1;               // x segment of this thread's column (-1: outside the image)
    if (cellkeys) {
#pragma unroll
        for (int r = 0; r < WARP_ROWS; r++) ck_row[r] = 0u;
    }
    const int ox = warp_window_origin_x(min(x, g.W - 1), g);      // (all lanes: the wave's first lane decides)
    const WarpRowsY rows(y0, g);
    if (xin) {
        float2 f[WR];
#pragma unroll
        for (int r = 0; r < WR; r++) f[r] = flow[(size_t)min(y0 + r, g.H - 1) * g.W + x];
        if (cellkeys) {
#pragma unroll
            for (int r = 0; r < WR; r++) ck_row[r] = y0 + r < g.H ? f2key(d_max_nan(f[r].x, f[r].y)) : 0u;
            ck_sx = d_segment(x, g.T, g.ov);
        }
        T res[WR];
        int ys[WR], oys[WR];
#pragma unroll
        for (int r = 0; r < WR; r++) { ys[r] = min(y0 + r, g.H - 1); oys[r] = rows.origin(y0 + r); }
        warp_tiled_rows<T, IDX32, WR>(img, g, f, x, ys, oys, ox, res);
#pragma unroll
        for (int r = 0; r < WR; r++)
            if (y0 + r < g.H) {
                out[(size_t)(y0 + r) * g.W + x] = res[r];
                if (MM) { lo = fminf(lo, (float)res[r]); hi = fmaxf(hi, (float)res[r]); }
            }
    }
    if (cellkeys) {
        // all 64 lanes are here.  One atomic per wave and (x segment, y segment) pair: the lanes of a wave hold 64
        // consecutive columns, i.e. one x segment, rarely two or three; the block's 8 rows one y segment, rarely two.
        // (Per-lane atomics on the few hundred cell addresses serialise in L2: 4.5 ms instead of 0.3 per launch.)
        const int sy_first = d_segment(y0, g.T, g.ov), sy_last = d_segment(min(y0 + WR - 1, g.H - 1), g.T, g.ov);
        for (int sy = sy_first; sy <= sy_last; sy++) {
            unsigned k = 0u;
#pragma unroll
            for (int r = 0; r < WR; r++)
                if (sy_first == sy_last || d_segment(min(y0 + r, g.H - 1), g.T, g.ov) == sy) k = max(k, ck_row[r]);
            unsigned long long todo = __ballot(ck_sx >= 0);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int seg = __shfl(ck_sx, leader);
                const bool mine = ck_sx == seg;
                unsigned v = mine ? k : 0u;
                for (int off = 32; off > 0; off >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, off));
                // ~1250 waves share a cell: the atomics go to one of CELL_REPLICAS copies of the cell array, picked by the
                // block row, so that they spread over L2 channels instead of queueing on one address
                if ((int)(threadIdx.x & 63) == leader && v)
                    atomicMax(&cellkeys[((size_t)(blockIdx.y % CELL_REPLICAS) * nsegy + sy) * nsegx + seg], v);
                todo &= ~__ballot(mine);
            }
        }
    }
    if (MM) d_block_minmax(lo, hi, part + 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x));
}

// the rows [y_begin, y_end) of warp_tiled_kernel's result (bands of the page-warp driver, ma_warp_pages_host)
template <typename T, bool IDX32>
__global__ __launch_bounds__(256) void warp_band_kernel(const T* __restrict__ img, MaTiling g,
                                                        const float2* __restrict__ flow, T* __restrict__ out,
                                                        int y_begin, int y_end)
{
    constexpr int WR = WARP_ROWS;
    const int x = blockIdx.x * 256 + threadIdx.x, y0 = y_begin + blockIdx.y * WR;
    const int ox = warp_window_origin_x(min(x, g.W - 1), g);
    if (x >= g.W) return;
    const WarpRowsY rows(y0, g);
    float2 f[WR];
#pragma unroll
    for (int r = 0; r < WR; r++) f[r] = flow[(size_t)min(y0 + r, y_end - 1) * g.W + x];
    T res[WR];
    int ys[WR], oys[WR];
#pragma unroll
    for (int r = 0; r < WR; r++) { ys[r] = min(y0 + r, y_end - 1); oys[r] = rows.origin(ys[r]); }
    warp_tiled_rows<T, IDX32, WR>(img, g, f, x, ys, oys, ox, res);
#pragma unroll
    for (int r = 0; r < WR; r++)
        if (y0 + r < y_end) out[(size_t)(y0 + r) * g.W + x] = res[r];
}

// ---- flow merge ---------------------------------------------------------------------------------
// per window: max over the zero-padded window of both flow components with numpy's .max() semantics: a NaN
// anywhere in the window makes the maximum NaN, `nan == 0` is False and the window takes the general remap branch
// (optflow_registrator.py:38-47).
// grid: (window, row band of WM_ROWS rows); maxkeys[2*window + {0,1}] must be zero-initialised
// (key 0 decodes to a NaN that loses against every real value's key).
constexpr int WM_ROWS = 64;
__global__ __launch_bounds__(256) void window_max_kernel(const float2* __restrict__ f1, const float2* __restrict__ f2,
                                                         MaTiling g, unsigned* __restrict__ maxkeys)
{
    const int widx = blockIdx.x;
    int oy = 0, ox = 0;
    if (g.T > 0) { int ty = widx / g.ntx, tx = widx - ty * g.ntx; oy = ty * g.T - g.ov; ox = tx * g.T - g.ov; }
    const int yb = min(oy + g.Ph, g.H), xa = max(ox, 0), xb = min(ox + g.Pw, g.W);
    const int ya = max(oy, 0) + blockIdx.y * WM_ROWS;
    if (ya >= yb) return;
    const int ye = min(ya + WM_ROWS, yb);
    const bool padded = oy < 0 || ox < 0 || oy + g.Ph > g.H || ox + g.Pw > g.W;
    const float init = padded ? 0.f : -INFINITY;  // zero padding takes part in numpy's .max()
    float m1 = init, m2 = init;
    for (int yy = ya + (threadIdx.x >> 6); yy < ye; yy += 4)
        for (int xx = xa + (threadIdx.x & 63); xx < xb; xx += 64) {
            float2 a = f1[(size_t)yy * g.W + xx], b = f2[(size_t)yy * g.W + xx];
            m1 = d_max_nan(m1, d_max_nan(a.x, a.y));
            m2 = d_max_nan(m2, d_max_nan(b.x, b.y));
        }
    for (int off = 32; off > 0; off >>= 1) {
        m1 = d_max_nan(m1, __shfl_down(m1, off));
        m2 = d_max_nan(m2, __shfl_down(m2, off));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&maxkeys[widx * 2], f2key(m1));
        atomicMax(&maxkeys[widx * 2 + 1], f2key(m2));
    }
}

// Window maxima from non-overlapping cells (T > 2*ov): along each axis the window borders k*T - ov and k*T + ov cut
// the image into segments  zone_k = [k*T - ov, k*T + ov)  (index 2k) and  core_k = [k*T + ov, (k+1)*T - ov)
// (index 2k+1); window t is the union of segments 2t, 2t+1, 2t+2, so every flow value is read once (the banded
// kernel above reads the overlaps 1.44x at tile 1000 / overlap 100).
// cell_max_kernel: grid (x segments, row bands of the y segments); cellkeys zero-initialised.
constexpr int CM_ROWS = 64;
__device__ __forceinline__ void seg_range(int s, int T, int ov, int len, int& a, int& b)
{
    const int k = s >> 1;
    if (s & 1) { a = k * T + ov; b = (k + 1) * T - ov; }
    else { a = k * T - ov; b = k * T + ov; }
    a = max(a, 0); b = min(b, len);
}
__global__ __launch_bounds__(256) void cell_max_kernel(const float2* __restrict__ f1, const float2* __restrict__ f2,
                                                       MaTiling g, int nsegx, unsigned* __restrict__ cellkeys)
{
    const int bz = (2 * g.ov + CM_ROWS - 1) / CM_ROWS, bc = (g.T - 2 * g.ov + CM_ROWS - 1) / CM_ROWS;
    const int per = blockIdx.y / (bz + bc), r = blockIdx.y - per * (bz + bc);
    const int segy = r < bz ? 2 * per : 2 * per + 1, band = r < bz ? r : r - bz;
    int ya, yb, xa, xb;
    seg_range(segy, g.T, g.ov, g.H, ya, yb);
    seg_range(blockIdx.x, g.T, g.ov, g.W, xa, xb);
    // zone 0 starts at -ov: bands are counted from the unclipped segment start so that they tile it exactly
    const int y0 = (segy & 1 ? (segy >> 1) * g.T + g.ov : (segy >> 1) * g.T - g.ov) + band * CM_ROWS;
    ya = max(ya, y0); yb = min(yb, y0 + CM_ROWS);
    if (ya >= yb || xa >= xb) return;
    float m1 = -INFINITY, m2 = -INFINITY;
    for (int yy = ya + (threadIdx.x >> 6); yy < yb; yy += 4)
        for (int xx = xa + (threadIdx.x & 63); xx < xb; xx += 64) {
            const float2 a = f1[(size_t)yy * g.W + xx], b = f2[(size_t)yy * g.W + xx];
            m1 = d_max_nan(m1, d_max_nan(a.x, a.y));
            m2 = d_max_nan(m2, d_max_nan(b.x, b.y));
        }
    for (int off = 32; off > 0; off >>= 1) {
        m1 = d_max_nan(m1, __shfl_down(m1, off));
        m2 = d_max_nan(m2, __shfl_down(m2, off));
    }
    if ((threadIdx.x & 63) == 0) {
        unsigned* c = cellkeys + ((size_t)segy * nsegx + blockIdx.x) * 2;
        atomicMax(&c[0], f2key(m1));
        atomicMax(&c[1], f2key(m2));
    }
}
// one thread per window: max over its 3 x 3 cells; zero padding takes part in numpy's .max()
// STRIDE = 2: cellkeys holds (flow1, flow2) pairs per cell (cell_max_kernel); STRIDE = 1: one array per flow (the
// by-product of warp_tiled_kernel)
template <int STRIDE>
__global__ void window_from_cells_kernel(const unsigned* __restrict__ cellkeys, const unsigned* __restrict__ cellkeys2,
                                         MaTiling g, int nsegx, unsigned* __restrict__ maxkeys)
{
    const int widx = blockIdx.x * blockDim.x + threadIdx.x;
    if (widx >= g.ntx * g.nty) return;
    const int ty = widx / g.ntx, tx = widx - ty * g.ntx;
    const int oy = ty * g.T - g.ov, ox = tx * g.T - g.ov;
    const bool padded = oy < 0 || ox < 0 || oy + g.Ph > g.H || ox + g.Pw > g.W;
    unsigned k1 = padded ? f2key(0.f) : 0u, k2 = k1;
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++) {
            const size_t cell = (size_t)(2 * ty + j) * nsegx + 2 * tx + i;
            if (STRIDE == 2) { k1 = max(k1, cellkeys[cell * 2]); k2 = max(k2, cellkeys[cell * 2 + 1]); }
            else {
                const size_t ncell = (size_t)nsegx * (2 * g.nty + 1);
                for (int rep = 0; rep < CELL_REPLICAS; rep++) {
                    k1 = max(k1, cellkeys[rep * ncell + cell]);
                    k2 = max(k2, cellkeys2[rep * ncell + cell]);
                }
            }
        }
    maxkeys[widx * 2] = k1;
    maxkeys[widx * 2 + 1] = k2;
}

constexpr int MERGE_ROWS = 8;
__global__ __launch_bounds__(256) void merge_flows_kernel(const float2* __restrict__ f1, const float2* __restrict__ f2,
                                                          MaTiling g, const unsigned* __restrict__ maxkeys,
                                                          float2* __restrict__ out)
{
    constexpr int MR = MERGE_ROWS;  // rows per thread, loads of all rows issued up front (as warp_tiled_kernel)
    const int x = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * MR;
    if (x >= g.W) return;
    const int tx = g.T > 0 ? x / g.T : 0;
    const int ox = g.T > 0 ? tx * g.T - g.ov : 0;
    float2 a[MR], res[MR];
#pragma unroll
    for (int r = 0; r < MR; r++) a[r] = f1[(size_t)min(y0 + r, g.H - 1) * g.W + x];
#pragma unroll
    for (int r = 0; r < MR; r++) {
        const int y = min(y0 + r, g.H - 1);
        int oy = 0, widx = 0;
        if (g.T > 0) {
            const int ty = y / g.T;
            widx = ty * g.ntx + tx;
            oy = ty * g.T - g.ov;
        }
        const size_t p = (size_t)y * g.W + x;
        if (key2f(maxkeys[widx * 2]) == 0.f) res[r] = f2[p];               // flow1.max() == 0 -> flow2
        else if (key2f(maxkeys[widx * 2 + 1]) == 0.f) res[r] = a[r];       // flow2.max() == 0 -> flow1
        else {
            // flow1 + cv.remap(flow2, -flow1): the map is -flow1 itself (absolute window coordinates, quirk Q1)
            Tap t = quantise(-a[r].x, -a[r].y);
            float2 s = make_float2(0.f, 0.f);
            if (!(t.sx >= g.Pw || t.sx + 1 < 0 || t.sy >= g.Ph || t.sy + 1 < 0)) {
                float2 v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int sx = t.sx + (k & 1), sy = t.sy + (k >> 1);
                    int ix = ox + sx, iy = oy + sy;
                    bool ok = sx >= 0 && sx < g.Pw && sy >= 0 && sy < g.Ph && (unsigned)ix < (unsigned)g.W &&
                              (unsigned)iy < (unsigned)g.H;
                    v[k] = ok ? f2[(size_t)iy * g.W + ix] : make_float2(0.f, 0.f);
                }
                s.x = Interp<float>::run(v[0].x, v[1].x, v[2].x, v[3].x, t.fx, t.fy);
                s.y = Interp<float>::run(v[0].y, v[1].y, v[2].y, v[3].y, t.fx, t.fy);
            }
            res[r] = make_float2(a[r].x + s.x, a[r].y + s.y);
        }
    }
#pragma unroll
    for (int r = 0; r < MR; r++)
        if (y0 + r < g.H) out[(size_t)(y0 + r) * g.W + x] = res[r];
}

} // namespace

extern "C" {

int ma_remap_bilinear(ma_ctx* ctx, const void* src, int dtype, int cn, int sh, int sw, const float* map_xy, int dh,
                      int dw, void* dst)
{
    MA_REQUIRE(ctx && src && map_xy && dst, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(cn == 1 || cn == 2, "cn must be 1 or 2");
    MA_REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0, "empty image");
    MA_REQUIRE(sh < 32767 && sw < 32767 && dh < 32767 && dw < 32767, "cv2.remap requires all dimensions < 32767");
    MA_HIP(hipSetDevice(ctx->device));
    MaProfScope ps(ctx, MA_K_WARP, (double)dh * dw);
    dim3 grid((dw + 255) / 256, dh), block(256);
    const float2* map = (const float2*)map_xy;
#define LAUNCH(T, CN) hipLaunchKernelGGL((remap_kernel<T, CN>), grid, block, 0, ctx->stream, (const T*)src, sh, sw, map, dh, dw, (T*)dst)
    if (dtype == MA_U8) { if (cn == 1) LAUNCH(uint8_t, 1); else LAUNCH(uint8_t, 2); }
    else if (dtype == MA_U16) { if (cn == 1) LAUNCH(uint16_t, 1); else LAUNCH(uint16_t, 2); }
    else { if (cn == 1) LAUNCH(float, 1); else LAUNCH(float, 2); }
#undef LAUNCH
    MA_HIP(hipGetLastError());
    return MA_OK;
}

int ma_warp_affine_cv(ma_ctx* ctx, const void* src, int dtype, int sh, int sw, const double* m2x3_host, int dh, int dw,
                      void* dst)
{
    MA_REQUIRE(ctx && src && m2x3_host && dst, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0 && dh <= 65535, "bad image size");
    MA_HIP(hipSetDevice(ctx->device));
    // cv::warpAffine inverts the forward matrix in double before the fixed-point stage
    AffineCv A;
    double M[6];
    for (int i = 0; i < 6; i++) M[i] = m2x3_host[i];
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    for (int i = 0; i < 6; i++) A.m[i] = M[i];
    MaProfScope ps(ctx, MA_K_WARP, (double)dh * dw);
    dim3 grid((dw + 255) / 256, dh), block(256);
    if (dtype == MA_U8) hipLaunchKernelGGL((warp_affine_cv_kernel<uint8_t>), grid, block, 0, ctx->stream, (const uint8_t*)src, sh, sw, A, dh, dw, (uint8_t*)dst);
    else if (dtype == MA_U16) hipLaunchKernelGGL((warp_affine_cv_kernel<uint16_t>), grid, block, 0, ctx->stream, (const uint16_t*)src, sh, sw, A, dh, dw, (uint16_t*)dst);
    else hipLaunchKernelGGL((warp_affine_cv_kernel<float>), grid, block, 0, ctx->stream, (const float*)src, sh, sw, A, dh, dw, (float*)dst);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

static int warp_tiled_impl(ma_ctx* ctx, const void* img, int dtype, int H, int W, const float* flow, int tile, int overlap,
                           void* out, float* minmax_dev, unsigned* flow_cellkeys_dev = nullptr)
{
    MA_REQUIRE(ctx && img && flow && out, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(H > 0 && W > 0, "bad image size");
    MA_REQUIRE(H <= MA_GRID_Y_MAX * WARP_ROWS, "image too tall");
    MA_REQUIRE(tile >= 0 && overlap >= 0, "tile/overlap must be >= 0");
    MaTiling g = ma_make_tiling(H, W, tile, overlap);
    MA_REQUIRE(g.Ph < 32767 && g.Pw < 32767, "cv2.remap requires window dimensions < 32767");
    MA_HIP(hipSetDevice(ctx->device));
    dim3 grid((W + 255) / 256, (H + WARP_ROWS - 1) / WARP_ROWS), block(256);
    const size_t nblk = (size_t)grid.x * grid.y;
    float* part = nullptr;
    if (minmax_dev) {
        MA_TRY(ma_ws_reserve(ctx, nblk * 2 * sizeof(float)));
        part = (float*)ctx->ws;
    }
    const int nsegx = 2 * g.ntx + 1, nsegy = 2 * g.nty + 1;
    if (flow_cellkeys_dev) MA_REQUIRE(tile > 2 * overlap && overlap > 0, "flow cell maxima need tile > 2*overlap > 0");
    MaProfScope ps(ctx, MA_K_WARP, (double)H * W);
    if (flow_cellkeys_dev)
        MA_HIP(hipMemsetAsync(flow_cellkeys_dev, 0, (size_t)CELL_REPLICAS * nsegx * nsegy * sizeof(unsigned), ctx->stream));
    const float2* f = (const float2*)flow;
    unsigned* ck = flow_cellkeys_dev;
    const bool idx32 = (unsigned long long)H * (unsigned long long)W < (1ull << 31);
#define MA_WARP2(T, MMF, IX) hipLaunchKernelGGL((warp_tiled_kernel<T, MMF, IX>), grid, block, 0, ctx->stream, (const T*)img, g, f, (T*)out, part, ck, nsegx, nsegy)
#define MA_WARP(T) do { if (part) { if (idx32) MA_WARP2(T, true, true); else MA_WARP2(T, true, false); } \
                        else { if (idx32) MA_WARP2(T, false, true); else MA_WARP2(T, false, false); } } while (0)
    if (dtype == MA_U8) MA_WARP(uint8_t);
    else if (dtype == MA_U16) MA_WARP(uint16_t);
    else MA_WARP(float);
#undef MA_WARP
#undef MA_WARP2
    MA_HIP(hipGetLastError());
    if (part) MA_TRY(ma_launch_minmax_final(ctx, part, (int)nblk, minmax_dev));
    return MA_OK;
}

int ma_warp_tiled(ma_ctx* ctx, const void* img, int dtype, int H, int W, const float* flow, int tile, int overlap,
                  void* out)
{
    return warp_tiled_impl(ctx, img, dtype, H, W, flow, tile, overlap, out, nullptr);
}

int ma_warp_tiled_minmax(ma_ctx* ctx, const void* img, int dtype, int H, int W, const float* flow, int tile,
                         int overlap, void* out, float* minmax_dev)
{
    MA_REQUIRE(minmax_dev, "NULL argument");
    return warp_tiled_impl(ctx, img, dtype, H, W, flow, tile, overlap, out, minmax_dev);
}

int ma_warp_tiled_flowcells(ma_ctx* ctx, const void* img, int dtype, int H, int W, const float* flow, int tile,
                            int overlap, void* out, float* minmax_dev, unsigned* flow_cellkeys_dev)
{
    MA_REQUIRE(flow_cellkeys_dev, "NULL argument");
    return warp_tiled_impl(ctx, img, dtype, H, W, flow, tile, overlap, out, minmax_dev, flow_cellkeys_dev);
}

// ---- page-warp driver ---------------------------------------------------------------------
// warp_and_save_pages (microaligner/__main__.py:288-302): every channel / z page of a cycle is warped with the
// SAME flow.  The flow stays in HBM; pages stream through NS slots (a device buffer pair each) on the three engines
// of the context, so the upload of one piece, the kernel of the previous one and the download of the one before
// overlap.  The unit of the pipeline is a BAND of whole tile rows, not a page: an output pixel only reads source
// pixels of its own window (warper.py:29-76 cuts the windows before cv2.remap sees them), so the rows of tile row
// ty are complete once source rows < (ty + 1) * tile + overlap have arrived -- a single page (Warper.warp() of a
// host image, the reference's own per-page loop) overlaps its own upload, kernel and download, and the first and
// last page of a longer run lose only a band to filling and draining.
int ma_warp_pages_plan(int dtype, int H, int W, int tile, int overlap, size_t band_bytes, int* band_rows, int* n_bands)
{
    MA_REQUIRE(band_rows && n_bands, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(H > 0 && W > 0 && H <= MA_GRID_Y_MAX * WARP_ROWS, "bad size");
    MA_REQUIRE(tile >= 0 && overlap >= 0 && band_bytes > 0, "tile/overlap must be >= 0, the band size positive");
    const size_t rowb = (size_t)W * ma_esize(dtype);
    int rows = H;
    if (tile > 0) {
        const size_t tr = (band_bytes + (size_t)tile * rowb - 1) / ((size_t)tile * rowb);   // tile rows per band
        rows = (int)std::min<size_t>((size_t)H, tr * (size_t)tile);
    }
    *band_rows = rows;
    *n_bands = (H + rows - 1) / rows;
    return MA_OK;
}

int ma_warp_pages_host(ma_ctx* ctx, const void* const* pages_host, void* const* out_host, int n_pages, int dtype,
                       int H, int W, const float* flow, int tile, int overlap)
{
    MA_REQUIRE(ctx && pages_host && out_host && flow, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(H > 0 && W > 0 && H <= MA_GRID_Y_MAX * WARP_ROWS && n_pages >= 0, "bad size");
    MA_REQUIRE(tile >= 0 && overlap >= 0, "tile/overlap must be >= 0");
    MaTiling g = ma_make_tiling(H, W, tile, overlap);
    MA_REQUIRE(g.Ph < 32767 && g.Pw < 32767, "cv2.remap requires window dimensions < 32767");
    for (int i = 0; i < n_pages; i++) MA_REQUIRE(pages_host[i] && out_host[i], "NULL page pointer");
    if (n_pages == 0) return MA_OK;
    MA_HIP(hipSetDevice(ctx->device));
    // An upload thread copies band after band into input slot page % NS on the H2D stream, this thread launches the
    // band's warp on the ctx stream, a download thread copies its rows out on the D2H stream; events order the streams,
    // counters under one mutex order the threads.  Pageable pages (numpy arrays, rows of a memmapped TIFF) are staged
    // by the engines through page-locked chunks -- the runtime's own staging of pageable memory reached 14 GB/s per
    // direction here, the engines 2 - 3 x that (profiles/r04_notes.md).
    constexpr int NS = 3;
    const int ns = n_pages < NS ? n_pages : NS;
    const size_t esz = ma_esize(dtype), rowb = (size_t)W * esz;
    const size_t nb = (size_t)H * rowb;
    const size_t bytes = ma_align_up(nb, 256);
    // bands: whole tile rows, at least MA_OPT_WARP_BAND_BYTES each (32 MiB: a transfer below that no longer runs at the
    // link rate); an untiled warp (tile == 0: one window) is one band
    int band_rows = H, nband = 1;
    MA_TRY(ma_warp_pages_plan(dtype, H, W, tile, overlap, ctx->warp_band_bytes, &band_rows, &nband));
    const long long n_units = (long long)n_pages * nband;
    MA_TRY(ma_ws_reserve(ctx, bytes * 2 * ns));  // device buffers come from the context workspace
    void *din[NS], *dout[NS];
    std::vector<hipEvent_t> ev_up((size_t)ns * nband, nullptr), ev_k((size_t)ns * nband, nullptr);
    auto cleanup = [&]() {
        for (hipEvent_t e : ev_up) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_k) if (e) (void)hipEventDestroy(e);
    };
    for (int k = 0; k < ns; k++) {
        din[k] = (char*)ctx->ws + bytes * (2 * k);
        dout[k] = (char*)ctx->ws + bytes * (2 * k + 1);
    }
    // The slots live in the context workspace, which kernels enqueued earlier on the compute stream (a tiled Farneback,
    // a dog(), the NMI) may still be using: both transfer engines start behind everything the compute stream holds now
    // (ma_ws_reserve itself synchronises only when the workspace has to grow).
    {
        hipEvent_t ws_idle = ma_ctx_sync_event(ctx, MA_EV_WARP_PAGES);
        hipStream_t s_up = ma_engine_stream(ctx, MA_ENGINE_H2D), s_down = ma_engine_stream(ctx, MA_ENGINE_D2H);
        if (!ws_idle || !s_up || !s_down) return MA_EHIP;
        MA_HIP(hipEventRecord(ws_idle, ctx->stream));
        MA_HIP(hipStreamWaitEvent(s_up, ws_idle, 0));
        MA_HIP(hipStreamWaitEvent(s_down, ws_idle, 0));
    }
    for (size_t e = 0; e < ev_up.size(); e++)
        if (hipEventCreateWithFlags(&ev_up[e], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_k[e], hipEventDisableTiming) != hipSuccess) {
            cleanup();
            ma_set_error("hipEventCreate failed");
            return MA_EHIP;
        }
    // output rows of band b, and the source rows that must be resident before it runs
    auto band_begin = [&](int b) { return b * band_rows; };
    auto band_end = [&](int b) { return std::min(H, (b + 1) * band_rows); };
    auto src_end = [&](int b) { return b == nband - 1 ? H : std::min(H, (b + 1) * band_rows + g.ov); };
    // byte offsets at which the bands end: in the source (a band's window reaches `overlap` rows further) and in the result
    std::vector<size_t> cuts_src(nband), cuts_out(nband);
    for (int b = 0; b < nband; b++) {
        cuts_src[b] = (size_t)src_end(b) * rowb;
        cuts_out[b] = (size_t)band_end(b) * rowb;
    }
    const bool TRACE = getenv("MICROALIGNER_TRACE_PAGES") != nullptr;   // timeline of the three threads on stderr
    const auto T0 = std::chrono::steady_clock::now();
    auto now_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count(); };
    if (TRACE) fprintf(stderr, "[pages] %d pages, %d bands of %d rows\n", n_pages, nband, band_rows);
    std::mutex mu;
    std::condition_variable cv;
    long long uploaded = 0, launched = 0;   // in units (page * nband + band)
    int downloaded = 0;                     // in pages
    int failed = MA_OK;
    std::string what;
    auto fail = [&](int rc) {   // called with mu held
        if (failed == MA_OK) { failed = rc; what = ma_last_error(); }
        cv.notify_all();
    };
    // A page is ONE copy per direction (ma_engine_*_pieces): the staging of pageable memory keeps its chunks in flight
    // across the band boundaries, the bands only decide when the events are recorded and waited for.
    std::thread up([&]() {
        for (int i = 0; i < n_pages; i++) {
            const int k = i % ns;
            {   // slot k is free again once page i - ns has been downloaded
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return failed != MA_OK || downloaded > i - ns; });
                if (failed != MA_OK) return;
            }
            const int rc = ma_engine_h2d_pieces(ctx, MA_ENGINE_H2D, din[k], pages_host[i], nb, cuts_src.data(), nband, [&](int b) {
                const int r = ma_engine_record(ctx, MA_ENGINE_H2D, ev_up[(size_t)k * nband + b]);
                if (TRACE) fprintf(stderr, "[pages] %8.2f up   p%d b%d\n", now_ms(), i, b);
                std::lock_guard<std::mutex> lk(mu);
                if (r != MA_OK) return r;
                if (failed != MA_OK) return failed;
                uploaded = (long long)i * nband + b + 1;
                cv.notify_all();
                return (int)MA_OK;
            }, false);   // no wait at the page boundary: the next page's first chunk is staged under this page's last DMAs
            if (rc != MA_OK) {
                std::lock_guard<std::mutex> lk(mu);
                fail(rc);
                return;
            }
        }
        const int rc = ma_engine_sync(ctx, MA_ENGINE_H2D);
        if (rc != MA_OK) {
            std::lock_guard<std::mutex> lk(mu);
            fail(rc);
        }
    });
    std::thread down([&]() {
        for (int i = 0; i < n_pages; i++) {
            const int k = i % ns;
            const int rc = ma_engine_d2h_pieces(ctx, MA_ENGINE_D2H, out_host[i], dout[k], nb, cuts_out.data(), nband, [&](int b) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return failed != MA_OK || launched > (long long)i * nband + b; });
                    if (failed != MA_OK) return failed;
                }
                if (TRACE) fprintf(stderr, "[pages] %8.2f down p%d b%d (issued)\n", now_ms(), i, b);
                return ma_engine_wait(ctx, MA_ENGINE_D2H, ev_k[(size_t)k * nband + b]);
            });
            std::lock_guard<std::mutex> lk(mu);
            if (rc != MA_OK) { fail(rc); return; }
            if (TRACE) fprintf(stderr, "[pages] %8.2f down p%d complete\n", now_ms(), i);
            downloaded = i + 1;
            cv.notify_all();
        }
    });
    const float2* f = (const float2*)flow;
    const dim3 block(256);
    for (long long u = 0; u < n_units; u++) {
        const int i = (int)(u / nband), b = (int)(u % nband), k = i % ns;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return failed != MA_OK || uploaded > u; });
            if (failed != MA_OK) break;
        }
        const int y0 = band_begin(b), y1 = band_end(b);
        const dim3 grid((W + 255) / 256, (y1 - y0 + WARP_ROWS - 1) / WARP_ROWS);
        hipError_t e = hipStreamWaitEvent(ctx->stream, ev_up[(size_t)k * nband + b], 0);
        if (e == hipSuccess) {
#define MA_BAND(T) do { if (idx32) hipLaunchKernelGGL((warp_band_kernel<T, true>), grid, block, 0, ctx->stream, (const T*)din[k], g, f, (T*)dout[k], y0, y1); \
                        else hipLaunchKernelGGL((warp_band_kernel<T, false>), grid, block, 0, ctx->stream, (const T*)din[k], g, f, (T*)dout[k], y0, y1); } while (0)
            const bool idx32 = (unsigned long long)H * (unsigned long long)W < (1ull << 31);
            if (dtype == MA_U8) MA_BAND(uint8_t);
            else if (dtype == MA_U16) MA_BAND(uint16_t);
            else MA_BAND(float);
#undef MA_BAND
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(ev_k[(size_t)k * nband + b], ctx->stream);
        if (TRACE) fprintf(stderr, "[pages] %8.2f kern p%d b%d\n", now_ms(), i, b);
        std::lock_guard<std::mutex> lk(mu);
        if (e != hipSuccess) {
            ma_set_error("warp of page %d (rows %d..%d) failed: %s", i, y0, y1, hipGetErrorString(e));
            fail(MA_EHIP);
            break;
        }
        launched = u + 1;
        cv.notify_all();
    }
    up.join();
    down.join();
    if (TRACE) fprintf(stderr, "[pages] %8.2f joined\n", now_ms());
    // also when a thread gave up early: nothing of this call may still be reading the caller's pages or writing its
    // results once it has returned
    (void)ma_engine_sync(ctx, MA_ENGINE_H2D);
    (void)ma_engine_sync(ctx, MA_ENGINE_D2H);
    (void)hipStreamSynchronize(ctx->stream);
    cleanup();
    if (TRACE) fprintf(stderr, "[pages] %8.2f cleaned\n", now_ms());
    if (failed != MA_OK) {
        ma_set_error("%s", what.c_str());
        return failed;
    }
    return MA_OK;
}

int ma_merge_flows_tiled(ma_ctx* ctx, const float* flow1, const float* flow2, int H, int W, int tile, int overlap,
                         float* out)
{
    MA_REQUIRE(ctx && flow1 && flow2 && out, "NULL argument");
    MA_REQUIRE(H > 0 && W > 0 && H <= MA_GRID_Y_MAX * MERGE_ROWS, "bad image size");
    MA_REQUIRE(tile >= 0 && overlap >= 0, "tile/overlap must be >= 0");
    MaTiling g = ma_make_tiling(H, W, tile, overlap);
    MA_REQUIRE(g.Ph < 32767 && g.Pw < 32767, "cv2.remap requires window dimensions < 32767");
    MA_HIP(hipSetDevice(ctx->device));
    const int nwin = g.ntx * g.nty;
    const bool cells = tile > 0 && overlap > 0 && tile > 2 * overlap;
    const int nsegx = 2 * g.ntx + 1, nsegy = 2 * g.nty + 1;
    const size_t ncell = cells ? (size_t)nsegx * nsegy : 0;
    MA_TRY(ma_dconst_reserve(ctx, ((size_t)nwin + ncell) * 2 * sizeof(unsigned)));
    unsigned* maxes = (unsigned*)ctx->dconst;
    unsigned* cellkeys = maxes + (size_t)nwin * 2;
    MaProfScope ps(ctx, MA_K_MERGE, (double)H * W);
    MA_HIP(hipMemsetAsync(maxes, 0, ((size_t)nwin + ncell) * 2 * sizeof(unsigned), ctx->stream));
    if (cells) {
        const int bands = (2 * overlap + CM_ROWS - 1) / CM_ROWS + (tile - 2 * overlap + CM_ROWS - 1) / CM_ROWS;
        hipLaunchKernelGGL(cell_max_kernel, dim3(nsegx, (g.nty + 1) * bands), dim3(256), 0, ctx->stream,
                           (const float2*)flow1, (const float2*)flow2, g, nsegx, cellkeys);
        hipLaunchKernelGGL((window_from_cells_kernel<2>), dim3((nwin + 255) / 256), dim3(256), 0, ctx->stream, cellkeys,
                           (const unsigned*)nullptr, g, nsegx, maxes);
    } else {
        hipLaunchKernelGGL(window_max_kernel, dim3(nwin, (g.Ph + WM_ROWS - 1) / WM_ROWS), dim3(256), 0, ctx->stream,
                           (const float2*)flow1, (const float2*)flow2, g, maxes);
    }
    hipLaunchKernelGGL(merge_flows_kernel, dim3((W + 255) / 256, (H + MERGE_ROWS - 1) / MERGE_ROWS), dim3(256), 0, ctx->stream, (const float2*)flow1,
                       (const float2*)flow2, g, maxes, (float2*)out);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

int ma_merge_flows_tiled_cells(ma_ctx* ctx, const float* flow1, const float* flow2, int H, int W, int tile, int overlap,
                               const unsigned* cellkeys1, const unsigned* cellkeys2, float* out)
{
    MA_REQUIRE(ctx && flow1 && flow2 && out && cellkeys1 && cellkeys2, "NULL argument");
    MA_REQUIRE(H > 0 && W > 0 && H <= MA_GRID_Y_MAX * MERGE_ROWS, "bad image size");
    MA_REQUIRE(tile > 2 * overlap && overlap > 0, "flow cell maxima need tile > 2*overlap > 0");
    MaTiling g = ma_make_tiling(H, W, tile, overlap);
    MA_REQUIRE(g.Ph < 32767 && g.Pw < 32767, "cv2.remap requires window dimensions < 32767");
    MA_HIP(hipSetDevice(ctx->device));
    const int nwin = g.ntx * g.nty, nsegx = 2 * g.ntx + 1;
    MA_TRY(ma_dconst_reserve(ctx, (size_t)nwin * 2 * sizeof(unsigned)));
    unsigned* maxes = (unsigned*)ctx->dconst;
    MaProfScope ps(ctx, MA_K_MERGE, (double)H * W);
    hipLaunchKernelGGL((window_from_cells_kernel<1>), dim3((nwin + 255) / 256), dim3(256), 0, ctx->stream, cellkeys1,
                       cellkeys2, g, nsegx, maxes);
    hipLaunchKernelGGL(merge_flows_kernel, dim3((W + 255) / 256, (H + MERGE_ROWS - 1) / MERGE_ROWS), dim3(256), 0, ctx->stream, (const float2*)flow1,
                       (const float2*)flow2, g, maxes, (float2*)out);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

} // extern "C"
