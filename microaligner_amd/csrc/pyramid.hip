// Gaussian pyramid steps: cv2.pyrDown (optflow_registrator.py:194) and cv2.pyrUp of the 2-channel
// flow with the numpy pre-multiply fused (optflow_registrator.py:140,150,164,169,212,214).
// Semantics: SURVEY.md Appendix A.3 / A.4 (5-tap [1 4 6 4 1], reflect-101; pyrUp's asymmetric borders).
#include "ma_internal.h"

#include <type_traits>

namespace {

template <typename T> struct PyrAcc { using type = int; };
template <> struct PyrAcc<float> { using type = float; };

template <typename T> struct PyrVec;
template <> struct PyrVec<float> { using v2 = float2; using v4 = float4; };
template <> struct PyrVec<uint8_t> { using v2 = uchar2; using v4 = uchar4; };
template <> struct PyrVec<uint16_t> { using v2 = ushort2; using v4 = ushort4; };

template <typename T> __device__ __forceinline__ T pyr_down_finish(typename PyrAcc<T>::type sum)
{
    if constexpr (std::is_same<T, float>::value) return (T)(sum * (1.f / 256));
    else return (T)(((int)sum + 128) >> 8);  // (sum+128)>>8 of 8/16-bit inputs never leaves the type's range
}

// one destination pixel, any position (reflect-101 on both axes)
template <typename T>
__device__ __forceinline__ T pyr_down_px(const T* __restrict__ src, int h, int w, int x, int y)
{
    using A = typename PyrAcc<T>::type;
    int cx[5];
#pragma unroll
    for (int j = 0; j < 5; j++) cx[j] = d_reflect101(2 * x + j - 2, w);
    A rows[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const T* s = src + (size_t)d_reflect101(2 * y + k - 2, h) * w;
        A v0 = (A)s[cx[0]], v1 = (A)s[cx[1]], v2 = (A)s[cx[2]], v3 = (A)s[cx[3]], v4 = (A)s[cx[4]];
        rows[k] = v2 * 6 + (v1 + v3) * 4 + v0 + v4;
    }
    return pyr_down_finish<T>(rows[2] * 6 + (rows[1] + rows[3]) * 4 + rows[0] + rows[4]);
}

// A thread produces the destination pair (2p, 2p+1) of row y from source columns 4p-2 .. 4p+4: one 2-vector, one
// 4-vector and one scalar load per source row instead of ten scalar loads (vec: w % 4 == 0, so the vectors are
// aligned).  Pairs that touch the left/right border and everything when !vec take the per-pixel path.
// PD_ROWS destination rows per thread: their 2*PD_ROWS + 3 source rows are loaded once (all loads issued up front).
// MM: also reduce (min, max) of the block's outputs into part[2 * block] (see warp_tiled_kernel).
constexpr int PD_ROWS = 4;
template <typename T, bool MM>
__global__ __launch_bounds__(256) void pyr_down_kernel(const T* __restrict__ src, int h, int w, T* __restrict__ dst,
                                                       int dh, int dw, int vec, float* __restrict__ part)
{
    using A = typename PyrAcc<T>::type;
    using V2 = typename PyrVec<T>::v2;
    using V4 = typename PyrVec<T>::v4;
    constexpr int NR = 2 * PD_ROWS + 3;
    const int p = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * PD_ROWS;
    const int x0 = 2 * p;
    float lo = INFINITY, hi = -INFINITY;
    auto put = [&](T* at, T v) {
        *at = v;
        if (MM) { lo = fminf(lo, (float)v); hi = fmaxf(hi, (float)v); }
    };
    if (x0 < dw) {
        if (!vec || p == 0 || 4 * p + 4 >= w || x0 + 1 >= dw) {
            for (int r = 0; r < PD_ROWS && y0 + r < dh; r++) {
                T* drow = dst + (size_t)(y0 + r) * dw;
                put(drow + x0, pyr_down_px<T>(src, h, w, x0, y0 + r));
                if (x0 + 1 < dw) put(drow + x0 + 1, pyr_down_px<T>(src, h, w, x0 + 1, y0 + r));
            }
        } else {
            V2 l[NR];
            V4 m[NR];
            T e[NR];
#pragma unroll
            for (int k = 0; k < NR; k++) {
                // rows past the last destination row of a partial group are clamped (loaded, not used)
                const T* s = src + (size_t)d_reflect101(min(2 * y0 + k - 2, 2 * (dh - 1) + 2), h) * w + 4 * p;
                l[k] = *reinterpret_cast<const V2*>(s - 2);
                m[k] = *reinterpret_cast<const V4*>(s);
                e[k] = s[4];
            }
            A ra[NR], rb[NR];
#pragma unroll
            for (int k = 0; k < NR; k++) {
                const A c0 = (A)l[k].x, c1 = (A)l[k].y, c2 = (A)m[k].x, c3 = (A)m[k].y, c4 = (A)m[k].z, c5 = (A)m[k].w, c6 = (A)e[k];
                ra[k] = c2 * 6 + (c1 + c3) * 4 + c0 + c4;
                rb[k] = c4 * 6 + (c3 + c5) * 4 + c2 + c6;
            }
#pragma unroll
            for (int r = 0; r < PD_ROWS; r++) {
                if (y0 + r >= dh) break;
                const int k = 2 * r;
                const T oa = pyr_down_finish<T>(ra[k + 2] * 6 + (ra[k + 1] + ra[k + 3]) * 4 + ra[k] + ra[k + 4]);
                const T ob = pyr_down_finish<T>(rb[k + 2] * 6 + (rb[k + 1] + rb[k + 3]) * 4 + rb[k] + rb[k + 4]);
                T* drow = dst + (size_t)(y0 + r) * dw;
                if ((dw & 1) == 0) {
                    V2 o; o.x = oa; o.y = ob;
                    *reinterpret_cast<V2*>(drow + x0) = o;
                    if (MM) { lo = fminf(lo, fminf((float)oa, (float)ob)); hi = fmaxf(hi, fmaxf((float)oa, (float)ob)); }
                } else { put(drow + x0, oa); put(drow + x0 + 1, ob); }
            }
        }
    }
    if (MM) d_block_minmax(lo, hi, part + 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x));
}

// horizontally upsampled value of one source row at destination column X (two channels)
__device__ __forceinline__ float2 up_row(const float2* __restrict__ s, int w, int X, int dw, float scale)
{
    auto ld = [&](int i) { float2 v = s[i]; return make_float2(v.x * scale, v.y * scale); };
    if (w == 1) { float2 a = ld(0); return make_float2(a.x * 8, a.y * 8); }
    if (X >= 2 * w) X = 2 * w - 1;  // dw > 2w: the extra column repeats the last odd column
    const int x = X >> 1;
    if ((X & 1) == 0) {
        if (x == 0) { float2 a = ld(0), b = ld(1); return make_float2(a.x * 6 + b.x * 2, a.y * 6 + b.y * 2); }
        if (x == w - 1) { float2 a = ld(w - 2), b = ld(w - 1); return make_float2(a.x + b.x * 7, a.y + b.y * 7); }
        float2 a = ld(x - 1), b = ld(x), c = ld(x + 1);
        return make_float2(a.x + b.x * 6 + c.x, a.y + b.y * 6 + c.y);
    }
    if (x == w - 1) { float2 a = ld(w - 1); return make_float2(a.x * 8, a.y * 8); }
    float2 a = ld(x), b = ld(x + 1);
    return make_float2((a.x + b.x) * 4, (a.y + b.y) * 4);
}

// one destination pixel, any position
__device__ __forceinline__ float2 pyr_up_px(const float2* __restrict__ src, int h, int w, float scale, int X, int Y, int dw)
{
    if (Y >= 2 * h) Y = 2 * h - 2;  // dh > 2h: the extra row repeats destination row 2h-2
    const int y = Y >> 1;
    auto srow = [&](int sy) { return src + (size_t)(d_reflect101(sy * 2, h * 2) / 2) * w; };
    float2 r1 = up_row(srow(y), w, X, dw, scale), r2 = up_row(srow(y + 1), w, X, dw, scale);
    if ((Y & 1) == 0) {
        float2 r0 = up_row(srow(y - 1), w, X, dw, scale);
        return make_float2((r0.x + r1.x * 6 + r2.x) * (1.f / 64), (r0.y + r1.y * 6 + r2.y) * (1.f / 64));
    }
    return make_float2(((r1.x + r2.x) * 4) * (1.f / 64), ((r1.y + r2.y) * 4) * (1.f / 64));
}

// A thread produces PU_ROWS vertically adjacent 2x2 destination cells (2x..2x+1, 2y..2y+1) from the
// (PU_ROWS + 2) x 3 source neighbourhood: 3 loads per source row, all issued up front, 16-byte stores.  Cells on the
// source border (and the extra row/column of odd destination sizes) take the per-pixel path; same operations in
// the same order either way.
constexpr int PU_ROWS = 4;
__global__ __launch_bounds__(256) void pyr_up_flow_kernel(const float2* __restrict__ src, int h, int w, float scale,
                                                          float2* __restrict__ dst, int dh, int dw)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * PU_ROWS;
    const int X = 2 * x;
    if (X >= dw) return;
    if (x < 1 || x > w - 2 || y0 < 1 || y0 + PU_ROWS - 1 > h - 2) {
        for (int r = 0; r < PU_ROWS; r++)
            for (int j = 0; j < 2; j++)
                for (int i = 0; i < 2; i++) {
                    const int Y = 2 * (y0 + r) + j;
                    if (X + i < dw && Y < dh) dst[(size_t)Y * dw + X + i] = pyr_up_px(src, h, w, scale, X + i, Y, dw);
                }
        return;
    }
    float2 ev[PU_ROWS + 2], od[PU_ROWS + 2];  // horizontally upsampled source rows y0-1 .. y0+PU_ROWS at X and X+1
    {
        float2 a[PU_ROWS + 2], b[PU_ROWS + 2], c[PU_ROWS + 2];
#pragma unroll
        for (int k = 0; k < PU_ROWS + 2; k++) {
            const float2* s = src + (size_t)(y0 + k - 1) * w + x;
            a[k] = s[-1]; b[k] = s[0]; c[k] = s[1];
        }
#pragma unroll
        for (int k = 0; k < PU_ROWS + 2; k++) {
            const float ax = a[k].x * scale, ay = a[k].y * scale, bx = b[k].x * scale, by = b[k].y * scale,
                        cx = c[k].x * scale, cy = c[k].y * scale;
            ev[k] = make_float2(ax + bx * 6 + cx, ay + by * 6 + cy);
            od[k] = make_float2((bx + cx) * 4, (by + cy) * 4);
        }
    }
#pragma unroll
    for (int r = 0; r < PU_ROWS; r++) {
        float4 top, bot;   // rows of cell r: source rows r (above), r+1 (centre), r+2 (below)
        top.x = (ev[r].x + ev[r + 1].x * 6 + ev[r + 2].x) * (1.f / 64);
        top.y = (ev[r].y + ev[r + 1].y * 6 + ev[r + 2].y) * (1.f / 64);
        top.z = (od[r].x + od[r + 1].x * 6 + od[r + 2].x) * (1.f / 64);
        top.w = (od[r].y + od[r + 1].y * 6 + od[r + 2].y) * (1.f / 64);
        bot.x = ((ev[r + 1].x + ev[r + 2].x) * 4) * (1.f / 64);
        bot.y = ((ev[r + 1].y + ev[r + 2].y) * 4) * (1.f / 64);
        bot.z = ((od[r + 1].x + od[r + 2].x) * 4) * (1.f / 64);
        bot.w = ((od[r + 1].y + od[r + 2].y) * 4) * (1.f / 64);
        float2* d0 = dst + (size_t)(2 * (y0 + r)) * dw + X;
        if ((dw & 1) == 0) {
            // non-temporal: the upsampled flow (2.1 GB at full resolution) is read again only after other kernels have run
            // (0.91 -> 0.76 ms per step, profiles/r05_notes.md)
            typedef float nt_f4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store((nt_f4){top.x, top.y, top.z, top.w}, reinterpret_cast<nt_f4*>(d0));
            __builtin_nontemporal_store((nt_f4){bot.x, bot.y, bot.z, bot.w}, reinterpret_cast<nt_f4*>(d0 + dw));
        } else {
            d0[0] = make_float2(top.x, top.y); d0[1] = make_float2(top.z, top.w);
            d0[dw] = make_float2(bot.x, bot.y); d0[dw + 1] = make_float2(bot.z, bot.w);
        }
    }
}

} // namespace

extern "C" {

static int pyr_down_impl(ma_ctx* ctx, const void* src, int dtype, int h, int w, void* dst, float* minmax_dev)
{
    MA_REQUIRE(ctx && src && dst, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(h > 0 && w > 0, "empty image");
    const int dh = (h + 1) / 2, dw = (w + 1) / 2;
    MA_REQUIRE(dh <= MA_GRID_Y_MAX * PD_ROWS, "image too tall");
    MA_HIP(hipSetDevice(ctx->device));
    dim3 grid(((dw + 1) / 2 + 255) / 256, (dh + PD_ROWS - 1) / PD_ROWS), block(256);
    const size_t nblk = (size_t)grid.x * grid.y;
    float* part = nullptr;
    if (minmax_dev) {
        MA_TRY(ma_ws_reserve(ctx, nblk * 2 * sizeof(float)));
        part = (float*)ctx->ws;
    }
    MaProfScope ps(ctx, MA_K_PYR_DOWN, (double)h * w);
    const int vec = (w % 4 == 0) && ((size_t)src % 16 == 0) && ((size_t)dst % 8 == 0);
#define MA_PD(T) do { if (part) hipLaunchKernelGGL((pyr_down_kernel<T, true>), grid, block, 0, ctx->stream, (const T*)src, h, w, (T*)dst, dh, dw, vec, part); \
                      else hipLaunchKernelGGL((pyr_down_kernel<T, false>), grid, block, 0, ctx->stream, (const T*)src, h, w, (T*)dst, dh, dw, vec, part); } while (0)
    if (dtype == MA_U8) MA_PD(uint8_t);
    else if (dtype == MA_U16) MA_PD(uint16_t);
    else MA_PD(float);
#undef MA_PD
    MA_HIP(hipGetLastError());
    if (part) MA_TRY(ma_launch_minmax_final(ctx, part, (int)nblk, minmax_dev));
    return MA_OK;
}

int ma_pyr_down(ma_ctx* ctx, const void* src, int dtype, int h, int w, void* dst)
{
    return pyr_down_impl(ctx, src, dtype, h, w, dst, nullptr);
}

int ma_pyr_down_minmax(ma_ctx* ctx, const void* src, int dtype, int h, int w, void* dst, float* minmax_dev)
{
    MA_REQUIRE(minmax_dev, "NULL argument");
    return pyr_down_impl(ctx, src, dtype, h, w, dst, minmax_dev);
}

int ma_pyr_up_flow(ma_ctx* ctx, const float* src, int h, int w, float scale, float* dst, int dh, int dw)
{
    MA_REQUIRE(ctx && src && dst, "NULL argument");
    MA_REQUIRE(h > 0 && w > 0 && dh > 0 && dw > 0 && (dh + 1) / 2 <= MA_GRID_Y_MAX * PU_ROWS, "bad size");
    MA_REQUIRE(abs(dw - w * 2) == dw % 2 && abs(dh - h * 2) == dh % 2,
               "cv2.pyrUp requires |dst - 2*src| == dst % 2 on both axes");
    MA_HIP(hipSetDevice(ctx->device));
    MaProfScope ps(ctx, MA_K_PYR_UP, (double)dh * dw);
    MA_REQUIRE((size_t)dst % 16 == 0, "dst must be 16-byte aligned");
    hipLaunchKernelGGL(pyr_up_flow_kernel, dim3(((dw + 1) / 2 + 255) / 256, ((dh + 1) / 2 + PU_ROWS - 1) / PU_ROWS), dim3(256), 0, ctx->stream,
                       (const float2*)src, h, w, scale, (float2*)dst, dh, dw);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

} // extern "C"
