// Gaussian pyramid steps: cv2.pyrDown (optflow_registrator.py:194) and cv2.pyrUp of the 2-channel
// flow with the numpy pre-multiply fused (optflow_registrator.py:140,150,164,169,212,214).
// Semantics: SURVEY.md Appendix A.3 / A.4 (5-tap [1 4 6 4 1], reflect-101; pyrUp's asymmetric borders).
#include "ma_internal.h"

#include <type_traits>

namespace {

template <typename T> struct PyrAcc { using type = int; };
template <> struct PyrAcc<float> { using type = float; };

template <typename T>
__global__ __launch_bounds__(256) void pyr_down_kernel(const T* __restrict__ src, int h, int w, T* __restrict__ dst,
                                                       int dh, int dw)
{
    using A = typename PyrAcc<T>::type;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
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
    A sum = rows[2] * 6 + (rows[1] + rows[3]) * 4 + rows[0] + rows[4];
    if constexpr (std::is_same<T, float>::value) {
        dst[(size_t)y * dw + x] = (T)(sum * (1.f / 256));
    } else {
        int v = ((int)sum + 128) >> 8;
        dst[(size_t)y * dw + x] = (T)v;  // (sum+128)>>8 of 8/16-bit inputs never leaves the type's range
    }
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

__global__ __launch_bounds__(256) void pyr_up_flow_kernel(const float2* __restrict__ src, int h, int w, float scale,
                                                          float2* __restrict__ dst, int dh, int dw)
{
    const int X = blockIdx.x * 256 + threadIdx.x;
    int Y = blockIdx.y;
    if (X >= dw) return;
    const int Yout = Y;
    if (Y >= 2 * h) Y = 2 * h - 2;  // dh > 2h: the extra row repeats destination row 2h-2
    const int y = Y >> 1;
    auto srow = [&](int sy) { return src + (size_t)(d_reflect101(sy * 2, h * 2) / 2) * w; };
    float2 r1 = up_row(srow(y), w, X, dw, scale), r2 = up_row(srow(y + 1), w, X, dw, scale);
    float2 out;
    if ((Y & 1) == 0) {
        float2 r0 = up_row(srow(y - 1), w, X, dw, scale);
        out = make_float2((r0.x + r1.x * 6 + r2.x) * (1.f / 64), (r0.y + r1.y * 6 + r2.y) * (1.f / 64));
    } else {
        out = make_float2(((r1.x + r2.x) * 4) * (1.f / 64), ((r1.y + r2.y) * 4) * (1.f / 64));
    }
    dst[(size_t)Yout * dw + X] = out;
}

} // namespace

extern "C" {

int ma_pyr_down(ma_ctx* ctx, const void* src, int dtype, int h, int w, void* dst)
{
    MA_REQUIRE(ctx && src && dst, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(h > 0 && w > 0, "empty image");
    const int dh = (h + 1) / 2, dw = (w + 1) / 2;
    MA_REQUIRE(dh <= 65535, "image too tall");
    MA_HIP(hipSetDevice(ctx->device));
    MaProfScope ps(ctx, MA_K_PYR_DOWN, (double)h * w);
    dim3 grid((dw + 255) / 256, dh), block(256);
    if (dtype == MA_U8) hipLaunchKernelGGL((pyr_down_kernel<uint8_t>), grid, block, 0, ctx->stream, (const uint8_t*)src, h, w, (uint8_t*)dst, dh, dw);
    else if (dtype == MA_U16) hipLaunchKernelGGL((pyr_down_kernel<uint16_t>), grid, block, 0, ctx->stream, (const uint16_t*)src, h, w, (uint16_t*)dst, dh, dw);
    else hipLaunchKernelGGL((pyr_down_kernel<float>), grid, block, 0, ctx->stream, (const float*)src, h, w, (float*)dst, dh, dw);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

int ma_pyr_up_flow(ma_ctx* ctx, const float* src, int h, int w, float scale, float* dst, int dh, int dw)
{
    MA_REQUIRE(ctx && src && dst, "NULL argument");
    MA_REQUIRE(h > 0 && w > 0 && dh > 0 && dw > 0 && dh <= 65535, "bad size");
    MA_REQUIRE(abs(dw - w * 2) == dw % 2 && abs(dh - h * 2) == dh % 2,
               "cv2.pyrUp requires |dst - 2*src| == dst % 2 on both axes");
    MA_HIP(hipSetDevice(ctx->device));
    MaProfScope ps(ctx, MA_K_PYR_UP, (double)dh * dw);
    hipLaunchKernelGGL(pyr_up_flow_kernel, dim3((dw + 255) / 256, dh), dim3(256), 0, ctx->stream, (const float2*)src,
                       h, w, scale, (float2*)dst, dh, dw);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

} // extern "C"
