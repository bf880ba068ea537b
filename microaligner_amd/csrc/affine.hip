// transform_img_with_tmat (microaligner/shared_modules/utils.py:98-114): skimage.transform.warp with the inverse
// 3x3 matrix, bilinear interpolation, constant border 0, clip to the input range, cast back to the input dtype.
// Semantics restated from scikit-image 0.18 (_warps.py warp / _warp_fast / bilinear_interpolation /
// _clip_warp_output) and pinned by fixtures produced with the real library (tests/golden/make_affine_golden.py).
// Integer images are processed in f64, float32 images in f32 (skimage's convert_to_float).  SURVEY 8f-3.
#include "ma_internal.h"

#include <cmath>

namespace {

struct AffineArgs {
    double m[9];
    float mf[9];
    int mode;  // 0 metric (no shear), 1 affine, 2 projective -- chosen by exact comparisons like _warp_fast
};

template <typename F> struct MatOf;
template <> struct MatOf<double> { __device__ static const double* get(const AffineArgs& a) { return a.m; } };
template <> struct MatOf<float> { __device__ static const float* get(const AffineArgs& a) { return a.mf; } };

template <typename T, typename F>
__global__ __launch_bounds__(256) void warp_affine_kernel(const T* __restrict__ src, int h, int w, AffineArgs a,
                                                          const float* __restrict__ mm, T* __restrict__ dst)
{
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= w) return;
    const F* M = MatOf<F>::get(a);
    const F x = (F)c, y = (F)r;
    F sx, sy;
    if (a.mode == 0) { sx = M[0] * x + M[2]; sy = M[4] * y + M[5]; }
    else if (a.mode == 1) { sx = M[0] * x + M[1] * y + M[2]; sy = M[3] * x + M[4] * y + M[5]; }
    else {
        const F z = M[6] * x + M[7] * y + M[8];
        sx = (M[0] * x + M[1] * y + M[2]) / z;
        sy = (M[3] * x + M[4] * y + M[5]) / z;
    }
    // floor / ceil corners; a non-finite or absurd coordinate is treated as far outside the image (-> cval)
    const bool finite = fabs(sx) < (F)1e9 && fabs(sy) < (F)1e9;
    if (!finite) { sx = (F)-5; sy = (F)-5; }
    const long minr = (long)floor(sy), minc = (long)floor(sx), maxr = (long)ceil(sy), maxc = (long)ceil(sx);
    const F dr = sy - (F)minr, dc = sx - (F)minc;
    auto px = [&](long rr, long cc2) -> F {
        return (rr >= 0 && rr < h && cc2 >= 0 && cc2 < w) ? (F)src[(size_t)rr * w + cc2] : (F)0;
    };
    const F one = (F)1;
    const F top = (one - dc) * px(minr, minc) + dc * px(minr, maxc);
    const F bot = (one - dc) * px(maxr, minc) + dc * px(maxr, maxc);
    F out = (one - dr) * top + dr * bot;
    // _clip_warp_output: clip to [min, max] of the input, but keep pixels that equal cval (0) when 0 is outside it
    const F lo = (F)mm[0], hi = (F)mm[1];
    const bool keep0 = !(lo <= (F)0 && (F)0 <= hi) && out == (F)0;
    if (!keep0) out = out < lo ? lo : (out > hi ? hi : out);
    dst[(size_t)r * w + c] = (T)out;  // ndarray.astype: truncation toward zero for the integer dtypes
}

template <typename T>
__global__ __launch_bounds__(256) void mm_partial(const T* __restrict__ src, size_t n, float* __restrict__ part)
{
    float lo = INFINITY, hi = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = (float)src[i];
        lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
    for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_down(lo, off)); hi = fmaxf(hi, __shfl_down(hi, off)); }
    __shared__ float slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2] = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
        part[blockIdx.x * 2 + 1] = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
    }
}
__global__ __launch_bounds__(256) void mm_final(const float* __restrict__ part, int nparts, float* __restrict__ out)
{
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < nparts; i += 256) { lo = fminf(lo, part[i * 2]); hi = fmaxf(hi, part[i * 2 + 1]); }
    for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_down(lo, off)); hi = fmaxf(hi, __shfl_down(hi, off)); }
    __shared__ float slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
        out[1] = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
    }
}

} // namespace

extern "C" int ma_warp_affine(ma_ctx* ctx, const void* src, int dtype, int h, int w, const double* inverse_3x3,
                              void* dst)
{
    MA_REQUIRE(ctx && src && dst && inverse_3x3, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(h > 0 && w > 0 && h <= 65535, "bad image size");
    MA_HIP(hipSetDevice(ctx->device));
    AffineArgs a;
    for (int i = 0; i < 9; i++) { a.m[i] = inverse_3x3[i]; a.mf[i] = (float)inverse_3x3[i]; }
    // _warp_fast picks the transform on the matrix in the image's float type
    bool affine_row, noshear;
    if (dtype == MA_F32) { affine_row = a.mf[6] == 0 && a.mf[7] == 0 && a.mf[8] == 1; noshear = a.mf[1] == 0 && a.mf[3] == 0; }
    else { affine_row = a.m[6] == 0 && a.m[7] == 0 && a.m[8] == 1; noshear = a.m[1] == 0 && a.m[3] == 0; }
    a.mode = affine_row ? (noshear ? 0 : 1) : 2;
    const size_t n = (size_t)h * w;
    const int blocks = (int)std::min<size_t>(1024, (n + 2047) / 2048);
    MA_TRY(ma_dconst_reserve(ctx, (1024 * 2 + 8) * sizeof(float)));
    float* part = (float*)ctx->dconst;
    float* mm = part + 2048;
    MaProfScope ps(ctx, MA_K_OTHER, (double)n);
    dim3 grid((w + 255) / 256, h), block(256);
    if (dtype == MA_U8) {
        hipLaunchKernelGGL((mm_partial<uint8_t>), dim3(blocks), dim3(256), 0, ctx->stream, (const uint8_t*)src, n, part);
        hipLaunchKernelGGL(mm_final, dim3(1), dim3(256), 0, ctx->stream, part, blocks, mm);
        hipLaunchKernelGGL((warp_affine_kernel<uint8_t, double>), grid, block, 0, ctx->stream, (const uint8_t*)src, h, w, a, mm, (uint8_t*)dst);
    } else if (dtype == MA_U16) {
        hipLaunchKernelGGL((mm_partial<uint16_t>), dim3(blocks), dim3(256), 0, ctx->stream, (const uint16_t*)src, n, part);
        hipLaunchKernelGGL(mm_final, dim3(1), dim3(256), 0, ctx->stream, part, blocks, mm);
        hipLaunchKernelGGL((warp_affine_kernel<uint16_t, double>), grid, block, 0, ctx->stream, (const uint16_t*)src, h, w, a, mm, (uint16_t*)dst);
    } else {
        hipLaunchKernelGGL((mm_partial<float>), dim3(blocks), dim3(256), 0, ctx->stream, (const float*)src, n, part);
        hipLaunchKernelGGL(mm_final, dim3(1), dim3(256), 0, ctx->stream, part, blocks, mm);
        hipLaunchKernelGGL((warp_affine_kernel<float, float>), grid, block, 0, ctx->stream, (const float*)src, h, w, a, mm, (float*)dst);
    }
    MA_HIP(hipGetLastError());
    return MA_OK;
}
