// Difference-of-Gaussians preprocess, OptFlowRegistrator.dog (optflow_registrator.py:249-274):
//   normalize(0,1,MINMAX,32F) -> GaussianBlur(k,k,sigma_lo) and GaussianBlur(k,k,sigma_hi), k = 8*sigma_lo+1
//   -> hs - ls -> normalize(0,255,MINMAX,8U)
// plus the min/max reductions it needs and the input conditioning of SURVEY 8f-2
// (np.maximum fold over z, utils.py:92; cv2.normalize -> u8, utils.py:94).
// Semantics: SURVEY.md Appendix A.5.  Both sigmas share every load; rows first, then columns.
#include "ma_internal.h"

#include <cfloat>
#include <cmath>

namespace {

// ---- min / max ------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void minmax_partial(const T* __restrict__ src, size_t n, float* __restrict__ part)
{
    float lo = INFINITY, hi = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = (float)src[i];
        lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_down(lo, off));
        hi = fmaxf(hi, __shfl_down(hi, off));
    }
    __shared__ float slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2] = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
        part[blockIdx.x * 2 + 1] = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
    }
}

__global__ __launch_bounds__(256) void minmax_final(const float* __restrict__ part, int nparts, float* __restrict__ out)
{
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < nparts; i += 256) { lo = fminf(lo, part[i * 2]); hi = fmaxf(hi, part[i * 2 + 1]); }
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_down(lo, off));
        hi = fmaxf(hi, __shfl_down(hi, off));
    }
    __shared__ float slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
        out[1] = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
    }
}

constexpr int MM_BLOCKS = 1024;

// min/max of a device array -> host doubles (synchronises).  Uses ctx->dconst for the partials.
int minmax_impl(ma_ctx* ctx, const void* src, int dtype, size_t n, double* mn, double* mx)
{
    MA_TRY(ma_dconst_reserve(ctx, (MM_BLOCKS * 2 + 2) * sizeof(float)));
    MA_TRY(ma_pinned_reserve(ctx, 64));
    float* part = (float*)ctx->dconst;
    float* out = part + MM_BLOCKS * 2;
    int blocks = (int)((n + 256 * 8 - 1) / (256 * 8));
    if (blocks > MM_BLOCKS) blocks = MM_BLOCKS;
    if (blocks < 1) blocks = 1;
    if (dtype == MA_U8) hipLaunchKernelGGL((minmax_partial<uint8_t>), dim3(blocks), dim3(256), 0, ctx->stream, (const uint8_t*)src, n, part);
    else if (dtype == MA_U16) hipLaunchKernelGGL((minmax_partial<uint16_t>), dim3(blocks), dim3(256), 0, ctx->stream, (const uint16_t*)src, n, part);
    else hipLaunchKernelGGL((minmax_partial<float>), dim3(blocks), dim3(256), 0, ctx->stream, (const float*)src, n, part);
    hipLaunchKernelGGL(minmax_final, dim3(1), dim3(256), 0, ctx->stream, part, blocks, out);
    MA_HIP(hipGetLastError());
    float* h = (float*)ctx->pinned;
    MA_HIP(hipMemcpyAsync(h, out, 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    MA_HIP(hipStreamSynchronize(ctx->stream));
    *mn = h[0]; *mx = h[1];
    return MA_OK;
}

// ---- DOG blurs --------------------------------------------------------------------------------
// Row pass: normalise on the fly (v*a + b, two roundings) and accumulate both kernels left to right.
template <typename T>
__global__ __launch_bounds__(256) void dog_rows(const T* __restrict__ src, int h, int w, float a, float b, int ksize,
                                                const float* __restrict__ klo, const float* __restrict__ khi,
                                                float* __restrict__ tlo, float* __restrict__ thi)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const T* s = src + (size_t)y * w;
    const int r = ksize / 2;
    float accl = 0.f, acch = 0.f;
    for (int j = 0; j < ksize; j++) {
        float v = __fadd_rn(__fmul_rn((float)s[d_reflect101(x - r + j, w)], a), b);
        float pl = klo[j] * v, ph = khi[j] * v;
        if (j == 0) { accl = pl; acch = ph; }
        else { accl = accl + pl; acch = acch + ph; }
    }
    tlo[(size_t)y * w + x] = accl;
    thi[(size_t)y * w + x] = acch;
}

// Column pass (symmetric form), difference hs - ls, and per-block min/max of the difference.
__global__ __launch_bounds__(256) void dog_cols_diff(const float* __restrict__ tlo, const float* __restrict__ thi, int h,
                                                     int w, int ksize, const float* __restrict__ klo,
                                                     const float* __restrict__ khi, float* __restrict__ diff)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int r = ksize / 2;
    const size_t c = (size_t)y * w + x;
    float sl = klo[r] * tlo[c], sh = khi[r] * thi[c];
    for (int j = 1; j <= r; j++) {
        const size_t pa = (size_t)d_reflect101(y + j, h) * w + x, pb = (size_t)d_reflect101(y - j, h) * w + x;
        float pl = klo[r + j] * (tlo[pa] + tlo[pb]);
        float ph = khi[r + j] * (thi[pa] + thi[pb]);
        sl = sl + pl;
        sh = sh + ph;
    }
    diff[c] = sh - sl;
}

template <typename T>
__global__ __launch_bounds__(256) void scale_to_u8(const T* __restrict__ src, size_t n, float a, float b,
                                                   uint8_t* __restrict__ dst)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = __fadd_rn(__fmul_rn((float)src[i], a), b);
        dst[i] = (uint8_t)d_clamp(d_cvround(v), 0, 255);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void max_project_kernel(const T* __restrict__ planes, int nz, size_t n,
                                                          T* __restrict__ dst)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        T m = planes[i];
        for (int z = 1; z < nz; z++) { T v = planes[(size_t)z * n + i]; m = v > m ? v : m; }
        dst[i] = m;
    }
}

// getGaussianKernel(ksize, sigma, CV_32F) as OpenCV 4.x computes it (A.5): taps in double,
// sum = 2*sum(t)+1, multiplied by 1/sum, rounded to float once.
void gaussian_kernel(int ksize, double sigma, std::vector<float>& k)
{
    k.resize(ksize);
    double sigmaX = sigma > 0 ? sigma : ksize * 0.15 + 0.35;
    double scale2X = -0.125 / (sigmaX * sigmaX);
    int n2 = (ksize - 1) / 2;
    std::vector<double> v(n2 + 1);
    double sum = 0;
    for (int i = 0, x = 1 - ksize; i < n2; i++, x += 2) {
        v[i] = std::exp((double)(x * x) * scale2X);
        sum += v[i];
    }
    sum *= 2;
    sum += 1;
    double mul1 = 1. / sum;
    for (int i = 0; i < n2; i++) {
        double t = v[i] * mul1;
        k[i] = (float)t;
        k[ksize - 1 - i] = (float)t;
    }
    k[n2] = (float)mul1;
}

int grid_for(size_t n) { size_t b = (n + 256 * 4 - 1) / (256 * 4); return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

} // namespace

extern "C" {

int ma_minmax(ma_ctx* ctx, const void* src, int dtype, size_t n, double* mn_host, double* mx_host)
{
    MA_REQUIRE(ctx && src && mn_host && mx_host, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(n > 0, "empty array");
    MA_HIP(hipSetDevice(ctx->device));
    return minmax_impl(ctx, src, dtype, n, mn_host, mx_host);
}

int ma_dog_u8(ma_ctx* ctx, const void* src, int dtype, int h, int w, int low_sigma, int high_sigma, uint8_t* dst)
{
    MA_REQUIRE(ctx && src && dst, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(h > 0 && w > 0 && h <= 65535, "bad image size");
    MA_REQUIRE(low_sigma >= 1 && high_sigma >= 1, "sigmas must be >= 1");
    MA_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)h * w;
    const int ksize = low_sigma * 4 * 2 + 1;  // optflow_registrator.py:262

    double smin, smax;
    MA_TRY(minmax_impl(ctx, src, dtype, n, &smin, &smax));
    // normalize(src, 0, 1, NORM_MINMAX, CV_32F): scale/shift rounded to float as OpenCV does
    double scale = (1.0 - 0.0) * (smax - smin > DBL_EPSILON ? 1. / (smax - smin) : 0);
    scale = (float)scale;
    float shiftf = (float)0.0 - (float)(smin * scale);
    const float a = (float)scale, b = shiftf;

    std::vector<float> klo, khi;
    gaussian_kernel(ksize, low_sigma, klo);
    gaussian_kernel(ksize, high_sigma, khi);
    const float *dlo = nullptr, *dhi = nullptr;
    MA_TRY(ma_const_table(ctx, ((uint64_t)2 << 56) | ((uint64_t)ksize << 16) | (uint64_t)low_sigma, klo.data(), klo.size(), &dlo));
    MA_TRY(ma_const_table(ctx, ((uint64_t)2 << 56) | ((uint64_t)ksize << 16) | (uint64_t)high_sigma, khi.data(), khi.size(), &dhi));

    MA_TRY(ma_ws_reserve(ctx, n * 3 * sizeof(float)));
    float* tlo = (float*)ctx->ws;
    float* thi = tlo + n;
    float* diff = thi + n;
    {
        MaProfScope ps(ctx, MA_K_DOG, (double)n);
        dim3 grid((w + 255) / 256, h), block(256);
        if (dtype == MA_U8) hipLaunchKernelGGL((dog_rows<uint8_t>), grid, block, 0, ctx->stream, (const uint8_t*)src, h, w, a, b, ksize, dlo, dhi, tlo, thi);
        else if (dtype == MA_U16) hipLaunchKernelGGL((dog_rows<uint16_t>), grid, block, 0, ctx->stream, (const uint16_t*)src, h, w, a, b, ksize, dlo, dhi, tlo, thi);
        else hipLaunchKernelGGL((dog_rows<float>), grid, block, 0, ctx->stream, (const float*)src, h, w, a, b, ksize, dlo, dhi, tlo, thi);
        hipLaunchKernelGGL(dog_cols_diff, grid, block, 0, ctx->stream, tlo, thi, h, w, ksize, dlo, dhi, diff);
        MA_HIP(hipGetLastError());
    }
    double dmin, dmax;
    MA_TRY(minmax_impl(ctx, diff, MA_F32, n, &dmin, &dmax));
    // normalize(diff, 0, 255, NORM_MINMAX, CV_8U): scale/shift in double, applied in float
    double scale8 = 255. * (dmax - dmin > DBL_EPSILON ? 1. / (dmax - dmin) : 0);
    double shift8 = 0. - dmin * scale8;
    {
        MaProfScope ps(ctx, MA_K_DOG, 0);
        hipLaunchKernelGGL((scale_to_u8<float>), dim3(grid_for(n)), dim3(256), 0, ctx->stream, diff, n, (float)scale8,
                           (float)shift8, dst);
        MA_HIP(hipGetLastError());
    }
    return MA_OK;
}

int ma_max_project(ma_ctx* ctx, const void* planes, int dtype, int nz, size_t n, void* dst)
{
    MA_REQUIRE(ctx && planes && dst, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(nz >= 1 && n > 0, "empty stack");
    MA_HIP(hipSetDevice(ctx->device));
    MaProfScope ps(ctx, MA_K_OTHER, (double)n * nz);
    dim3 grid(grid_for(n)), block(256);
    if (dtype == MA_U8) hipLaunchKernelGGL((max_project_kernel<uint8_t>), grid, block, 0, ctx->stream, (const uint8_t*)planes, nz, n, (uint8_t*)dst);
    else if (dtype == MA_U16) hipLaunchKernelGGL((max_project_kernel<uint16_t>), grid, block, 0, ctx->stream, (const uint16_t*)planes, nz, n, (uint16_t*)dst);
    else hipLaunchKernelGGL((max_project_kernel<float>), grid, block, 0, ctx->stream, (const float*)planes, nz, n, (float*)dst);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

int ma_normalize_minmax_u8(ma_ctx* ctx, const void* src, int dtype, size_t n, uint8_t* dst)
{
    MA_REQUIRE(ctx && src && dst, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(n > 0, "empty array");
    MA_HIP(hipSetDevice(ctx->device));
    double smin, smax;
    MA_TRY(minmax_impl(ctx, src, dtype, n, &smin, &smax));
    double scale = 255. * (smax - smin > DBL_EPSILON ? 1. / (smax - smin) : 0);
    double shift = 0. - smin * scale;
    MaProfScope ps(ctx, MA_K_OTHER, (double)n);
    dim3 grid(grid_for(n)), block(256);
    if (dtype == MA_U8) hipLaunchKernelGGL((scale_to_u8<uint8_t>), grid, block, 0, ctx->stream, (const uint8_t*)src, n, (float)scale, (float)shift, dst);
    else if (dtype == MA_U16) hipLaunchKernelGGL((scale_to_u8<uint16_t>), grid, block, 0, ctx->stream, (const uint16_t*)src, n, (float)scale, (float)shift, dst);
    else hipLaunchKernelGGL((scale_to_u8<float>), grid, block, 0, ctx->stream, (const float*)src, n, (float)scale, (float)shift, dst);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

} // extern "C"
