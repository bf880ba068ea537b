// Difference-of-Gaussians preprocess, OptFlowRegistrator.dog (optflow_registrator.py:249-274):
//   normalize(0,1,MINMAX,32F) -> GaussianBlur(k,k,sigma_lo) and GaussianBlur(k,k,sigma_hi), k = 8*sigma_lo+1
//   -> hs - ls -> normalize(0,255,MINMAX,8U)
// plus the min/max reductions it needs and the input conditioning of SURVEY 8f-2
// (np.maximum fold over z, utils.py:92; cv2.normalize -> u8, utils.py:94).
// Semantics: SURVEY.md Appendix A.5.  Rows first (plain left-to-right accumulation), then columns
// (symmetric form); both sigmas share every load.  No host synchronisation inside the chain: the
// normalisation scalars are derived on the device from the reduced min/max exactly as OpenCV derives them.
#include "ma_internal.h"

#include <algorithm>
#include <cfloat>
#include <cmath>

namespace {

// ---- min / max ------------------------------------------------------------------------------
__device__ __forceinline__ void block_minmax_256(float lo, float hi, float* out2)
{
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_down(lo, off));
        hi = fmaxf(hi, __shfl_down(hi, off));
    }
    __shared__ float slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out2[0] = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
        out2[1] = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
    }
}

// 16 bytes per lane and load; head/tail elements that do not fill an aligned 16-byte group go one by one
template <typename T>
__global__ __launch_bounds__(256) void minmax_partial(const T* __restrict__ src, size_t n, float* __restrict__ part)
{
    constexpr int E = 16 / sizeof(T);
    float lo = INFINITY, hi = -INFINITY;
    const size_t mis = ((size_t)src & 15) / sizeof(T);
    size_t head = mis ? E - mis : 0;
    if (head > n) head = n;
    const size_t nvec = (n - head) / E;
    const uint4* v = reinterpret_cast<const uint4*>(src + head);
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (size_t i = tid; i < nvec; i += stride) {
        const uint4 q = v[i];
        const unsigned wd[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if constexpr (sizeof(T) == 4) {
                float f = __uint_as_float(wd[k]);
                lo = fminf(lo, f); hi = fmaxf(hi, f);
            } else if constexpr (sizeof(T) == 2) {
                float f0 = (float)(wd[k] & 0xffffu), f1 = (float)(wd[k] >> 16);
                lo = fminf(lo, fminf(f0, f1)); hi = fmaxf(hi, fmaxf(f0, f1));
            } else {
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    float f = (float)((wd[k] >> (8 * b)) & 0xffu);
                    lo = fminf(lo, f); hi = fmaxf(hi, f);
                }
            }
        }
    }
    for (size_t i = tid; i < head; i += stride) { float f = (float)src[i]; lo = fminf(lo, f); hi = fmaxf(hi, f); }
    for (size_t i = head + nvec * E + tid; i < n; i += stride) { float f = (float)src[i]; lo = fminf(lo, f); hi = fmaxf(hi, f); }
    block_minmax_256(lo, hi, part + blockIdx.x * 2);
}

// scalars of the DOG chain, produced and consumed on the device
struct DogScalars {
    float mm_src[2];   // min, max of the input
    float mm_diff[2];  // min, max of hs - ls
    float a, b;        // normalize(src, 0, 1, MINMAX, 32F):  v*a + b
    float a8, b8;      // normalize(diff, 0, 255, MINMAX, 8U): v*a8 + b8
    int src_max_is_zero;
};

// normalize(src, 0, 1, NORM_MINMAX, CV_32F): scale rounded to float, shift = (float)0 - (float)(smin*scale)
// sticky (may be NULL): set to 1 when the input's maximum is exactly 0 but the image is not all zero -- the one input
// for which the reference's dog() returns the image UNCHANGED (optflow_registrator.py:256-257) and the uint8 result
// of this chain is not what it goes on with; ma_optflow_register reports it (register.hip)
__device__ __forceinline__ void d_dog_params_in(DogScalars* s, float mn, float mx, int* sticky)
{
    if (sticky && mx == 0.f && mn < 0.f) atomicOr(sticky, 1);
    s->mm_src[0] = mn; s->mm_src[1] = mx;
    double smin = mn, smax = mx;
    double scale = (1.0 - 0.0) * (smax - smin > DBL_EPSILON ? 1. / (smax - smin) : 0);
    scale = (float)scale;
    s->a = (float)scale;
    s->b = (float)0.0 - (float)(smin * scale);
    s->src_max_is_zero = smax == 0.0;
}
// normalize(diff, 0, 255, NORM_MINMAX, CV_8U): scale/shift in double, applied in float
__device__ __forceinline__ void d_dog_params_out(DogScalars* s, float mn, float mx)
{
    s->mm_diff[0] = mn; s->mm_diff[1] = mx;
    double dmin = mn, dmax = mx;
    double scale = 255. * (dmax - dmin > DBL_EPSILON ? 1. / (dmax - dmin) : 0);
    double shift = 0. - dmin * scale;
    s->a8 = (float)scale;
    s->b8 = (float)shift;
}
// the input's (min, max) came from the kernel that produced it
__global__ void dog_params_in(DogScalars* s, const float* __restrict__ mm, int* sticky)
{
    d_dog_params_in(s, mm[0], mm[1], sticky);
}

// one block of 1024 threads folds all partial (min, max) pairs; 4 independent 8-byte loads per lane and step
constexpr int MMF_T = 1024;
// sc / what: the DOG scalars that follow from this (min, max) are computed by the same thread (what = 1: the input's
// normalisation, 2: the difference image's) instead of by a kernel of their own
__global__ __launch_bounds__(MMF_T) void minmax_final(const float* __restrict__ part, int nparts, float* __restrict__ out,
                                                      DogScalars* __restrict__ sc, int what, int* sticky = nullptr)
{
    float lo = INFINITY, hi = -INFINITY;
    const float2* p2 = reinterpret_cast<const float2*>(part);
    for (int i0 = threadIdx.x; i0 < nparts; i0 += 4 * MMF_T) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = p2[min(i0 + u * MMF_T, nparts - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) { lo = fminf(lo, v[u].x); hi = fmaxf(hi, v[u].y); }
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_down(lo, off));
        hi = fmaxf(hi, __shfl_down(hi, off));
    }
    __shared__ float slo[MMF_T / 64], shi[MMF_T / 64];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < MMF_T / 64; k++) { lo = fminf(lo, slo[k]); hi = fmaxf(hi, shi[k]); }
        if (out) { out[0] = lo; out[1] = hi; }
        if (what == 1) d_dog_params_in(sc, lo, hi, sticky);
        else if (what == 2) d_dog_params_out(sc, lo, hi);
    }
}

constexpr int MM_BLOCKS = 2048;

int launch_minmax(ma_ctx* ctx, const void* src, int dtype, size_t n, float* part, float* out2, DogScalars* sc = nullptr,
                  int what = 0, int* sticky = nullptr)
{
    int blocks = (int)((n + 256 * 8 - 1) / (256 * 8));
    if (blocks > MM_BLOCKS) blocks = MM_BLOCKS;
    if (blocks < 1) blocks = 1;
    if (dtype == MA_U8) hipLaunchKernelGGL((minmax_partial<uint8_t>), dim3(blocks), dim3(256), 0, ctx->stream, (const uint8_t*)src, n, part);
    else if (dtype == MA_U16) hipLaunchKernelGGL((minmax_partial<uint16_t>), dim3(blocks), dim3(256), 0, ctx->stream, (const uint16_t*)src, n, part);
    else hipLaunchKernelGGL((minmax_partial<float>), dim3(blocks), dim3(256), 0, ctx->stream, (const float*)src, n, part);
    hipLaunchKernelGGL(minmax_final, dim3(1), dim3(MMF_T), 0, ctx->stream, part, blocks, out2, sc, what, sticky);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

// min/max of a device array -> host doubles (synchronises).  Uses ctx->dconst for the partials.
int minmax_host(ma_ctx* ctx, const void* src, int dtype, size_t n, double* mn, double* mx)
{
    MA_TRY(ma_dconst_reserve(ctx, (MM_BLOCKS * 2 + 64) * sizeof(float)));
    MA_TRY(ma_pinned_reserve(ctx, 64));
    float* part = (float*)ctx->dconst;
    float* out = part + MM_BLOCKS * 2;
    MA_TRY(launch_minmax(ctx, src, dtype, n, part, out));
    float* h = (float*)ctx->pinned;
    MA_HIP(hipMemcpyAsync(h, out, 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    MA_HIP(hipStreamSynchronize(ctx->stream));
    *mn = h[0]; *mx = h[1];
    return MA_OK;
}

// v*a + b of cv2.normalize's convertTo, in either rounding model (MA_DOG_FUSED_SCALE)
__device__ __forceinline__ float d_scale(int fused, float v, float a, float b)
{
    return fused ? __fmaf_rn(v, a, b) : __fadd_rn(__fmul_rn(v, a), b);
}

// ---- DOG row pass, any kernel size (the fallback chain for sizes other than the reference's 41) ----------
// Block: 256 output columns x DR rows.  The normalised input row segment (+- r halo, reflect-101) is staged in
// LDS; a thread produces 4 consecutive columns of one row, accumulating both kernels left to right over the
// ksize taps (acc = k0*v0, then acc = acc + k_j*v_j for ascending j).
constexpr int DR = 4;
template <typename T, bool FUSED>
__global__ __launch_bounds__(256) void dog_rows(const T* __restrict__ src, int h, int w, int ksize,
                                                const DogScalars* __restrict__ sc, const float* __restrict__ klh,
                                                float* __restrict__ tlo, float* __restrict__ thi, int fused_scale)
{
    extern __shared__ float lds[];  // [DR][256 + 2r]
    const int r = ksize / 2, span = 256 + 2 * r;
    const int x0 = blockIdx.x * 256, y0 = blockIdx.y * DR;
    const float a = sc->a, b = sc->b;
    for (int row = 0; row < DR; row++) {
        const T* s = src + (size_t)min(y0 + row, h - 1) * w;
        for (int c = threadIdx.x; c < span; c += 256)
            lds[row * span + c] = d_scale(fused_scale, (float)s[d_reflect101(x0 - r + c, w)], a, b);
    }
    __syncthreads();
    const int row = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int y = y0 + row, x = x0 + 4 * lane;
    if (y >= h || x >= w) return;
    const ma_f2* __restrict__ kk = reinterpret_cast<const ma_f2*>(klh);   // klh[j] = (k_lo[j], k_hi[j])
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const float* v = lds + row * span + 4 * lane + o;   // columns past the image stay inside the staged span
        ma_f2 s2 = kk[0] * (ma_f2){v[0], v[0]};
        for (int j = 1; j < ksize; j++) {
            const float vj = v[j];
            if (FUSED) {
                s2 = __builtin_elementwise_fma(kk[j], (ma_f2){vj, vj}, s2);
            } else {
                const ma_f2 p = kk[j] * (ma_f2){vj, vj};
                s2 = s2 + p;
            }
        }
        if (x + o < w) { tlo[(size_t)y * w + x + o] = s2.x; thi[(size_t)y * w + x + o] = s2.y; }
    }
}

// ---- DOG column pass + difference + per-block min/max ---------------------------------------------
// Block: 64 columns x (NW*R) rows; per array the column strip (+- r halo rows, reflect-101) is staged in LDS and
// the symmetric filter slides a register window down the column (d_sym_fir_slide).
template <int R, int NW, bool FUSED>
__global__ __launch_bounds__(64 * NW) void dog_cols_diff(const float* __restrict__ tlo, const float* __restrict__ thi,
                                                         int h, int w, int ksize, const float* __restrict__ klo_c,
                                                         const float* __restrict__ khi_c, float* __restrict__ diff,
                                                         float* __restrict__ part)
{
    extern __shared__ float lds[];  // [(NW*R + 2r)][64]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = ksize / 2;
    // y-fastest work list walked contiguously per XCD (d_xcd_work_item): halo rows come from the XCD's L2
    const int nbx = (w + 63) / 64, nby = (h + NW * R - 1) / (NW * R);
    const int item = d_xcd_work_item(blockIdx.x, nbx * nby);
    if (item >= nbx * nby) return;
    const int x0 = (item / nby) * 64, y0 = (item % nby) * (NW * R);
    constexpr int G = 2;  // guard rows (d_sym_fir_slide contract)
    const int rows = NW * R + 2 * r + 2 * G;
    const int xc = min(x0 + lane, w - 1);
    float sl[R], sh[R];
    for (int arr = 0; arr < 2; arr++) {
        const float* src = arr == 0 ? tlo : thi;
        {
            constexpr int SB = 16;  // global loads issued per wave before the first LDS store
            const float* scol = src + xc;
            for (int j0 = wv; j0 < rows; j0 += NW * SB) {
                float v[SB];
#pragma unroll
                for (int k = 0; k < SB; k++) v[k] = scol[(size_t)d_reflect101(y0 - r - G + min(j0 + NW * k, rows - 1), h) * w];
#pragma unroll
                for (int k = 0; k < SB; k++)
                    if (j0 + NW * k < rows) lds[(j0 + NW * k) * 64 + lane] = v[k];
            }
        }
        __syncthreads();
        if (arr == 0) d_sym_fir_slide_pk<R, FUSED, true>(lds + lane, G + r + wv * R, r, klo_c, sl);
        else d_sym_fir_slide_pk<R, FUSED, true>(lds + lane, G + r + wv * R, r, khi_c, sh);
        __syncthreads();
    }
    float lo = INFINITY, hi = -INFINITY;
    const int x = x0 + lane;
#pragma unroll
    for (int q = 0; q < R; q++) {
        const int y = y0 + wv * R + q;
        if (x < w && y < h) {
            float d = sh[q] - sl[q];
            diff[(size_t)y * w + x] = d;
            lo = fminf(lo, d); hi = fmaxf(hi, d);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_down(lo, off));
        hi = fmaxf(hi, __shfl_down(hi, off));
    }
    __shared__ float slo[NW], shi[NW];
    if (lane == 0) { slo[wv] = lo; shi[wv] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < NW; k++) { lo = fminf(lo, slo[k]); hi = fmaxf(hi, shi[k]); }
        const size_t bid = (size_t)item;
        part[bid * 2] = lo;
        part[bid * 2 + 1] = hi;
    }
}

// ---- fused DOG: row pass -> LDS -> column pass -> difference + block min/max (ksize 41, the reference's sigmas) ----
// One block owns a 64-column strip and streams down LY rows of it.  The row-filtered rows of both sigmas live in
// LDS (CB, [sigma][rows][64]); per super-step of S = NW*R output rows the block
//   1. stages 2 x 32 normalised input rows (+-20 halo columns, reflect-101) and row-filters them into CB rows
//      [CARRY, CARRY + S)  -- 4 outputs per thread, one (sigma_lo, sigma_hi) accumulator pair per output; the 44
//      inputs come from 16-byte LDS reads of the staged row, tap pairs (k_lo, k_hi) from SGPRs, and a packed op takes
//      its input by half-select of the register pair that holds it.  The staged rows have a pitch of 128 floats: on gfx950 a 16-byte-per-lane LDS read is free of
//      bank conflicts only when the 256 bytes a group of 16 lanes reads start on a 128-byte boundary (SQ counters:
//      22 % of the LDS-active cycles were conflicts at pitches of 104 .. 136 floats, 0.5 % at 128);
//   2. runs the symmetric column filter for both sigmas from CB (d_sym_fir_slide_pk, as dog_cols_diff), writes
//      hs - ls and folds the min / max;
//   3. moves the last CARRY = 2r + 2 rows of CB to the top: they are the halo of the next super-step.
// The two intermediate images of the unfused chain (8 B/px written and read back) never reach HBM: per pixel the
// kernel reads the source once (+ the halo columns its neighbour strip also reads, served by L2) and writes
// 4 bytes.  Per output the operations and their order are those of dog_rows / dog_cols_diff: bit-identical.
// The global loads of the next chunk are issued before the row filter of the current one (register prefetch).
constexpr int DF_NW = 8, DF_R = 8, DF_S = DF_NW * DF_R;   // 64 output rows per super-step
constexpr int DF_KS = 41, DF_RAD = 20, DF_G = 1;          // guard row: d_sym_fir_slide_pk loads [jb-m-1, jb+R+m]
constexpr int DF_CARRY = 2 * DF_RAD + 2 * DF_G;           // 42
constexpr int DF_CBROWS = DF_S + DF_CARRY;                // 106
constexpr int DF_CH = 32;                                 // rows per row-filter chunk (512 threads x 4 columns)
constexpr int DF_PITCH = 128;                             // floats per staged row: rows start on 512-byte boundaries
constexpr size_t df_lds(int sp) { return (size_t)(DF_CH * sp + 2 * DF_CBROWS * 64) * sizeof(float); }  // 70 656 B: 2 blocks / CU

template <typename T, int DF_SP, bool FUSED>
__global__ __launch_bounds__(64 * DF_NW, 2) void dog_fused(const T* __restrict__ src, int h, int w, int LY, int nstrips,
                                                          int nseg, const DogScalars* __restrict__ sc,
                                                          const float* __restrict__ klh, const float* __restrict__ klo_c,
                                                          const float* __restrict__ khi_c, float* __restrict__ diff,
                                                          float* __restrict__ part, int fused_scale)
{
    extern __shared__ float lds[];
    float* A = lds;                         // [DF_CH][DF_SP]   A[row][c] = v[c]
    float* CB = A + DF_CH * DF_SP;          // [2][DF_CBROWS][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // strips fastest: neighbouring strips (which share 40 of their 104 input columns) are consecutive items of one XCD
    const int item = d_xcd_work_item(blockIdx.x, nstrips * nseg);
    if (item >= nstrips * nseg) return;
    const int strip = item % nstrips, seg = item / nstrips;
    const int x0 = strip * 64, Y0 = seg * LY, Y1 = min(h, Y0 + LY);
    const float a = sc->a, b = sc->b;
    const int xa = d_reflect101(x0 - DF_RAD + lane, w);
    const int xb = d_reflect101(x0 - DF_RAD + 64 + min(lane, 2 * DF_RAD - 1), w);

    float va[4], vb[4];
    // wave wv stages rows 4 wv .. 4 wv + 3 of the chunk whose first row is global row g0 (reflect-101 in y)
    auto load_chunk = [&](int g0) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const T* srow = src + (size_t)d_reflect101(g0 + wv * 4 + k, h) * w;
            va[k] = (float)srow[xa];
            vb[k] = (float)srow[xb];
        }
    };
    auto commit_chunk = [&]() {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int row = wv * 4 + k;
            const float fa = d_scale(fused_scale, va[k], a, b), fb = d_scale(fused_scale, vb[k], a, b);
            A[row * DF_SP + lane] = fa;
            if (lane < 2 * DF_RAD) A[row * DF_SP + 64 + lane] = fb;
        }
    };
    // row filter of the staged chunk into CB rows [cb0, cb0 + nrows): thread = 4 consecutive columns of one row
    const int rrow = tid >> 4, rq = tid & 15;
    const ma_f2* __restrict__ kk = reinterpret_cast<const ma_f2*>(klh);   // klh[j] = (k_lo[j], k_hi[j])
    auto row_filter = [&](int cb0, int nrows) {
        if (rrow >= nrows) return;
        constexpr int KMAX = (DF_KS + 3 + 3) / 4 * 4;   // 44 inputs per thread
        ma_f2 E[KMAX / 2];                               // E[q] = (v[2q], v[2q+1])
        const float4* a4 = reinterpret_cast<const float4*>(A + rrow * DF_SP + 4 * rq);
#pragma unroll
        for (int q = 0; q < KMAX / 4; q++) {
            const float4 t = a4[q];
            E[2 * q] = (ma_f2){t.x, t.y}; E[2 * q + 1] = (ma_f2){t.z, t.w};
        }
        // One accumulator pair per output: (sigma_lo, sigma_hi).  Tap j multiplies the tap pair (k_lo[j], k_hi[j]) from
        // SGPRs by input v[o + j] broadcast to both halves -- a half-select of the register pair that holds it, no
        // move -- so odd and even offsets cost the same and every input is read from LDS once.
        auto V = [&](int i) { const float x = (i & 1) ? E[i >> 1].y : E[i >> 1].x; return (ma_f2){x, x}; };
        ma_f2 acc[4];
        {
            const ma_f2 k = kk[0];
#pragma unroll
            for (int o = 0; o < 4; o++) acc[o] = V(o) * k;
        }
#pragma unroll
        for (int j = 1; j < DF_KS; j++) {
            const ma_f2 k = kk[j];
#pragma unroll
            for (int o = 0; o < 4; o++) {
                if (FUSED) acc[o] = __builtin_elementwise_fma(V(o + j), k, acc[o]);
                else acc[o] = acc[o] + V(o + j) * k;
            }
        }
        const ma_f2 lo01 = {acc[0].x, acc[1].x}, lo23 = {acc[2].x, acc[3].x};
        const ma_f2 hi01 = {acc[0].y, acc[1].y}, hi23 = {acc[2].y, acc[3].y};
        float* clo = CB + (cb0 + rrow) * 64 + 4 * rq;
        *reinterpret_cast<float4*>(clo) = make_float4(lo01.x, lo01.y, lo23.x, lo23.y);
        *reinterpret_cast<float4*>(clo + DF_CBROWS * 64) = make_float4(hi01.x, hi01.y, hi23.x, hi23.y);
    };

    float lo = INFINITY, hi = -INFINITY;
    const int gtop = Y0 - DF_RAD - DF_G;      // global row held by CB row 0 during the first super-step
    // prologue: CB rows [0, CARRY)
    load_chunk(gtop);
    commit_chunk();
    __syncthreads();
    load_chunk(gtop + DF_CH);
    row_filter(0, DF_CH);
    __syncthreads();
    commit_chunk();
    __syncthreads();
    load_chunk(gtop + DF_CARRY);
    row_filter(DF_CH, DF_CARRY - DF_CH);
    __syncthreads();
    for (int ys = Y0; ys < Y1; ys += DF_S) {
        const int g = ys - DF_RAD - DF_G;     // global row of CB row 0
        commit_chunk();                        // rows g + CARRY .. + 31
        __syncthreads();
        load_chunk(g + DF_CARRY + DF_CH);
        row_filter(DF_CARRY, DF_CH);
        __syncthreads();
        commit_chunk();                        // rows g + CARRY + 32 .. + 63
        __syncthreads();
        if (ys + DF_S < Y1) load_chunk(g + DF_S + DF_CARRY);   // first chunk of the next super-step
        row_filter(DF_CARRY + DF_CH, DF_CH);
        __syncthreads();
        {
            float sl[DF_R], sh[DF_R];
            d_sym_fir_slide_pk<DF_R, FUSED, true>(CB + lane, DF_G + DF_RAD + wv * DF_R, DF_RAD, klo_c, sl);
            d_sym_fir_slide_pk<DF_R, FUSED, true>(CB + DF_CBROWS * 64 + lane, DF_G + DF_RAD + wv * DF_R, DF_RAD, khi_c, sh);
            const int x = x0 + lane;
#pragma unroll
            for (int q = 0; q < DF_R; q++) {
                const int y = ys + wv * DF_R + q;
                if (x < w && y < Y1) {
                    const float d = sh[q] - sl[q];
                    diff[(size_t)y * w + x] = d;
                    lo = fminf(lo, d); hi = fmaxf(hi, d);
                }
            }
        }
        __syncthreads();
        if (ys + DF_S < Y1) {
            // the halo of the next super-step: CB rows [S, S + CARRY) -> [0, CARRY) (disjoint, S >= CARRY)
            for (int e = tid; e < 2 * DF_CARRY * 16; e += 64 * DF_NW) {
                const int sg = e / (DF_CARRY * 16), rem = e - sg * (DF_CARRY * 16);
                float4* base = reinterpret_cast<float4*>(CB + sg * DF_CBROWS * 64);
                base[rem] = base[DF_S * 16 + rem];
            }
            __syncthreads();
        }
    }
    // block min / max through the (now idle) dynamic LDS: no static allocation, the block's budget is exactly 2 per CU
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_down(lo, off));
        hi = fmaxf(hi, __shfl_down(hi, off));
    }
    __syncthreads();
    if (lane == 0) { lds[wv] = lo; lds[DF_NW + wv] = hi; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < DF_NW; k++) { lo = fminf(lo, lds[k]); hi = fmaxf(hi, lds[DF_NW + k]); }
        part[(size_t)item * 2] = lo;
        part[(size_t)item * 2 + 1] = hi;
    }
}

// dst = saturate_u8(round_half_even(src*a + b)); a/b either immediate or from the DOG scalars.
// Groups of four pixels: 16-byte loads for float sources, one packed 4-byte store (byte stores from 64 lanes
// use a quarter of a store instruction's width).
template <typename T>
__global__ __launch_bounds__(256) void scale_to_u8(const T* __restrict__ src, size_t n, float a, float b,
                                                   const DogScalars* __restrict__ sc, uint8_t* __restrict__ dst,
                                                   int fused_scale)
{
    bool zero = false;
    if (sc) { a = sc->a8; b = sc->b8; zero = sc->src_max_is_zero != 0; }
    auto cvt = [&](float x) -> unsigned {
        float v = d_scale(fused_scale, x, a, b);
        return zero ? 0u : (unsigned)d_clamp(d_cvround(v), 0, 255);
    };
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    const bool vec = (((size_t)src & (4 * sizeof(T) - 1)) | ((size_t)dst & 3)) == 0;
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t i = tid; i < n4; i += stride) {
        float x0, x1, x2, x3;
        if constexpr (sizeof(T) == 4) {
            const float4 q = reinterpret_cast<const float4*>(src)[i];
            x0 = q.x; x1 = q.y; x2 = q.z; x3 = q.w;
        } else {
            const T* s = src + 4 * i;
            x0 = (float)s[0]; x1 = (float)s[1]; x2 = (float)s[2]; x3 = (float)s[3];
        }
        __builtin_nontemporal_store(cvt(x0) | (cvt(x1) << 8) | (cvt(x2) << 16) | (cvt(x3) << 24), &reinterpret_cast<unsigned*>(dst)[i]);
    }
    for (size_t i = n4 * 4 + tid; i < n; i += stride) dst[i] = (uint8_t)cvt((float)src[i]);
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

// Mat::convertTo(CV_32F) of an integer image: exact
template <typename T>
__global__ __launch_bounds__(256) void convert_f32_kernel(const T* __restrict__ src, size_t n, float* __restrict__ dst)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (float)src[i];
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

constexpr int DC_NW = 8;  // column pass: 64 columns x 8*R rows per block; R = 10 when r % 10 == 0 (no tail taps) else 16
                          // (8 waves x 10 rows measured 2 % faster than 4 x 20, profiles/r01_notes.md)

// launch geometry and workspace of one dog() call
struct DogPlan {
    bool fused;
    int nstrips, LY, nseg;
    dim3 cgrid;
    size_t nblk, npart, bytes;
};
DogPlan dog_plan(int h, int w, int low_sigma)
{
    DogPlan p;
    const size_t n = (size_t)h * w;
    const int ksize = low_sigma * 4 * 2 + 1;
    const int r = ksize / 2;
    const int DC_R = (r % 10 == 0) ? 10 : 16;
    p.fused = ksize == DF_KS;
    p.nstrips = (w + 63) / 64;
    // rows per block: enough blocks to fill 256 CUs x 2 a few times over, few enough that the 42 halo rows a block
    // filters before its first output row stay a small fraction
    {
        const int want_seg = std::max(1, 1536 / p.nstrips);
        p.LY = (int)ma_align_up((size_t)std::max(1, (h + want_seg - 1) / want_seg), DF_S);
        if (p.LY < 4 * DF_S) p.LY = std::min(4 * DF_S, (int)ma_align_up((size_t)h, DF_S));
    }
    p.nseg = (h + p.LY - 1) / p.LY;
    // workspace: [tlo, thi,] diff (f32 each), block partials, scalars
    p.cgrid = dim3((w + 63) / 64, (h + DC_NW * DC_R - 1) / (DC_NW * DC_R));
    p.nblk = p.fused ? (size_t)p.nstrips * p.nseg : (size_t)p.cgrid.x * p.cgrid.y;
    p.npart = p.nblk > MM_BLOCKS ? p.nblk : MM_BLOCKS;
    const size_t nimg = p.fused ? 1 : 3;
    p.bytes = n * nimg * sizeof(float) + p.npart * 2 * sizeof(float) + 256 + 16;
    return p;
}

} // namespace

size_t ma_dog_workspace_bytes(int h, int w, int low_sigma) { return dog_plan(h, w, low_sigma).bytes; }

int ma_launch_minmax_final(ma_ctx* ctx, const float* part, int nparts, float* out2)
{
    hipLaunchKernelGGL(minmax_final, dim3(1), dim3(MMF_T), 0, ctx->stream, part, nparts, out2, (DogScalars*)nullptr, 0,
                       (int*)nullptr);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

extern "C" {

int ma_minmax(ma_ctx* ctx, const void* src, int dtype, size_t n, double* mn_host, double* mx_host)
{
    MA_REQUIRE(ctx && src && mn_host && mx_host, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(n > 0, "empty array");
    MA_HIP(hipSetDevice(ctx->device));
    return minmax_host(ctx, src, dtype, n, mn_host, mx_host);
}

static int dog_u8_impl(ma_ctx* ctx, const void* src, int dtype, int h, int w, int low_sigma, int high_sigma, uint8_t* dst,
                       int* src_max_is_zero_host, const float* src_minmax_dev, int flags)
{
    MA_REQUIRE((flags & ~(MA_DOG_FUSED_BLUR | MA_DOG_FUSED_SCALE | MA_DOG_REPORT_ASYNC)) == 0, "unknown DOG flag");
    MA_REQUIRE(!(flags & MA_DOG_REPORT_ASYNC) || src_max_is_zero_host, "MA_DOG_REPORT_ASYNC needs a page-locked flag word");
    const bool fblur = (flags & MA_DOG_FUSED_BLUR) != 0;
    const int fscale = (flags & MA_DOG_FUSED_SCALE) != 0;
    MA_REQUIRE(ctx && src && dst, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(h > 0 && w > 0, "bad image size");
    MA_REQUIRE(low_sigma >= 1 && high_sigma >= 1, "sigmas must be >= 1");
    MA_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)h * w;
    const int ksize = low_sigma * 4 * 2 + 1;  // optflow_registrator.py:262
    const int r = ksize / 2;
    const size_t lds_rows = (size_t)DR * (256 + 2 * r) * sizeof(float);
    const int DC_R = (r % 10 == 0) ? 10 : 16;
    const size_t lds_cols = (size_t)(DC_NW * DC_R + 2 * r + 4) * 64 * sizeof(float);
    MA_REQUIRE(lds_rows <= 160 * 1024 && lds_cols <= 160 * 1024, "low_sigma too large for the LDS-staged DOG kernels");

    std::vector<float> klo, khi;
    gaussian_kernel(ksize, low_sigma, klo);
    gaussian_kernel(ksize, high_sigma, khi);
    const float *dlo = nullptr, *dloc = nullptr, *dhic = nullptr;
    const uint64_t key = ((uint64_t)ksize << 16);
    std::vector<float> klh(2 * (size_t)ksize);  // row pass: (lo, hi) interleaved
    for (int j = 0; j < ksize; j++) { klh[2 * j] = klo[j]; klh[2 * j + 1] = khi[j]; }
    MA_TRY(ma_const_table(ctx, ((uint64_t)2 << 56) | key | ((uint64_t)low_sigma << 8) | (uint64_t)high_sigma, klh.data(),
                          klh.size(), &dlo));
    // centre-first halves for the symmetric column pass: c[i] = k[r + i]
    std::vector<float> clo = ma_layout_taps(std::vector<float>(klo.begin() + r, klo.end()));
    std::vector<float> chi = ma_layout_taps(std::vector<float>(khi.begin() + r, khi.end()));
    MA_TRY(ma_const_table(ctx, ((uint64_t)3 << 56) | key | (uint64_t)low_sigma, clo.data(), clo.size(), &dloc));
    MA_TRY(ma_const_table(ctx, ((uint64_t)3 << 56) | key | (uint64_t)high_sigma, chi.data(), chi.size(), &dhic));

    // The reference's kernel size (41: sigmas 5 / 9) takes the fused kernel; other sizes the two-kernel chain.
    const DogPlan pl = dog_plan(h, w, low_sigma);
    const bool fused = pl.fused;
    const int nstrips = pl.nstrips, LY = pl.LY, nseg = pl.nseg;
    const size_t nblk = pl.nblk, npart = pl.npart, bytes = pl.bytes;
    MA_TRY(ma_ws_reserve(ctx, bytes));
    float* tlo = (float*)ctx->ws;
    float* thi = tlo + n;
    float* diff = fused ? (float*)ctx->ws : thi + n;
    float* part = (float*)ma_align_up((size_t)(diff + n), 16);  // read as float2 pairs by minmax_final
    DogScalars* sc = (DogScalars*)(part + npart * 2);
    MA_REQUIRE((h + DR - 1) / DR <= 65535, "image too tall");

    MaProfScope ps(ctx, MA_K_DOG, (double)n);
    // the scalars of the two normalisations are computed by the last thread of the reduction they follow from
    if (src_minmax_dev) {  // the producer of `src` already reduced it
        hipLaunchKernelGGL(dog_params_in, dim3(1), dim3(1), 0, ctx->stream, sc, src_minmax_dev, ctx->dog_sticky);
    } else {
        MA_TRY(launch_minmax(ctx, src, dtype, n, part, nullptr, sc, 1, ctx->dog_sticky));
    }
    if (fused) {
        const dim3 grid(ma_xcd_grid((long long)nblk)), block(64 * DF_NW);
#define MA_DOG_FUSED(T, SP)                                                                                                  \
    do {                                                                                                                     \
        if (fblur) hipLaunchKernelGGL((dog_fused<T, SP, true>), grid, block, df_lds(SP), ctx->stream, (const T*)src, h, w,   \
                                      LY, nstrips, nseg, sc, dlo, dloc, dhic, diff, part, fscale);                          \
        else hipLaunchKernelGGL((dog_fused<T, SP, false>), grid, block, df_lds(SP), ctx->stream, (const T*)src, h, w, LY,    \
                                nstrips, nseg, sc, dlo, dloc, dhic, diff, part, fscale);                                    \
    } while (0)
        if (dtype == MA_U8) MA_DOG_FUSED(uint8_t, DF_PITCH);
        else if (dtype == MA_U16) MA_DOG_FUSED(uint16_t, DF_PITCH);
        else MA_DOG_FUSED(float, DF_PITCH);
#undef MA_DOG_FUSED
    } else {
        {
            dim3 grid((w + 255) / 256, (h + DR - 1) / DR), block(256);
#define MA_DOG_ROWS(T)                                                                                                       \
    do {                                                                                                                     \
        if (fblur) hipLaunchKernelGGL((dog_rows<T, true>), grid, block, lds_rows, ctx->stream, (const T*)src, h, w, ksize,   \
                                      sc, dlo, tlo, thi, fscale);                                                           \
        else hipLaunchKernelGGL((dog_rows<T, false>), grid, block, lds_rows, ctx->stream, (const T*)src, h, w, ksize, sc,    \
                                dlo, tlo, thi, fscale);                                                                     \
    } while (0)
            if (dtype == MA_U8) MA_DOG_ROWS(uint8_t);
            else if (dtype == MA_U16) MA_DOG_ROWS(uint16_t);
            else MA_DOG_ROWS(float);
#undef MA_DOG_ROWS
        }
#define MA_DOG_COLS(RR, FF) hipLaunchKernelGGL((dog_cols_diff<RR, DC_NW, FF>), dim3(ma_xcd_grid((long long)nblk)), dim3(64 * DC_NW), \
                                               lds_cols, ctx->stream, tlo, thi, h, w, ksize, dloc, dhic, diff, part)
        if (DC_R == 10) { if (fblur) MA_DOG_COLS(10, true); else MA_DOG_COLS(10, false); }
        else { if (fblur) MA_DOG_COLS(16, true); else MA_DOG_COLS(16, false); }
#undef MA_DOG_COLS
    }
    hipLaunchKernelGGL(minmax_final, dim3(1), dim3(MMF_T), 0, ctx->stream, part, (int)nblk, (float*)nullptr, sc, 2,
                       (int*)nullptr);
    hipLaunchKernelGGL((scale_to_u8<float>), dim3(grid_for(n)), dim3(256), 0, ctx->stream, diff, n, 0.f, 0.f, sc, dst, fscale);
    MA_HIP(hipGetLastError());
    if (src_max_is_zero_host && (flags & MA_DOG_REPORT_ASYNC)) {
        // stream ordered: the scalars live in the workspace, which later calls reuse only behind this copy
        MA_HIP(hipMemcpyAsync(src_max_is_zero_host, &sc->src_max_is_zero, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    } else if (src_max_is_zero_host) {
        MA_TRY(ma_pinned_reserve(ctx, 64));
        MA_HIP(hipMemcpyAsync(ctx->pinned, &sc->src_max_is_zero, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        MA_HIP(hipStreamSynchronize(ctx->stream));
        *src_max_is_zero_host = *(int*)ctx->pinned;
    }
    return MA_OK;
}

int ma_dog_u8(ma_ctx* ctx, const void* src, int dtype, int h, int w, int low_sigma, int high_sigma, uint8_t* dst,
              int* src_max_is_zero_host)
{
    return dog_u8_impl(ctx, src, dtype, h, w, low_sigma, high_sigma, dst, src_max_is_zero_host, nullptr, 0);
}

int ma_dog_u8_ex(ma_ctx* ctx, const void* src, int dtype, int h, int w, int low_sigma, int high_sigma, int flags,
                 const float* src_minmax_dev, uint8_t* dst, int* src_max_is_zero_host)
{
    return dog_u8_impl(ctx, src, dtype, h, w, low_sigma, high_sigma, dst, src_max_is_zero_host, src_minmax_dev, flags);
}

int ma_dog_u8_minmax(ma_ctx* ctx, const void* src, int dtype, int h, int w, int low_sigma, int high_sigma,
                     const float* src_minmax_dev, uint8_t* dst, int* src_max_is_zero_host)
{
    MA_REQUIRE(src_minmax_dev, "NULL argument");
    return dog_u8_impl(ctx, src, dtype, h, w, low_sigma, high_sigma, dst, src_max_is_zero_host, src_minmax_dev, 0);
}

int ma_convert_f32(ma_ctx* ctx, const void* src, int dtype, size_t n, float* dst)
{
    MA_REQUIRE(ctx && src && dst, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(n > 0, "empty array");
    MA_HIP(hipSetDevice(ctx->device));
    if (dtype == MA_F32) return ma_memcpy_d2d(ctx, dst, src, n * sizeof(float));
    MaProfScope ps(ctx, MA_K_OTHER, (double)n);
    dim3 grid(grid_for(n)), block(256);
    if (dtype == MA_U8) hipLaunchKernelGGL((convert_f32_kernel<uint8_t>), grid, block, 0, ctx->stream, (const uint8_t*)src, n, dst);
    else hipLaunchKernelGGL((convert_f32_kernel<uint16_t>), grid, block, 0, ctx->stream, (const uint16_t*)src, n, dst);
    MA_HIP(hipGetLastError());
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
    MA_TRY(minmax_host(ctx, src, dtype, n, &smin, &smax));
    double scale = 255. * (smax - smin > DBL_EPSILON ? 1. / (smax - smin) : 0);
    double shift = 0. - smin * scale;
    MaProfScope ps(ctx, MA_K_OTHER, (double)n);
    dim3 grid(grid_for(n)), block(256);
    const DogScalars* none = nullptr;
    if (dtype == MA_U8) hipLaunchKernelGGL((scale_to_u8<uint8_t>), grid, block, 0, ctx->stream, (const uint8_t*)src, n, (float)scale, (float)shift, none, dst, 0);
    else if (dtype == MA_U16) hipLaunchKernelGGL((scale_to_u8<uint16_t>), grid, block, 0, ctx->stream, (const uint16_t*)src, n, (float)scale, (float)shift, none, dst, 0);
    else hipLaunchKernelGGL((scale_to_u8<float>), grid, block, 0, ctx->stream, (const float*)src, n, (float)scale, (float)shift, none, dst, 0);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

} // extern "C"
