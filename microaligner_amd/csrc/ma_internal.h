// Internal declarations shared by the HIP translation units of libmicroaligner_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/microaligner_hip.h"

void ma_set_error(const char* fmt, ...);

#define MA_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ma_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return MA_EHIP;                                                                   \
        }                                                                                     \
    } while (0)

#define MA_REQUIRE(cond, msg)                              \
    do {                                                   \
        if (!(cond)) {                                     \
            ma_set_error("invalid argument: %s", msg);     \
            return MA_EINVAL;                              \
        }                                                  \
    } while (0)

#define MA_TRY(expr)            \
    do {                        \
        int _rc = (expr);       \
        if (_rc != MA_OK) return _rc; \
    } while (0)

struct ma_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // grow-only device workspace reused by every call (tile batches, DOG temporaries, histograms)
    void* ws = nullptr;
    size_t ws_bytes = 0;
    size_t ws_limit = (size_t)48 << 30;
    // small pinned host buffer for scalar results (min/max, NMI scores)
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    // small device buffer for filter taps / scalars
    void* dconst = nullptr;
    size_t dconst_bytes = 0;
    // per-kernel event accounting
    bool profile = false;
    struct Rec { hipEvent_t a, b; int id; };
    std::vector<Rec> pending;
    std::vector<hipEvent_t> free_events;
    double prof_ms[MA_K_COUNT] = {0};
    long long prof_n[MA_K_COUNT] = {0};
    double prof_px[MA_K_COUNT] = {0};
};

int ma_ws_reserve(ma_ctx* ctx, size_t bytes);      // ensures ctx->ws has >= bytes
int ma_pinned_reserve(ma_ctx* ctx, size_t bytes);
int ma_dconst_reserve(ma_ctx* ctx, size_t bytes);
// immutable device copies of small float tables (filter taps), cached per (device, key) for the process lifetime
int ma_const_table(ma_ctx* ctx, uint64_t key, const float* host, size_t n, const float** dev);

// RAII-ish launch bracket for per-kernel accounting
struct MaProfScope {
    ma_ctx* ctx; int id; hipEvent_t a = nullptr, b = nullptr; bool on;
    MaProfScope(ma_ctx* c, int kid, double px);
    ~MaProfScope();
};
int ma_profile_flush(ma_ctx* ctx);

static inline size_t ma_esize(int dtype) { return dtype == MA_U8 ? 1 : (dtype == MA_U16 ? 2 : 4); }
static inline size_t ma_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- device helpers ---------------------------------------------------------
__device__ __forceinline__ int d_reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}
__device__ __forceinline__ int d_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// cvRound semantics of the x86 cvtss2si: round-half-even, out of range/NaN -> INT_MIN
__device__ __forceinline__ int d_cvround(float v)
{
    if (!(fabsf(v) < 2147483648.0f)) return (int)0x80000000;
    return (int)rintf(v);
}
__device__ __forceinline__ int d_cvfloor(float v)
{
    if (!(fabsf(v) < 2147483648.0f)) return (int)0x80000000;
    return (int)floorf(v);
}

template <typename T>
__device__ __forceinline__ float d_to_f32(T v) { return (float)v; }

// multiply-add with selectable rounding model (see MA_FB_MULADD_FUSED)
template <bool FUSED>
__device__ __forceinline__ float d_muladd(float a, float b, float c)
{
    if (FUSED) return __fmaf_rn(a, b, c);
    return __fadd_rn(__fmul_rn(a, b), c);
}

// Symmetric FIR along one axis with a rotating register window (used by the Farneback window blur and the DOG
// column pass).  `col` points at this thread's line in LDS, element j is col[j * stride].  For R consecutive
// outputs starting at element jb:   s = c * k0;   s = (x[+i] + x[-i]) * k_i + s   for i = 1..m   (ascending i).
// Full groups of R taps rotate the +i / -i windows through statically indexed registers (two LDS reads per
// tap); the remaining m % R taps are read straight from LDS.  `last` is the largest valid element index.
template <int R, bool FUSED>
__device__ __forceinline__ void d_sym_fir_slide(const float* __restrict__ col, const int stride, const int jb,
                                                const int m, const int last, const float* __restrict__ taps,
                                                float acc[R])
{
    float P[R], Q[R];
    const float k0 = taps[0];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = col[(jb + r) * stride] * k0;
#pragma unroll
    for (int o = 1; o <= R; o++) P[o % R] = col[min(jb + o, last) * stride];
#pragma unroll
    for (int o = -1; o <= R - 2; o++) Q[(o + R) % R] = col[max(jb + o, 0) * stride];
    const int full = m / R;
    for (int g = 0; g < full; g++) {
#pragma unroll
        for (int ii = 0; ii < R; ii++) {
            const int i = g * R + ii + 1;
            const float ki = taps[i];
#pragma unroll
            for (int r = 0; r < R; r++)
                acc[r] = d_muladd<FUSED>(P[(r + ii + 1) % R] + Q[((r - ii - 1) % R + R) % R], ki, acc[r]);
            P[(ii + 1) % R] = col[min(jb + i + R, last) * stride];
            Q[((-ii - 2) % R + R) % R] = col[max(jb - i - 1, 0) * stride];
        }
    }
    for (int i = full * R + 1; i <= m; i++) {
        const float ki = taps[i];
#pragma unroll
        for (int r = 0; r < R; r++)
            acc[r] = d_muladd<FUSED>(col[(jb + r + i) * stride] + col[(jb + r - i) * stride], ki, acc[r]);
    }
}

// Tile geometry shared by the tiled kernels (slicer.py / stitcher.py semantics).
struct MaTiling {
    int H, W;      // image size
    int T, ov;     // tile size and overlap; T == 0 -> a single untiled window == the image
    int ntx, nty;  // tiles per axis
    int Ph, Pw;    // window (padded tile) height/width
};

static inline MaTiling ma_make_tiling(int H, int W, int tile, int overlap)
{
    MaTiling g;
    g.H = H; g.W = W;
    if (tile <= 0) {
        g.T = 0; g.ov = 0; g.ntx = g.nty = 1; g.Ph = H; g.Pw = W;
    } else {
        g.T = tile; g.ov = overlap;
        g.ntx = (W + tile - 1) / tile; g.nty = (H + tile - 1) / tile;
        g.Ph = g.Pw = tile + 2 * overlap;
    }
    return g;
}
