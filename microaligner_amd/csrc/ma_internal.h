// Internal declarations shared by the HIP translation units of libmicroaligner_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <functional>
#include <vector>

#include "../../include/microaligner_hip.h"

void ma_set_error(const char* fmt, ...);
struct MaStageRing;   // ring of page-locked chunks for staged transfers of pageable memory (ma_api.hip)

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
    // size-bucketed cache of device buffers for the intermediates of ma_optflow_register (pyramids, DOG images, flows):
    // stream-ordered reuse on the ctx stream, returned to the driver by ma_ctx_trim / ma_ctx_destroy
    std::multimap<size_t, void*> pool_free;
    std::unordered_map<void*, size_t> pool_live;
    // bytes moved by the explicit host <-> device copies of this ctx (ma_memcpy_*, ma_engine_memcpy_*,
    // ma_warp_pages_host); atomic: the transfer engines are driven from their own host threads
    std::atomic<unsigned long long> h2d_bytes{0}, d2h_bytes{0};
    // transfer engines (MA_ENGINE_H2D / MA_ENGINE_D2H): streams created on first use under `mu`
    hipStream_t engine[3] = {nullptr, nullptr, nullptr};
    std::mutex mu;
    // staging rings per engine and direction ([engine][0] host -> device, [engine][1] device -> host)
    MaStageRing* stage[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    // a ring serves one copy at a time: a second host thread on the same engine and direction (ctx.asdevice() / .numpy() on
    // the default context while parallel.stream_pairs drives it) waits its turn instead of sharing chunks
    std::mutex stage_mu[3][2];
    // MA_OPT_COMPANION_STREAM
    bool companion = true;
    // MA_OPT_WARP_BAND_BYTES
    size_t warp_band_bytes = (size_t)32 << 20;
    // device int that the dog() chain sets when an input has max() == 0 but is not all zero (see d_dog_params_in);
    // NULL outside ma_optflow_register
    int* dog_sticky = nullptr;
    int* dog_sticky_buf = nullptr;   // the allocation behind it (owned by the ctx)
    // two page-locked ints for the max() == 0 reports of the dog() calls inside ma_feature_round (written in stream order,
    // read after the round's synchronisation); allocated on first use, owned by the ctx
    int* round_flags = nullptr;
    // companion ctx (own stream and workspace, same device) for the work of ma_optflow_register that does not depend on
    // the flow -- dog(ref) and dog(mov) of every level -- and the events that order the two streams; created on first use
    ma_ctx* side = nullptr;
    std::vector<hipEvent_t> sync_events;
};
// the companion ctx of `ctx` (created on first use; nullptr + error set on failure)
ma_ctx* ma_ctx_side(ma_ctx* ctx);
void ma_stage_rings_destroy(ma_ctx* ctx);
// stream of an engine (MA_ENGINE_*); nullptr + error set on failure
// gridDim.y limit: kernels that put rows on the y axis take R rows per block, an image may have MA_GRID_Y_MAX * R rows
constexpr int MA_GRID_Y_MAX = 65535;
hipStream_t ma_engine_stream(ma_ctx* ctx, int engine);
// One host <-> device copy on a transfer engine whose caller wants to act between its pieces (the bands of the page-warp
// driver) without giving up the overlap of staging copy and DMA at every piece, as a sequence of ma_engine_memcpy_* calls
// would.  cuts[0 .. ncuts): ascending end offsets of the pieces, cuts[ncuts - 1] == bytes.
//   h2d: enqueued(j) is called, in order, as soon as every byte of piece j has been handed to the engine's stream (an event
//        recorded in the callback fires when piece j is in HBM);
//   d2h: before(j) is called, in order, before the first byte of piece j is read (the callback makes the engine's stream wait
//        for whatever produces the piece).
// Both return when the whole copy is complete; a callback's non-zero code stops the copy and is returned.  h2d with
// wait == false returns as soon as the last byte has been handed to the stream (a pageable source has been read completely
// by then, a page-locked one must stay untouched until the stream has been waited for): a sequence of pages keeps the engine
// busy across the page boundaries, the caller ends it with ma_engine_sync.
typedef std::function<int(int)> MaPieceFn;
int ma_engine_h2d_pieces(ma_ctx* ctx, int engine, void* dst, const void* src_host, size_t bytes, const size_t* cuts, int ncuts,
                         const MaPieceFn& enqueued, bool wait = true);
int ma_engine_d2h_pieces(ma_ctx* ctx, int engine, void* dst_host, const void* src, size_t bytes, const size_t* cuts, int ncuts,
                         const MaPieceFn& before);
// event i of the ctx's pool of timing-free events for ordering its two streams (created on demand; nullptr on failure)
hipEvent_t ma_ctx_sync_event(ma_ctx* ctx, size_t i);
// fixed indices beyond the events ma_optflow_register uses (2 per level + 1 <= 65)
constexpr size_t MA_EV_WARP_PAGES = 125, MA_EV_COPY_IN = 126, MA_EV_COPY_OUT = 127;

// device buffer from the ctx cache (64 KiB buckets); nullptr + error set on failure
void* ma_pool_alloc(ma_ctx* ctx, size_t bytes);
void ma_pool_free(ma_ctx* ctx, void* p);
// NMI of consecutive chunks, enqueue only: scores land in scores_pinned_host (page-locked) once the ctx stream has
// passed this point (nmi.hip)
int ma_nmi_u8_enqueue(ma_ctx* ctx, const uint8_t* a, const uint8_t* b, size_t n, size_t chunk, double* scores_pinned_host,
                      int max_scores, int* n_scores);
// the two halves of the gate in one pair of launches: NMI(a, b0) -> scores0, NMI(a, b1) -> scores1 (b1 may be NULL)
int ma_nmi_u8_enqueue2(ma_ctx* ctx, const uint8_t* a, const uint8_t* b0, const uint8_t* b1, size_t n, size_t chunk,
                       double* scores0_pinned_host, double* scores1_pinned_host, int max_scores, int* n_scores);

int ma_ws_reserve(ma_ctx* ctx, size_t bytes);      // ensures ctx->ws has >= bytes
// workspace one dog() call of an (h, w) image takes from ctx->ws (dog.hip)
size_t ma_dog_workspace_bytes(int h, int w, int low_sigma);
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

// Global access as wave-uniform base + 32-bit per-lane BYTE offset: lets the compiler pick the
// `global_load_dword v, voffset, s[base:base+1]` form (no 64-bit VALU address arithmetic per access).
__device__ __forceinline__ float d_ldg(const float* ubase, unsigned byte_off)
{
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ubase) + byte_off);
}
__device__ __forceinline__ void d_stg(float* ubase, unsigned byte_off, float v)
{
    *reinterpret_cast<float*>(reinterpret_cast<char*>(ubase) + byte_off) = v;
}

// multiply-add with selectable rounding model (see MA_FB_MULADD_FUSED)
template <bool FUSED>
__device__ __forceinline__ float d_muladd(float a, float b, float c)
{
    if (FUSED) return __fmaf_rn(a, b, c);
    return __fadd_rn(__fmul_rn(a, b), c);
}

// Device layout of a symmetric tap table k[0..m]: t[0] = k0, t[8 + (i-1)] = k_i for i >= 1, so that every group of
// 4 or 8 consecutive taps starting at i = 1 (mod 8) is 16/32-byte aligned and loads with one s_load_dwordx4/x8.
#define MA_TAP(t, i) ((t)[7 + (i)])
static inline std::vector<float> ma_layout_taps(const std::vector<float>& k)
{
    std::vector<float> t(8 + (k.size() > 0 ? k.size() - 1 : 0) + 8, 0.f);
    t[0] = k[0];
    for (size_t i = 1; i < k.size(); i++) t[7 + i] = k[i];
    return t;
}

// Symmetric FIR along one axis with a rotating register window (used by the Farneback window blur and the DOG
// column pass).  `col` points at this thread's line in LDS, element j is col[j * stride].  For R consecutive
// outputs starting at element jb:   s = c * k0;   s = (x[+i] + x[-i]) * k_i + s   for i = 1..m   (ascending i).
// Full groups of R taps rotate the +i / -i windows through statically indexed registers (two LDS reads per
// tap); the remaining m % R taps are read straight from LDS.  The caller guarantees that elements
// [jb - m - 2, jb + R + m + 1] are addressable (two guard elements beyond the halo on either side; their
// values are loaded into the rotating windows but never used).
template <int R, bool FUSED>
__device__ __forceinline__ void d_sym_fir_slide(const float* __restrict__ col, const int stride, const int jb,
                                                const int m, const float* __restrict__ taps, float acc[R])
{
    float P[R], Q[R];
    const float k0 = taps[0];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = col[(jb + r) * stride] * k0;
#pragma unroll
    for (int o = 1; o <= R; o++) P[o % R] = col[(jb + o) * stride];
#pragma unroll
    for (int o = -1; o <= R - 2; o++) Q[(o + R) % R] = col[(jb + o) * stride];
    const int full = m / R;
    for (int g = 0; g < full; g++) {
#pragma unroll
        for (int ii = 0; ii < R; ii++) {
            const int i = g * R + ii + 1;
            const float ki = MA_TAP(taps, i);
#pragma unroll
            for (int r = 0; r < R; r++)
                acc[r] = d_muladd<FUSED>(P[(r + ii + 1) % R] + Q[((r - ii - 1) % R + R) % R], ki, acc[r]);
            P[(ii + 1) % R] = col[(jb + i + R) * stride];
            Q[((-ii - 2) % R + R) % R] = col[(jb - i - 1) * stride];
        }
    }
    for (int i = full * R + 1; i <= m; i++) {
        const float ki = MA_TAP(taps, i);
#pragma unroll
        for (int r = 0; r < R; r++)
            acc[r] = d_muladd<FUSED>(col[(jb + r + i) * stride] + col[(jb + r - i) * stride], ki, acc[r]);
    }
}

// Packed-math form of d_sym_fir_slide: v_pk_add_f32 / v_pk_mul_f32 (or v_pk_fma_f32) work on register pairs
// and issue at ~1.2x the cost of a scalar op for twice the work on gfx950 (profiles/r01_ubench_valu.txt).
// Outputs are paired H = R/2 apart: pair r = (out[r], out[r+H]).  At tap i it needs the input pairs
// (x[r+i], x[r+H+i]) and (x[r-i], x[r+H-i]); with windows of pairs W[o] = (x[o], x[o+H]) both are W[r+i] and
// W[r-i], so ONE alignment class suffices: the +i window holds o in [i, i+H-1], the -i window o in [-i, H-1-i],
// each rotates one slot per tap through statically indexed registers (period H taps) and loads one new pair per
// tap.  3R registers, like the scalar form.  Per component the operations and their order are exactly those of
// the scalar form: results are bit-identical.
// The pair loads are ds_read2(st64)_b32 written as inline asm: left to itself the compiler re-pairs the LDS reads
// by adjacency and rebuilds the H-apart pairs with ~10 v_mov per tap.  Each tap issues its two loads first and
// waits for them (s_waitcnt lgkmcnt(0), tied to the loaded registers) after its arithmetic.
// ST64 = true : element stride is 64 floats (column of a [rows][64] LDS tile)  -> ds_read2st64_b32
// ST64 = false: element stride is 1 float  (row of an LDS tile)                 -> ds_read2_b32
// Addressable range required from the caller: [jb - m - 1, jb + R + m].
typedef float ma_f2 __attribute__((ext_vector_type(2)));
typedef unsigned ma_u2 __attribute__((ext_vector_type(2)));

template <bool ST64, int O0, int O1>
__device__ __forceinline__ ma_f2 d_lds_read_pair(unsigned addr)
{
    static_assert(O0 >= 0 && O0 < 256 && O1 >= 0 && O1 < 256, "ds_read2 offsets are 8 bit");
    ma_f2 v;
    if (ST64) asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1));
    else asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1));
    return v;
}

template <int H, bool FUSED, bool ST64, int II>
__device__ __forceinline__ void d_fir_tap_pk(ma_f2 (&A)[H], ma_f2 (&P)[H], ma_f2 (&Q)[H], const float ki,
                                             const unsigned addrP, const unsigned addrQ)
{
    // group-relative element offsets: +window entry o = i + H, i = g*H + II + 1 (addrP points at jb + g*H);
    // -window entry o = -i - 1 (addrQ points at jb - g*H - H - 1).
    // The outgoing +pair (slot SP) is read only by output pair 0 and the outgoing -pair (slot SQ) only by output
    // pair H-1: those two sums go first, then the new pairs are loaded INTO THE SAME registers, so the windows
    // never change registers across the loop back-edge (otherwise: 4 v_mov per tap).
    constexpr int SP = (II + 1) % H, SQ = ((-II - 2) % H + H) % H;
    static_assert(((H - 1 - II - 1) % H + H) % H == SQ, "slot bookkeeping");
    const ma_f2 kk = {ki, ki};
    ma_f2 t[H];
    t[0] = P[SP] + Q[((0 - II - 1) % H + H) % H];
    if (H > 1) t[H - 1] = P[(H - 1 + II + 1) % H] + Q[SQ];
    __builtin_amdgcn_sched_barrier(0);
    P[SP] = d_lds_read_pair<ST64, II + 1 + H, II + 1 + 2 * H>(addrP);
    Q[SQ] = d_lds_read_pair<ST64, H - 1 - II, 2 * H - 1 - II>(addrQ);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 1; r < H - 1; r++) t[r] = P[(r + II + 1) % H] + Q[((r - II - 1) % H + H) % H];
    if (FUSED) {
#pragma unroll
        for (int r = 0; r < H; r++) A[r] = __builtin_elementwise_fma(t[r], kk, A[r]);
    } else {
#pragma unroll
        for (int r = 0; r < H; r++) t[r] = t[r] * kk;
#pragma unroll
        for (int r = 0; r < H; r++) A[r] = t[r] + A[r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(P[SP]), "+v"(Q[SQ]));
}

template <int H, bool FUSED, bool ST64, int... II>
__device__ __forceinline__ void d_fir_group_pk(ma_f2 (&A)[H], ma_f2 (&P)[H], ma_f2 (&Q)[H], const float* __restrict__ k,
                                               const unsigned addrP, const unsigned addrQ,
                                               std::integer_sequence<int, II...>)
{
    (d_fir_tap_pk<H, FUSED, ST64, II>(A, P, Q, k[II], addrP, addrQ), ...);
}

template <int R, bool FUSED, bool ST64>
__device__ __forceinline__ void d_sym_fir_slide_pk(const float* __restrict__ col, const int jb, const int m,
                                                   const float* __restrict__ taps, float acc[R])
{
    static_assert(R % 2 == 0, "R must be even");
    constexpr int H = R / 2;
    constexpr int stride = ST64 ? 64 : 1;
    auto W2 = [&](int o) { return (ma_f2){col[(jb + o) * stride], col[(jb + o + H) * stride]}; };
    ma_f2 A[H], P[H], Q[H];
    const float k0 = taps[0];
#pragma unroll
    for (int r = 0; r < H; r++) A[r] = W2(r) * (ma_f2){k0, k0};
#pragma unroll
    for (int o = 1; o <= H; o++) P[o % H] = W2(o);               // tap 1: o in [1, H]
#pragma unroll
    for (int o = -1; o <= H - 2; o++) Q[(o + H) % H] = W2(o);    // tap 1: o in [-1, H-2]
    const int full = m / H;
    // LDS byte addresses (low 32 bits of a flat shared pointer are the LDS offset)
    unsigned addrP = (unsigned)(size_t)(col + jb * stride);
    unsigned addrQ = (unsigned)(size_t)(col + (jb - H - 1) * stride);
    for (int g = 0; g < full; g++) {
        float kg[H];  // this group's taps: one aligned scalar load
#pragma unroll
        for (int j = 0; j < H; j++) kg[j] = MA_TAP(taps, g * H + 1 + j);
        d_fir_group_pk<H, FUSED, ST64>(A, P, Q, kg, addrP, addrQ, std::make_integer_sequence<int, H>{});
        addrP += H * stride * 4;
        addrQ -= H * stride * 4;
    }
#pragma unroll
    for (int r = 0; r < H; r++) { acc[r] = A[r].x; acc[r + H] = A[r].y; }
    for (int i = full * H + 1; i <= m; i++) {
        const float ki = MA_TAP(taps, i);
#pragma unroll
        for (int r = 0; r < R; r++)
            acc[r] = d_muladd<FUSED>(col[(jb + r + i) * stride] + col[(jb + r - i) * stride], ki, acc[r]);
    }
}

// d_sym_fir_slide_pk over a RING of C rows (ST64 layout: row stride 64 floats) whose first 3*H + 1 rows are mirrored
// behind row C - 1: element j of the line is at ring row j mod C, and any run of up to 3*H + 1 consecutive elements
// that starts inside the ring is contiguous in LDS (it may run into the mirror).  jb: ring row of the first output,
// 0 <= jb < C, wave-uniform.  Requires m % H == 0.  Same operations in the same order as the linear form.
template <int R, bool FUSED>
__device__ __forceinline__ void d_sym_fir_ring_pk(const float* __restrict__ col, const int jb, const int C, const int m,
                                                  const float* __restrict__ taps, float acc[R])
{
    static_assert(R % 2 == 0, "R must be even");
    constexpr int H = R / 2;
    auto W2 = [&](int o) {
        int i = jb + o;
        if (i < 0) i += C;
        return (ma_f2){col[i * 64], col[(i + H) * 64]};
    };
    ma_f2 A[H], P[H], Q[H];
    const float k0 = taps[0];
#pragma unroll
    for (int r = 0; r < H; r++) A[r] = W2(r) * (ma_f2){k0, k0};
#pragma unroll
    for (int o = 1; o <= H; o++) P[o % H] = W2(o);
#pragma unroll
    for (int o = -1; o <= H - 2; o++) Q[(o + H) % H] = W2(o);
    const unsigned base = (unsigned)(size_t)col;
    int rowP = jb, rowQ = jb - H - 1;
    if (rowQ < 0) rowQ += C;
    const int full = m / H;
    for (int g = 0; g < full; g++) {
        float kg[H];
#pragma unroll
        for (int j = 0; j < H; j++) kg[j] = MA_TAP(taps, g * H + 1 + j);
        d_fir_group_pk<H, FUSED, true>(A, P, Q, kg, base + (unsigned)rowP * 256u, base + (unsigned)rowQ * 256u,
                                       std::make_integer_sequence<int, H>{});
        rowP += H;
        if (rowP >= C) rowP -= C;
        rowQ -= H;
        if (rowQ < 0) rowQ += C;
    }
#pragma unroll
    for (int r = 0; r < H; r++) { acc[r] = A[r].x; acc[r + H] = A[r].y; }
}

// Block-wide (min, max) of per-thread values -> part[0], part[1] (thread 0 writes).  All threads of the block must
// call it; blocks of up to 1024 threads.
__device__ __forceinline__ void d_block_minmax(float lo, float hi, float* part)
{
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_down(lo, off));
        hi = fmaxf(hi, __shfl_down(hi, off));
    }
    __shared__ float s_lo[16], s_hi[16];
    const int nw = (blockDim.x * blockDim.y * blockDim.z + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < nw; k++) { lo = fminf(lo, s_lo[k]); hi = fmaxf(hi, s_hi[k]); }
        part[0] = lo; part[1] = hi;
    }
}
// folds nparts (min, max) pairs on the device into out2[0..1] (dog.hip); stream ordered
int ma_launch_minmax_final(ma_ctx* ctx, const float* part, int nparts, float* out2);

// XCD-aware work mapping (speed only, never correctness).  Workgroups of a 1-D grid are dealt round-robin to the
// 8 XCDs (block b -> XCD b % 8), each with its own 4 MiB L2.  Stencil blocks that share a halo should therefore
// be consecutive *within one XCD*: block b takes work item (b % 8) * ceil(N/8) + b / 8, so every XCD walks a
// contiguous range of the work list in dispatch order.  Launch ma_xcd_grid(N) blocks; items >= N are skipped.
__device__ __forceinline__ int d_xcd_work_item(int b, int n_items)
{
    const int per = (n_items + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}
static inline unsigned ma_xcd_grid(long long n_items) { return (unsigned)(((n_items + 7) / 8) * 8); }

// Slot -> unit permutation for work lists whose units (windows, planes) have unequal cost (border windows are
// mostly skipped, see window_extent): with units padded to a multiple of 8, XCD k's contiguous range of the list
// holds the units k, k+8, k+16, ... so cheap and expensive units are spread evenly over the XCDs.
__device__ __forceinline__ int d_xcd_unit(int slot, int n_units)
{
    const int per = (n_units + 7) >> 3;
    return (slot % per) * 8 + slot / per;  // >= n_units for padding slots
}
static inline long long ma_xcd_slots(long long n_units) { return ((n_units + 7) / 8) * 8; }

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
