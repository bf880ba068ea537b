// Single-scale Farneback optical flow (cv2.calcOpticalFlowFarneback with levels=0,
// poly_n=1, OPTFLOW_FARNEBACK_GAUSSIAN -- reference call site
// microaligner/optflow_reg/flow_calc.py:33-44) batched over the zero-padded
// overlapping windows TileFlowCalc cuts (flow_calc.py:59-98, slicer.py, stitcher.py).
//
// Pipeline per batch of windows (all fields planar f32, [window][plane][Ph][pitch]):
//   fb_polyexp_m0     : image pair -> R0, R1 (5 planes each) and the first matrix field M (5 planes)
//                       3x3 pre-blur (reflect-101) + 3x3 polynomial expansion (replicate), LDS staged.
//   repeat iterations:
//     fb_blur_v       : M -> V   win-tap Gaussian along y (replicate), register sliding window over LDS
//     fb_blur_h_solve : V -> flow (2x2 solve in double) -> next M (UpdateMatrices fused), or, on the
//                       last iteration, the centre crop of the window straight into the stitched flow.
//
// Arithmetic follows SURVEY.md Appendix A.1 operation by operation (compiled with
// -ffp-contract=off), so results are bit-identical to oracle/ma_oracle.c.
#include "ma_internal.h"

#include <algorithm>
#include <cmath>
#include <cfloat>
#include <type_traits>

namespace {

struct FbGeom {
    MaTiling t;
    int pitch;          // floats per plane row
    int tile0;          // global index of the first window of this batch
    size_t plane;       // floats per plane = Ph * pitch
    int margin;         // active-extent margin (see window_extent); >= Ph, Pw disables the optimisation
};

struct PolyConsts {
    float g0, g1, xg1, xxg1;
    double ig11, ig03, ig33, ig55;
};

// planes per window in the workspace
enum { PL_R0 = 0, PL_R1 = 5, PL_M = 10, PL_V = 15, PL_COUNT = 20 };

__device__ __forceinline__ float* plane_ptr(float* ws, const FbGeom& g, int wl, int pl)
{
    return ws + ((size_t)wl * PL_COUNT + pl) * g.plane;
}

template <typename T>
__device__ __forceinline__ float fetch_window(const T* img, const MaTiling& t, int oy, int ox, int ly, int lx)
{
    int y = oy + ly, x = ox + lx;
    if ((unsigned)y < (unsigned)t.H && (unsigned)x < (unsigned)t.W) return (float)img[(size_t)y * t.W + x];
    return 0.f;
}

__device__ __forceinline__ void window_origin(const MaTiling& t, int widx, int& oy, int& ox)
{
    if (t.T == 0) { oy = 0; ox = 0; return; }
    int ty = widx / t.ntx, tx = widx - ty * t.ntx;
    oy = ty * t.T - t.ov;
    ox = tx * t.T - t.ov;
}

// Active extent of a window.  Right/bottom border windows are mostly zero padding (slicer.py pads every window to
// (tile+2*overlap)^2).  With vx = number of window columns that lie inside the image, the polynomial expansions
// vanish for x >= vx + 2 and, by induction over the iterations (a matrix entry can only become non-zero where the
// blurred previous field, hence the flow, is non-zero), M_k vanishes for x >= vx + 2 + k*m.  All fields are
// therefore exactly +0 beyond  ex = vx + (iterations-1)*m + 3  (same along y), nothing outside [0,ex) x [0,ey)
// can influence a pixel inside the image, and the kernels neither compute nor read there: loads beyond the
// extent are replaced by 0.f, work items beyond it exit.  Bit-identical to processing the full window
// (tests compare against the oracle, which does process it); saves 9 % (16384^2) to 31 % (4096^2) of the work.
__device__ __forceinline__ void window_extent(const FbGeom& g, int oy, int ox, int& ey, int& ex)
{
    const int vy = min(g.t.Ph, g.t.H - oy), vx = min(g.t.Pw, g.t.W - ox);
    ey = min(g.t.Ph, vy + min(g.margin, g.t.Ph));
    ex = min(g.t.Pw, vx + min(g.margin, g.t.Pw));
}

// Needed extent of a window (the mirror image of the active extent: that one drops what cannot be non-zero, this
// one drops what nobody reads).  Of a tiled window only the centre crop [ov, ov+T) (clipped to the image) reaches
// the stitched flow (stitcher.py:62-65).  Walking the data flow backwards -- crop <- horizontal pass <- vertical
// pass <- UpdateMatrices (pointwise in the flow) <- previous iteration -- the horizontal pass of the iteration
// that has `rem` iterations after it is needed on  crop +- rem*m, its vertical pass on the same rows and m more
// columns either side, and the M it writes exactly where the next vertical pass reads.  Blocks outside exit;
// values outside the rectangle are stale but only ever feed outputs that are outside the next rectangle.
// Bit-identical to blurring the whole window; 13 % less blur work at 3 iterations, tile 1000, overlap 100.
struct FbRect { int y0, y1, x0, x1; };
__host__ __device__ __forceinline__ FbRect needed_rect_h(const MaTiling& t, int oy, int ox, int reach)
{
    FbRect r;
    if (t.T == 0) { r.y0 = 0; r.y1 = t.Ph; r.x0 = 0; r.x1 = t.Pw; return r; }
    const int cy1 = t.ov + t.T < t.H - oy ? t.ov + t.T : t.H - oy;
    const int cx1 = t.ov + t.T < t.W - ox ? t.ov + t.T : t.W - ox;
    r.y0 = t.ov - reach > 0 ? t.ov - reach : 0;
    r.x0 = t.ov - reach > 0 ? t.ov - reach : 0;
    r.y1 = cy1 + reach < t.Ph ? cy1 + reach : t.Ph;
    r.x1 = cx1 + reach < t.Pw ? cx1 + reach : t.Pw;
    return r;
}
__host__ __device__ __forceinline__ FbRect needed_rect_v(const MaTiling& t, int oy, int ox, int reach, int m)
{
    FbRect r = needed_rect_h(t, oy, ox, reach);
    r.x0 = r.x0 - m > 0 ? r.x0 - m : 0;
    r.x1 = r.x1 + m < t.Pw ? r.x1 + m : t.Pw;
    return r;
}
// block grids start at the rectangle's corner, columns aligned down to 128 bytes for the row accesses
constexpr int NEED_XALIGN = 32;

// ---------------------------------------------------------------------------------------------
// K1: pre-blur + polynomial expansion of both images + first UpdateMatrices (flow == 0)
// ---------------------------------------------------------------------------------------------
constexpr int K1_TX = 64, K1_TY = 16, K1_THREADS = 256;

// NEAR_BORDER = false: the caller guarantees that the pixel is at least 5 px away from every window edge (a
// block-uniform fact for all but the outermost blocks), and the attenuation test is not even compiled
template <bool NEAR_BORDER = true>
__device__ __forceinline__ void update_matrices_px(const float r0[5], float r2, float r3, float r4, float r5,
                                                   float r6, bool inside, float dx, float dy, int x, int y,
                                                   int w, int h, float out[5])
{
    // A.1 step 3; r2..r6 are the (bilinearly sampled) R1 values when `inside`
    if (inside) {
        r4 = (r0[2] + r4) * 0.5f;
        r5 = (r0[3] + r5) * 0.5f;
        r6 = (r0[4] + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = r0[2];
        r5 = r0[3];
        r6 = r0[4] * 0.5f;
    }
    r2 = (r0[0] - r2) * 0.5f;
    r3 = (r0[1] - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    const int BORDER = 5;
    if (NEAR_BORDER && ((unsigned)(x - BORDER) >= (unsigned)(w - BORDER * 2) || (unsigned)(y - BORDER) >= (unsigned)(h - BORDER * 2))) {
        const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
        float scale = (x < BORDER ? border[x] : 1.f) * (x >= w - BORDER ? border[w - x - 1] : 1.f) *
                      (y < BORDER ? border[y] : 1.f) * (y >= h - BORDER ? border[h - y - 1] : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    out[0] = r4 * r4 + r6 * r6;
    out[1] = (r4 + r5) * r6;
    out[2] = r5 * r5 + r6 * r6;
    out[3] = r4 * r2 + r6 * r3;
    out[4] = r6 * r2 + r5 * r3;
}

template <typename T>
__global__ __launch_bounds__(K1_THREADS, 4) void fb_polyexp_m0(const T* __restrict__ prev, const T* __restrict__ next,
                                                            FbGeom g, PolyConsts pc, float* __restrict__ ws)
{
    constexpr int RW = K1_TX + 4, RH = K1_TY + 4;   // raw tile
    constexpr int TW = K1_TX + 2;                   // row-blurred / blurred / vertical-pass tiles
    __shared__ float raw[RH][RW];
    __shared__ float tb[RH][TW];
    __shared__ float bl[K1_TY + 2][TW];
    __shared__ float vt[3][K1_TY][TW];

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * K1_TX, y0 = blockIdx.y * K1_TY;
    const int wl = blockIdx.z;
    const int Ph = g.t.Ph, Pw = g.t.Pw;
    int oy, ox, ey, ex;
    window_origin(g.t, g.tile0 + wl, oy, ox);
    window_extent(g, oy, ox, ey, ex);
    if (x0 >= ex || y0 >= ey) return;  // nothing but exact zeros there

    const int lx = tid & 63;
    const int lyg = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index; output mapping: column lx, rows lyg*4 .. +3
    float r0v[4][5];  // R0 of this thread's four pixels, kept for the first UpdateMatrices in the second pass

    // Staging index math is done once: each wave owns rows lyg, lyg+4, ... of every LDS tile (row math on the
    // SALU) and a lane owns column lx; the 2-4 halo columns past 64 are handled by the first 4*rows (2*rows)
    // threads of the block, one element each, so no wave issues a nearly empty instruction.
    const int rtx0 = d_reflect101(x0 - 2 + lx, Pw);
    const int tcx0 = d_clamp(d_clamp(x0 - 1 + lx, 0, Pw - 1) - (x0 - 2), 1, RW - 2);
    // halo element of this thread: raw tile row tid/4, column 64 + tid%4; other tiles row tid/2, column 64 + tid%2
    const int hr4 = tid >> 2, hc4 = 64 + (tid & 3), hr2 = tid >> 1, hc2 = 64 + (tid & 1);
    const int rtxh = d_reflect101(x0 - 2 + hc4, Pw);
    const int tcxh = d_clamp(d_clamp(x0 - 1 + hc2, 0, Pw - 1) - (x0 - 2), 1, RW - 2);

    // stores go through one buffer resource spanning the window's planes (plane offsets in SGPRs), NON-TEMPORAL (round 5): the
    // kernel is bound by its 60 B/px of writes, nothing reads them before 8 GB of other planes have gone by, and with the
    // `nt` policy the 15 write streams of a block no longer fight for L2 lines: 6.68 -> 5.76 ms per step (5.7 TB/s)
    constexpr int ST_NT = 2;
    const int plane4 = (int)(g.plane * sizeof(float));
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(plane_ptr(ws, g, wl, 0), 0, PL_COUNT * plane4, 0x00020000);
    // block whose 20 x 68 raw tile needs neither reflection nor zero padding
    const bool interior = x0 >= 2 && x0 + K1_TX + 2 <= Pw && y0 >= 2 && y0 + K1_TY + 2 <= Ph && ox + x0 - 2 >= 0 &&
                          ox + x0 + K1_TX + 2 <= g.t.W && oy + y0 - 2 >= 0 && oy + y0 + K1_TY + 2 <= g.t.H;

    for (int img = 0; img < 2; img++) {
        const T* src = img == 0 ? prev : next;
        // 1. raw tile at virtual coordinates (reflect-101 of the window), zero outside the image
        if (sizeof(T) == 1 && interior) {
            // uint8 images (the DOG inputs), block fully inside window and image: the 20 x 68 bytes are fetched as
            // 20 x 17 (unaligned) dwords -- byte-per-lane loads cost this kernel ~35 % (profiles/r01_notes.md)
            constexpr int DW = RW / 4;
            const uint8_t* base = reinterpret_cast<const uint8_t*>(src) + (size_t)(oy + y0 - 2) * g.t.W + (ox + x0 - 2);
            for (int e = tid; e < RH * DW; e += K1_THREADS) {
                const int j = e / DW, dw = e - j * DW;
                uint32_t word;
                __builtin_memcpy(&word, base + (size_t)j * g.t.W + 4 * dw, 4);
                raw[j][4 * dw + 0] = (float)(word & 255u);
                raw[j][4 * dw + 1] = (float)((word >> 8) & 255u);
                raw[j][4 * dw + 2] = (float)((word >> 16) & 255u);
                raw[j][4 * dw + 3] = (float)(word >> 24);
            }
        } else {
            float v[RH / 4], vh = 0.f;
#pragma unroll
            for (int k = 0; k < RH / 4; k++)
                v[k] = fetch_window(src, g.t, oy, ox, d_reflect101(y0 - 2 + lyg + 4 * k, Ph), rtx0);
            if (hr4 < RH) vh = fetch_window(src, g.t, oy, ox, d_reflect101(y0 - 2 + hr4, Ph), rtxh);
#pragma unroll
            for (int k = 0; k < RH / 4; k++) raw[lyg + 4 * k][lx] = v[k];
            if (hr4 < RH) raw[hr4][hc4] = vh;
        }
        __syncthreads();
        // 2. horizontal [1/4 1/2 1/4] at clamped columns
#pragma unroll
        for (int k = 0; k < RH / 4; k++) {
            const int j = lyg + 4 * k;
            tb[j][lx] = raw[j][tcx0] * 0.5f + (raw[j][tcx0 - 1] + raw[j][tcx0 + 1]) * 0.25f;
        }
        if (hr2 < RH) tb[hr2][hc2] = raw[hr2][tcxh] * 0.5f + (raw[hr2][tcxh - 1] + raw[hr2][tcxh + 1]) * 0.25f;
        __syncthreads();
        // 3. vertical [1/4 1/2 1/4] at clamped rows -> blurred image (replicate semantics for step 4/5)
#pragma unroll
        for (int k = 0; k < (K1_TY + 2 + 3) / 4; k++) {
            const int j = lyg + 4 * k;
            if (j < K1_TY + 2) {
                const int cy = d_clamp(d_clamp(y0 - 1 + j, 0, Ph - 1) - (y0 - 2), 1, RH - 2);
                bl[j][lx] = tb[cy][lx] * 0.5f + (tb[cy - 1][lx] + tb[cy + 1][lx]) * 0.25f;
            }
        }
        if (hr2 < K1_TY + 2) {
            const int cy = d_clamp(d_clamp(y0 - 1 + hr2, 0, Ph - 1) - (y0 - 2), 1, RH - 2);
            bl[hr2][hc2] = tb[cy][hc2] * 0.5f + (tb[cy - 1][hc2] + tb[cy + 1][hc2]) * 0.25f;
        }
        __syncthreads();
        // 4. vertical part of the polynomial expansion (float)
        auto vpass = [&](int j, int c) {
            float up = bl[j][c], ce = bl[j + 1][c], dn = bl[j + 2][c];
            float p = up + dn;
            float row0 = ce * pc.g0;
            vt[0][j][c] = row0 + pc.g1 * p;
            vt[1][j][c] = 0.f + pc.xg1 * (dn - up);
            vt[2][j][c] = 0.f + pc.xxg1 * p;
        };
#pragma unroll
        for (int k = 0; k < K1_TY / 4; k++) vpass(lyg + 4 * k, lx);
        if (hr2 < K1_TY) vpass(hr2, hc2);
        __syncthreads();
        // 5. horizontal part (double accumulators where OpenCV has them) -> R
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            // one pixel at a time: without the fence the scheduler interleaves the four f64 sections (144 VGPRs)
            __builtin_amdgcn_sched_barrier(0);
            const int j = lyg * 4 + rr;
            const int y = y0 + j, x = x0 + lx;
            const int c = lx + 1;
            float a0m = vt[0][j][c - 1], a0 = vt[0][j][c], a0p = vt[0][j][c + 1];
            float a1m = vt[1][j][c - 1], a1 = vt[1][j][c], a1p = vt[1][j][c + 1];
            float a2m = vt[2][j][c - 1], a2 = vt[2][j][c], a2p = vt[2][j][c + 1];
            double b1 = (double)(a0 * pc.g0), b3 = (double)(a1 * pc.g0), b5 = (double)(a2 * pc.g0);
            double tg = (double)(a0p + a0m);
            b1 += tg * (double)pc.g1;
            double b4 = 0.0 + tg * (double)pc.xxg1;
            double b2 = 0.0 + (double)((a0p - a0m) * pc.xg1);
            b3 += (double)((a1p + a1m) * pc.g1);
            double b6 = 0.0 + (double)((a1p - a1m) * pc.xg1);
            b5 += (double)((a2p + a2m) * pc.g1);
            float R[5];
            R[1] = (float)(b2 * pc.ig11);
            R[0] = (float)(b3 * pc.ig11);
            R[3] = (float)(b1 * pc.ig03 + b4 * pc.ig33);
            R[2] = (float)(b1 * pc.ig03 + b5 * pc.ig33);
            R[4] = (float)(b6 * pc.ig55);
            const bool ok = y < Ph && x < Pw;
            const int pix4 = (y * g.pitch + x) * 4;
            if (ok) {
#pragma unroll
                for (int k = 0; k < 5; k++)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(R[k]), wrsrc, pix4,
                                                          ((img == 0 ? PL_R0 : PL_R1) + k) * plane4, ST_NT);
            }
            if (img == 0) {
#pragma unroll
                for (int k = 0; k < 5; k++) r0v[rr][k] = R[k];
            } else if (ok) {
                // first UpdateMatrices with flow == 0: the bilinear sample degenerates to R1 itself
                const bool inside = x < Pw - 1 && y < Ph - 1;
                float Mv[5];
                update_matrices_px(r0v[rr], R[0], R[1], R[2], R[3], R[4], inside, 0.f, 0.f, x, y, Pw, Ph, Mv);
#pragma unroll
                for (int k = 0; k < 5; k++)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(Mv[k]), wrsrc, pix4, (PL_M + k) * plane4, ST_NT);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// K2: vertical window blur  M -> V   (one plane per blockIdx.z)
// Lanes run along x; each thread owns one column and R consecutive output rows, sliding a register
// window over an LDS-staged column strip:  s = c*k0;  s = (dn_i + up_i)*k_i + s  for i = 1..m.
// ---------------------------------------------------------------------------------------------
// occupancy targets (waves per SIMD) passed to __launch_bounds__; tuned on MI355X (profiles/r01_notes.md)
constexpr int BV_WAVES = 6, BH_WAVES = 4;
template <int R, int NW, bool FUSED>
__global__ __launch_bounds__(64 * NW, BV_WAVES) void fb_blur_v(FbGeom g, int m, const float* __restrict__ taps,
                                                     float* __restrict__ ws, int nplanes, int reach)
{
    extern __shared__ float lds[];  // [(NW*R + 2m)][64]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index: keeps row math on the SALU
    const int Ph = g.t.Ph, Pw = g.t.Pw;
    // work list ordered y-fastest inside a column strip, walked contiguously per XCD: the 2m halo rows that
    // vertically adjacent blocks share are then served by that XCD's L2 instead of being re-fetched
    const int nbx = (Pw + 63) / 64, nby = (Ph + NW * R - 1) / (NW * R);
    // (window, plane) units interleaved over the XCDs (d_xcd_unit): border windows, which are mostly skipped
    // (window_extent), are spread evenly.  Measured alternatives (profiles/r01_notes.md): contiguous unit ranges
    // per XCD leave one XCD idle early; window-major interleaving is 25 % slower on this kernel.
    const int nslots = (int)(((nplanes + 7) / 8) * 8);
    const int item = d_xcd_work_item(blockIdx.x, nbx * nby * nslots);
    const int by = item % nby, bx = (item / nby) % nbx;
    const int bz = d_xcd_unit(item / (nby * nbx), nplanes);
    if (bz >= nplanes) return;
    const int wl = bz / 5, ch = bz - wl * 5;
    const float* src = plane_ptr(ws, g, wl, PL_M + ch);
    float* dst = plane_ptr(ws, g, wl, PL_V + ch);
    int oy, ox, ey, ex;
    window_origin(g.t, g.tile0 + wl, oy, ox);
    window_extent(g, oy, ox, ey, ex);
    const FbRect need = needed_rect_v(g.t, oy, ox, reach, m);
    const int x0 = (need.x0 & ~(NEED_XALIGN - 1)) + bx * 64, y0 = need.y0 + by * (NW * R);
    if (x0 >= min(ex, need.x1) || y0 >= min(ey, need.y1)) return;

    constexpr int G = 2;  // guard rows on either side (d_sym_fir_slide contract)
    const int rows = NW * R + 2 * m + 2 * G;
    const int xc = min(x0 + lane, Pw - 1);
    const bool xin = xc < ex;
    {
        // rows in batches of SB per wave: all SB global loads are issued before the first LDS store
        constexpr int SB = 42;
        const unsigned xo = (unsigned)xc * 4u;  // lane byte offset inside a row
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0,
                                                                               (int)(g.plane * sizeof(float)), 0x00020000);
        for (int j0 = w; j0 < rows; j0 += NW * SB) {
            float v[SB];
#pragma unroll
            for (int k = 0; k < SB; k++) {
                const int y = d_clamp(y0 - m - G + j0 + NW * k, 0, Ph - 1);
                // buffer load: the row offset rides in an SGPR, the lane offset is one VGPR for all rows
                v[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)xo, min(y, ey - 1) * g.pitch * 4, 0));
                if (y >= ey || !xin) v[k] = 0.f;  // beyond the active extent M is exactly zero (and was not written)
            }
#pragma unroll
            for (int k = 0; k < SB; k++)
                if (j0 + NW * k < rows) lds[(j0 + NW * k) * 64 + lane] = v[k];
        }
    }
    __syncthreads();

    if (y0 + w * R >= min(ey, need.y1)) return;  // this wave's rows are not needed (no barrier follows)
    float acc[R];
    d_sym_fir_slide_pk<R, FUSED, true>(lds + lane, G + m + w * R, m, taps, acc);
    const int x = x0 + lane;
    const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(g.plane * sizeof(float)), 0x00020000);
    if (x >= need.x0 && x < need.x1) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int y = y0 + w * R + r;
            if (y < need.y1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[r]), drsrc, x * 4, y * g.pitch * 4, 0);
        }
    }
}

// K2, streaming form (used when the tap pairs are whole groups of R/2, i.e. at the reference's window of 99): one block
// walks DOWN a 64-column strip of one plane.  The strip's rows live in an LDS ring of C = NW*R + 2m + 4 rows (plus the
// mirror rows d_sym_fir_ring_pk needs); per chunk of NW*R output rows only the NW*R new rows are staged -- the tiled
// form above stages all C rows for every chunk, 2.8 x the output -- and their global loads are issued before the
// filter of the previous chunk runs (R registers per lane), so the memory latency hides behind the arithmetic.
template <int R, int NW, bool FUSED>
__global__ __launch_bounds__(64 * NW, BV_WAVES) void fb_blur_v_stream(FbGeom g, int m, const float* __restrict__ taps,
                                                            float* __restrict__ ws, int nplanes, int reach)
{
    extern __shared__ float lds[];  // [C + MIR][64]
    constexpr int H = R / 2, G = 2, CH = NW * R, MIR = 3 * H + 1;
    const int C = CH + 2 * m + 2 * G;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Ph = g.t.Ph, Pw = g.t.Pw;
    const int nbx = (Pw + 63) / 64;
    // plain dispatch order (round 5): the strips of this form share no halo, so nothing is gained by keeping neighbours on one
    // XCD, and the V rows they write leave faster through eight L2s than through one (24.0 -> 23.7 ms per step; the expansion
    // kernel shows the same preference, profiles/r05_notes.md).  The tiled form above keeps the XCD-aware list for its halos.
    const int item = blockIdx.x;
    const int bx = item % nbx;
    const int bz = item / nbx;
    if (bz >= nplanes) return;
    const int wl = bz / 5, ch = bz - wl * 5;
    const float* src = plane_ptr(ws, g, wl, PL_M + ch);
    float* dst = plane_ptr(ws, g, wl, PL_V + ch);
    int oy, ox, ey, ex;
    window_origin(g.t, g.tile0 + wl, oy, ox);
    window_extent(g, oy, ox, ey, ex);
    const FbRect need = needed_rect_v(g.t, oy, ox, reach, m);
    const int x0 = (need.x0 & ~(NEED_XALIGN - 1)) + bx * 64;
    const int yend = min(ey, need.y1);
    if (x0 >= min(ex, need.x1) || need.y0 >= yend) return;
    const int ybase = need.y0 - m - G;       // window row held by virtual row 0 of the walk
    const int xc = min(x0 + lane, Pw - 1);
    const bool xin = xc < ex;
    const unsigned xo = (unsigned)xc * 4u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0,
                                                                           (int)(g.plane * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(g.plane * sizeof(float)), 0x00020000);
    // Loads are unconditional and their results are used unconditionally: beyond the active extent (where M is
    // exactly zero and was never written) the row is clamped into the extent and the value is masked to +0 when it
    // is committed to LDS, a chunk later.  (Written as `if (beyond) val = 0` next to the load, the compiler makes the
    // load itself conditional and waits for every single one right where it is issued -- fourteen dependent round
    // trips in front of the filter instead of one hidden behind it: 0.3 ms per launch, profiles/r03_notes.md.)
    auto load_row = [&](int v) -> float {    // virtual row v = window row ybase + v, replicated at the window border
        const int y = d_clamp(ybase + v, 0, Ph - 1);
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)xo, min(y, ey - 1) * g.pitch * 4, 0));
    };
    const unsigned lanemask = xin ? 0xffffffffu : 0u;
    auto row_mask = [&](int v) -> unsigned {  // wave-uniform: all ones unless virtual row v lies below the active extent
        return d_clamp(ybase + v, 0, Ph - 1) >= ey ? 0u : 0xffffffffu;
    };
    auto commit_row = [&](int pos, float val, unsigned rmask) {   // pos: ring row, 0 <= pos < C
        val = __uint_as_float(__float_as_uint(val) & (lanemask & rmask));
        lds[pos * 64 + lane] = val;
        if (pos < MIR) lds[(pos + C) * 64 + lane] = val;
    };
    // prologue: virtual rows [0, 2m + 2G) sit at ring rows of the same number
    {
        const int npro = 2 * m + 2 * G;
        constexpr int SB = 13;
        for (int j0 = w; j0 < npro; j0 += NW * SB) {
            float v[SB];
#pragma unroll
            for (int k = 0; k < SB; k++) v[k] = load_row(min(j0 + NW * k, npro - 1));
#pragma unroll
            for (int k = 0; k < SB; k++)
                if (j0 + NW * k < npro) commit_row(j0 + NW * k, v[k], row_mask(j0 + NW * k));
        }
    }
    float nv[R];                              // the NW*R new rows of the coming chunk, R per wave
    int vnew = 2 * m + 2 * G;                 // first virtual row of those
    int pnew = vnew;                          // its ring row (C > 2m + 2G)
    int jbase = m + G;                        // ring row of the chunk's first output row
#pragma unroll
    for (int k = 0; k < R; k++) nv[k] = load_row(vnew + w + NW * k);
    const int x = x0 + lane;
    for (int y0 = need.y0; y0 < yend; y0 += CH) {
#pragma unroll
        for (int k = 0; k < R; k++) {
            int p = pnew + w + NW * k;
            if (p >= C) p -= C;
            commit_row(p, nv[k], row_mask(vnew + w + NW * k));
        }
        __syncthreads();
        vnew += CH;
        pnew += CH;
        if (pnew >= C) pnew -= C;
        if (y0 + CH < yend) {
#pragma unroll
            for (int k = 0; k < R; k++) nv[k] = load_row(vnew + w + NW * k);
        }
        if (y0 + w * R < yend) {              // else: this wave's rows of the last chunk are not needed
            int jb = jbase + w * R;
            if (jb >= C) jb -= C;
            float acc[R];
            d_sym_fir_ring_pk<R, FUSED>(lds + lane, jb, C, m, taps, acc);
            if (x >= need.x0 && x < need.x1) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int y = y0 + w * R + r;
                    if (y < need.y1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[r]), drsrc, x * 4, y * g.pitch * 4, 0);
                }
            }
        }
        jbase += CH;
        if (jbase >= C) jbase -= C;
        __syncthreads();                      // every wave is done with the rows the next commit overwrites
    }
}

// ---------------------------------------------------------------------------------------------
// K3: horizontal window blur of V (5 planes) + 2x2 solve (double) + UpdateMatrices / final store.
// Lanes run along y (one row each) so the register window slides along x; results are transposed
// through LDS so that all global traffic (R0, R1 gather, M / flow stores) is coalesced along x.
// Block: 64 rows x (NW*R) columns.
// ---------------------------------------------------------------------------------------------
// Q = ceil((NW*R + 2m + 4) / 64): column chunks of the staged row tile
// LAST: the final iteration only stores the centre crop of the flow; it is its own instantiation, so it carries
// neither the code nor the registers of UpdateMatrices.
// ROWS = 64: a lane is a row.  ROWS = 32: a wave is two half-waves of 32 rows, the upper one NW * R columns further right
// (ds_read2_b32 is banked per group of 32 lanes, so the two halves never conflict): the same 112 output columns and the
// same staged width per block with half the rows, half the waves and half the LDS -- four independent blocks per CU instead
// of two, so that the memory phases of a block (staging, UpdateMatrices) more often find another block filtering.
template <int R, int NW, bool FUSED, int Q, bool LAST, int ROWS>
__global__ __launch_bounds__(64 * NW, BH_WAVES) void fb_blur_h_solve(FbGeom g, int m, const float* __restrict__ taps,
                                                           float* __restrict__ ws, float* __restrict__ flow_out,
                                                           int nwin, int reach)
{
    extern __shared__ float lds[];
    static_assert(ROWS == 64 || ROWS == 32, "a wave covers 64 rows, or 32 rows twice");
    constexpr int HALVES = 64 / ROWS;
    constexpr int TXW = NW * R * HALVES;  // output columns per block
    constexpr int NT = 64 * NW;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int row = lane & (ROWS - 1);                       // tile row of this lane in the filter and the solve
    const int hcol = (lane / ROWS) * (NW * R);               // first output column of this lane's half-wave
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index: keeps row/column math on the SALU
    const int Ph = g.t.Ph, Pw = g.t.Pw;
    // x-fastest work list walked contiguously per XCD (see fb_blur_v): horizontally adjacent blocks share 2m columns
    const int nbx = (Pw + TXW - 1) / TXW, nby = (Ph + ROWS - 1) / ROWS;
    const int nslots = (int)(((nwin + 7) / 8) * 8);
    const int item = d_xcd_work_item(blockIdx.x, nbx * nby * nslots);
    const int bx = item % nbx, by = (item / nbx) % nby;
    const int wl = d_xcd_unit(item / (nbx * nby), nwin);  // windows interleaved over the XCDs (see fb_blur_v)
    if (wl >= nwin) return;
    int oy, ox, ey, ex;
    window_origin(g.t, g.tile0 + wl, oy, ox);
    window_extent(g, oy, ox, ey, ex);
    const FbRect need = needed_rect_h(g.t, oy, ox, reach);
    const int x0 = (need.x0 & ~(NEED_XALIGN - 1)) + bx * TXW, y0 = need.y0 + by * ROWS;
    const int xend = min(ex, need.x1), yend = min(ey, need.y1);   // needed AND possibly non-zero
    if (x0 >= xend || y0 >= yend) return;
    constexpr int G = 2;  // guard columns on either side (d_sym_fir_slide contract)
    const int cols = TXW + 2 * m + 2 * G;
    const int lp = cols | 1;              // odd LDS pitch: lanes (rows) hit distinct banks

    // Staging of one plane: ROWS/NW rows x Q column chunks per wave; every global load is issued before the first
    // LDS store.  (Issuing plane ch+1's loads before plane ch is filtered -- register prefetch -- was measured: no gain
    // at 4 waves/SIMD, spills at 6.)
    constexpr int RW = ROWS / NW;
    float v[RW][Q];
    // one buffer resource spans the window's 20 planes: plane and row offsets ride in SGPRs, each access needs
    // only a 32-bit lane offset (no 64-bit VALU address arithmetic)
    const int plane4 = (int)(g.plane * sizeof(float));
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(plane_ptr(ws, g, wl, 0), 0, PL_COUNT * plane4, 0x00020000);
    // lane byte offset of column chunk q.  A column right of the active extent holds exactly zero (and was never written):
    // its lane offset lies beyond the buffer's range, where a buffer load returns 0 by itself -- no select per load
    int xo[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int xi = d_clamp(x0 - m - G + lane + 64 * q, 0, Pw - 1);
        xo[q] = xi >= ex ? (int)0x80000000 : xi * 4;
    }
    auto issue = [&](int ch) {
#pragma unroll
        for (int k = 0; k < RW; k++) {
            const int soff = (PL_V + ch) * plane4 + min(y0 + w + NW * k, Ph - 1) * g.pitch * 4;
#pragma unroll
            for (int q = 0; q < Q; q++)
                v[k][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrsrc, xo[q], soff, 0));
        }
    };
    auto commit = [&]() {
        // one predicate per column chunk (only the last chunk is partial), not one branch per store
#pragma unroll
        for (int q = 0; q < Q; q++) {
            if (lane + 64 * q < cols) {
#pragma unroll
                for (int k = 0; k < RW; k++) lds[(w + NW * k) * lp + lane + 64 * q] = v[k][q];
            }
        }
    };

    float hs[5][R];
    const bool wave_needed = x0 + w * R < xend;  // else: this wave's columns feed nothing (it still stages)
    issue(0);
#pragma unroll
    for (int ch = 0; ch < 5; ch++) {
        commit();
        __syncthreads();
        float acc[R];
        if (wave_needed) {
            d_sym_fir_slide_pk<R, FUSED, false>(lds + row * lp + hcol, G + m + w * R, m, taps, acc);
        } else {
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < R; r++) hs[ch][r] = acc[r];
        // the next plane's loads go out before the barrier: a wave that finishes its filter early waits with its
        // loads in flight instead of idle (they land in registers; LDS is rewritten only after the barrier): -3 %
        if (ch < 4) issue(ch + 1);
        __syncthreads();
    }

    // 2x2 solve where the filter left its sums: in registers, lane = row, r = column.  Only the flow crosses LDS
    // (one 8-byte store per pixel, all 64 lanes, one pass) on its way to the x-major mapping of the epilogue.
    // Pitch TXW + 1 pairs: the 16 lanes of a store group start 2 banks apart.
    constexpr int TP = TXW + 1;
    ma_f2* tb = reinterpret_cast<ma_f2*>(lds);
    if (wave_needed) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const double g11 = hs[0][r], g12 = hs[1][r], g22 = hs[2][r], h1 = hs[3][r], h2 = hs[4][r];
            const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            const float dx = (float)((g11 * h2 - g12 * h1) * idet);
            const float dy = (float)((g22 * h1 - g12 * h2) * idet);
            tb[row * TP + hcol + w * R + r] = (ma_f2){dx, dy};
        }
    }
    __syncthreads();

    // Epilogue, lanes along x: every global access (R0, the R1 gather, M / flow stores) is coalesced.  Pixels right
    // of / below the active extent are skipped: nothing reads M there (loads beyond the extent are replaced by 0)
    // and the stitched flow ends with the image.
    if (LAST) {
        // centre crop of the window -> stitched flow (stitcher.py:62-65).  With reach == 0 the needed rectangle IS the
        // crop clipped to the image (needed_rect_h), so membership needs no second test.  The flow is addressed
        // through a buffer resource that starts at this block's first image row (offsets stay below 2^31 for any
        // image the 32-bit pixel coordinates allow).
        float* frow = flow_out + (size_t)(oy + y0) * g.t.W * 2;
        const long long fbytes = (long long)min(ROWS, yend - y0) * g.t.W * 8;
        const __amdgpu_buffer_rsrc_t frsrc = __builtin_amdgcn_make_buffer_rsrc(
            frow, 0, (int)(fbytes < 0x7fffffffLL ? fbytes : 0x7fffffffLL), 0x00020000);
        for (int p = tid; p < ROWS * TXW; p += NT) {
            const int rh = p / TXW, c = p - rh * TXW;
            const int y = y0 + rh, x = x0 + c;
            if (y < yend && x >= need.x0 && x < xend) {
                const ma_f2 f = tb[rh * TP + c];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ma_u2, f), frsrc, (rh * g.t.W + ox + x) * 8, 0, 0);
            }
        }
    } else {
        auto epilogue = [&](auto near_border_tag) {
            constexpr bool NEAR_BORDER = decltype(near_border_tag)::value;
            // pixel p = tid + k * NT of the tile, row-major: (row, column) and the byte offset of the pixel advance by
            // constants (NT = DR rows + DC columns; one more row when the column wraps) -- no division, no multiply
            constexpr int DR = NT / TXW, DC = NT % TXW;
            const int pitch4 = g.pitch * 4;
            int rh = tid / TXW, c = tid - rh * TXW;
            int pix4 = ((y0 + rh) * g.pitch + x0 + c) * 4;
            int tbi = rh * TP + c;            // index of the pixel's flow in the LDS tile
            for (int p = tid; p < ROWS * TXW; p += NT, rh += DR, c += DC, pix4 += DR * pitch4 + DC * 4, tbi += DR * TP + DC) {
                if (c >= TXW) { c -= TXW; rh += 1; pix4 += pitch4 - TXW * 4; tbi += TP - TXW; }
                const int y = y0 + rh, x = x0 + c;
                if (y < yend && x >= need.x0 && x < xend) {
                    const ma_f2 f = tb[tbi];
                    const float dx = f.x, dy = f.y;
                    // UpdateMatrices (A.1 step 3) at this pixel
                    float r0[5];
#pragma unroll
                    for (int k = 0; k < 5; k++)
                        r0[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrsrc, pix4, (PL_R0 + k) * plane4, 0));
                    float fx = (float)x + dx, fy = (float)y + dy;
                    int x1 = d_cvfloor(fx), y1 = d_cvfloor(fy);
                    fx -= (float)x1; fy -= (float)y1;
                    float r[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
                    const bool inside = (unsigned)x1 < (unsigned)(Pw - 1) && (unsigned)y1 < (unsigned)(Ph - 1);
                    // R1 is exactly zero beyond the active extent (and was not written there)
                    const bool r1zero = x1 + 1 >= ex || y1 + 1 >= ey;
                    if (inside && !r1zero) {
                        float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy,
                              a11 = fx * fy;
                        // 0 <= y1 < Ph < 2^15 and pitch < 2^16 here: a 24-bit multiply-add (full rate) serves
                        const int q0 = (int)(__umul24((unsigned)y1, (unsigned)g.pitch) + (unsigned)x1) * 4, q1 = q0 + pitch4;
#pragma unroll
                        for (int k = 0; k < 5; k++) {
                            const int so = (PL_R1 + k) * plane4;
                            const float s00 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrsrc, q0, so, 0));
                            const float s01 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrsrc, q0 + 4, so, 0));
                            const float s10 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrsrc, q1, so, 0));
                            const float s11 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrsrc, q1 + 4, so, 0));
                            r[k] = a00 * s00 + a01 * s01 + a10 * s10 + a11 * s11;
                        }
                    }
                    float Mv[5];
                    update_matrices_px<NEAR_BORDER>(r0, r[0], r[1], r[2], r[3], r[4], inside, dx, dy, x, y, Pw, Ph, Mv);
#pragma unroll
                    for (int k = 0; k < 5; k++)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(Mv[k]), wrsrc, pix4, (PL_M + k) * plane4, 0);
                }
            }
        };
        // the 5-px border attenuation of UpdateMatrices concerns only the outermost blocks of a window
        const bool near_border = x0 < 5 || x0 + TXW > Pw - 5 || y0 < 5 || y0 + ROWS > Ph - 5;
        if (near_border) epilogue(std::true_type{});
        else epilogue(std::false_type{});
    }
}

// ---------------------------------------------------------------------------------------------
// Fallback kernels for windows whose halo does not fit LDS (winsize > ~450): plain per-pixel loops.
// ---------------------------------------------------------------------------------------------
template <bool FUSED>
__global__ void fb_blur_v_simple(FbGeom g, int m, const float* __restrict__ taps, float* __restrict__ ws)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    const int wl = blockIdx.z / 5, ch = blockIdx.z - wl * 5;
    if (x >= g.t.Pw) return;
    const float* src = plane_ptr(ws, g, wl, PL_M + ch);
    float* dst = plane_ptr(ws, g, wl, PL_V + ch);
    float s = src[(size_t)y * g.pitch + x] * taps[0];
    for (int i = 1; i <= m; i++) {
        float dn = src[(size_t)min(y + i, g.t.Ph - 1) * g.pitch + x];
        float up = src[(size_t)max(y - i, 0) * g.pitch + x];
        s = d_muladd<FUSED>(dn + up, MA_TAP(taps, i), s);
    }
    dst[(size_t)y * g.pitch + x] = s;
}

template <bool FUSED>
__global__ void fb_blur_h_solve_simple(FbGeom g, int m, const float* __restrict__ taps, float* __restrict__ ws,
                                       int last, float* __restrict__ flow_out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    const int wl = blockIdx.z;
    const int Ph = g.t.Ph, Pw = g.t.Pw;
    if (x >= Pw) return;
    float hsum[5];
    for (int ch = 0; ch < 5; ch++) {
        const float* src = plane_ptr(ws, g, wl, PL_V + ch) + (size_t)y * g.pitch;
        float s = src[x] * taps[0];
        for (int i = 1; i <= m; i++) s = d_muladd<FUSED>(src[max(x - i, 0)] + src[min(x + i, Pw - 1)], MA_TAP(taps, i), s);
        hsum[ch] = s;
    }
    double g11 = hsum[0], g12 = hsum[1], g22 = hsum[2], h1 = hsum[3], h2 = hsum[4];
    double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
    float dx = (float)((g11 * h2 - g12 * h1) * idet);
    float dy = (float)((g22 * h1 - g12 * h2) * idet);
    int oy, ox;
    window_origin(g.t, g.tile0 + wl, oy, ox);
    if (last) {
        bool keep = g.t.T == 0 || (y >= g.t.ov && y < g.t.ov + g.t.T && x >= g.t.ov && x < g.t.ov + g.t.T);
        int iy = oy + y, ix = ox + x;
        if (keep && iy < g.t.H && ix < g.t.W)
            reinterpret_cast<float2*>(flow_out)[(size_t)iy * g.t.W + ix] = make_float2(dx, dy);
        return;
    }
    const float* R0p = plane_ptr(ws, g, wl, PL_R0);
    const float* R1p = plane_ptr(ws, g, wl, PL_R1);
    float* Mp = plane_ptr(ws, g, wl, PL_M);
    const size_t pix = (size_t)y * g.pitch + x;
    float r0[5];
    for (int k = 0; k < 5; k++) r0[k] = R0p[k * g.plane + pix];
    float fx = (float)x + dx, fy = (float)y + dy;
    int x1 = d_cvfloor(fx), y1 = d_cvfloor(fy);
    fx -= (float)x1; fy -= (float)y1;
    float r[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const bool inside = (unsigned)x1 < (unsigned)(Pw - 1) && (unsigned)y1 < (unsigned)(Ph - 1);
    if (inside) {
        float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        const float* q = R1p + (size_t)y1 * g.pitch + x1;
        for (int k = 0; k < 5; k++) {
            const float* qk = q + k * g.plane;
            r[k] = a00 * qk[0] + a01 * qk[1] + a10 * qk[g.pitch] + a11 * qk[g.pitch + 1];
        }
    }
    float Mv[5];
    update_matrices_px(r0, r[0], r[1], r[2], r[3], r[4], inside, dx, dy, x, y, Pw, Ph, Mv);
    for (int k = 0; k < 5; k++) Mp[k * g.plane + pix] = Mv[k];
}

// planar workspace -> user buffers (debug variant)
__global__ void fb_copy_planes(FbGeom g, const float* __restrict__ ws, int pl, float* __restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
    if (x >= g.t.Pw) return;
    out[((size_t)k * g.t.Ph + y) * g.t.Pw + x] = ws[((size_t)pl + k) * g.plane + (size_t)y * g.pitch + x];
}

// ---- host side ------------------------------------------------------------------------------

// hal::Cholesky64f-style SPD solve, as used by Mat::inv(DECOMP_CHOLESKY) (A.1 step 2)
bool chol_inverse6(double* A, double* b)
{
    const int m = 6, n = 6;
    for (int i = 0; i < m; i++) {
        int j;
        for (j = 0; j < i; j++) {
            double s = A[i * m + j];
            for (int k = 0; k < j; k++) s -= A[i * m + k] * A[j * m + k];
            A[i * m + j] = s * A[j * m + j];
        }
        double s = A[i * m + i];
        for (int k = 0; k < j; k++) { double t = A[i * m + k]; s -= t * t; }
        if (s < DBL_EPSILON) return false;
        A[i * m + i] = 1. / std::sqrt(s);
    }
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = b[i * n + j];
            for (int k = 0; k < i; k++) s -= A[i * m + k] * b[k * n + j];
            b[i * n + j] = s * A[i * m + i];
        }
    for (int i = m - 1; i >= 0; i--)
        for (int j = 0; j < n; j++) {
            double s = b[i * n + j];
            for (int k = m - 1; k > i; k--) s -= A[k * m + i] * b[k * n + j];
            b[i * n + j] = s * A[i * m + i];
        }
    return true;
}

bool make_poly_consts(double sigma, PolyConsts* pc)
{
    const int n = 1;
    float gb[3], xgb[3], xxgb[3];
    float *gk = gb + n, *xg = xgb + n, *xxg = xxgb + n;
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        gk[x] = (float)std::exp(-x * x / (2 * sigma * sigma));
        s += gk[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        gk[x] = (float)(gk[x] * s);
        xg[x] = (float)(x * gk[x]);
        xxg[x] = (float)(x * x * gk[x]);
    }
    double G[36] = {0}, I6[36] = {0};
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            G[0] += gk[y] * gk[x];
            G[7] += gk[y] * gk[x] * x * x;
            G[21] += gk[y] * gk[x] * x * x * x * x;
            G[35] += gk[y] * gk[x] * x * x * y * y;
        }
    G[14] = G[3] = G[4] = G[18] = G[24] = G[7];
    G[28] = G[21];
    G[22] = G[27] = G[35];
    for (int i = 0; i < 6; i++) I6[i * 6 + i] = 1.;
    if (!chol_inverse6(G, I6)) return false;
    pc->g0 = gk[0]; pc->g1 = gk[1]; pc->xg1 = xg[1]; pc->xxg1 = xxg[1];
    pc->ig11 = I6[7]; pc->ig03 = I6[3]; pc->ig33 = I6[21]; pc->ig55 = I6[35];
    return true;
}

void make_window_taps(int winsize, std::vector<float>& k)
{
    int m = winsize / 2;
    k.resize(m + 1);
    double sigma = m * 0.3, s = 1;
    k[0] = (float)s;
    for (int i = 1; i <= m; i++) {
        float t = (float)std::exp(-i * i / (2 * sigma * sigma));
        k[i] = t;
        s += t * 2;
    }
    s = 1. / s;
    for (int i = 0; i <= m; i++) k[i] = (float)(k[i] * s);
}

constexpr int BV_R = 14, BV_NW = 4;   // fb_blur_v: 64 columns x (NW*R) = 56 rows per block: at the default window (99 taps) the
                                      // 49 tap pairs are 7 full groups of R/2 and the strip (158 rows) lets 4 blocks share a CU
constexpr int BVS_NW = 8;             // fb_blur_v_stream: chunks of 8 x 14 = 112 rows, ring of 214 + 22 rows = 60 KB, 2 blocks / CU
                                      // (measured per launch: 4 waves 1.55 ms, 6 waves 1.70, 8 waves 1.53, 16 waves 1.84; tiled form 1.67)
#ifndef MA_BH_ROWS
#define MA_BH_ROWS 32
#endif
constexpr int BH_ROWS = MA_BH_ROWS;   // rows per block of fb_blur_h_solve: 32 (half-wave rows, 4 waves) or 64 (8 waves)
constexpr int BH_R = 14, BH_NW = 8 * BH_ROWS / 64;
                                      // fb_blur_h_solve: 112 columns per block either way, 4 waves / SIMD at ~126 VGPRs.  Round 5: 32 rows x 4
                                      // waves (29 KB, FOUR blocks / CU) instead of 64 rows x 8 waves (58 KB, two): the SIMDs issue a VALU
                                      // instruction in 85 % (non-last) / 87 % (last) of their cycles instead of 73 % / 79 %
                                      // (profiles/r05_sq_counters_cfg3.txt), 30.1 -> 27.9 ms per step.  Earlier, per launch: R 8 x 8 waves
                                      // 2.142 ms, R 14 x 4 waves x 64 rows (2.8 x staging) 2.227, R 14 x 8 waves 2.104
constexpr int BH_TXW = BH_NW * BH_R * (64 / BH_ROWS);
constexpr size_t LDS_MAX = 160 * 1024;

template <typename T, bool FUSED>
int run_batch(ma_ctx* ctx, const T* prev, const T* next, FbGeom g, int nwin, const PolyConsts& pc, int m,
              const float* taps, int iters, float* flow_out)
{
    float* ws = (float*)ctx->ws;
    const int Ph = g.t.Ph, Pw = g.t.Pw;
    const size_t lds_v = (size_t)(BV_NW * BV_R + 2 * m + 4) * 64 * sizeof(float);
    const int colsh = BH_TXW + 2 * m + 4;
    size_t lds_h = (size_t)BH_ROWS * (colsh | 1) * sizeof(float);
    const size_t lds_t = (size_t)BH_ROWS * (BH_TXW + 1) * 2 * sizeof(float);   // the flow on its way to the x-major epilogue
    if (lds_t > lds_h) lds_h = lds_t;
    const bool fast = m >= 1 && lds_v <= LDS_MAX && lds_h <= LDS_MAX && colsh <= 320;  // 320 = 5 chunks (winsize <= 253)
    // the LDS-staged kernels honour the active extent; the fallback kernels process whole windows
    g.margin = fast ? (iters - 1) * m + 3 : (1 << 28);
    // pixels actually processed -- the unit of the per-kernel accounting: the active extent for the expansion
    // kernel, active extent x needed rectangle for the two passes of every iteration
    double px = 0;
    std::vector<double> px_v(iters, 0.0), px_h(iters, 0.0);
    for (int wl = 0; wl < nwin; wl++) {
        int oy = 0, ox = 0;
        if (g.t.T > 0) { int ty = (g.tile0 + wl) / g.t.ntx, tx = (g.tile0 + wl) % g.t.ntx; oy = ty * g.t.T - g.t.ov; ox = tx * g.t.T - g.t.ov; }
        const int vy = std::min(Ph, g.t.H - oy), vx = std::min(Pw, g.t.W - ox);
        const int ey = std::min(Ph, vy + std::min(g.margin, Ph)), ex = std::min(Pw, vx + std::min(g.margin, Pw));
        px += (double)ey * ex;
        for (int it = 0; it < iters; it++) {
            const int reach = fast ? (iters - 1 - it) * m : (1 << 28) / 2;
            const FbRect rh = needed_rect_h(g.t, oy, ox, reach), rv = needed_rect_v(g.t, oy, ox, reach, m);
            px_h[it] += (double)std::max(0, std::min(ey, rh.y1) - rh.y0) * std::max(0, std::min(ex, rh.x1) - rh.x0);
            px_v[it] += (double)std::max(0, std::min(ey, rv.y1) - rv.y0) * std::max(0, std::min(ex, rv.x1) - rv.x0);
        }
    }
    {
        MaProfScope ps(ctx, MA_K_POLYEXP_M0, px);
        dim3 grid((Pw + K1_TX - 1) / K1_TX, (Ph + K1_TY - 1) / K1_TY, nwin);
        hipLaunchKernelGGL((fb_polyexp_m0<T>), grid, dim3(K1_THREADS), 0, ctx->stream, prev, next, g, pc, ws);
    }
    for (int it = 0; it < iters; it++) {
        const int last = it == iters - 1;
        const int reach = (iters - 1 - it) * m;  // see needed_rect_h
        if (fast) {
            {
                MaProfScope ps(ctx, MA_K_BLUR_V, px_v[it]);
                // the streaming form needs whole tap groups (m % (R/2) == 0: the reference's window of 99) and its ring in LDS
                const size_t lds_vs = (size_t)(BVS_NW * BV_R + 2 * m + 4 + 3 * (BV_R / 2) + 1) * 64 * sizeof(float);
                if (m % (BV_R / 2) == 0 && lds_vs <= LDS_MAX) {
                    const long long items = (long long)((Pw + 63) / 64) * nwin * 5;
                    hipLaunchKernelGGL((fb_blur_v_stream<BV_R, BVS_NW, FUSED>), dim3((unsigned)items), dim3(64 * BVS_NW),
                                       lds_vs, ctx->stream, g, m, taps, ws, nwin * 5, reach);
                } else {
                    const long long items = (long long)((Pw + 63) / 64) * ((Ph + BV_NW * BV_R - 1) / (BV_NW * BV_R)) * ma_xcd_slots(nwin * 5);
                    hipLaunchKernelGGL((fb_blur_v<BV_R, BV_NW, FUSED>), dim3(ma_xcd_grid(items)), dim3(64 * BV_NW), lds_v,
                                       ctx->stream, g, m, taps, ws, nwin * 5, reach);
                }
            }
            {
                MaProfScope ps(ctx, MA_K_BLUR_H_SOLVE, px_h[it]);
                const long long items = (long long)((Pw + BH_TXW - 1) / BH_TXW) * ((Ph + BH_ROWS - 1) / BH_ROWS) * ma_xcd_slots(nwin);
                // Q = 64-column chunks of the staged row tile (112 + 2m + 4 columns)
#define MA_BLUR_H(QQ)                                                                                               \
    do {                                                                                                            \
        if (last) hipLaunchKernelGGL((fb_blur_h_solve<BH_R, BH_NW, FUSED, QQ, true, BH_ROWS>), dim3(ma_xcd_grid(items)),     \
                                     dim3(64 * BH_NW), lds_h, ctx->stream, g, m, taps, ws, flow_out, nwin, reach);  \
        else hipLaunchKernelGGL((fb_blur_h_solve<BH_R, BH_NW, FUSED, QQ, false, BH_ROWS>), dim3(ma_xcd_grid(items)),         \
                                dim3(64 * BH_NW), lds_h, ctx->stream, g, m, taps, ws, flow_out, nwin, reach);       \
    } while (0)
                if (colsh <= 192) MA_BLUR_H(3);
                else if (colsh <= 256) MA_BLUR_H(4);
                else MA_BLUR_H(5);
#undef MA_BLUR_H
            }
        } else {
            {
                MaProfScope ps(ctx, MA_K_BLUR_V, px);
                dim3 grid((Pw + 255) / 256, Ph, nwin * 5);
                hipLaunchKernelGGL((fb_blur_v_simple<FUSED>), grid, dim3(256), 0, ctx->stream, g, m, taps, ws);
            }
            {
                MaProfScope ps(ctx, MA_K_BLUR_H_SOLVE, px);
                dim3 grid((Pw + 255) / 256, Ph, nwin);
                hipLaunchKernelGGL((fb_blur_h_solve_simple<FUSED>), grid, dim3(256), 0, ctx->stream, g, m, taps, ws,
                                   last, flow_out);
            }
        }
    }
    MA_HIP(hipGetLastError());
    return MA_OK;
}

int get_taps(ma_ctx* ctx, int winsize, const float** out)
{
    std::vector<float> k;
    make_window_taps(winsize, k);
    std::vector<float> t = ma_layout_taps(k);
    return ma_const_table(ctx, ((uint64_t)1 << 56) | (uint64_t)winsize, t.data(), t.size(), out);
}

int farneback_impl(ma_ctx* ctx, const void* prev, const void* next, int dtype, int H, int W, int tile, int overlap,
                   int winsize, int iterations, int poly_n, double poly_sigma, int flags, float* flow_out,
                   float* R0_out, float* R1_out, float* M0_out)
{
    MA_REQUIRE(ctx && prev && next && flow_out, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(H > 0 && W > 0, "image must be non-empty");
    MA_REQUIRE(tile >= 0 && overlap >= 0, "tile/overlap must be >= 0");
    MA_REQUIRE(winsize >= 1, "winsize must be >= 1");
    MA_REQUIRE(iterations >= 1, "iterations must be >= 1");
    MA_REQUIRE(poly_n == 1, "only poly_n == 1 is supported (the value microaligner passes)");
    MA_HIP(hipSetDevice(ctx->device));

    PolyConsts pc;
    if (!make_poly_consts(poly_sigma, &pc)) { ma_set_error("polynomial-expansion Gram matrix is singular"); return MA_EINVAL; }
    const float* taps = nullptr;
    MA_TRY(get_taps(ctx, winsize, &taps));
    const int m = winsize / 2;

    FbGeom g;
    g.t = ma_make_tiling(H, W, tile, overlap);
    g.pitch = (int)ma_align_up((size_t)g.t.Pw, 64);
    g.plane = (size_t)g.t.Ph * g.pitch;
    g.margin = 1 << 28;  // set per batch family in run_batch
    const int nwin_total = g.t.ntx * g.t.nty;
    const size_t per_win = g.plane * PL_COUNT * sizeof(float);
    MA_REQUIRE(per_win <= ctx->ws_limit, "one window does not fit the workspace limit");
    int batch = (int)(ctx->ws_limit / per_win);
    if (batch > nwin_total) batch = nwin_total;
    if ((long long)batch * 5 > 65535) batch = 65535 / 5;  // gridDim.z
    MA_TRY(ma_ws_reserve(ctx, per_win * batch));
    const bool fused = (flags & MA_FB_MULADD_FUSED) != 0;

    for (int t0 = 0; t0 < nwin_total; t0 += batch) {
        int n = nwin_total - t0 < batch ? nwin_total - t0 : batch;
        g.tile0 = t0;
        int rc;
#define RUN(TYPE)                                                                                           \
    rc = fused ? run_batch<TYPE, true>(ctx, (const TYPE*)prev, (const TYPE*)next, g, n, pc, m, taps, iterations, flow_out) \
               : run_batch<TYPE, false>(ctx, (const TYPE*)prev, (const TYPE*)next, g, n, pc, m, taps, iterations, flow_out)
        if (dtype == MA_U8) { RUN(uint8_t); }
        else if (dtype == MA_U16) { RUN(uint16_t); }
        else { RUN(float); }
#undef RUN
        if (rc != MA_OK) return rc;
    }
    if (R0_out || R1_out || M0_out) {
        // debug dump is only meaningful for iterations == 1 semantics of M; R0/R1 are always valid
        dim3 grid((g.t.Pw + 255) / 256, g.t.Ph, 5);
        float* ws = (float*)ctx->ws;
        if (R0_out) hipLaunchKernelGGL(fb_copy_planes, grid, dim3(256), 0, ctx->stream, g, ws, (int)PL_R0, R0_out);
        if (R1_out) hipLaunchKernelGGL(fb_copy_planes, grid, dim3(256), 0, ctx->stream, g, ws, (int)PL_R1, R1_out);
        if (M0_out) hipLaunchKernelGGL(fb_copy_planes, grid, dim3(256), 0, ctx->stream, g, ws, (int)PL_M, M0_out);
        MA_HIP(hipGetLastError());
    }
    return MA_OK;
}

} // namespace

extern "C" {

int ma_farneback_tiled(ma_ctx* ctx, const void* prev, const void* next, int dtype, int H, int W, int tile,
                       int overlap, int winsize, int iterations, int poly_n, double poly_sigma, int flags,
                       float* flow_out)
{
    return farneback_impl(ctx, prev, next, dtype, H, W, tile, overlap, winsize, iterations, poly_n, poly_sigma,
                          flags, flow_out, nullptr, nullptr, nullptr);
}

int ma_farneback_debug(ma_ctx* ctx, const void* prev, const void* next, int dtype, int H, int W, int winsize,
                       int iterations, double poly_sigma, int flags, float* flow_out, float* R0_planar,
                       float* R1_planar, float* M0_planar)
{
    return farneback_impl(ctx, prev, next, dtype, H, W, 0, 0, winsize, iterations, 1, poly_sigma, flags, flow_out,
                          R0_planar, R1_planar, M0_planar);
}

} // extern "C"
