// OptFlowRegistrator.register() behind one C entry point (ma_optflow_register).
//
// The level loop of microaligner/optflow_reg/optflow_registrator.py:93-173 -- pyramid (:175-202), per level
// warp -> dog -> tiled Farneback -> warp -> dog x2 -> mutual-information gate (shared_modules/similarity_scoring.py:
// 27-68) -> accept / reject bookkeeping with the reference's quirks (Q1 absolute-coordinate merge,
// Q2 no x2 when upscaling to full resolution, Q3 x4 in the middle-level reject branch) -- restated in C++ on top of
// the primitives of this library.  Everything is enqueued on the ctx stream; the only host round trip per level is
// the read of the two lists of NMI chunk scores (page-locked memory) that decides the gate.  A C or C++ host can
// drive the whole path with this call; microaligner_amd.OptFlowRegistrator.register() is a thin wrapper around it.
#include "ma_internal.h"

#include <cmath>
#include <cstring>

namespace {

// RAII handle of a buffer from the ctx cache
struct Buf {
    ma_ctx* ctx = nullptr;
    void* p = nullptr;
    Buf() = default;
    Buf(ma_ctx* c, size_t bytes) : ctx(c), p(ma_pool_alloc(c, bytes)) {}
    Buf(const Buf&) = delete;
    Buf& operator=(const Buf&) = delete;
    Buf(Buf&& o) noexcept : ctx(o.ctx), p(o.p) { o.p = nullptr; }
    Buf& operator=(Buf&& o) noexcept
    {
        if (this != &o) { reset(); ctx = o.ctx; p = o.p; o.p = nullptr; }
        return *this;
    }
    ~Buf() { reset(); }
    void reset() { if (p) ma_pool_free(ctx, p); p = nullptr; }
    explicit operator bool() const { return p != nullptr; }
};

#define MA_ALLOC(var, ctx, bytes)                 \
    Buf var((ctx), (bytes));                      \
    if (!var) return MA_ENOMEM

// A flow (h, w, 2) float32 on the device: `p` is what kernels read and write, `own` keeps it alive when it comes from
// the ctx cache (empty when `p` is the caller's output buffer).  cellkeys: per-cell maxima folded by a warp that read
// this flow (ma_warp_tiled_flowcells), for the merge that follows.
struct Flow {
    Buf own;
    float* p = nullptr;
    int h = 0, w = 0;
    Buf cellkeys;
    int ck_tile = -1, ck_overlap = -1;
    int alloc(ma_ctx* ctx, int hh, int ww)
    {
        h = hh; w = ww;
        own = Buf(ctx, (size_t)hh * ww * 2 * sizeof(float));
        p = (float*)own.p;
        return p ? MA_OK : MA_ENOMEM;
    }
};
// A warped image with the (min, max) its producer left on the device for the dog() that follows
struct Warped {
    Buf data, minmax;
};

inline bool is_tiled(int h, int w, int tile) { return (double)(h > w ? h : w) / (double)tile >= 2.0; }

// numpy's pairwise summation (numpy/core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum), unit stride
double np_pairwise_sum(const double* a, long n)
{
    if (n < 8) {
        double res = 0.;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        long i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

}  // namespace

// np.mean(scores) for a contiguous float64 vector exactly as numpy evaluates it: add.reduce is the pairwise sum of
// the whole vector, followed by one true division by n (checked against numpy in tests/test_host_logic.py).
double ma_np_mean(const double* v, long n)
{
    if (n <= 0) return NAN;
    return np_pairwise_sum(v, n) / (double)n;
}

namespace {

int warp_level(ma_ctx* ctx, const void* img, int dtype, int h, int w, Flow& flow, int tile, int overlap, Warped& out)
{
    // Warper.warp() on device arrays (optflow_registrator.py:113,125).  The kernel also leaves the min / max of its
    // output (for the dog() of the warped image) and, where the tiling has cells, the per-cell maxima of the flow
    // it reads (for a following merge of that flow).
    out.data = Buf(ctx, (size_t)h * w * ma_esize(dtype));
    out.minmax = Buf(ctx, 2 * sizeof(float));
    if (!out.data || !out.minmax) return MA_ENOMEM;
    if (tile > 2 * overlap && overlap > 0) {
        const size_t ncell = (size_t)(2 * ((w + tile - 1) / tile) + 1) * (2 * ((h + tile - 1) / tile) + 1);
        flow.cellkeys = Buf(ctx, (size_t)MA_FLOW_CELL_REPLICAS * ncell * sizeof(unsigned));
        if (!flow.cellkeys) return MA_ENOMEM;
        flow.ck_tile = tile; flow.ck_overlap = overlap;
        return ma_warp_tiled_flowcells(ctx, img, dtype, h, w, flow.p, tile, overlap, out.data.p, (float*)out.minmax.p,
                                       (unsigned*)flow.cellkeys.p);
    }
    return ma_warp_tiled_minmax(ctx, img, dtype, h, w, flow.p, tile, overlap, out.data.p, (float*)out.minmax.p);
}

int dog_level(ma_ctx* ctx, const void* img, int dtype, int h, int w, const float* minmax_dev, int flags, Buf& out,
              ma_ctx* run_on = nullptr)
{
    // dog(img, True) inside register() (optflow_registrator.py:118-119,128-130): default sigmas 5 / 9.  The result always
    // comes from ctx's buffer cache; run_on: the ctx whose stream and workspace do the work (the companion, for the
    // images that do not depend on the flow)
    out = Buf(ctx, (size_t)h * w);
    if (!out) return MA_ENOMEM;
    return ma_dog_u8_ex(run_on ? run_on : ctx, img, dtype, h, w, 5, 9, flags, minmax_dev, (uint8_t*)out.p, nullptr);
}

// points the dog() chain of ctx and of its companion at the ctx's sticky "max() == 0, not all zero" flag for the
// duration of a call
struct StickyScope {
    ma_ctx *ctx, *side;
    bool on;
    StickyScope(ma_ctx* c, ma_ctx* s, bool enable) : ctx(c), side(s), on(enable) {}
    void arm() { ctx->dog_sticky = ctx->dog_sticky_buf; side->dog_sticky = ctx->dog_sticky_buf; }
    ~StickyScope() { ctx->dog_sticky = nullptr; side->dog_sticky = nullptr; }
};

// waits for the companion stream before the buffers it writes can go back to the cache (error paths; a no-op otherwise)
struct SideDrain {
    ma_ctx* side;
    ~SideDrain() { if (side) (void)hipStreamSynchronize(side->stream); }
};

int merge_level(ma_ctx* ctx, Flow& f1, Flow& f2, int h, int w, int tile, int overlap, float* out)
{
    // _merge_flow_in_tiles (optflow_registrator.py:217-233)
    if (f1.cellkeys && f2.cellkeys && f1.ck_tile == tile && f2.ck_tile == tile && f1.ck_overlap == overlap &&
        f2.ck_overlap == overlap)
        return ma_merge_flows_tiled_cells(ctx, f1.p, f2.p, h, w, tile, overlap, (const unsigned*)f1.cellkeys.p,
                                          (const unsigned*)f2.cellkeys.p, out);
    return ma_merge_flows_tiled(ctx, f1.p, f2.p, h, w, tile, overlap, out);
}

struct Level { Buf data; const void* ptr = nullptr; int h = 0, w = 0, factor = 1; };

// _generate_img_pyr (optflow_registrator.py:175-202): smallest level first; a level is kept while both sides stay
// >= 100 px; the full-resolution image is appended when use_full_res_img
int build_pyramid(ma_ctx* ctx, const void* full, int dtype, int H, int W, const ma_params& p, std::vector<Level>& lv)
{
    std::vector<Level> down;
    const void* cur = full;
    int ch = H, cw = W;
    for (int l = 0; l < p.num_pyr_lvl; l++) {
        const int factor = 1 << (l + 1);
        if ((double)H / factor < 100. || (double)W / factor < 100.) break;
        Level L;
        L.h = (ch + 1) / 2; L.w = (cw + 1) / 2; L.factor = factor;
        L.data = Buf(ctx, (size_t)L.h * L.w * ma_esize(dtype));
        if (!L.data) return MA_ENOMEM;
        MA_TRY(ma_pyr_down(ctx, cur, dtype, ch, cw, L.data.p));
        L.ptr = L.data.p;
        cur = L.ptr; ch = L.h; cw = L.w;
        down.push_back(std::move(L));
    }
    for (size_t i = down.size(); i-- > 0;) lv.push_back(std::move(down[i]));
    if (p.use_full_res_img) {
        Level L;
        L.ptr = full; L.h = H; L.w = W; L.factor = 1;
        lv.push_back(std::move(L));
    }
    return MA_OK;
}

}  // namespace

extern "C" {

void ma_params_default(ma_params* p)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->num_pyr_lvl = 4;       // optflow_registrator.py:54-59
    p->num_iterations = 3;
    p->tile_size = 1000;
    p->overlap = 100;
    p->use_full_res_img = 0;
    p->use_dog = 0;
}

int ma_host_np_mean(const double* v, long n, double* out)
{
    MA_REQUIRE(v && out && n > 0, "bad argument");
    *out = ma_np_mean(v, n);
    return MA_OK;
}

int ma_optflow_register(ma_ctx* ctx, const void* ref, const void* mov, int dtype, int H, int W, const ma_params* params,
                        float* flow_out, ma_level_report* reports, int max_reports, int* n_reports)
{
    MA_REQUIRE(ctx && ref && mov && params && flow_out, "NULL argument");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(H > 0 && W > 0, "image must be non-empty");
    const ma_params p = *params;
    // optflow_registrator.py:177-184
    MA_REQUIRE(p.num_pyr_lvl >= 0, "Number of pyramid levels cannot be less than 0");
    MA_REQUIRE(!(p.num_pyr_lvl == 0 && !p.use_full_res_img),
               "Number of pyramid levels is 0 and use_full_res_img is False. Please change one of the parameters");
    MA_REQUIRE(p.num_pyr_lvl < 31, "too many pyramid levels");
    MA_REQUIRE(p.tile_size > 0 && p.overlap >= 0 && p.num_iterations >= 1, "bad tile_size / overlap / num_iterations");
    if (n_reports) *n_reports = 0;
    MA_HIP(hipSetDevice(ctx->device));

    std::vector<Level> ref_pyr, mov_pyr;
    MA_TRY(build_pyramid(ctx, ref, dtype, H, W, p, ref_pyr));
    MA_TRY(build_pyramid(ctx, mov, dtype, H, W, p, mov_pyr));
    const int n_lvl = (int)ref_pyr.size();
    if (n_lvl == 0) {
        // the reference dies here with UnboundLocalError (m_flow is never assigned, optflow_registrator.py:173)
        ma_set_error("invalid argument: image of shape (%d, %d) is too small for num_pyr_lvl=%d (every pyramid level must "
                     "keep >= 100 px per side) and use_full_res_img is False", H, W, p.num_pyr_lvl);
        return MA_EINVAL;
    }
    MA_REQUIRE(!reports || max_reports >= n_lvl, "reports buffer too small");
    const int tile = p.tile_size, ov = p.overlap;
    const int win = ov - (1 - ov % 2);   // largest odd number <= overlap (optflow_registrator.py:91)
    MA_REQUIRE(win >= 1, "overlap must be >= 1 (it sets the Farneback window)");

    // page-locked landing area for the chunk scores of the two NMIs of a level
    size_t max_chunks = 1;
    for (const Level& L : ref_pyr) {
        const size_t n = (size_t)L.h * L.w;
        const size_t c = is_tiled(L.h, L.w, tile) ? (n + (size_t)tile * tile - 1) / ((size_t)tile * tile) : 1;
        if (c > max_chunks) max_chunks = c;
    }
    MA_REQUIRE(max_chunks <= 65535, "too many NMI chunks");
    MA_TRY(ma_pinned_reserve(ctx, 2 * max_chunks * sizeof(double) + 64));
    double* sc_after = (double*)ctx->pinned;
    double* sc_before = sc_after + max_chunks;
    int* sticky_host = (int*)(sc_before + max_chunks);
    *sticky_host = 0;

    // dog(ref) and dog(mov) of every level depend on the pyramids only: the companion stream computes them, coarsest level
    // first, while this stream walks the levels -- the two full-resolution images are filtered under the coarse levels'
    // Farneback, whose few windows leave most of the chip idle.  Events order the streams: the companion starts when the
    // pyramids exist, every level waits for its two images.  Same kernels, same inputs: the results do not change.
    // MA_OPT_COMPANION_STREAM = 0: the same launches in the same order on the ctx stream itself (every kernel alone on the
    // chip: the per-kernel timings of a profile are then comparable from run to run)
    ma_ctx* side = ctx->companion ? ma_ctx_side(ctx) : ctx;
    if (!side) return MA_EHIP;
    {
        // the companion's workspace is sized ONCE, for its largest image, before anything is enqueued on it: growing it
        // level by level would wait for the stream (and reallocate) n_lvl times inside this prologue
        size_t need = 0;
        for (const Level& L : ref_pyr) {
            const size_t b = ma_dog_workspace_bytes(L.h, L.w, 5);
            if (b > need) need = b;
        }
        MA_TRY(ma_ws_reserve(side, need));
    }
    // float images only: every dog() of this call reports an input whose max() is 0 without being all zero (integer
    // images cannot be negative, there max() == 0 IS all zero and the uint8 zero image is what the reference goes on with)
    StickyScope sticky(ctx, side, dtype == MA_F32);
    if (dtype == MA_F32) {
        if (!ctx->dog_sticky_buf) MA_HIP(hipMalloc((void**)&ctx->dog_sticky_buf, 64));
        MA_HIP(hipMemsetAsync(ctx->dog_sticky_buf, 0, sizeof(int), ctx->stream));
        sticky.arm();
    }
    std::vector<Buf> ref_dogs(n_lvl), raw_dogs(n_lvl);
    SideDrain drain{side != ctx ? side : nullptr};   // declared after the buffers: destroyed (drained) before they are released
    {
        hipEvent_t ready = ma_ctx_sync_event(ctx, 0);
        if (!ready) return MA_EHIP;
        MA_HIP(hipEventRecord(ready, ctx->stream));
        MA_HIP(hipStreamWaitEvent(side->stream, ready, 0));
        for (int lvl = 0; lvl < n_lvl; lvl++) {
            const Level& R = ref_pyr[lvl];
            const Level& M = mov_pyr[lvl];
            hipEvent_t e_ref = ma_ctx_sync_event(ctx, 1 + 2 * (size_t)lvl), e_raw = ma_ctx_sync_event(ctx, 2 + 2 * (size_t)lvl);
            if (!e_ref || !e_raw) return MA_EHIP;
            MA_TRY(dog_level(ctx, R.ptr, dtype, R.h, R.w, nullptr, p.dog_flags, ref_dogs[lvl], side));
            MA_HIP(hipEventRecord(e_ref, side->stream));
            MA_TRY(dog_level(ctx, M.ptr, dtype, M.h, M.w, nullptr, p.dog_flags, raw_dogs[lvl], side));
            MA_HIP(hipEventRecord(e_raw, side->stream));
        }
    }

    Flow m_flow;   // the merged flow carried from level to level
    for (int lvl = 0; lvl < n_lvl; lvl++) {
        const bool last = lvl == n_lvl - 1;
        const Level& R = ref_pyr[lvl];
        const Level& M = mov_pyr[lvl];
        const int h = R.h, w = R.w;
        const size_t npx = (size_t)h * w;

        // mov_lvl = mov_raw on the first level, else warp(mov_raw, m_flow) (optflow_registrator.py:110-115)
        Warped pre;
        const void* mov_lvl = M.ptr;
        const float* mov_lvl_mm = nullptr;
        if (lvl > 0) {
            MA_TRY(warp_level(ctx, M.ptr, dtype, h, w, m_flow, tile, ov, pre));
            mov_lvl = pre.data.p;
            mov_lvl_mm = (const float*)pre.minmax.p;
        }
        // the gate always needs dog(ref); with use_dog it doubles as the Farneback input (:118-119,128)
        Buf ref_dog = std::move(ref_dogs[lvl]);
        MA_HIP(hipStreamWaitEvent(ctx->stream, ma_ctx_sync_event(ctx, 1 + 2 * (size_t)lvl), 0));
        // a single level at full resolution: its flow is the result if accepted, so it is computed in the caller's buffer
        Flow this_flow;
        if (last && lvl == 0 && h == H && w == W) {
            this_flow.p = flow_out; this_flow.h = h; this_flow.w = w;
        } else {
            MA_TRY(this_flow.alloc(ctx, h, w));
        }
        {
            const bool tiled = is_tiled(h, w, tile);   // flow_calc.py:60-64
            Buf mov_dog;
            const void *fb_ref = R.ptr, *fb_mov = mov_lvl;
            int fb_dtype = dtype;
            if (p.use_dog) {
                if (lvl == 0) {
                    // the first level's moving image IS the raw one: its dog() is the companion stream's
                    MA_HIP(hipStreamWaitEvent(ctx->stream, ma_ctx_sync_event(ctx, 2), 0));
                    fb_mov = raw_dogs[0].p;
                } else {
                    MA_TRY(dog_level(ctx, mov_lvl, dtype, h, w, mov_lvl_mm, p.dog_flags, mov_dog));
                    fb_mov = mov_dog.p;
                }
                fb_ref = ref_dog.p; fb_dtype = MA_U8;
            }
            // prev = moving image, next = reference image (flow_calc.py:34-35)
            MA_TRY(ma_farneback_tiled(ctx, fb_mov, fb_ref, fb_dtype, h, w, tiled ? tile : 0, tiled ? ov : 0, win,
                                      p.num_iterations, 1, 1.7, p.fb_flags, this_flow.p));
        }
        // gate (:125-132): mov_warped = warp(mov_lvl, this_flow); "before" is the RAW level, not the pre-warped one
        int n_after = 0, n_before = 0;
        {
            Warped warped;
            Buf warped_dog, raw_dog = std::move(raw_dogs[lvl]);
            MA_TRY(warp_level(ctx, mov_lvl, dtype, h, w, this_flow, tile, ov, warped));
            MA_TRY(dog_level(ctx, warped.data.p, dtype, h, w, (const float*)warped.minmax.p, p.dog_flags, warped_dog));
            const size_t chunk = is_tiled(h, w, tile) ? (size_t)tile * tile : 0;   // similarity_scoring.py:27-50
            MA_HIP(hipStreamWaitEvent(ctx->stream, ma_ctx_sync_event(ctx, 2 + 2 * (size_t)lvl), 0));
            // both halves of the gate share the reference labels and one pair of launches
            MA_TRY(ma_nmi_u8_enqueue2(ctx, (const uint8_t*)ref_dog.p, (const uint8_t*)warped_dog.p, (const uint8_t*)raw_dog.p,
                                      npx, chunk, sc_after, sc_before, (int)max_chunks, &n_after));
            n_before = n_after;
        }
        if (dtype == MA_F32)
            MA_HIP(hipMemcpyAsync(sticky_host, ctx->dog_sticky_buf, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        MA_HIP(hipStreamSynchronize(ctx->stream));   // the one host round trip of the level
        if (*sticky_host) {
            ma_set_error("invalid argument: a float image on the path (level %d of %d, %d x %d) has max() == 0 without being "
                         "all zero; the reference's dog() returns it unchanged (optflow_registrator.py:256-257), which only "
                         "the Python level loop models (engine='python')", lvl, n_lvl, h, w);
            return MA_EINVAL;
        }
        // the stream is idle: settle the per-kernel accounting now, so that its events are reused level after level
        // (a profiled run that only ever creates events stalls for ~0.2 s once the runtime's signal pool has to grow)
        if (ctx->profile) MA_TRY(ma_profile_flush(ctx));
        const double after = n_after == 1 ? sc_after[0] : ma_np_mean(sc_after, n_after);
        const double before = n_before == 1 ? sc_before[0] : ma_np_mean(sc_before, n_before);
        const bool accepted = after > before;
        if (reports) {
            ma_level_report& r = reports[lvl];
            r.factor = R.factor; r.h = h; r.w = w; r.mi_after = after; r.mi_before = before; r.accepted = accepted ? 1 : 0;
        }
        if (n_reports) *n_reports = lvl + 1;
        pre = Warped();
        ref_dog.reset();

        const int nh = last ? H : mov_pyr[lvl + 1].h, nw = last ? W : mov_pyr[lvl + 1].w;
        // Where the reference calls _upscale_flow_to_full_res (:204-215) the flow is pyrUp'ed ONCE to the full-resolution
        // size and not doubled (Q2); at full resolution it is returned as it is.
        auto upscale_to_full = [&](const float* src, int sh, int sw) -> int {
            if (sh == H && sw == W) {
                if (src != flow_out) MA_TRY(ma_memcpy_d2d(ctx, flow_out, src, (size_t)H * W * 2 * sizeof(float)));
                return MA_OK;
            }
            return ma_pyr_up_flow(ctx, src, sh, sw, 1.0f, flow_out, H, W);
        };
        if (accepted) {
            if (lvl == 0) {
                if (last) {
                    MA_TRY(upscale_to_full(this_flow.p, h, w));
                } else {
                    Flow nf;
                    MA_TRY(nf.alloc(ctx, nh, nw));
                    MA_TRY(ma_pyr_up_flow(ctx, this_flow.p, h, w, 2.0f, nf.p, nh, nw));   // :140
                    m_flow = std::move(nf);
                }
            } else if (last && h == H && w == W) {
                MA_TRY(merge_level(ctx, m_flow, this_flow, h, w, tile, ov, flow_out));       // :146, the result as it is
            } else {
                MA_ALLOC(merged, ctx, npx * 2 * sizeof(float));
                MA_TRY(merge_level(ctx, m_flow, this_flow, h, w, tile, ov, (float*)merged.p));
                if (last) {
                    MA_TRY(upscale_to_full((const float*)merged.p, h, w));
                } else {
                    Flow nf;
                    MA_TRY(nf.alloc(ctx, nh, nw));
                    MA_TRY(ma_pyr_up_flow(ctx, (const float*)merged.p, h, w, 2.0f, nf.p, nh, nw));   // :150
                    m_flow = std::move(nf);
                }
            }
        } else if (lvl == 0) {
            // zero flow of the next level's shape, or of the full shape on the last level (:155-160)
            if (last) {
                MA_TRY(ma_memset(ctx, flow_out, 0, (size_t)H * W * 2 * sizeof(float)));
            } else {
                Flow nf;
                MA_TRY(nf.alloc(ctx, nh, nw));
                MA_TRY(ma_memset(ctx, nf.p, 0, (size_t)nh * nw * 2 * sizeof(float)));
                m_flow = std::move(nf);
            }
        } else if (last) {
            if (m_flow.h == H && m_flow.w == W)                                                  // use_full_res_img: unchanged
                MA_TRY(ma_memcpy_d2d(ctx, flow_out, m_flow.p, (size_t)H * W * 2 * sizeof(float)));
            else
                MA_TRY(ma_pyr_up_flow(ctx, m_flow.p, m_flow.h, m_flow.w, 2.0f, flow_out, H, W));    // :164
        } else {
            Flow nf;
            MA_TRY(nf.alloc(ctx, nh, nw));
            MA_TRY(ma_pyr_up_flow(ctx, m_flow.p, m_flow.h, m_flow.w, 4.0f, nf.p, nh, nw));           // sic: x4 (:169)
            m_flow = std::move(nf);
        }
    }
    return MA_OK;
}

}  // extern "C"
