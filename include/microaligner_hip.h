/*
 * microaligner_hip.h -- C-ABI of the MI355X (gfx950) implementation of
 * microaligner's optical-flow registration hot path.
 *
 * The reference (VasylVaskivskyi/microaligner v1.0.0) is pure Python and has no
 * FFI layer of its own: on this path it calls six OpenCV functions and one
 * scikit-learn function from the optflow_reg package and
 * microaligner/shared_modules/similarity_scoring.py.  Each entry point below
 * replaces one of those call sites (cited per function) with a batched HIP
 * kernel pipeline; microaligner_amd/ binds them with ctypes and keeps the
 * reference's Python class API on top (INTEGRATION.md shows the binding).
 *
 * Conventions
 *  - Every function returns MA_OK (0) or a negative ma_status; the message of
 *    the last failure on the calling thread is ma_last_error().  No C++
 *    exception crosses this boundary.
 *  - Image/flow pointers are DEVICE pointers obtained from ma_malloc (or any
 *    hipMalloc'ed memory of the ctx's device) unless a parameter is named
 *    *_host.  Images are dense row-major (row stride = width * element size);
 *    flows and maps are (h, w, 2) float32 interleaved (x, y) as OpenCV returns
 *    them.
 *  - Work is enqueued on the ctx's HIP stream; functions that return data to
 *    the host synchronise that stream, the others do not (call ma_sync).
 *  - One ma_ctx per device per process; a ctx is not thread-safe.
 *  - dtype: MA_U8 / MA_U16 / MA_F32.
 */
#ifndef MICROALIGNER_HIP_H
#define MICROALIGNER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ma_ctx ma_ctx;

enum ma_dtype { MA_U8 = 0, MA_U16 = 1, MA_F32 = 2 };

enum ma_status {
    MA_OK = 0,
    MA_EINVAL = -1, /* bad argument (Python raises ValueError) */
    MA_ENOMEM = -2, /* host or device allocation failed */
    MA_EHIP = -3,   /* a HIP runtime call failed (Python raises RuntimeError) */
    MA_ENODEV = -4  /* no usable HIP device */
};

/* flags for ma_farneback_tiled */
enum ma_farneback_flags {
    MA_FB_MULADD_FUSED = 1 /* window blur uses fused multiply-add (models OpenCV builds whose
                              v_muladd is an FMA); default is multiply-then-add (x86 SSE baseline) */
};

/* flags for ma_dog_u8_ex: rounding model of the dog() chain.  OpenCV keeps GaussianBlur's separable filters
 * (filter.simd.hpp) and cv2.normalize's scaling (convert_scale.simd.hpp) in CPU-dispatched objects: the SSE2
 * baseline multiplies then adds (flags 0, the default of every other entry point), the AVX2 + FMA3 objects that an
 * x86-64 host with AVX2 selects at run time use fused multiply-adds (both flags). */
enum ma_dog_flags {
    MA_DOG_FUSED_BLUR = 1, /* row filter acc = fma(x_j, k_j, acc); column filter acc = fma(a + b, k_j, acc) */
    MA_DOG_FUSED_SCALE = 2, /* both normalize() steps: dst = fma(src, a, b) */
    MA_DOG_REPORT_ASYNC = 4 /* src_max_is_zero_host is PAGE-LOCKED memory (ma_host_alloc): the flag is copied there in stream
                               order and the call does not synchronise; valid after the next synchronisation of the ctx */
};

/* ---- library / context ------------------------------------------------- */
const char* ma_version(void);
const char* ma_last_error(void);
int ma_device_count(int* count);
int ma_ctx_create(int device, ma_ctx** out);
void ma_ctx_destroy(ma_ctx* ctx);
int ma_sync(ma_ctx* ctx);
/* Upper bound, in bytes, for the internal tile-batch workspace (default 48 GiB). */
int ma_ctx_set_workspace_limit(ma_ctx* ctx, size_t bytes);
/* The ctx's hipStream_t as an opaque pointer (for event timing by the caller). */
void* ma_ctx_stream(ma_ctx* ctx);
/* Returns the device buffers the ctx caches for the intermediates of ma_optflow_register to the driver (synchronises). */
int ma_ctx_trim(ma_ctx* ctx);
/* Bytes moved so far by the explicit host <-> device copies of this ctx (ma_memcpy_h2d / _d2h / _d2h_async and the
 * page transfers of ma_warp_pages_host); reset != 0 clears the counters after reading.  Lets a caller (and the tests)
 * check that a loop such as warp_and_save_pages (__main__.py:288-302) uploads its flow once, not once per page. */
int ma_ctx_transfer_stats(ma_ctx* ctx, unsigned long long* h2d_bytes, unsigned long long* d2h_bytes, int reset);

/* Run-time switches of a ctx.  MA_OPT_COMPANION_STREAM (default 1): ma_optflow_register computes the dog() of the
 * reference and of the moving image of every level on a second, low-priority HIP stream under the level loop; 0 keeps
 * everything on the ctx stream (same kernels, same inputs, same results -- for alone-on-the-chip kernel timings that are
 * comparable from run to run).  MA_OPT_WORKSPACE_LIMIT: same as ma_ctx_set_workspace_limit.
 * MA_OPT_WARP_BAND_BYTES (default 32 MiB): ma_warp_pages_host moves a page in bands of whole tile rows of at least this
 * many bytes (1: one tile row per band -- for tests of the band logic on small pages). */
enum ma_option { MA_OPT_COMPANION_STREAM = 1, MA_OPT_WORKSPACE_LIMIT = 2, MA_OPT_WARP_BAND_BYTES = 3 };
int ma_ctx_set_option(ma_ctx* ctx, int option, long long value);
int ma_ctx_get_option(ma_ctx* ctx, int option, long long* value);

/* ---- transfer engines -------------------------------------------------------
 * A stream of pairs arrives from the host and leaves to the host (the reference reads every page from TIFF and writes
 * every result back, __main__.py:398-433; one page in memory at a time, README.md:5).  To keep the kernels of pair k,
 * the upload of pair k+1 and the download of pair k-1 in flight together a ctx owns two more HIP streams besides its
 * compute stream: MA_ENGINE_H2D and MA_ENGINE_D2H.  Copies issued through ma_engine_memcpy_* run on the named engine's
 * stream and return when THAT copy is complete (the calling host thread waits for its own copy only: pageable host
 * memory is staged by the runtime, page-locked memory from ma_host_alloc goes by DMA); events order the engines:
 * ma_engine_record(ctx, e, ev) marks the work enqueued on engine e so far, ma_engine_wait(ctx, e, ev) makes everything
 * enqueued on engine e afterwards wait for it (no host wait).  ev: from ma_event_create.  The three engines of one ctx
 * may be driven from three host threads (one per engine); the counters behind ma_ctx_transfer_stats are atomic.
 * microaligner_amd.parallel.stream_pairs is the pipeline built on this. */
enum ma_engine { MA_ENGINE_COMPUTE = 0, MA_ENGINE_H2D = 1, MA_ENGINE_D2H = 2 };
int ma_engine_memcpy_h2d(ma_ctx* ctx, int engine, void* dst, const void* src_host, size_t bytes);
int ma_engine_memcpy_d2h(ma_ctx* ctx, int engine, void* dst_host, const void* src, size_t bytes);
int ma_engine_record(ma_ctx* ctx, int engine, void* ev);
int ma_engine_wait(ma_ctx* ctx, int engine, void* ev);
int ma_engine_sync(ma_ctx* ctx, int engine);
/* Pageable host memory handed to ma_engine_memcpy_* on a transfer engine is staged by the library itself: a ring of
 * page-locked 32 MiB chunks per direction, filled / drained by a small pool of host threads
 * (MICROALIGNER_COPY_THREADS, default 8 on hosts with >= 32 hardware threads) while the DMA engine moves the previous
 * chunk, so that the device only ever sees page-locked copies (the runtime's own staging of pageable memory shares
 * the shader engines with running kernels).  Each direction has its own pool; the drain of a download writes the
 * caller's array with non-temporal stores (the lines are not fetched for ownership nor kept in cache:
 * MICROALIGNER_COPY_NT, bit 0 = fills, bit 1 = drains, default 2).  ma_host_parallel_copy is the upload pool's memcpy,
 * ma_host_stream_copy the download pool's streaming copy, exported so that they can be checked without a GPU.
 * Host-only. */
int ma_host_parallel_copy(void* dst, const void* src, size_t bytes);
int ma_host_stream_copy(void* dst, const void* src, size_t bytes);
/* Host wait for an event (any engine). */
int ma_event_sync(ma_ctx* ctx, void* ev);

/* Identity of a device, for the per-rank lines of a multi-GPU run: marketing name, PCI bus id ("0000:c1:00.0"), free
 * and total HBM in bytes, compute units.  Any output pointer may be NULL.  Needs no ctx. */
int ma_device_info(int device, char* name, size_t name_len, char* pci_bus_id, size_t pci_len, size_t* mem_free,
                   size_t* mem_total, int* compute_units);
/* Shader clock (GHz) the chip sustains under a packed-FP32 load with the instruction mix of the window-blur kernels
 * (v_pk_add / v_pk_mul / v_pk_add, 4 waves per SIMD on every CU) for about `milliseconds`: s_memtime against the
 * 100 MHz s_memrealtime, median over the blocks.  bench.py prices the VALU-bound kernels against the peak at THIS
 * clock, measured in the same run, next to the 2.4 GHz nominal one.  Synchronises. */
int ma_clock_probe(ma_ctx* ctx, double milliseconds, double* sustained_ghz);

/* ---- device memory (caller owns host buffers; library owns nothing it returns
 *      except the error string) ------------------------------------------- */
int ma_malloc(ma_ctx* ctx, size_t bytes, void** dptr);
int ma_free(ma_ctx* ctx, void* dptr);
int ma_memcpy_h2d(ma_ctx* ctx, void* dst, const void* src_host, size_t bytes);
int ma_memcpy_d2h(ma_ctx* ctx, void* dst_host, const void* src, size_t bytes); /* synchronises */
int ma_memcpy_d2d(ma_ctx* ctx, void* dst, const void* src, size_t bytes);
/* Result transfer of the numpy-in / numpy-out API (register() returns the flow, optflow_registrator.py:173;
 * Warper.warp() returns the image, warper.py:53): enqueue only, the caller waits with ma_sync.  dst_host should
 * be page-locked (ma_host_alloc) for the copy to overlap later kernels; pageable memory makes HIP stage it. */
int ma_memcpy_d2h_async(ma_ctx* ctx, void* dst_host, const void* src, size_t bytes);
/* Page-locked host memory, portable across devices, for result arrays that are reused from call to call (a fresh
 * pageable array costs a first-touch page fault per 4 KiB on every call).  Not tied to a ctx. */
int ma_host_alloc(size_t bytes, void** hptr);
int ma_host_free(void* hptr);
/* Page-lock memory the CALLER owns, in place (hipHostRegister): a node-wide shared result array
 * (microaligner_amd.parallel.shared_array: the counterpart of the reference's memmapped output, __main__.py:116-132), a
 * reused input buffer.  From then on transfers to and from it go by DMA directly -- no staging copy, one pass over host
 * DRAM per byte instead of three, which is what an 8-rank node needs.  Registration pins the pages
 * (cost ~ a first touch of every page): for buffers that live across many transfers, not for one-shot arrays.  The range
 * must stay mapped until ma_host_unregister; each process registers its own mapping of shared memory. */
int ma_host_register(void* hptr, size_t bytes);
int ma_host_unregister(void* hptr);
/* How a transfer of [hptr, hptr + bytes) will be carried out: *direct = 1 when the WHOLE range is page-locked (inside one
 * ma_host_register registration, or inside one allocation the HIP runtime reports as host memory) and goes by DMA as it
 * is; 0 when any part of it is pageable or of unknown extent -- e.g. a copy that starts inside a registered row of a shared
 * array and runs past its end -- and the transfer is staged through the ctx's page-locked ring; 2 when a blocking copy of
 * the range would page-lock it for its own duration (opt-in, MICROALIGNER_TRANSIENT_PIN=1, arrays of >= 64 MiB:
 * hipHostRegister, one DMA, hipHostUnregister -- one pass over host DRAM instead of the staged path's three to four, at
 * the price of the registration). */
int ma_host_transfer_is_direct(const void* hptr, size_t bytes, int* direct);
/* Process-wide account of the transient page-locking: copies carried out under a registration of their own, how many of their
 * registrations were slower than 30 ms per GiB, the time spent registering and the volume, and whether the mechanism is (still)
 * active for big arrays in this process. */
int ma_transient_pin_stats(long long* copies, long long* slow_registrations, double* register_ms, double* gib, int* active);
int ma_memset(ma_ctx* ctx, void* dst, int value, size_t bytes);

/* ---- timing (HIP events on the ctx stream) ------------------------------ */
int ma_event_create(ma_ctx* ctx, void** ev);
int ma_event_destroy(ma_ctx* ctx, void* ev);
int ma_event_record(ma_ctx* ctx, void* ev);
int ma_event_elapsed_ms(ma_ctx* ctx, void* ev_start, void* ev_stop, float* ms); /* synchronises on ev_stop */
/* Per-kernel accounting: when enabled every kernel launch of the Farneback
 * pipeline is bracketed by events; ma_profile_get returns accumulated time and
 * launch count per kernel id (enum ma_kernel_id). */
enum ma_kernel_id {
    MA_K_POLYEXP_M0 = 0, MA_K_BLUR_V = 1, MA_K_BLUR_H_SOLVE = 2, MA_K_WARP = 3, MA_K_MERGE = 4,
    MA_K_PYR_DOWN = 5, MA_K_PYR_UP = 6, MA_K_DOG = 7, MA_K_NMI = 8, MA_K_OTHER = 9, MA_K_COUNT = 10
};
int ma_profile_enable(ma_ctx* ctx, int on);
int ma_profile_reset(ma_ctx* ctx);
int ma_profile_get(ma_ctx* ctx, int kernel_id, double* total_ms, long long* launches, double* px);

/* ---- Farneback ----------------------------------------------------------
 * Replaces cv2.calcOpticalFlowFarneback(prev, next, None, pyr_scale=0.5,
 * levels=0, winsize, iterations, poly_n, poly_sigma, OPTFLOW_FARNEBACK_GAUSSIAN)
 * as called from microaligner/optflow_reg/flow_calc.py:33-44, together with
 * TileFlowCalc's split -> fan-out -> stitch (flow_calc.py:59-98,
 * shared_modules/slicer.py:69-118, stitcher.py:72-118).
 *
 * tile == 0: one Farneback call on the whole (H, W) image (flow_calc.py:60-64).
 * tile  > 0: the image is cut into ceil(H/tile) x ceil(W/tile) windows of
 *            (tile + 2*overlap)^2 pixels, zero padded outside the image; each
 *            window is an independent Farneback problem with its own borders;
 *            the centre [overlap, overlap+tile) of every window is written to
 *            flow_out.  Bit-for-bit the tile-local semantics of the reference.
 * prev = moving image, next = reference image (flow_calc.py:34-35).
 * poly_n must be 1 (the only value the reference passes, flow_calc.py:41).
 */
int ma_farneback_tiled(ma_ctx* ctx, const void* prev, const void* next, int dtype, int H, int W,
                       int tile, int overlap, int winsize, int iterations, int poly_n,
                       double poly_sigma, int flags, float* flow_out);

/* Debug/validation variant for one untiled plane pair: also returns the
 * polynomial expansions (planar, 5 x H x W) and the first matrix field M. */
int ma_farneback_debug(ma_ctx* ctx, const void* prev, const void* next, int dtype, int H, int W,
                       int winsize, int iterations, double poly_sigma, int flags, float* flow_out,
                       float* R0_planar, float* R1_planar, float* M0_planar);

/* ---- the whole of OptFlowRegistrator.register() ------------------------------------------------
 * Replaces the level loop of microaligner/optflow_reg/optflow_registrator.py:93-173 (with _generate_img_pyr
 * :175-202, _upscale_flow_to_full_res :204-215, _merge_list_of_flows :235-240 and the mutual-information gate of
 * shared_modules/similarity_scoring.py:27-68): Gaussian pyramid of both images, and per level, smallest first,
 * warp by the flow so far -> dog() -> tiled Farneback -> warp -> dog() x 2 -> NMI gate -> merge / pyrUp or the
 * reject branch, including the reference's quirks (merge in absolute coordinates; no doubling when the result is
 * upscaled to full resolution; x4 in the middle-level reject branch).  ma_params carries the attributes of the
 * reference's class (optflow_registrator.py:54-59) plus the two rounding-model flag sets of this library.
 *
 * ref, mov: (H, W) device images of `dtype`; flow_out: (H, W, 2) float32 device buffer owned by the caller, with
 * mov(p) ~ ref(p + flow(p)).  reports (may be NULL): one record per pyramid level, smallest level first -- what the
 * reference prints (factor, MI scores, accept / reject); *n_reports receives the number of levels.  The call
 * synchronises the ctx stream once per level (the gate decision needs the NMI scores on the host); on return the
 * flow is enqueued, not necessarily complete (ma_sync).  MA_EINVAL when the pyramid would be empty: where the
 * reference dies with an UnboundLocalError (:173).  MA_EINVAL with "max() == 0" in ma_last_error() when a float32 image
 * on the path (a level, or a warped level) has a maximum of exactly 0 without being all zero: the reference's dog()
 * returns such an image unchanged (:256-257) and goes on with float labels, which this entry point does not model --
 * microaligner_amd.OptFlowRegistrator then repeats the call with its Python level loop, which does. */
typedef struct ma_params {
    int num_pyr_lvl;      /* 4    */
    int num_iterations;   /* 3    */
    int tile_size;        /* 1000 */
    int overlap;          /* 100: also sets the Farneback window, the largest odd number <= overlap (:91) */
    int use_full_res_img; /* 0    */
    int use_dog;          /* 0    */
    int fb_flags;         /* enum ma_farneback_flags */
    int dog_flags;        /* enum ma_dog_flags */
} ma_params;
typedef struct ma_level_report {
    int factor;           /* pyramid factor of the level: 2^k, 1 = full resolution */
    int h, w;             /* level shape */
    double mi_after;      /* mean NMI(dog(ref), dog(warped moving image)) */
    double mi_before;     /* mean NMI(dog(ref), dog(raw moving level)) */
    int accepted;         /* mi_after > mi_before */
} ma_level_report;
void ma_params_default(ma_params* p); /* the defaults of OptFlowRegistrator.__init__ (:54-59), flags 0 */
int ma_optflow_register(ma_ctx* ctx, const void* ref, const void* mov, int dtype, int H, int W, const ma_params* params,
                        float* flow_out, ma_level_report* reports, int max_reports, int* n_reports);
/* np.mean of a contiguous float64 vector exactly as numpy evaluates it (pairwise sum, one division): the reduction mi_tiled applies to the chunk scores (similarity_scoring.py:49).  Host-only helper,
 * exported so that the gate arithmetic of ma_optflow_register can be checked against numpy without a GPU. */
int ma_host_np_mean(const double* v, long n, double* out);

/* ---- remap / warp --------------------------------------------------------
 * ma_remap_bilinear replaces cv2.remap(src, map, None, cv2.INTER_LINEAR)
 * (border constant 0): warper.py:65 and optflow_registrator.py:45.
 * src: (sh, sw, cn) of dtype; map: (dh, dw, 2) float32 absolute source
 * coordinates; dst: (dh, dw, cn) of dtype.  cn is 1 or 2.  All dims < 32767. */
int ma_remap_bilinear(ma_ctx* ctx, const void* src, int dtype, int cn, int sh, int sw,
                      const float* map_xy, int dh, int dw, void* dst);

/* Warper.warp() (warper.py:37-76): split image and flow into zero-padded
 * (tile+2*overlap)^2 windows, map = float32(x_local - flow) per window
 * (warper.py:55-60), cv2.remap per window, stitch the centres. */
int ma_warp_tiled(ma_ctx* ctx, const void* img, int dtype, int H, int W, const float* flow,
                  int tile, int overlap, void* out);

/* ma_warp_tiled that also leaves by-products on the device for the steps that follow it inside register():
 * minmax_dev (may be NULL): (min, max) of the warped image, which the dog() of that image takes (ma_dog_u8_minmax);
 * flow_cellkeys_dev: maximum of both FLOW components over the (2*ntx+1) x (2*nty+1) cells the window borders
 * k*tile -+ overlap cut the image into (MA_FLOW_CELL_REPLICAS partial copies of the row-major cell array, to be folded
 * with max; order-preserving unsigned keys, NaN = largest), from which
 * ma_merge_flows_tiled_cells derives the per-window flow.max() tests of merge_two_flows
 * (optflow_registrator.py:38-42) without reading the flows again.  Requires tile > 2*overlap > 0. */
#define MA_FLOW_CELL_REPLICAS 8
int ma_warp_tiled_flowcells(ma_ctx* ctx, const void* img, int dtype, int H, int W, const float* flow, int tile,
                            int overlap, void* out, float* minmax_dev, unsigned* flow_cellkeys_dev);

/* Page-warp driver, warp_and_save_pages (__main__.py:288-302, 427-433): warps n_pages HOST images (the channel
 * and z pages of one cycle) with ONE device-resident flow, writing into caller-provided HOST buffers (e.g. rows
 * of the memmapped output TIFF).  Upload, kernel and download overlap on the ctx's three engines, in BANDS of whole
 * tile rows (MA_OPT_WARP_BAND_BYTES): a window never reads outside itself (warper.py:29-76), so a band runs as soon as
 * its source rows have arrived -- also within a single page (n_pages == 1 is Warper.warp() of a host image).  Pageable
 * and page-locked buffers are both accepted, per page.  MICROALIGNER_TRACE_PAGES=1 prints the timeline of the three
 * host threads to stderr.  Synchronous: returns when every output page is complete. */
int ma_warp_pages_host(ma_ctx* ctx, const void* const* pages_host, void* const* out_host, int n_pages, int dtype,
                       int H, int W, const float* flow, int tile, int overlap);
/* The bands ma_warp_pages_host cuts a page into for a band size of band_bytes: band b holds the output rows
 * [b * band_rows, min(H, (b + 1) * band_rows)) and runs once the source rows below min(H, (b + 1) * band_rows + overlap)
 * are in HBM.  band_rows is a whole number of tile rows holding at least band_bytes (or H: one band; always for
 * tile == 0).  Host-only (no ctx): the geometry can be checked without a GPU. */
int ma_warp_pages_plan(int dtype, int H, int W, int tile, int overlap, size_t band_bytes, int* band_rows, int* n_bands);

/* OptFlowRegistrator._merge_flow_in_tiles / merge_two_flows
 * (optflow_registrator.py:37-47,217-233): per window, out = flow2 if
 * flow1.max()==0, flow1 if flow2.max()==0, else flow1 + remap(flow2, -flow1). */
int ma_merge_flows_tiled(ma_ctx* ctx, const float* flow1, const float* flow2, int H, int W,
                         int tile, int overlap, float* out);
/* Same result with the cell maxima of both flows already on the device (ma_warp_tiled_flowcells). */
int ma_merge_flows_tiled_cells(ma_ctx* ctx, const float* flow1, const float* flow2, int H, int W, int tile, int overlap,
                               const unsigned* cellkeys1, const unsigned* cellkeys2, float* out);

/* ---- pyramids -------------------------------------------------------------
 * cv2.pyrDown(img) (optflow_registrator.py:194): dst is ((h+1)/2, (w+1)/2). */
int ma_pyr_down(ma_ctx* ctx, const void* src, int dtype, int h, int w, void* dst);
/* cv2.pyrUp(flow * scale, dstsize=(dw, dh)) on a 2-channel float32 flow
 * (optflow_registrator.py:140,150,164,169,212,214); the numpy pre-multiply
 * (2, 4 or 1) is fused.  Requires |dw-2w| == dw%2 and |dh-2h| == dh%2. */
int ma_pyr_up_flow(ma_ctx* ctx, const float* src, int h, int w, float scale, float* dst, int dh, int dw);

/* ---- DOG -------------------------------------------------------------------
 * img.max() etc. for the dog() shortcut (optflow_registrator.py:256). */
int ma_minmax(ma_ctx* ctx, const void* src, int dtype, size_t n, double* mn_host, double* mx_host);
/* The body of OptFlowRegistrator.dog (optflow_registrator.py:259-274):
 * normalize(0,1,MINMAX,32F) -> GaussianBlur(k,k,low) & GaussianBlur(k,k,high)
 * with k = low_sigma*8+1 -> hs-ls -> normalize(0,255,MINMAX,8U).  The whole chain is stream ordered
 * (its scalars never visit the host).  If src_max_is_zero_host is not NULL the call synchronises and
 * reports whether src.max() == 0, the case in which the reference returns the image unchanged (:256-257);
 * dst is all zero then. */
int ma_dog_u8(ma_ctx* ctx, const void* src, int dtype, int h, int w, int low_sigma, int high_sigma,
              uint8_t* dst, int* src_max_is_zero_host);

/* ---- NMI gate -----------------------------------------------------------------
 * mi_tiled (similarity_scoring.py:27-50): normalized_mutual_info_score of two
 * u8 label arrays over consecutive runs of `chunk` elements of the flattened
 * arrays (chunk == 0 or chunk >= n: one score over everything).  Writes
 * ceil(n/chunk) scores to scores_host (caller takes the mean). */
int ma_nmi_u8(ma_ctx* ctx, const uint8_t* a, const uint8_t* b, size_t n, size_t chunk,
              double* scores_host, int max_scores, int* n_scores);
/* mutual_information_test (similarity_scoring.py:52-58): both halves of the gate -- NMI(a, b0) and NMI(a, b1) -- share the
 * reference labels `a`, one pair of launches and one synchronisation. */
int ma_nmi_u8_pair(ma_ctx* ctx, const uint8_t* a, const uint8_t* b0, const uint8_t* b1, size_t n, size_t chunk,
                   double* scores0_host, double* scores1_host, int max_scores, int* n_scores);

/* ---- "next" rows ------------------------------------------------
 * np.maximum fold over z-planes (utils.py:92) and
 * cv2.normalize(.., 0, 255, NORM_MINMAX, CV_8U) (utils.py:94). */
int ma_max_project(ma_ctx* ctx, const void* planes, int dtype, int nz, size_t n, void* dst);
int ma_normalize_minmax_u8(ma_ctx* ctx, const void* src, int dtype, size_t n, uint8_t* dst);

/* Mat::convertTo(CV_32F) of an integer image (exact).  cv2.calcOpticalFlowFarneback converts each of its two inputs to
 * float32 on its own (optflowgf.cpp), so the reference accepts a uint8 reference image next to a uint16 moving image
 * (flow_calc.py:33-44); the Python layer converts such a pair with this before ma_farneback_tiled. */
int ma_convert_f32(ma_ctx* ctx, const void* src, int dtype, size_t n, float* dst);

/* transform_img_with_tmat (utils.py:98-114) after padding: skimage.transform.warp(img,
 * AffineTransform(inverse_3x3), output_shape=img.shape, preserve_range=True).astype(dtype) -- bilinear,
 * constant border 0, clipped to the input range.  inverse_3x3_host: 9 doubles, row major, the matrix the
 * reference obtains from np.linalg.pinv (output pixel -> input coordinate). */
int ma_warp_affine(ma_ctx* ctx, const void* src, int dtype, int h, int w, const double* inverse_3x3_host,
                   void* dst);

/* Producer/consumer variants that keep the min / max of an image on the device.  dog() starts with
 * cv2.normalize(img, 0, 1, NORM_MINMAX) (optflow_registrator.py:259): when the image was just written by a warp
 * (optflow_registrator.py:113,125 via Warper.warp) or by cv2.pyrDown (:194), the producing kernel can reduce its
 * own output -- ma_warp_tiled_minmax / ma_pyr_down_minmax store (min, max) as two floats at minmax_dev (device
 * memory) -- and ma_dog_u8_minmax takes them instead of reading the image once more.  Results are identical to
 * the plain entry points. */
int ma_warp_tiled_minmax(ma_ctx* ctx, const void* img, int dtype, int H, int W, const float* flow, int tile,
                         int overlap, void* out, float* minmax_dev);
int ma_pyr_down_minmax(ma_ctx* ctx, const void* src, int dtype, int h, int w, void* dst, float* minmax_dev);
int ma_dog_u8_minmax(ma_ctx* ctx, const void* src, int dtype, int h, int w, int low_sigma, int high_sigma,
                     const float* src_minmax_dev, uint8_t* dst, int* src_max_is_zero_host);

/* ma_dog_u8 / ma_dog_u8_minmax with the rounding model as a parameter (enum ma_dog_flags); src_minmax_dev may be
 * NULL (the image is reduced first).  Replaces the same call sites: optflow_registrator.py:259-274. */
int ma_dog_u8_ex(ma_ctx* ctx, const void* src, int dtype, int h, int w, int low_sigma, int high_sigma, int flags,
                 const float* src_minmax_dev, uint8_t* dst, int* src_max_is_zero_host);

/* cv2.warpAffine(src, M, dsize=(dw, dh)) with the default flags (INTER_LINEAR, BORDER_CONSTANT 0), the call of
 * FeatureRegistrator.transform_img (feature_reg/feature_registrator.py:128-132) for images up to 32000 px.
 * m2x3_host: the FORWARD 2x3 matrix (6 doubles, row major) exactly as passed to cv2.warpAffine; it is inverted in
 * double and evaluated in OpenCV's 10-bit fixed point (WarpAffineInvoker), then sampled like cv2.remap. */
int ma_warp_affine_cv(ma_ctx* ctx, const void* src, int dtype, int sh, int sw, const double* m2x3_host, int dh, int dw,
                      void* dst);

/* ---- descriptor matching (FeatureRegistrator) -------------------------------------------------
 * Replaces cv2.FlannBasedMatcher().knnMatch(query, train, k=2) (feature_reg/feature_detection.py:137-141) with the
 * EXACT search: for every query row the two train rows at the smallest L2 distance, ties to the lower index.
 * query: (nq, dim) float32, train: (nt, dim) float32, dim a multiple of 4 (pad with zeros), nt >= 2; idx_out:
 * (nq, 2) int32, dist_out: (nq, 2) float32 SQUARED distances, accumulated in float32 over ascending dimension
 * (d2 = d2 + diff*diff, two roundings).  All device pointers. */
int ma_knn2_l2(ma_ctx* ctx, const float* query, int nq, const float* train, int nt, int dim, int* idx_out,
               float* dist_out);

/* The same search with the strategy chosen by the caller.  The RESULT is defined by the exact kernel in every mode.
 * MA_KNN_FILTERED (what MA_KNN_AUTO picks for big sets): the train rows are first ranked by |t|^2 - 2 q.t on the matrix
 * cores, the four best of every split of the train set are re-evaluated with the defining sum, and a rounding bound
 * certifies that no other row can enter or tie the pair; queries without a certificate (near-ties, duplicated
 * descriptors) are served by the exact kernel.  Bit-identical to MA_KNN_EXACT by construction.  The ranking runs on the
 * FP16 matrix cores with every operand scaled by a power of two and split into two float16 numbers (three
 * v_mfma_f32_32x32x16_f16 per sixteen dimensions, dim <= 208; round 6) or, MA_KNN_FILTERED_F32 (dim <= 216; round 3), on
 * the FP32 ones (v_mfma_f32_32x32x2_f32, an exact fmaf chain); the certificate's bound follows the ranking's error.
 * uncertified_host: NULL, or where to put the number of queries that took the exact fallback (this makes the call
 * synchronous). */
enum ma_knn_mode { MA_KNN_AUTO = 0, MA_KNN_EXACT = 1, MA_KNN_FILTERED = 2, MA_KNN_FILTERED_F32 = 3 };
int ma_knn2_l2_ex(ma_ctx* ctx, const float* query, int nq, const float* train, int nt, int dim, int* idx_out,
                  float* dist_out, int mode, int* uncertified_host);

/* Matching step after the 2-NN search (feature_detection.py:142-158): Lowe's ratio test sqrt(d0) < ratio * sqrt(d1) over the
 * (squared) distances ma_knn2_l2 left on the device, then the counterpart of cv.estimateAffinePartial2D(query points -> train
 * points, RANSAC, confidence) -- bit for bit microaligner_amd/feature_reg/sparse_cpu.py:estimate_affine_partial_2d, whose
 * random sequence numpy's Generator(PCG64(seed)) defines: rng_state = {state hi, state lo, inc hi, inc lo} of
 * numpy.random.PCG64(seed).state; NULL: the state of seed 0, the reference-side default (a C host needs no numpy).
 * idx / dist_sq: (nq, 2) device arrays; query_pts / train_pts: (n, 2) float64 (x, y) device
 * arrays.  Results on the host: the 2 x 3 matrix (row major), the number of good matches and
 * status 0 = matrix valid, 1 = fewer than 3 good matches (the reference returns the identity), 2 = no model (cv2 returns
 * None), 3 = coordinates not integer-valued or too large for exact sums: not computed, use the host statement. */
int ma_match_similarity(ma_ctx* ctx, const int* idx, const float* dist_sq, int nq, const double* query_pts,
                        const double* train_pts, int nt, float ratio, double confidence, double reproj_threshold,
                        int max_iters, const unsigned long long rng_state[4], double* m2x3_host, int* n_good_host,
                        int* status_host);
/* Host-only pieces of the above, exported for the CPU tests: `count` draws of Generator.choice(n, 2, replace=False) from the
 * given PCG64 state (pairs_out: count x 2), and the adaptive iteration count of the RANSAC loop. */
int ma_host_pcg64_choice2(const unsigned long long state[4], int n, int count, int* pairs_out);
int ma_host_ransac_iterations(int count, int n, double confidence, int max_iters, int it, int* iters);
/* One round of FeatureRegistrator's level loop (feature_reg/feature_registrator.py:162-207) in one call: features of the
 * current moving image (ma_feature_extract on dog(current) -- or on the uint8 image itself without use_dog), 2-NN, ratio test
 * and RANSAC against the level's reference features (ma_knn2_l2, ma_match_similarity, seed 0), candidate =
 * cv2.warpAffine(current, estimate) (ma_warp_affine_cv), dog(candidate), and both halves of the mutual-information gate
 * (NMI(ref_gate, dog(candidate)), NMI(ref_gate, dog(current)), chunks of nmi_chunk elements, 0 = whole image).  Everything
 * stays on the device; the result struct and the chunk scores are all that comes back (the call synchronises).
 * current: (H, W) device image; current_gate: dog(current) if the caller has it, else NULL and it is written to
 * current_gate_out; ref_gate: dog(reference level); ref_desc / ref_pts / n_ref: the reference features as ma_feature_extract
 * left them; tables as for ma_daisy_describe.  candidate_out (H, W, dtype) and candidate_gate_out (H, W uint8) are written
 * unless the estimate is the identity (then the "after" scores compare the current image with itself, as the reference's
 * loop does).  status: ma_match_similarity's (0 matrix valid, 1 fewer than 3 good matches, 2 no model, 3 not computed: use the
 * host statement -- nothing after the matching has run then) or 4 = the current image has no features or the reference fewer
 * than two (the reference loop takes the identity without a "Good matches" line).  zero_max: bit 0 = dog(current) met an image
 * whose max() is 0, bit 1 = dog(candidate) did -- the reference's dog() returns such an image unchanged (:288-291); the caller
 * repeats the level with the step-by-step entry points. */
typedef struct ma_feature_round_result {
    double m2x3[6];
    int n_query, n_good, status, is_identity, n_scores, zero_max;
} ma_feature_round_result;
int ma_feature_round(ma_ctx* ctx, const void* current, int dtype, int H, int W, const uint8_t* current_gate,
                     uint8_t* current_gate_out, const uint8_t* ref_gate, const float* ref_desc, const double* ref_pts,
                     int n_ref, int tile, int use_dog, size_t nmi_chunk, const double* const* weights_host, const int* radii,
                     const double* cos_sin_host, const double* offs_host, size_t workspace_bytes, void* candidate_out,
                     uint8_t* candidate_gate_out, double* scores_after_host, double* scores_before_host, int max_scores,
                     ma_feature_round_result* res);

/* ---- dense halves of the feature stage (FeatureRegistrator), batched over nt square tiles of side P ----
 * ma_fast_nms: FAST-9/16 corner score of the tile interiors (tile[margin:-margin, margin:-margin], as
 * feature_detection.py:105 cuts them) kept only at strict 3x3 local maxima -- the pixels
 * cv.FastFeatureDetector_create(threshold, True, TYPE_9_16).detect() returns, with their responses.
 * tiles: (nt, P, P) uint8; score_out: (nt, P-2*margin, P-2*margin) int32, 0 = no keypoint. */
int ma_fast_nms(ma_ctx* ctx, const uint8_t* tiles, int nt, int P, int margin, int threshold, int* score_out);
/* The same detector followed by the reference's selection (feature_detection.py:105-106: sorted by response, strongest
 * first, Python's stable sort keeping row-major order among equals, cut to nfeatures_limit) on the device: kp_out
 * (device, nt x limit x 3 int32) receives (x, y, response) per keypoint in interior coordinates, counts_host[t] how
 * many of tile t's `limit` slots are filled.  limit <= 8192.  Synchronises (the counts). */
int ma_fast_keypoints(ma_ctx* ctx, const uint8_t* tiles, int nt, int P, int margin, int threshold, int limit, int* kp_out,
                      int* counts_host);
/* split_image_into_tiles (tile_registration.py:27-34, slicer.py:69-118) for a uint8 device image: the n_tiles windows
 * first_tile .. of the row-major tile grid, each (tile + 2*overlap)^2, zero outside the image, into tiles_out
 * (n_tiles, P, P). */
int ma_cut_tiles_u8(ma_ctx* ctx, const uint8_t* img, int H, int W, int tile, int overlap, int first_tile, int n_tiles,
                    uint8_t* tiles_out);
/* find_features of a whole uint8 device image in one call (tile_registration.py:78-97 -> feature_detection.py:88-120,
 * 161-168): the feature windows are cut (ma_cut_tiles_u8), the corners detected, ranked and cut to `limit` per tile
 * (ma_fast_keypoints), compacted tile by tile in combine_features' order (tiles with fewer than 3 keypoints dropped) and
 * described (ma_daisy_describe, the smoothing limited to each window's image content) -- everything stays on the device, in
 * batches of tiles sized from workspace_bytes (0: 8 GiB); only the number of keypoints comes back (one synchronisation).
 * desc_out: (capacity, 200) float32, pts_out: (capacity, 2) float64 (x, y) in image coordinates, resp_out: (capacity) int32
 * responses, all device; capacity >= tiles x limit.  Tables as for ma_daisy_describe. */
int ma_feature_extract(ma_ctx* ctx, const uint8_t* img, int H, int W, int tile, int overlap, int threshold, int limit,
                       const double* const* weights_host, const int* radii, const double* cos_sin_host, const double* offs_host,
                       size_t workspace_bytes, int capacity, float* desc_out, double* pts_out, int* resp_out, int* n_out_host);
/* The same work ENQUEUED: no synchronisation, the number of keypoints lands in n_out_pinned -- page-locked host memory
 * (ma_host_alloc) -- in stream order; the caller reads it after waiting for the ctx's stream or for an event recorded behind the
 * call (ma_event_record / ma_event_sync).  This is how FeatureRegistrator computes the REFERENCE image's features of every level on
 * a second ctx while the first one works through the moving image's coarse levels (feature_registrator.py:70-76 computes them up
 * front; nothing in a level's rounds needs the next level's reference features). */
int ma_feature_extract_enqueue(ma_ctx* ctx, const uint8_t* img, int H, int W, int tile, int overlap, int threshold, int limit,
                               const double* const* weights_host, const int* radii, const double* cos_sin_host,
                               const double* offs_host, size_t workspace_bytes, int capacity, float* desc_out, double* pts_out,
                               int* resp_out, int* n_out_pinned);
/* ma_daisy_describe: DAISY descriptors (radius 21, 3 rings x 8 locations + centre, 8 orientation bins = 200 floats,
 * no normalisation, bilinear sampling; feature_detection.py:107-110) at nkp keypoints.  tiles: (nt, P, P) uint8 or
 * float32 (device).  weights_host[c] / radii[c]: centre-first half of the c-th incremental Gaussian kernel (host
 * doubles, radii[c] + 1 of them); cos_sin_host: 8 x (cos, sin) of the orientation bins; offs_host: 25 x (dy, dx)
 * sampling offsets (entry 0 = the keypoint itself, cube 0; entries 1 + 8 r + j sample cube r).  kp_tile: tile index
 * per keypoint, kp_xy: (x, y) float64 per keypoint in tile coordinates (device).  desc_out: (nkp, 200) float32. */
int ma_daisy_describe(ma_ctx* ctx, const void* tiles, int dtype, int nt, int P, const double* const* weights_host,
                      const int* radii, const double* cos_sin_host, const double* offs_host, const int* kp_tile,
                      const double* kp_xy, int nkp, float* desc_out);

#ifdef __cplusplus
}
#endif
#endif /* MICROALIGNER_HIP_H */
