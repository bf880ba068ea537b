/*
 * ma_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the third-party arithmetic that microaligner's
 * optical-flow hot path delegates to:
 *   - cv2.calcOpticalFlowFarneback  (reference call site: microaligner/optflow_reg/flow_calc.py:33-44)
 *   - cv2.remap INTER_LINEAR        (microaligner/optflow_reg/warper.py:65, optflow_registrator.py:45)
 *   - cv2.pyrDown / cv2.pyrUp       (optflow_registrator.py:194 / :140,150,164,169,212,214)
 *   - cv2.normalize + cv2.GaussianBlur (the dog() chain, optflow_registrator.py:249-274)
 *   - sklearn normalized_mutual_info_score (microaligner/shared_modules/similarity_scoring.py:36,44)
 *
 * PARITY STATUS: "parity unpinned" for every OpenCV primitive.  OpenCV
 * (opencv-contrib-python==4.5.5.64, environment.yaml:75) is an un-vendored
 * dependency that is absent from /root/reference and from this image, and the
 * reference ships no tests or golden vectors.  The functions below restate the
 * published OpenCV 4.5.5 algorithms (modules/video/src/optflowgf.cpp,
 * modules/imgproc/src/{imgwarp,pyramids,smooth.dispatch}.cpp + filter.simd.hpp,
 * modules/core/src/{norm,convert_scale}.cpp) as specified in SURVEY.md
 * Appendix A: same operation order, same float/double placement, x86 SSE
 * baseline semantics (multiply and add are separate roundings, no FMA; the
 * `fused` switches model the builds where OpenCV's v_muladd lowers to FMA).
 * They are pinned only by closed-form known-answer tests (tests/test_oracle_kat.py).
 * The NMI function IS pinned: tests check it against the installed scikit-learn.
 * The pin for the rest is one run away on any machine with the reference's OpenCV:
 * tests/golden/make_cv2_golden.py (numpy + cv2 only) writes tests/golden/cv2_4.5.5.npz,
 * tests/test_cv2_golden.py holds every function below (and the HIP path) against it.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Host threads for the row-parallel loops below (image rows, NMI chunks, Farneback windows are independent, so the
 * results do not depend on this number).  Default 1. */
static int g_threads = 1;
/* The row loops stop scaling long before the window fan-out does (measured on a 2 x 64-core EPYC: a 4096^2 dog()
 * takes 0.07 s on 16-64 threads and 1 s on 256), so they use at most ORC_ROW_THREADS_MAX threads. */
#define ORC_ROW_THREADS_MAX 32
static int g_row_threads = 1;
void orc_set_threads(int n)
{
    g_threads = n < 1 ? 1 : n;
    g_row_threads = g_threads < ORC_ROW_THREADS_MAX ? g_threads : ORC_ROW_THREADS_MAX;
}
int orc_get_threads(void) { return g_threads; }

/* ISA clones of the hot loops (resolved at load time, so the library runs on any x86-64): wider vectors only --
 * -ffp-contract=off holds for every clone, each element still sees the same IEEE operations in the same order. */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define ORC_CLONES __attribute__((target_clones("default", "avx2", "avx512f")))
#else
#define ORC_CLONES
#endif

#define ORC_U8 0
#define ORC_U16 1
#define ORC_F32 2

#define ORC_OK 0
#define ORC_EINVAL (-1)
#define ORC_ENOMEM (-2)

/* multiply-add with a selectable rounding model: OpenCV's v_muladd is
 * mul-then-add on the x86 SSE baseline and a fused op on FMA3/NEON builds. */
static inline float muladd_f(float a, float b, float c, int fused)
{
    if (fused) return fmaf(a, b, c);
    float p = a * b;
    return p + c;
}

/* cvRound: round-half-to-even as cvtss2si/cvtsd2si do; out of range -> INT_MIN */
static inline int cv_round_f(float v)
{
    if (!(fabsf(v) < 2147483648.0f)) return INT_MIN;
    return (int)lrintf(v);
}
static inline int cv_round_d(double v)
{
    if (!(fabs(v) < 2147483648.0)) return INT_MIN;
    return (int)lrint(v);
}
static inline int cv_floor_f(float v)
{
    if (!(fabsf(v) < 2147483648.0f)) return INT_MIN;
    int i = (int)v;
    return i - (v < (float)i);
}
static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline float load_as_f32(const void* p, int dtype, size_t i)
{
    switch (dtype) {
    case ORC_U8: return (float)((const uint8_t*)p)[i];
    case ORC_U16: return (float)((const uint16_t*)p)[i];
    default: return ((const float*)p)[i];
    }
}

/* ------------------------------------------------------------------------- */
/* Farneback, single scale (levels=0), OPTFLOW_FARNEBACK_GAUSSIAN             */
/* SURVEY.md Appendix A.1                                                     */
/* ------------------------------------------------------------------------- */

/* Cholesky solve as OpenCV's hal::Cholesky64f (inverse of an SPD matrix with
 * b = identity).  A is m x m (overwritten by L), b is m x n (overwritten). */
static int chol_solve(double* A, int m, double* b, int n)
{
    int i, j, k;
    double s;
    for (i = 0; i < m; i++) {
        for (j = 0; j < i; j++) {
            s = A[i * m + j];
            for (k = 0; k < j; k++) s -= A[i * m + k] * A[j * m + k];
            A[i * m + j] = s * A[j * m + j];
        }
        s = A[i * m + i];
        for (k = 0; k < j; k++) {
            double t = A[i * m + k];
            s -= t * t;
        }
        if (s < DBL_EPSILON) return 0;
        A[i * m + i] = 1. / sqrt(s);
    }
    for (i = 0; i < m; i++)
        for (j = 0; j < n; j++) {
            s = b[i * n + j];
            for (k = 0; k < i; k++) s -= A[i * m + k] * b[k * n + j];
            b[i * n + j] = s * A[i * m + i];
        }
    for (i = m - 1; i >= 0; i--)
        for (j = 0; j < n; j++) {
            s = b[i * n + j];
            for (k = m - 1; k > i; k--) s -= A[k * m + i] * b[k * n + j];
            b[i * n + j] = s * A[i * m + i];
        }
    return 1;
}

/* 1-D kernels g, xg, xxg (index -n..n, pointers are to the centre) and the four
 * used entries of the inverse Gram matrix.  A.1 step 2. */
int orc_farneback_prepare_gaussian(int n, double sigma, float* g, float* xg, float* xxg,
                                   double* ig11, double* ig03, double* ig33, double* ig55)
{
    int x, y;
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    double s = 0.;
    for (x = -n; x <= n; x++) {
        g[x] = (float)exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1. / s;
    for (x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G[36], I6[36];
    memset(G, 0, sizeof(G));
    memset(I6, 0, sizeof(I6));
    for (y = -n; y <= n; y++)
        for (x = -n; x <= n; x++) {
            /* float products accumulated into double, as the C++ expression types give */
            G[0] += g[y] * g[x];
            G[1 * 6 + 1] += g[y] * g[x] * x * x;
            G[3 * 6 + 3] += g[y] * g[x] * x * x * x * x;
            G[5 * 6 + 5] += g[y] * g[x] * x * x * y * y;
        }
    G[2 * 6 + 2] = G[0 * 6 + 3] = G[0 * 6 + 4] = G[3 * 6 + 0] = G[4 * 6 + 0] = G[1 * 6 + 1];
    G[4 * 6 + 4] = G[3 * 6 + 3];
    G[3 * 6 + 4] = G[4 * 6 + 3] = G[5 * 6 + 5];
    for (x = 0; x < 6; x++) I6[x * 6 + x] = 1.;
    if (!chol_solve(G, 6, I6, 6)) return ORC_EINVAL;
    *ig11 = I6[1 * 6 + 1];
    *ig03 = I6[0 * 6 + 3];
    *ig33 = I6[3 * 6 + 3];
    *ig55 = I6[5 * 6 + 5];
    return ORC_OK;
}

/* 3x3 Gaussian pre-blur with the fixed kernel [1/4 1/2 1/4], rows then
 * columns, BORDER_REFLECT_101.  A.1 step 1 (sigma=0, smooth_sz=3). */
static void preblur3(const float* src, float* dst, int h, int w, float* tmp)
{
    const float k0 = 0.5f, k1 = 0.25f;
#pragma omp parallel for schedule(static) num_threads(g_row_threads)
    for (int y = 0; y < h; y++) {
        const float* s = src + (size_t)y * w;
        float* t = tmp + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            float a = s[reflect101(x - 1, w)], b = s[reflect101(x + 1, w)];
            t[x] = s[x] * k0 + (a + b) * k1;
        }
    }
#pragma omp parallel for schedule(static) num_threads(g_row_threads)
    for (int y = 0; y < h; y++) {
        const float* t0 = tmp + (size_t)reflect101(y - 1, h) * w;
        const float* t1 = tmp + (size_t)y * w;
        const float* t2 = tmp + (size_t)reflect101(y + 1, h) * w;
        float* d = dst + (size_t)y * w;
        for (int x = 0; x < w; x++) d[x] = t1[x] * k0 + (t0[x] + t2[x]) * k1;
    }
}

/* Polynomial expansion -> R, 5 floats per pixel interleaved.  A.1 step 2. */
static int poly_exp(const float* src, float* dst, int h, int w, int n, double sigma)
{
    float* kbuf = (float*)malloc(sizeof(float) * (n * 6 + 3));
    const size_t rowlen = (size_t)(w + n * 2) * 3;
    float* rowbuf = (float*)malloc(sizeof(float) * rowlen * g_row_threads);
    if (!kbuf || !rowbuf) { free(kbuf); free(rowbuf); return ORC_ENOMEM; }
    float* g = kbuf + n;
    float* xg = g + n * 2 + 1;
    float* xxg = xg + n * 2 + 1;
    double ig11, ig03, ig33, ig55;
    int rc = orc_farneback_prepare_gaussian(n, sigma, g, xg, xxg, &ig11, &ig03, &ig33, &ig55);
    if (rc) { free(kbuf); free(rowbuf); return rc; }

#pragma omp parallel for schedule(static) num_threads(g_row_threads)
    for (int y = 0; y < h; y++) {
        int k, x;
#ifdef _OPENMP
        float* row = rowbuf + rowlen * omp_get_thread_num() + n * 3;
#else
        float* row = rowbuf + n * 3;
#endif
        float g0 = g[0], g1, g2;
        const float* srow0 = src + (size_t)y * w;
        const float* srow1;
        float* drow = dst + (size_t)y * w * 5;

        for (x = 0; x < w; x++) {
            row[x * 3] = srow0[x] * g0;
            row[x * 3 + 1] = row[x * 3 + 2] = 0.f;
        }
        for (k = 1; k <= n; k++) {
            g0 = g[k]; g1 = xg[k]; g2 = xxg[k];
            srow0 = src + (size_t)(y - k > 0 ? y - k : 0) * w;
            srow1 = src + (size_t)(y + k < h - 1 ? y + k : h - 1) * w;
            for (x = 0; x < w; x++) {
                float p = srow0[x] + srow1[x];
                float t0 = row[x * 3] + g0 * p;
                float t1 = row[x * 3 + 1] + g1 * (srow1[x] - srow0[x]);
                float t2 = row[x * 3 + 2] + g2 * p;
                row[x * 3] = t0;
                row[x * 3 + 1] = t1;
                row[x * 3 + 2] = t2;
            }
        }
        for (x = 0; x < n * 3; x++) {
            row[-1 - x] = row[2 - x];
            row[w * 3 + x] = row[w * 3 + x - 3];
        }
        for (x = 0; x < w; x++) {
            g0 = g[0];
            double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0,
                   b4 = 0, b5 = row[x * 3 + 2] * g0, b6 = 0;
            for (k = 1; k <= n; k++) {
                double tg = row[(x + k) * 3] + row[(x - k) * 3];
                g0 = g[k];
                b1 += tg * g0;
                b4 += tg * xxg[k];
                b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[k];
                b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * g0;
                b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[k];
                b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * g0;
            }
            drow[x * 5 + 1] = (float)(b2 * ig11);
            drow[x * 5] = (float)(b3 * ig11);
            drow[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
            drow[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
            drow[x * 5 + 4] = (float)(b6 * ig55);
        }
    }
    free(kbuf);
    free(rowbuf);
    return ORC_OK;
}

/* A.1 step 3 */
static void update_matrices(const float* R0a, const float* R1, const float* flowa, float* Ma,
                            int h, int w, int y0, int y1)
{
    enum { BORDER = 5 };
    static const float border[BORDER] = { 0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f };
    const size_t step1 = (size_t)w * 5;
#pragma omp parallel for schedule(static) num_threads(g_row_threads)
    for (int y = y0; y < y1; y++) {
        const float* flow = flowa + (size_t)y * w * 2;
        const float* R0 = R0a + (size_t)y * w * 5;
        float* M = Ma + (size_t)y * w * 5;
        for (int x = 0; x < w; x++) {
            float dx = flow[x * 2], dy = flow[x * 2 + 1];
            float fx = x + dx, fy = y + dy;
            int x1 = cv_floor_f(fx), yy1 = cv_floor_f(fy);
            float r2, r3, r4, r5, r6;
            fx -= x1; fy -= yy1;
            if ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)yy1 < (unsigned)(h - 1)) {
                const float* ptr = R1 + (size_t)yy1 * step1 + (size_t)x1 * 5;
                float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy),
                      a10 = (1.f - fx) * fy, a11 = fx * fy;
                r2 = a00 * ptr[0] + a01 * ptr[5] + a10 * ptr[step1] + a11 * ptr[step1 + 5];
                r3 = a00 * ptr[1] + a01 * ptr[6] + a10 * ptr[step1 + 1] + a11 * ptr[step1 + 6];
                r4 = a00 * ptr[2] + a01 * ptr[7] + a10 * ptr[step1 + 2] + a11 * ptr[step1 + 7];
                r5 = a00 * ptr[3] + a01 * ptr[8] + a10 * ptr[step1 + 3] + a11 * ptr[step1 + 8];
                r6 = a00 * ptr[4] + a01 * ptr[9] + a10 * ptr[step1 + 4] + a11 * ptr[step1 + 9];
                r4 = (R0[x * 5 + 2] + r4) * 0.5f;
                r5 = (R0[x * 5 + 3] + r5) * 0.5f;
                r6 = (R0[x * 5 + 4] + r6) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = R0[x * 5 + 2];
                r5 = R0[x * 5 + 3];
                r6 = R0[x * 5 + 4] * 0.5f;
            }
            r2 = (R0[x * 5] - r2) * 0.5f;
            r3 = (R0[x * 5 + 1] - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            if ((unsigned)(x - BORDER) >= (unsigned)(w - BORDER * 2) ||
                (unsigned)(y - BORDER) >= (unsigned)(h - BORDER * 2)) {
                float scale = (x < BORDER ? border[x] : 1.f) *
                              (x >= w - BORDER ? border[w - x - 1] : 1.f) *
                              (y < BORDER ? border[y] : 1.f) *
                              (y >= h - BORDER ? border[h - y - 1] : 1.f);
                r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
            }
            M[x * 5] = r4 * r4 + r6 * r6;
            M[x * 5 + 1] = (r4 + r5) * r6;
            M[x * 5 + 2] = r5 * r5 + r6 * r6;
            M[x * 5 + 3] = r4 * r2 + r6 * r3;
            M[x * 5 + 4] = r6 * r2 + r5 * r3;
        }
    }
}

/* window kernel of A.1 step 4: k[0..m], sigma = 0.3 m, normalised in double */
void orc_farneback_window_kernel(int winsize, float* kernel /* m+1 */)
{
    int m = winsize / 2;
    double sigma = m * 0.3, s = 1;
    kernel[0] = (float)s;
    for (int i = 1; i <= m; i++) {
        float t = (float)exp(-i * i / (2 * sigma * sigma));
        kernel[i] = t;
        s += t * 2;
    }
    s = 1. / s;
    for (int i = 0; i <= m; i++) kernel[i] = (float)(kernel[i] * s);
}

/* A.1 step 4: separable window blur of M (replicate borders) + 2x2 solve in
 * double.  Two-phase form (blur all rows, then optionally rebuild M), which is
 * exactly equivalent to OpenCV's lagging row-stripe update. */
/* One line of the window blur: acc[x] = c[x]*k0; acc[x] = (a_i[x] + b_i[x])*k_i + acc[x] for i = 1..m, where a_i / b_i are
 * the lines i steps after / before the centre line (rows for the vertical pass, pixels of the padded vsum row for the
 * horizontal one).  The tap loop is the outer one so that the inner loop runs over contiguous memory; per element the
 * operations and their order are those of the per-pixel loop. */
ORC_CLONES static void blur_line(const float* const* after, const float* const* before, const float* centre,
                                 const float* kernel, int m, float* acc, int n, int fused)
{
    const float k0 = kernel[0];
    for (int x = 0; x < n; x++) acc[x] = centre[x] * k0;
    for (int i = 1; i <= m; i++) {
        const float* a = after[i];
        const float* b = before[i];
        const float k = kernel[i];
        if (fused) {
            for (int x = 0; x < n; x++) acc[x] = fmaf(a[x] + b[x], k, acc[x]);
        } else {
            for (int x = 0; x < n; x++) {
                float t = (a[x] + b[x]) * k;
                acc[x] = t + acc[x];
            }
        }
    }
}

static int update_flow_gaussian(const float* R0, const float* R1, float* flow, float* M,
                                int h, int w, int winsize, int update, int fused)
{
    const int m = winsize / 2;
    float* kernel = (float*)malloc(sizeof(float) * (m + 1));
    if (!kernel) return ORC_ENOMEM;
    orc_farneback_window_kernel(winsize, kernel);
    int rc_all = ORC_OK;
    /* rows are independent: M is only read here (it is rebuilt after the loop).  Inside orc_farneback_batch this
     * region is nested and therefore runs on the calling thread alone. */
#pragma omp parallel num_threads(g_row_threads)
    {
        float* vsumbuf = (float*)malloc(sizeof(float) * ((size_t)(w + m * 2 + 2) * 5));
        float* hsum = (float*)malloc(sizeof(float) * (size_t)w * 5);
        const float** srow = (const float**)malloc(sizeof(float*) * (m * 2 + 1) * 2);
        if (!vsumbuf || !hsum || !srow) {
#pragma omp critical
            rc_all = ORC_ENOMEM;
        } else {
            float* vsum = vsumbuf + (m + 1) * 5;
            const float** hrow = srow + (m * 2 + 1);
#pragma omp for schedule(static)
            for (int y = 0; y < h; y++) {
                for (int i = 0; i <= m; i++) {
                    srow[m - i] = M + (size_t)(y - i > 0 ? y - i : 0) * w * 5;
                    srow[m + i] = M + (size_t)(y + i < h - 1 ? y + i : h - 1) * w * 5;
                }
                /* s0 = srow[m][x]*k0; s0 = (srow[m+i][x] + srow[m-i][x])*k_i + s0 */
                for (int i = 0; i <= m; i++) hrow[i] = srow[m - i];
                blur_line(srow + m, hrow, srow[m], kernel, m, vsum, w * 5, fused);
                for (int x = 0; x < m * 5; x++) { /* replicate the first/last pixel */
                    vsum[-1 - x] = vsum[4 - x % 5];
                    vsum[w * 5 + x] = vsum[w * 5 - 5 + x % 5];
                }
                /* sum = vsum[x]*k0; sum = (vsum[x - 5i] + vsum[x + 5i])*k_i + sum */
                for (int i = 0; i <= m; i++) { srow[i] = vsum - i * 5; hrow[i] = vsum + i * 5; }
                blur_line(srow, hrow, vsum, kernel, m, hsum, w * 5, fused);
                float* f = flow + (size_t)y * w * 2;
                for (int x = 0; x < w; x++) {
                    double g11 = hsum[x * 5], g12 = hsum[x * 5 + 1], g22 = hsum[x * 5 + 2],
                           h1 = hsum[x * 5 + 3], h2 = hsum[x * 5 + 4];
                    double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
                    f[x * 2] = (float)((g11 * h2 - g12 * h1) * idet);
                    f[x * 2 + 1] = (float)((g22 * h1 - g12 * h2) * idet);
                }
            }
        }
        free(vsumbuf); free(hsum); free((void*)srow);
    }
    free(kernel);
    if (rc_all) return rc_all;
    if (update) update_matrices(R0, R1, flow, M, h, w, 0, h);
    return ORC_OK;
}

/* vsum border replicate: OpenCV writes vsum[-1-x] for x in [0, m*5): the k-th
 * padded pixel to the left repeats pixel 0 channel-wise; the expression above
 * (4 - x%5) maps x=0 -> ch4, x=1 -> ch3 ... i.e. vsum[-1]=ch4 of pixel 0. */

/* Full single-scale Farneback on one (h,w) plane pair.
 * prev/next: dtype u8/u16/f32, contiguous.  flow_out: h*w*2 float, (dx,dy).
 * Optional dumps (may be NULL): R0,R1 (h*w*5 interleaved), M0 (first M). */
int orc_farneback(const void* prev, const void* next, int dtype, int h, int w,
                  int winsize, int iters, int poly_n, double poly_sigma, int fused,
                  float* flow_out, float* R0_out, float* R1_out, float* M0_out)
{
    if (h <= 0 || w <= 0 || iters < 0 || poly_n < 1 || winsize < 1) return ORC_EINVAL;
    size_t npx = (size_t)h * w;
    float* fimg = (float*)malloc(sizeof(float) * npx);
    float* blur = (float*)malloc(sizeof(float) * npx);
    float* tmp = (float*)malloc(sizeof(float) * npx);
    float* R[2];
    R[0] = (float*)malloc(sizeof(float) * npx * 5);
    R[1] = (float*)malloc(sizeof(float) * npx * 5);
    float* M = (float*)malloc(sizeof(float) * npx * 5);
    int rc = ORC_OK;
    if (!fimg || !blur || !tmp || !R[0] || !R[1] || !M) { rc = ORC_ENOMEM; goto done; }
    const void* img[2] = { prev, next };
    for (int i = 0; i < 2; i++) {
        for (size_t p = 0; p < npx; p++) fimg[p] = load_as_f32(img[i], dtype, p);
        preblur3(fimg, blur, h, w, tmp);
        rc = poly_exp(blur, R[i], h, w, poly_n, poly_sigma);
        if (rc) goto done;
    }
    memset(flow_out, 0, sizeof(float) * npx * 2);
    update_matrices(R[0], R[1], flow_out, M, h, w, 0, h);
    if (R0_out) memcpy(R0_out, R[0], sizeof(float) * npx * 5);
    if (R1_out) memcpy(R1_out, R[1], sizeof(float) * npx * 5);
    if (M0_out) memcpy(M0_out, M, sizeof(float) * npx * 5);
    for (int i = 0; i < iters; i++) {
        rc = update_flow_gaussian(R[0], R[1], flow_out, M, h, w, winsize, i < iters - 1, fused);
        if (rc) goto done;
    }
done:
    free(fimg); free(blur); free(tmp); free(R[0]); free(R[1]); free(M);
    return rc;
}

/* Batch over equal-size tiles, OpenMP fan-out (the analogue of the reference's
 * dask fan-out, flow_calc.py:88-98).  prev/next: n * h*w contiguous planes. */
int orc_farneback_batch(const void* prev, const void* next, int dtype, int n, int h, int w,
                        int winsize, int iters, int poly_n, double poly_sigma, int fused,
                        float* flow_out, int nthreads)
{
    size_t esz = dtype == ORC_U8 ? 1 : (dtype == ORC_U16 ? 2 : 4);
    size_t plane = (size_t)h * w;
    int rc_all = ORC_OK;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int t = 0; t < n; t++) {
        int rc = orc_farneback((const char*)prev + plane * esz * t, (const char*)next + plane * esz * t,
                               dtype, h, w, winsize, iters, poly_n, poly_sigma, fused,
                               flow_out + plane * 2 * t, NULL, NULL, NULL);
        if (rc) {
#pragma omp critical
            rc_all = rc;
        }
    }
    return rc_all;
}

/* ------------------------------------------------------------------------- */
/* remap, INTER_LINEAR, BORDER_CONSTANT(0), map = interleaved (x,y) float     */
/* SURVEY.md Appendix A.2                                                     */
/* ------------------------------------------------------------------------- */
#define INTER_BITS 5
#define INTER_TAB_SIZE 32
#define INTER_REMAP_COEF_BITS 15
#define INTER_REMAP_COEF_SCALE (1 << INTER_REMAP_COEF_BITS)

static float g_tab_f[INTER_TAB_SIZE * INTER_TAB_SIZE][4];
static short g_tab_i[INTER_TAB_SIZE * INTER_TAB_SIZE + 2][4]; /* +2: the sum-fix probes past the entry */
static int g_tab_ready = 0;

static short sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

static void init_bilinear_tab(void)
{
    if (g_tab_ready) return;
    const float scale = 1.f / INTER_TAB_SIZE;
    float tab1[INTER_TAB_SIZE][2];
    for (int i = 0; i < INTER_TAB_SIZE; i++) {
        float x = i * scale;
        tab1[i][0] = 1.f - x;
        tab1[i][1] = x;
    }
    memset(g_tab_i, 0, sizeof(g_tab_i));
    for (int i = 0; i < INTER_TAB_SIZE; i++)
        for (int j = 0; j < INTER_TAB_SIZE; j++) {
            float* tab = g_tab_f[i * INTER_TAB_SIZE + j];
            short* itab = g_tab_i[i * INTER_TAB_SIZE + j];
            int isum = 0;
            const int ksize = 2;
            for (int k1 = 0; k1 < ksize; k1++) {
                float vy = tab1[i][k1];
                for (int k2 = 0; k2 < ksize; k2++) {
                    float v = vy * tab1[j][k2];
                    tab[k1 * ksize + k2] = v;
                    isum += itab[k1 * ksize + k2] = sat_short(cv_round_f(v * INTER_REMAP_COEF_SCALE));
                }
            }
            if (isum != INTER_REMAP_COEF_SCALE) {
                /* OpenCV's sum fix-up probes k1,k2 in [ksize/2, ksize/2+2), which for
                 * ksize=2 runs past this entry into the (still zero) next ones. */
                int diff = isum - INTER_REMAP_COEF_SCALE;
                int ksize2 = ksize / 2, Mk1 = ksize2, Mk2 = ksize2, mk1 = ksize2, mk2 = ksize2;
                for (int k1 = ksize2; k1 < ksize2 + 2; k1++)
                    for (int k2 = ksize2; k2 < ksize2 + 2; k2++) {
                        if (itab[k1 * ksize + k2] < itab[mk1 * ksize + mk2]) mk1 = k1, mk2 = k2;
                        else if (itab[k1 * ksize + k2] > itab[Mk1 * ksize + Mk2]) Mk1 = k1, Mk2 = k2;
                    }
                if (diff < 0) itab[Mk1 * ksize + Mk2] = (short)(itab[Mk1 * ksize + Mk2] - diff);
                else itab[mk1 * ksize + mk2] = (short)(itab[mk1 * ksize + mk2] - diff);
            }
        }
    g_tab_ready = 1;
}

/* expose the tables so tests can check them / the product can be compared */
void orc_remap_tables(float* tab_f /*1024*4*/, short* tab_i /*1024*4*/)
{
    init_bilinear_tab();
    memcpy(tab_f, g_tab_f, sizeof(g_tab_f));
    memcpy(tab_i, g_tab_i, sizeof(short) * INTER_TAB_SIZE * INTER_TAB_SIZE * 4);
}

/* fixed-point source coordinates (5 fractional bits) of destination pixel (x, y): from a float map ... */
#define COORDS_FROM_MAP                                                                         \
    const float* mrow = map + (size_t)y * dw * 2;                                               \
    int sxq = cv_round_f(mrow[x * 2] * INTER_TAB_SIZE);                                         \
    int syq = cv_round_f(mrow[x * 2 + 1] * INTER_TAB_SIZE);
/* ... or from the inverted affine matrix as WarpAffineInvoker computes them (imgwarp.cpp, AB_BITS = 10) */
#define COORDS_FROM_AFFINE                                                                      \
    int sxq = (X0 + adelta[x]) >> (AB_BITS - INTER_BITS);                                       \
    int syq = (Y0 + bdelta[x]) >> (AB_BITS - INTER_BITS);

#define REMAP_BODY(T, WT, KT, LOADW, CASTEXPR) REMAP_BODY_X(T, WT, KT, LOADW, CASTEXPR, , COORDS_FROM_MAP)
#define REMAP_BODY_X(T, WT, KT, LOADW, CASTEXPR, ROWSETUP, COORDS)                                  \
    _Pragma("omp parallel for schedule(static) num_threads(g_row_threads)")                         \
    for (int y = 0; y < dh; y++) {                                                              \
        T* drow = (T*)dst + (size_t)y * dw * cn;                                                \
        ROWSETUP                                                                                \
        for (int x = 0; x < dw; x++) {                                                          \
            COORDS                                                                              \
            int a = (syq & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (sxq & (INTER_TAB_SIZE - 1)); \
            int sx = sat_short(sxq >> INTER_BITS), sy = sat_short(syq >> INTER_BITS);           \
            const KT* wgt = LOADW[a];                                                           \
            for (int k = 0; k < cn; k++) {                                                      \
                T out;                                                                          \
                if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {                         \
                    out = 0;                                                                    \
                } else {                                                                        \
                    const T* S = (const T*)src;                                                 \
                    int x0ok = sx >= 0, x1ok = sx + 1 < sw, y0ok = sy >= 0, y1ok = sy + 1 < sh; \
                    T v0 = (x0ok && y0ok) ? S[((size_t)sy * sw + sx) * cn + k] : 0;             \
                    T v1 = (x1ok && y0ok) ? S[((size_t)sy * sw + sx + 1) * cn + k] : 0;         \
                    T v2 = (x0ok && y1ok) ? S[((size_t)(sy + 1) * sw + sx) * cn + k] : 0;       \
                    T v3 = (x1ok && y1ok) ? S[((size_t)(sy + 1) * sw + sx + 1) * cn + k] : 0;   \
                    WT acc = v0 * wgt[0] + v1 * wgt[1] + v2 * wgt[2] + v3 * wgt[3];             \
                    out = CASTEXPR;                                                             \
                }                                                                               \
                drow[x * cn + k] = out;                                                         \
            }                                                                                   \
        }                                                                                       \
    }

int orc_remap_bilinear(const void* src, int dtype, int cn, int sh, int sw,
                       const float* map, int dh, int dw, void* dst)
{
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || cn < 1) return ORC_EINVAL;
    if (sh >= 32767 || sw >= 32767 || dh >= 32767 || dw >= 32767) return ORC_EINVAL; /* SHRT_MAX assert */
    init_bilinear_tab();
    if (dtype == ORC_U8) {
        REMAP_BODY(uint8_t, int, short, g_tab_i,
                   (uint8_t)clampi((acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS, 0, 255))
    } else if (dtype == ORC_U16) {
        REMAP_BODY(uint16_t, float, float, g_tab_f, (uint16_t)clampi(cv_round_f(acc), 0, 65535))
    } else if (dtype == ORC_F32) {
        REMAP_BODY(float, float, float, g_tab_f, acc)
    } else
        return ORC_EINVAL;
    return ORC_OK;
}

/* cv2.warpAffine(src, M, dsize) with the default flags (INTER_LINEAR, BORDER_CONSTANT 0), as
 * feature_registrator.py:130 calls it [OCV-mem, imgwarp.cpp cv::warpAffine + WarpAffineInvoker]:
 * M (forward, 2x3 double) is inverted in double; per destination column adelta/bdelta =
 * saturate_cast<int>(M0*x*1024), (M3*x*1024); per row X0 = saturate_cast<int>((M1*y + M2)*1024) + 16 (half of
 * 1/32 px in 10-bit fixed point); X = (X0 + adelta[x]) >> 5 carries 5 fractional bits and feeds the same
 * bilinear tables as remap. */
#define AB_BITS 10
#define AB_SCALE (1 << AB_BITS)
static int sat_int_d(double v)
{
    if (!(v > -2147483648.0)) return INT32_MIN;
    if (!(v < 2147483647.0)) return INT32_MAX;
    return (int)lrint(v);
}

int orc_warp_affine_cv(const void* src, int dtype, int sh, int sw, const double* M_fwd, int dh, int dw, void* dst)
{
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || !M_fwd) return ORC_EINVAL;
    const int cn = 1;
    init_bilinear_tab();
    double M[6];
    memcpy(M, M_fwd, sizeof(M));
    {
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0 ? 1. / D : 0;
        double A11 = M[4] * D, A22 = M[0] * D;
        M[0] = A11; M[1] *= -D;
        M[3] *= -D; M[4] = A22;
        double b1 = -M[0] * M[2] - M[1] * M[5];
        double b2 = -M[3] * M[2] - M[4] * M[5];
        M[2] = b1; M[5] = b2;
    }
    int* adelta = (int*)malloc(sizeof(int) * 2 * (size_t)dw);
    if (!adelta) return ORC_EINVAL;
    int* bdelta = adelta + dw;
    for (int x = 0; x < dw; x++) {
        adelta[x] = sat_int_d(M[0] * x * AB_SCALE);
        bdelta[x] = sat_int_d(M[3] * x * AB_SCALE);
    }
    const int round_delta = AB_SCALE / INTER_TAB_SIZE / 2;
#define AFFINE_ROW                                                                              \
    const int X0 = sat_int_d((M[1] * y + M[2]) * AB_SCALE) + round_delta;                       \
    const int Y0 = sat_int_d((M[4] * y + M[5]) * AB_SCALE) + round_delta;
    int rc = ORC_OK;
    if (dtype == ORC_U8) {
        REMAP_BODY_X(uint8_t, int, short, g_tab_i,
                     (uint8_t)clampi((acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS, 0, 255),
                     AFFINE_ROW, COORDS_FROM_AFFINE)
    } else if (dtype == ORC_U16) {
        REMAP_BODY_X(uint16_t, float, float, g_tab_f, (uint16_t)clampi(cv_round_f(acc), 0, 65535), AFFINE_ROW,
                     COORDS_FROM_AFFINE)
    } else if (dtype == ORC_F32) {
        REMAP_BODY_X(float, float, float, g_tab_f, acc, AFFINE_ROW, COORDS_FROM_AFFINE)
    } else
        rc = ORC_EINVAL;
    free(adelta);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* pyrDown (A.3) and pyrUp (A.4)                                              */
/* ------------------------------------------------------------------------- */
int orc_pyr_down(const void* src, int dtype, int h, int w, void* dst)
{
    if (h <= 0 || w <= 0) return ORC_EINVAL;
    int dh = (h + 1) / 2, dw = (w + 1) / 2;
    if (dtype == ORC_F32) {
        const float* s = (const float*)src;
        float* rows_all = (float*)malloc(sizeof(float) * (size_t)dw * 5 * g_row_threads);
        if (!rows_all) return ORC_ENOMEM;
#pragma omp parallel for schedule(static) num_threads(g_row_threads)
        for (int y = 0; y < dh; y++) {
#ifdef _OPENMP
            float* rows = rows_all + (size_t)omp_get_thread_num() * dw * 5;
#else
            float* rows = rows_all;
#endif
            for (int k = 0; k < 5; k++) {
                const float* sr = s + (size_t)reflect101(2 * y + k - 2, h) * w;
                float* r = rows + (size_t)k * dw;
                for (int x = 0; x < dw; x++) {
                    float c = sr[reflect101(2 * x, w)];
                    float l1 = sr[reflect101(2 * x - 1, w)], r1 = sr[reflect101(2 * x + 1, w)];
                    float l2 = sr[reflect101(2 * x - 2, w)], r2 = sr[reflect101(2 * x + 2, w)];
                    r[x] = c * 6 + (l1 + r1) * 4 + l2 + r2;
                }
            }
            float* d = (float*)dst + (size_t)y * dw;
            const float *r0 = rows, *r1 = rows + dw, *r2 = rows + 2 * (size_t)dw, *r3 = rows + 3 * (size_t)dw,
                        *r4 = rows + 4 * (size_t)dw;
            for (int x = 0; x < dw; x++)
                d[x] = (r2[x] * 6 + (r1[x] + r3[x]) * 4 + r0[x] + r4[x]) * (1.f / 256);
        }
        free(rows_all);
        return ORC_OK;
    }
    if (dtype != ORC_U8 && dtype != ORC_U16) return ORC_EINVAL;
    int* rows_all = (int*)malloc(sizeof(int) * (size_t)dw * 5 * g_row_threads);
    if (!rows_all) return ORC_ENOMEM;
#pragma omp parallel for schedule(static) num_threads(g_row_threads)
    for (int y = 0; y < dh; y++) {
#ifdef _OPENMP
        int* rows = rows_all + (size_t)omp_get_thread_num() * dw * 5;
#else
        int* rows = rows_all;
#endif
        for (int k = 0; k < 5; k++) {
            size_t ro = (size_t)reflect101(2 * y + k - 2, h) * w;
            int* r = rows + (size_t)k * dw;
            for (int x = 0; x < dw; x++) {
                int idx[5];
                for (int j = 0; j < 5; j++) idx[j] = reflect101(2 * x + j - 2, w);
                int v[5];
                for (int j = 0; j < 5; j++)
                    v[j] = dtype == ORC_U8 ? ((const uint8_t*)src)[ro + idx[j]] : ((const uint16_t*)src)[ro + idx[j]];
                r[x] = v[2] * 6 + (v[1] + v[3]) * 4 + v[0] + v[4];
            }
        }
        for (int x = 0; x < dw; x++) {
            int sum = rows[2 * (size_t)dw + x] * 6 + (rows[(size_t)dw + x] + rows[3 * (size_t)dw + x]) * 4 +
                      rows[x] + rows[4 * (size_t)dw + x];
            int v = (sum + 128) >> 8;
            if (dtype == ORC_U8) ((uint8_t*)dst)[(size_t)y * dw + x] = (uint8_t)clampi(v, 0, 255);
            else ((uint16_t*)dst)[(size_t)y * dw + x] = (uint16_t)clampi(v, 0, 65535);
        }
    }
    free(rows_all);
    return ORC_OK;
}

/* pyrUp of a cn-channel float image to (dh,dw); requires |dw-2w| == dw%2 etc. */
int orc_pyr_up_f32(const float* src, int cn, int h, int w, float* dst, int dh, int dw)
{
    if (h <= 0 || w <= 0 || cn < 1) return ORC_EINVAL;
    if (abs(dw - w * 2) != dw % 2 || abs(dh - h * 2) != dh % 2) return ORC_EINVAL;
    int bufw = (dw + 1 > 2 * w ? dw + 1 : 2 * w) * cn;
    float* buf_all = (float*)malloc(sizeof(float) * (size_t)bufw * 3 * g_row_threads);
    if (!buf_all) return ORC_ENOMEM;
    /* horizontal pass of source row sy into `row` (2w columns, +1 if dw > 2w); source rows are independent
     * (row y writes destination rows 2y and min(2y+1, dh-1); for the last y both may coincide, same thread) */
#pragma omp parallel for schedule(static) num_threads(g_row_threads)
    for (int y = 0; y < h; y++) {
#ifdef _OPENMP
        float* buf = buf_all + (size_t)omp_get_thread_num() * bufw * 3;
#else
        float* buf = buf_all;
#endif
        float* rws[3];
        for (int k = 0; k < 3; k++) {
            int sy = y - 1 + k;
            int _sy = reflect101(sy * 2, h * 2) / 2;
            const float* s = src + (size_t)_sy * w * cn;
            float* row = buf + (size_t)k * bufw;
            rws[k] = row;
            if (w == 1) {
                for (int c = 0; c < cn; c++) row[c] = row[c + cn] = s[c] * 8;
                continue;
            }
            for (int c = 0; c < cn; c++) {
                float t0 = s[c] * 6 + s[c + cn] * 2;
                float t1 = (s[c] + s[c + cn]) * 4;
                row[c] = t0; row[c + cn] = t1;
                int sx = (w - 1) * cn + c;
                int dx = (w - 1) * 2 * cn + c;
                t0 = s[sx - cn] + s[sx] * 7;
                t1 = s[sx] * 8;
                row[dx] = t0; row[dx + cn] = t1;
                if (dw > w * 2) row[(dw - 1) * cn + c] = row[dx + cn];
            }
            for (int x = 1; x < w - 1; x++)
                for (int c = 0; c < cn; c++) {
                    int sx = x * cn + c, dx = x * 2 * cn + c;
                    float t0 = s[sx - cn] + s[sx] * 6 + s[sx + cn];
                    float t1 = (s[sx] + s[sx + cn]) * 4;
                    row[dx] = t0; row[dx + cn] = t1;
                }
        }
        float* dst0 = dst + (size_t)(y * 2) * dw * cn;
        int y1 = y * 2 + 1 < dh - 1 ? y * 2 + 1 : dh - 1;
        float* dst1 = dst + (size_t)y1 * dw * cn;
        for (int x = 0; x < dw * cn; x++) {
            float t1 = ((rws[1][x] + rws[2][x]) * 4) * (1.f / 64);
            float t0 = (rws[0][x] + rws[1][x] * 6 + rws[2][x]) * (1.f / 64);
            dst1[x] = t1;
            dst0[x] = t0;
        }
    }
    if (dh > h * 2) {
        const float* d0 = dst + (size_t)(h * 2 - 2) * dw * cn;
        float* d2 = dst + (size_t)(h * 2) * dw * cn;
        for (int x = 0; x < dw * cn; x++) d2[x] = d0[x];
    }
    free(buf_all);
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* normalize(NORM_MINMAX) + GaussianBlur: the dog() chain (A.5)               */
/* ------------------------------------------------------------------------- */
int orc_minmax(const void* src, int dtype, size_t n, double* mn, double* mx)
{
    if (n == 0) return ORC_EINVAL;
    double lo = load_as_f32(src, dtype, 0), hi = lo;
#pragma omp parallel for schedule(static) num_threads(g_row_threads) reduction(min : lo) reduction(max : hi)
    for (size_t i = 1; i < n; i++) {
        double v = load_as_f32(src, dtype, i);
        if (v < lo) lo = v;
        if (v > hi) hi = v;
    }
    *mn = lo; *mx = hi;
    return ORC_OK;
}

/* Rounding models of the dog() chain (ORC_DOG_* flags).  OpenCV keeps GaussianBlur's separable filters
 * (filter.simd.hpp: RowVec_32f, SymmColumnVec_32f) and convertTo's scale (convert_scale.simd.hpp: cvt_32f) in
 * CPU-dispatched translation units.  The SSE2 baseline objects multiply then add; the AVX2 objects, which every
 * x86-64 host with AVX2 + FMA3 selects at run time, are compiled with FMA3 and use fused multiply-adds
 * (_mm256_fmadd_ps / v_fma, and the scalar tails contract the same way):
 *   ORC_DOG_FUSED_BLUR   row filter  acc = fma(x_j, k_j, acc)  (left to right, first term k_0*x_0),
 *                        column filter  acc = fma(a + b, k_j, acc)  (first term k_r*c)
 *   ORC_DOG_FUSED_SCALE  both normalize() steps  dst = fma(src, a, b)
 * flags == 0 is the SSE2 model, ORC_DOG_FUSED_BLUR | ORC_DOG_FUSED_SCALE the AVX2 one. */
enum { ORC_DOG_FUSED_BLUR = 1, ORC_DOG_FUSED_SCALE = 2 };

/* normalize(src, alpha=0, beta=1, NORM_MINMAX, CV_32F) */
int orc_normalize_minmax_to_f32_ex(const void* src, int dtype, size_t n, double alpha, double beta, int fused, float* dst)
{
    double smin, smax;
    int rc = orc_minmax(src, dtype, n, &smin, &smax);
    if (rc) return rc;
    double dmin = alpha < beta ? alpha : beta, dmax = alpha < beta ? beta : alpha;
    double scale = (dmax - dmin) * (smax - smin > DBL_EPSILON ? 1. / (smax - smin) : 0);
    scale = (float)scale;
    double shift = (float)dmin - (float)(smin * scale);
    float a = (float)scale, b = (float)shift;
#pragma omp parallel for schedule(static) num_threads(g_row_threads)
    for (size_t i = 0; i < n; i++) {
        float v = load_as_f32(src, dtype, i);
        if (fused) {
            dst[i] = fmaf(v, a, b);
        } else {
            float p = v * a;
            dst[i] = p + b;
        }
    }
    return ORC_OK;
}
int orc_normalize_minmax_to_f32(const void* src, int dtype, size_t n, double alpha, double beta, float* dst)
{
    return orc_normalize_minmax_to_f32_ex(src, dtype, n, alpha, beta, 0, dst);
}

/* normalize(src f32, 0, 255, NORM_MINMAX, CV_8U) */
int orc_normalize_minmax_f32_to_u8_ex(const float* src, size_t n, int fused, uint8_t* dst)
{
    double smin, smax;
    int rc = orc_minmax(src, ORC_F32, n, &smin, &smax);
    if (rc) return rc;
    double scale = 255. * (smax - smin > DBL_EPSILON ? 1. / (smax - smin) : 0);
    double shift = 0. - smin * scale;
    float a = (float)scale, b = (float)shift;
#pragma omp parallel for schedule(static) num_threads(g_row_threads)
    for (size_t i = 0; i < n; i++) {
        float v;
        if (fused) {
            v = fmaf(src[i], a, b);
        } else {
            float p = src[i] * a;
            v = p + b;
        }
        dst[i] = (uint8_t)clampi(cv_round_f(v), 0, 255);
    }
    return ORC_OK;
}
int orc_normalize_minmax_f32_to_u8(const float* src, size_t n, uint8_t* dst)
{
    return orc_normalize_minmax_f32_to_u8_ex(src, n, 0, dst);
}

/* getGaussianKernel(ksize, sigma, CV_32F), odd ksize, sigma > 0.  OpenCV 4.x
 * computes the taps in (soft)double -- t = exp(-0.125 (2i-(n-1))^2 / sigma^2),
 * sum = 2*sum(t) + 1, k = t * (1/sum) -- and rounds to float once at the end
 * (getGaussianKernelBitExact in smooth.dispatch.cpp). */
void orc_gaussian_kernel(int ksize, double sigma, float* k)
{
    double sigmaX = sigma > 0 ? sigma : ksize * 0.15 + 0.35;
    double scale2X = -0.125 / (sigmaX * sigmaX);
    int n2 = (ksize - 1) / 2;
    double sum = 0;
    double* v = (double*)malloc(sizeof(double) * (n2 + 1));
    for (int i = 0, x = 1 - ksize; i < n2; i++, x += 2) {
        v[i] = exp((double)(x * x) * scale2X);
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
    free(v);
}

/* t[x] = t[x] + k*p[x]   and   d[x] = d[x] + k*(a[x] + b[x]): the line updates of the two GaussianBlur passes
 * (fused: one rounding per update) */
ORC_CLONES static void gb_axpy(float* t, const float* p, float k, int n, int fused)
{
    if (fused) {
        for (int x = 0; x < n; x++) t[x] = fmaf(k, p[x], t[x]);
    } else {
        for (int x = 0; x < n; x++) {
            float v = k * p[x];
            t[x] = t[x] + v;
        }
    }
}
ORC_CLONES static void gb_axpy2(float* d, const float* a, const float* b, float k, int n, int fused)
{
    if (fused) {
        for (int x = 0; x < n; x++) d[x] = fmaf(k, a[x] + b[x], d[x]);
    } else {
        for (int x = 0; x < n; x++) {
            float v = k * (a[x] + b[x]);
            d[x] = d[x] + v;
        }
    }
}

/* GaussianBlur(f32, (ksize,ksize), sigma), BORDER_REFLECT_101.  Row filter
 * (plain left-to-right accumulation) then symmetric column filter. */
int orc_gaussian_blur_f32_ex(const float* src, int h, int w, int ksize, double sigma, int fused, float* dst)
{
    if (h <= 0 || w <= 0 || ksize < 1 || !(ksize & 1)) return ORC_EINVAL;
    float* k = (float*)malloc(sizeof(float) * ksize);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)h * w);
    if (!k || !tmp) { free(k); free(tmp); return ORC_ENOMEM; }
    orc_gaussian_kernel(ksize, sigma, k);
    const int r = ksize / 2;
    /* row filter: acc = k[0]*s[x-r]; acc += k[j]*s[x-r+j] for j = 1..ksize-1, left to right.  The row is copied
     * once into a reflect-101 padded buffer and the tap loop is the outer one, so that the inner loop runs over x
     * on contiguous memory: per output pixel the operations and their order are unchanged. */
    const int pw = w + 2 * r;
    float* pad_all = (float*)malloc(sizeof(float) * (size_t)pw * g_row_threads);
    if (!pad_all) { free(k); free(tmp); return ORC_ENOMEM; }
#pragma omp parallel for schedule(static) num_threads(g_row_threads)
    for (int y = 0; y < h; y++) {
#ifdef _OPENMP
        float* pad = pad_all + (size_t)omp_get_thread_num() * pw;
#else
        float* pad = pad_all;
#endif
        const float* s = src + (size_t)y * w;
        float* t = tmp + (size_t)y * w;
        for (int i = 0; i < pw; i++) pad[i] = s[reflect101(i - r, w)];
        const float k0 = k[0];
        for (int x = 0; x < w; x++) t[x] = k0 * pad[x];
        for (int j = 1; j < ksize; j++) gb_axpy(t, pad + j, k[j], w, fused);
    }
    free(pad_all);
#pragma omp parallel for schedule(static) num_threads(g_row_threads)
    for (int y = 0; y < h; y++) {
        float* d = dst + (size_t)y * w;
        const float* c = tmp + (size_t)y * w;
        for (int x = 0; x < w; x++) d[x] = k[r] * c[x];
        for (int j = 1; j <= r; j++) {
            const float* a = tmp + (size_t)reflect101(y + j, h) * w;
            const float* b = tmp + (size_t)reflect101(y - j, h) * w;
            gb_axpy2(d, a, b, k[r + j], w, fused);
        }
    }
    free(k); free(tmp);
    return ORC_OK;
}

int orc_gaussian_blur_f32(const float* src, int h, int w, int ksize, double sigma, float* dst)
{
    return orc_gaussian_blur_f32_ex(src, h, w, ksize, sigma, 0, dst);
}

/* dog(img, True, low_sigma, high_sigma) of optflow_registrator.py:249-274 for a
 * non-all-zero input (the max()==0 shortcut is the caller's).  Output u8.  flags: ORC_DOG_*. */
int orc_dog_u8_ex(const void* src, int dtype, int h, int w, int low_sigma, int high_sigma, int flags, uint8_t* dst)
{
    size_t n = (size_t)h * w;
    float* fimg = (float*)malloc(sizeof(float) * n);
    float* ls = (float*)malloc(sizeof(float) * n);
    float* hs = (float*)malloc(sizeof(float) * n);
    int rc = ORC_OK;
    const int fb = (flags & ORC_DOG_FUSED_BLUR) != 0, fs = (flags & ORC_DOG_FUSED_SCALE) != 0;
    if (!fimg || !ls || !hs) { rc = ORC_ENOMEM; goto done; }
    rc = orc_normalize_minmax_to_f32_ex(src, dtype, n, 0, 1, fs, fimg);
    if (rc) goto done;
    int ksize = low_sigma * 4 * 2 + 1;
    rc = orc_gaussian_blur_f32_ex(fimg, h, w, ksize, low_sigma, fb, ls);
    if (rc) goto done;
    rc = orc_gaussian_blur_f32_ex(fimg, h, w, ksize, high_sigma, fb, hs);
    if (rc) goto done;
#pragma omp parallel for schedule(static) num_threads(g_row_threads)
    for (size_t i = 0; i < n; i++) hs[i] = hs[i] - ls[i];
    rc = orc_normalize_minmax_f32_to_u8_ex(hs, n, fs, dst);
done:
    free(fimg); free(ls); free(hs);
    return rc;
}
int orc_dog_u8(const void* src, int dtype, int h, int w, int low_sigma, int high_sigma, uint8_t* dst)
{
    return orc_dog_u8_ex(src, dtype, h, w, low_sigma, high_sigma, 0, dst);
}

/* ------------------------------------------------------------------------- */
/* normalized_mutual_info_score for u8 labels (A.6; sklearn                   */
/* metrics/cluster/_supervised.py, arithmetic-mean normaliser, natural log)   */
/* ------------------------------------------------------------------------- */
static double entropy_counts(const int64_t* cnt, int nb, int64_t total)
{
    int nz = 0;
    for (int i = 0; i < nb; i++) nz += cnt[i] > 0;
    if (nz == 1) return 0.0;
    double lt = log((double)total), s = 0;
    for (int i = 0; i < nb; i++)
        if (cnt[i] > 0) s += ((double)cnt[i] / (double)total) * (log((double)cnt[i]) - lt);
    return -s;
}

int orc_nmi_u8(const uint8_t* a, const uint8_t* b, size_t n, double* score)
{
    if (n == 0) return ORC_EINVAL;
    int64_t* joint = (int64_t*)calloc(65536, sizeof(int64_t));
    if (!joint) return ORC_ENOMEM;
    int64_t pa[256] = { 0 }, pb[256] = { 0 };
    for (size_t i = 0; i < n; i++) {
        joint[(size_t)a[i] * 256 + b[i]]++;
        pa[a[i]]++;
        pb[b[i]]++;
    }
    int ca = 0, cb = 0;
    for (int i = 0; i < 256; i++) { ca += pa[i] > 0; cb += pb[i] > 0; }
    if (ca == 1 && cb == 1) { *score = 1.0; free(joint); return ORC_OK; }
    double N = (double)n, logN = log(N);
    /* pi.sum() == pj.sum() == N */
    double mi = 0;
    for (int i = 0; i < 256; i++) {
        if (!pa[i]) continue;
        for (int j = 0; j < 256; j++) {
            int64_t nij = joint[(size_t)i * 256 + j];
            if (!nij) continue;
            double log_nm = log((double)nij);
            double nm = (double)nij / N;
            double outer = (double)(pa[i] * pb[j]);
            double log_outer = -log(outer) + logN + logN;
            double term = nm * (log_nm - logN) + nm * log_outer;
            if (fabs(term) < DBL_EPSILON) term = 0.0;
            mi += term;
        }
    }
    free(joint);
    if (mi < 0) mi = 0;
    /* sklearn >= 1.3 returns 0.0 early when |mi| < eps; 1.0.2 reaches 0/normaliser = 0 */
    if (fabs(mi) < DBL_EPSILON) { *score = 0.0; return ORC_OK; }
    double ha = entropy_counts(pa, 256, (int64_t)n), hb = entropy_counts(pb, 256, (int64_t)n);
    double norm = 0.5 * (ha + hb);
    if (norm < DBL_EPSILON) norm = DBL_EPSILON;
    *score = mi / norm;
    return ORC_OK;
}

/* mi_tiled's loop (similarity_scoring.py:38-48): one score per contiguous chunk of `chunk` elements of the flattened
 * arrays (the last chunk may be shorter), chunks fanned out over threads like the reference's dask tasks. */
int orc_nmi_u8_chunks(const uint8_t* a, const uint8_t* b, size_t n, size_t chunk, double* scores, int nscores)
{
    if (n == 0 || chunk == 0) return ORC_EINVAL;
    const size_t nch = (n + chunk - 1) / chunk;
    if ((size_t)nscores < nch) return ORC_EINVAL;
    int rc_all = ORC_OK;
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (long long c = 0; c < (long long)nch; c++) {
        const size_t o = (size_t)c * chunk, len = o + chunk <= n ? chunk : n - o;
        int rc = orc_nmi_u8(a + o, b + o, len, scores + c);
        if (rc) {
#pragma omp critical
            rc_all = rc;
        }
    }
    return rc_all;
}
