/* A plain C host of one matching round of the feature stage: nothing but include/microaligner_hip.h and libmicroaligner_hip.so.
 *
 *   feature_host H W tile ref_u8.bin mov_u8.bin tables.bin
 *
 * reads two raw uint8 images (the DOG outputs FeatureRegistrator extracts features from) and the DAISY tables -- three
 * int32 radii r0 r1 r2, then the centre-first half kernels (r0 + 1, r1 + 1, r2 + 1 doubles), 16 doubles (cos, sin) and 50
 * doubles (dy, dx) offsets, as feature_reg/feature_detection.py:_daisy_tables makes them -- and runs
 *   ma_feature_extract (both images) -> ma_knn2_l2 -> ma_match_similarity (rng_state NULL: numpy's seed 0)
 * entirely on the device.  Prints: n_ref n_mov n_good status and the six matrix entries (%.17g).
 * tests/test_feature_reg.py builds and runs it and compares the line with the Python path. */
#include <stdio.h>
#include <stdlib.h>

#include "microaligner_hip.h"

#define CHECK(call)                                                          \
    do {                                                                     \
        int rc_ = (call);                                                    \
        if (rc_ != MA_OK) {                                                  \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ma_last_error()); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

static void* slurp(const char* path, size_t bytes)
{
    FILE* f = fopen(path, "rb");
    void* p = malloc(bytes);
    if (!f || !p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(f);
    return p;
}

int main(int argc, char** argv)
{
    if (argc != 7) { fprintf(stderr, "usage: see the header comment\n"); return 2; }
    const int H = atoi(argv[1]), W = atoi(argv[2]), tile = atoi(argv[3]), overlap = 51;
    const size_t n = (size_t)H * W;
    unsigned char *ref_h = slurp(argv[4], n), *mov_h = slurp(argv[5], n);
    FILE* f = fopen(argv[6], "rb");
    int radii[3];
    if (!f || fread(radii, sizeof(int), 3, f) != 3) { fprintf(stderr, "cannot read the tables\n"); return 2; }
    double* w[3];
    for (int c = 0; c < 3; c++) {
        w[c] = malloc(sizeof(double) * (size_t)(radii[c] + 1));
        if (fread(w[c], sizeof(double), (size_t)radii[c] + 1, f) != (size_t)radii[c] + 1) return 2;
    }
    double cos_sin[16], offs[50];
    if (fread(cos_sin, sizeof(double), 16, f) != 16 || fread(offs, sizeof(double), 50, f) != 50) return 2;
    fclose(f);
    const double* wp[3] = {w[0], w[1], w[2]};

    const int ntiles = ((H + tile - 1) / tile) * ((W + tile - 1) / tile);
    int limit = 1000000 / ntiles;              /* feature_detection.py:161-168 */
    if (limit > 5000) limit = 5000;
    const int cap = ntiles * limit;

    ma_ctx* ctx = NULL;
    CHECK(ma_ctx_create(0, &ctx));
    void *img[2], *desc[2], *pts[2], *resp[2];
    int cnt[2];
    const unsigned char* host[2] = {ref_h, mov_h};
    for (int k = 0; k < 2; k++) {
        CHECK(ma_malloc(ctx, n, &img[k]));
        CHECK(ma_malloc(ctx, (size_t)cap * 200 * sizeof(float), &desc[k]));
        CHECK(ma_malloc(ctx, (size_t)cap * 2 * sizeof(double), &pts[k]));
        CHECK(ma_malloc(ctx, (size_t)cap * sizeof(int), &resp[k]));
        CHECK(ma_memcpy_h2d(ctx, img[k], host[k], n));
        CHECK(ma_feature_extract(ctx, img[k], H, W, tile, overlap, 1, limit, wp, radii, cos_sin, offs, 0, cap, desc[k], pts[k],
                                 resp[k], &cnt[k]));
    }
    if (cnt[0] < 2 || cnt[1] < 1) { printf("%d %d 0 1\n", cnt[0], cnt[1]); return 0; }
    void *idx = NULL, *dist = NULL;
    CHECK(ma_malloc(ctx, (size_t)cnt[1] * 2 * sizeof(int), &idx));
    CHECK(ma_malloc(ctx, (size_t)cnt[1] * 2 * sizeof(float), &dist));
    /* knnMatch(des_mov, des_ref, k=2): queries = the moving image's descriptors */
    CHECK(ma_knn2_l2(ctx, desc[1], cnt[1], desc[0], cnt[0], 200, idx, dist));
    double m[6] = {1, 0, 0, 0, 1, 0};
    int n_good = 0, status = 0;
    CHECK(ma_match_similarity(ctx, idx, dist, cnt[1], pts[1], pts[0], cnt[0], 0.5f, 0.99, 3.0, 2000, NULL, m, &n_good, &status));
    printf("%d %d %d %d %.17g %.17g %.17g %.17g %.17g %.17g\n", cnt[0], cnt[1], n_good, status, m[0], m[1], m[2], m[3], m[4], m[5]);
    ma_ctx_destroy(ctx);
    return 0;
}
