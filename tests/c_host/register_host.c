/* A plain C host of the hot path: nothing but include/microaligner_hip.h and libmicroaligner_hip.so.
 *
 *   register_host H W dtype ref.bin mov.bin flow.bin warped.bin num_pyr_lvl use_full_res use_dog tile overlap
 *
 * reads two raw row-major images, runs ma_optflow_register + ma_warp_tiled and writes the flow (H*W*2 float32) and the
 * warped moving image; prints one line per pyramid level: factor h w mi_after mi_before accepted.
 * tests/test_gpu_register.py builds and runs it and compares the files with the oracle. */
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

static void spit(const char* path, const void* p, size_t bytes)
{
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    fclose(f);
}

int main(int argc, char** argv)
{
    if (argc != 13) { fprintf(stderr, "usage: see the header comment\n"); return 2; }
    const int H = atoi(argv[1]), W = atoi(argv[2]), dtype = atoi(argv[3]);
    const size_t esz = dtype == MA_U8 ? 1 : dtype == MA_U16 ? 2 : 4, n = (size_t)H * W;
    void *ref_h = slurp(argv[4], n * esz), *mov_h = slurp(argv[5], n * esz);
    ma_params p;
    ma_params_default(&p);
    p.num_pyr_lvl = atoi(argv[8]);
    p.use_full_res_img = atoi(argv[9]);
    p.use_dog = atoi(argv[10]);
    p.tile_size = atoi(argv[11]);
    p.overlap = atoi(argv[12]);

    ma_ctx* ctx = NULL;
    CHECK(ma_ctx_create(0, &ctx));
    void *ref = NULL, *mov = NULL, *flow = NULL, *warped = NULL;
    CHECK(ma_malloc(ctx, n * esz, &ref));
    CHECK(ma_malloc(ctx, n * esz, &mov));
    CHECK(ma_malloc(ctx, n * 2 * sizeof(float), &flow));
    CHECK(ma_malloc(ctx, n * esz, &warped));
    CHECK(ma_memcpy_h2d(ctx, ref, ref_h, n * esz));
    CHECK(ma_memcpy_h2d(ctx, mov, mov_h, n * esz));

    ma_level_report rep[16];
    int n_rep = 0;
    CHECK(ma_optflow_register(ctx, ref, mov, dtype, H, W, &p, (float*)flow, rep, 16, &n_rep));
    CHECK(ma_warp_tiled(ctx, mov, dtype, H, W, (const float*)flow, p.tile_size, p.overlap, warped));
    for (int i = 0; i < n_rep; i++)
        printf("%d %d %d %.17g %.17g %d\n", rep[i].factor, rep[i].h, rep[i].w, rep[i].mi_after, rep[i].mi_before, rep[i].accepted);

    float* flow_h = (float*)malloc(n * 2 * sizeof(float));
    void* warped_h = malloc(n * esz);
    CHECK(ma_memcpy_d2h(ctx, flow_h, flow, n * 2 * sizeof(float)));
    CHECK(ma_memcpy_d2h(ctx, warped_h, warped, n * esz));
    spit(argv[6], flow_h, n * 2 * sizeof(float));
    spit(argv[7], warped_h, n * esz);
    CHECK(ma_free(ctx, ref));
    CHECK(ma_free(ctx, mov));
    CHECK(ma_free(ctx, flow));
    CHECK(ma_free(ctx, warped));
    ma_ctx_destroy(ctx);
    free(ref_h); free(mov_h); free(flow_h); free(warped_h);
    return 0;
}
