// One round of FeatureRegistrator's level loop behind ONE entry point (reference: microaligner/feature_reg/
// feature_registrator.py:162-207 -- per iteration: features of the current moving image, 2-NN + ratio test + RANSAC against the
// level's reference features, cv2.warpAffine by the estimate, the mutual-information gate over dog(reference), dog(candidate),
// dog(current)).  Nothing here is new arithmetic: the function chains the entry points the Python level loop used to call one by
// one -- ma_dog_u8_ex, ma_feature_extract, ma_knn2_l2, ma_match_similarity, ma_warp_affine_cv, the two NMIs of a gate in one pair
// of launches -- keeps every intermediate on the device and returns what the loop decides on: the estimate, the match counts,
// the chunk scores of both halves of the gate and whether a dog() met an image whose max() is 0 (the reference's shortcut,
// :288-291: the caller then repeats the level the careful way).  As ma_optflow_register did for the flow path, this takes the
// per-call overhead of a dozen boundary crossings per round out of the dependency chain.
#include "ma_internal.h"

#include <cstring>

extern "C" int ma_feature_round(ma_ctx* ctx, const void* current, int dtype, int H, int W, const uint8_t* current_gate,
                                uint8_t* current_gate_out, const uint8_t* ref_gate, const float* ref_desc,
                                const double* ref_pts, int n_ref, int tile, int use_dog, size_t nmi_chunk,
                                const double* const* weights_host, const int* radii, const double* cos_sin_host,
                                const double* offs_host, size_t workspace_bytes, void* candidate_out,
                                uint8_t* candidate_gate_out, double* scores_after_host, double* scores_before_host,
                                int max_scores, ma_feature_round_result* res)
{
    MA_REQUIRE(ctx && current && ref_gate && candidate_out && candidate_gate_out && scores_after_host && scores_before_host && res,
               "NULL argument");
    MA_REQUIRE(current_gate || current_gate_out, "dog(current) must be given or a buffer for it");
    MA_REQUIRE(dtype == MA_U8 || dtype == MA_U16 || dtype == MA_F32, "dtype must be u8/u16/f32");
    MA_REQUIRE(H > 0 && W > 0 && tile > 0 && (H > W ? H : W) <= 32000, "bad image size (cv2.warpAffine serves up to 32000 px)");
    MA_REQUIRE(use_dog || dtype == MA_U8, "FAST works on uint8 images: without the DOG preprocess the image itself must be uint8");
    MA_REQUIRE(n_ref == 0 || (ref_desc && ref_pts), "NULL reference features");
    MA_HIP(hipSetDevice(ctx->device));
    if (!ctx->round_flags) MA_HIP(hipHostMalloc((void**)&ctx->round_flags, 64, hipHostMallocDefault));
    int* flags = ctx->round_flags;
    flags[0] = flags[1] = 0;
    std::memset(res, 0, sizeof(*res));
    res->m2x3[0] = res->m2x3[4] = 1.0;
    res->is_identity = 1;
    const size_t n = (size_t)H * W;

    // dog(current): the gate's "before" image and, with use_dog, the image the features come from
    const uint8_t* gate = current_gate;
    if (!gate) {
        MA_TRY(ma_dog_u8_ex(ctx, current, dtype, H, W, 5, 9, MA_DOG_REPORT_ASYNC, nullptr, current_gate_out, &flags[0]));
        gate = current_gate_out;
    }
    const uint8_t* feat_src = use_dog ? gate : static_cast<const uint8_t*>(current);

    // features of the current image (tile_registration.find_features), all on the device
    const int overlap = 51;                                    // tile_registration.py:31
    const long long n_tiles = (long long)((W + tile - 1) / tile) * ((H + tile - 1) / tile);
    const int limit = (int)std::min<long long>(1000000 / n_tiles, 5000);      // feature_detection.py:161-168
    int nq = 0;
    char* feat = nullptr;
    float* desc = nullptr;
    double* pts = nullptr;
    if (limit >= 1 && tile + 2 * overlap <= 65535) {
        const size_t cap = (size_t)n_tiles * limit;
        const size_t b_desc = ma_align_up(cap * 200 * sizeof(float), 256), b_pts = ma_align_up(cap * 16, 256),
                     b_resp = ma_align_up(cap * 4, 256);
        feat = static_cast<char*>(ma_pool_alloc(ctx, b_desc + b_pts + b_resp));
        if (!feat) return MA_ENOMEM;
        desc = reinterpret_cast<float*>(feat);
        pts = reinterpret_cast<double*>(feat + b_desc);
        int* resp = reinterpret_cast<int*>(feat + b_desc + b_pts);
        const int rc = ma_feature_extract(ctx, feat_src, H, W, tile, overlap, 1, limit, weights_host, radii, cos_sin_host, offs_host,
                                          workspace_bytes, (int)cap, desc, pts, resp, &nq);
        if (rc != MA_OK) { ma_pool_free(ctx, feat); return rc; }
    }
    res->n_query = nq;

    // match_features (feature_detection.py:123-158): identity when either side has no features or the reference has fewer than two
    int rc = MA_OK;
    res->status = 4;
    if (nq >= 1 && n_ref >= 2) {
        const size_t b_idx = ma_align_up((size_t)nq * 2 * sizeof(int), 256);
        char* nn = static_cast<char*>(ma_pool_alloc(ctx, 2 * b_idx));
        if (!nn) { ma_pool_free(ctx, feat); return MA_ENOMEM; }
        int* idx = reinterpret_cast<int*>(nn);
        float* dist = reinterpret_cast<float*>(nn + b_idx);
        rc = ma_knn2_l2(ctx, desc, nq, ref_desc, n_ref, 200, idx, dist);
        double m[6];
        if (rc == MA_OK)
            rc = ma_match_similarity(ctx, idx, dist, nq, pts, ref_pts, n_ref, 0.5f, 0.99, 3.0, 2000, nullptr, m, &res->n_good,
                                     &res->status);
        ma_pool_free(ctx, nn);
        if (rc == MA_OK && res->status == 0) {
            std::memcpy(res->m2x3, m, sizeof(m));
            res->is_identity = (m[0] == 1.0 && m[1] == 0.0 && m[2] == 0.0 && m[3] == 0.0 && m[4] == 1.0 && m[5] == 0.0) ? 1 : 0;
        }
    }
    if (feat) ma_pool_free(ctx, feat);
    if (rc != MA_OK) return rc;
    if (res->status == 3) {                    // coordinates the device fit does not take: the caller's host statement decides
        MA_HIP(hipStreamSynchronize(ctx->stream));          // dog(current)'s report has landed before anybody resets the flags
        res->zero_max = flags[0] ? 1 : 0;
        return MA_OK;
    }

    // candidate = cv2.warpAffine(current, estimate), its dog(); an identity estimate compares the current image with itself
    const uint8_t* cand_gate = gate;
    if (!res->is_identity) {
        MA_TRY(ma_warp_affine_cv(ctx, current, dtype, H, W, res->m2x3, H, W, candidate_out));
        MA_TRY(ma_dog_u8_ex(ctx, candidate_out, dtype, H, W, 5, 9, MA_DOG_REPORT_ASYNC, nullptr, candidate_gate_out, &flags[1]));
        cand_gate = candidate_gate_out;
    }

    // the gate: NMI(reference, candidate) and NMI(reference, current), chunk by chunk, one pair of launches
    const size_t nchunks = (nmi_chunk == 0 || nmi_chunk >= n) ? 1 : (n + nmi_chunk - 1) / nmi_chunk;
    MA_REQUIRE(nchunks <= 65535 && (size_t)max_scores >= nchunks, "scores buffer too small");
    MA_TRY(ma_pinned_reserve(ctx, 2 * nchunks * sizeof(double)));
    double* p0 = static_cast<double*>(ctx->pinned);
    double* p1 = p0 + nchunks;
    MA_TRY(ma_nmi_u8_enqueue2(ctx, ref_gate, cand_gate, gate, n, nmi_chunk, p0, p1, max_scores, &res->n_scores));
    MA_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->profile) MA_TRY(ma_profile_flush(ctx));
    for (int i = 0; i < res->n_scores; i++) { scores_after_host[i] = p0[i]; scores_before_host[i] = p1[i]; }
    res->zero_max = (flags[0] ? 1 : 0) | (flags[1] ? 2 : 0);
    return MA_OK;
}
