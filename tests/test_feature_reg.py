"""FeatureRegistrator: the sparse CPU stage on its own (no GPU needed), the cv2.warpAffine
kernel against the oracle, and the whole class on a known similarity transform (GPU)."""
import numpy as np
import pytest

from oracle import oracle as O
from microaligner_amd import synthetic
from microaligner_amd.feature_reg import sparse_cpu as SP
from microaligner_amd.feature_reg import tile_registration as TR
from microaligner_amd.feature_reg.feature_detection import Features, find_features, match_features


def _brute_fast(img, t=1):
    h, w = img.shape
    out = np.zeros((h, w), np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            v = int(img[y, x])
            d = [v - int(img[y + dy, x + dx]) for dx, dy in SP._RING]
            best = max(max(min(d[(s + j) % 16] for j in range(9)), min(-d[(s + j) % 16] for j in range(9)))
                       for s in range(16))
            out[y, x] = best - 1 if best > t else 0
    return out


def test_fast_score_is_the_segment_test_definition():
    rng = np.random.default_rng(0)
    for img in ((rng.random((40, 47)) * 255).astype(np.uint8),
                np.clip(rng.normal(128, 2, (50, 50)), 0, 255).astype(np.uint8)):
        assert np.array_equal(SP.fast_score_map(img, 1), _brute_fast(img, 1))
    flat = np.full((30, 30), 77, np.uint8)
    assert SP.fast_detect(flat) == [] and not SP.fast_score_map(np.zeros((5, 5), np.uint8)).any()
    spot = flat.copy()
    spot[15, 15] = 200                              # an isolated bright pixel: one corner, all ring pixels darker
    kps = SP.fast_detect(spot)
    assert [k.pt for k in kps] == [(15.0, 15.0)] and kps[0].response == 200 - 77 - 1 and kps[0].size == 7.0


def test_nonmax_suppression_keeps_strict_local_maxima():
    img = synthetic.make_cells(120, 130, seed=2, dtype=np.uint8)
    s = SP.fast_score_map(img, 1)
    pts = {(int(k.pt[1]), int(k.pt[0])) for k in SP.fast_detect(img)}
    assert pts and len(pts) < (s > 0).sum()
    for y, x in pts:
        nb = s[y - 1:y + 2, x - 1:x + 2].copy()
        nb[1, 1] = -1
        assert s[y, x] > nb.max()


def test_knn_and_similarity_fit():
    rng = np.random.default_rng(1)
    q, t = rng.random((150, 200)).astype(np.float32), rng.random((260, 200)).astype(np.float32)
    idx, dist = SP.knn2(q, t, block=64)
    D = np.sqrt(((q[:, None, :] - t[None]) ** 2).sum(-1))
    o = np.argsort(D, 1)[:, :2]
    assert np.array_equal(idx, o) and np.allclose(dist, np.take_along_axis(D, o, 1), atol=1e-3)
    src = rng.random((300, 2)) * 800
    th = np.deg2rad(4.0)
    M = np.array([[1.03 * np.cos(th), -1.03 * np.sin(th), 20.5], [1.03 * np.sin(th), 1.03 * np.cos(th), -11.25]])
    dst = src @ M[:, :2].T + M[:, 2] + rng.normal(0, 0.2, (300, 2))
    dst[:100] = rng.random((100, 2)) * 800           # a third of the matches are wrong
    est, mask = SP.estimate_affine_partial_2d(src, dst)
    assert np.abs(est - M).max() < 0.05 and 180 <= mask.sum() <= 215 and not mask[:100].sum() > 10
    assert SP.estimate_affine_partial_2d(src[:1], dst[:1])[0] is None


def test_tiles_and_feature_bookkeeping():
    img = synthetic.make_cells(450, 620, seed=3, dtype=np.uint8)
    tiles, info = TR.split_image_into_tiles(img, 300)
    assert len(tiles) == 6 and info["ntiles"] == dict(x=3, y=2) and tiles[0].shape == (402, 402)
    assert np.array_equal(tiles[4][51:201, 51:351], img[300:450, 300:600]) and not tiles[0][:51].any()
    f = TR.find_features(img, 300)
    assert f.is_valid() and f.descriptors.shape == (len(f.keypoints), 200) and f.descriptors.dtype == np.float32
    xs = np.array([k.pt for k in f.keypoints])
    assert xs[:, 0].max() < 620 + 3 and xs[:, 1].max() < 450 + 3 and xs.min() >= 0
    assert not find_features(np.zeros((200, 200), np.uint8)).is_valid()
    assert np.array_equal(match_features(Features(), f, verbose=False), np.eye(2, 3))


def test_sparse_stage_recovers_a_translation():
    ref = synthetic.make_cells(600, 700, seed=4)
    M = np.array([[1.0, 0.0, 9.0], [0.0, 1.0, -6.0]])
    mov = O.warp_affine(ref, M)
    fr, fm = TR.find_features(O.dog(ref, True), 400), TR.find_features(O.dog(mov, True), 400)
    T = TR.register_img_pair(fr, fm, verbose=False)          # maps moving -> reference coordinates
    assert np.abs(T - np.array([[1, 0, -9.0], [0, 1, 6.0]])).max() < 0.15


def test_oracle_warp_affine_known_answers():
    rng = np.random.default_rng(5)
    img = (rng.random((40, 50)) * 255).astype(np.uint8)
    assert np.array_equal(O.warp_affine(img, np.eye(2, 3)), img)
    out = O.warp_affine(img, np.array([[1, 0, 3.0], [0, 1, -2.0]]))
    exp = np.zeros_like(img)
    exp[:-2, 3:] = img[2:, :-3]
    assert np.array_equal(out, exp)
    f = img.astype(np.float32)
    half = O.warp_affine(f, np.array([[1, 0, 0.5], [0, 1, 0.0]]))
    assert np.array_equal(half[:, 1:], 0.5 * (f[:, :-1] + f[:, 1:]))
    big = O.warp_affine(img, np.array([[2.0, 0, 0], [0, 2.0, 0]]), dsize=(100, 80))
    assert big.shape == (80, 100) and np.array_equal(big[::2, ::2][:40, :50], img)


# ---- the independent oracle of the feature stage (oracle/feature_oracle.py) and what pins it -------------------------
import os  # noqa: E402

SK = np.load(os.path.join(os.path.dirname(__file__), "golden", "feature_skimage.npz"))


@pytest.mark.parametrize("case", range(len(SK["cases"])))
def test_feature_oracle_is_pinned_by_scikit_image_fixtures(case):
    """oracle/feature_oracle.py against fixtures made with scikit-image 0.18.3 (the conda interpreter of the build
    container, tests/golden/make_feature_skimage_golden.py): the set of pixels that pass the 9-of-16 segment test equals
    corner_fast(n=9) exactly, and scikit-image's DAISY variant run on the oracle's smoothing / ring-geometry / layout
    helpers reproduces skimage.feature.daisy to 1e-12 (float64 on both sides; a few additions differ in order)."""
    from oracle import feature_oracle as FO
    seed, _, _, _, t = SK["cases"][case]
    seed, t = int(seed), int(t)
    tile = SK[f"tile{seed}"]
    exp = np.unpackbits(SK[f"fast{seed}"])[:tile.size].reshape(tile.shape).astype(bool)
    assert np.array_equal(FO.fast_is_corner(tile, t), exp) and exp.sum() > 500
    np.testing.assert_allclose(FO.daisy_skimage(tile, int(SK["step"])), SK[f"daisy{seed}"], rtol=0, atol=1e-12)
    # the score is the largest threshold that keeps the corner: positive exactly on the corners, and every sampled corner
    # passes the test at its score but not one above
    score = FO.fast_score(tile, t)
    assert np.array_equal(score > 0, exp)
    ys, xs = np.nonzero(exp)
    for y, x in list(zip(ys, xs))[::211]:
        sc = int(score[y, x])
        assert FO.fast_is_corner(tile, sc)[y, x] and not FO.fast_is_corner(tile, sc + 1)[y, x]


def test_host_feature_code_equals_the_oracle():
    """The product's host statement of the stage (feature_reg/sparse_cpu.py: FAST by sliding minima, DAISY on
    scipy.ndimage) against the oracle's direct definitions: corner scores after non-maximum suppression identical,
    descriptors identical to the last bit."""
    from oracle import feature_oracle as FO
    for seed in SK["cases"][:, 0].astype(int):
        tile = SK[f"tile{seed}"]
        exp = FO.fast_nms(FO.fast_score(tile, 1))
        got = np.zeros_like(exp)
        for k in SP.fast_detect(tile, threshold=1, nonmax=True):
            got[int(k.pt[1]), int(k.pt[0])] = int(k.response)
        assert np.array_equal(got, exp) and (exp > 0).sum() > 50
        ys, xs = np.nonzero(exp)
        pts = np.stack([xs, ys], 1).astype(np.float64)[::3]
        pts[::2] += 0.37                              # off-grid points exercise the bilinear weights
        des = SP.Daisy(radius=21, q_radius=3, q_theta=8, q_hist=8).compute(tile, [SP.KeyPoint((float(x), float(y))) for x, y in pts])
        assert np.array_equal(des, FO.daisy_describe(tile, pts))


# ---- GPU ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_device_features_equal_the_oracle(ctx):
    """The dense kernels of the feature stage, batched over all tiles of a level (ma_fast_nms, ma_daisy_describe),
    against the independent oracle of the stage (oracle/feature_oracle.py, pinned by scikit-image fixtures where a
    third-party implementation exists): non-maximum-suppressed FAST score maps identical, keypoints, responses and
    descriptors of every tile identical to the last bit.  Includes an all-zero tile and a constant one (no features)."""
    from oracle import feature_oracle as FO
    from microaligner_amd.feature_reg import feature_detection as FD
    img = O.dog(synthetic.make_cells(560, 640, seed=13), True)
    tiles, _ = TR.split_image_into_tiles(img, 250)
    tiles.append(np.zeros_like(tiles[0]))
    tiles.append(np.full_like(tiles[0], 7))
    stack = np.ascontiguousarray(np.stack(tiles))
    score = ctx.fast_nms(ctx.asdevice(stack), FD.TILE_OVERLAP, threshold=1)
    checked = (0, 4, len(tiles) - 2, len(tiles) - 1)
    maps = {t: FO.fast_detect(tiles[t], FD.TILE_OVERLAP, 1) for t in checked}
    for t in checked:
        assert np.array_equal(score[t], maps[t])
    dev = FD.find_features_device(tiles, ctx)
    assert len(dev) == len(tiles) and not dev[-1].is_valid() and not dev[-2].is_valid()
    limit = min(1000000 // len(tiles), 5000)
    for t in checked[:2]:
        ys, xs = np.nonzero(maps[t])
        order = np.argsort(-maps[t][ys, xs], kind="stable")[:limit]     # strongest first, row-major among equals
        assert dev[t].is_valid() and len(dev[t].pts) == len(order) > 100
        assert np.array_equal(dev[t].pts, np.stack([xs[order], ys[order]], 1).astype(np.float64))
        assert np.array_equal(dev[t].responses, maps[t][ys, xs][order].astype(np.float64))
        assert np.array_equal(dev[t].descriptors, FO.daisy_describe(tiles[t], dev[t].pts))
    # batches sized from a workspace budget (a 25 000^2 level does not fit one): same features whatever the batch size
    P = tiles[0].shape[0]
    assert [list(r) for r in FD._device_batches(5, P, 2 * 128 * P * P)] == [[0, 1], [2, 3], [4]]
    assert len(FD._device_batches(20000, 302, 1 << 50)) == 3            # 8 planes per tile, 65535 blocks along z
    small = FD.find_features_device(tiles, ctx, workspace_bytes=3 * 128 * P * P)
    for a, b in zip(dev, small):
        assert a.is_valid() == b.is_valid()
        if a.is_valid():
            assert np.array_equal(a.pts, b.pts) and np.array_equal(a.descriptors, b.descriptors)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,tile,dense", [((1130, 1210), 1000, True), ((700, 820), 300, False), ((260, 240), 200, False)])
def test_keypoints_selected_on_the_device_equal_the_oracle_selection(ctx, shape, tile, dense):
    """The route FeatureRegistrator takes: the DOG image stays on the device, the feature windows are cut there
    (ma_cut_tiles_u8), the corners are detected, ranked and cut to the per-tile limit there (ma_fast_keypoints: histogram
    cut-off, ordered collection, bitonic sort) and the descriptors stay there for the 2-NN search.  Against the oracle:
    the reference's selection (strongest first, row-major among equal responses, nfeatures_limit) tile by tile -- the
    dense case has ten times more corners than the limit of 5000 and thousands of ties at the cut-off score."""
    from oracle import feature_oracle as FO
    from microaligner_amd.feature_reg import feature_detection as FD
    rng = np.random.default_rng(shape[0])
    if dense:
        from scipy.ndimage import gaussian_filter
        img = gaussian_filter(rng.standard_normal(shape), 1.2)
        img = np.round((img - img.min()) / (img.max() - img.min()) * 255).astype(np.uint8)
    else:
        img = O.dog(synthetic.make_cells(*shape, seed=shape[1]), True)
    d_img = ctx.asdevice(img)
    got = FD.find_features_of_device_image(d_img, tile, ctx)
    tiles, info = TR.split_image_into_tiles(img, tile)
    assert np.array_equal(ctx.cut_tiles(d_img, tile, FD.TILE_OVERLAP, 0, len(tiles)).numpy(), np.stack(tiles))
    limit = min(1000000 // len(tiles), 5000)
    pts, resp, des = [], [], []
    for t, tl in enumerate(tiles[:4] if dense else tiles):
        m = FO.fast_detect(tl, FD.TILE_OVERLAP, 1)
        ys, xs = np.nonzero(m)
        order = np.argsort(-m[ys, xs], kind="stable")[:limit]
        if dense and t == 0:
            assert len(ys) > 5 * limit and (m[ys, xs] == m[ys, xs][order][-1]).sum() > 50       # a real cut with ties
        if len(order) < 3:
            continue
        origin = np.array([t % info["ntiles"]["x"] * tile, t // info["ntiles"]["x"] * tile], np.float64)
        p = np.stack([xs[order], ys[order]], 1).astype(np.float64)
        pts.append(p + origin)
        resp.append(m[ys, xs][order].astype(np.float64))
        des.append(FO.daisy_describe(tl, p[::7]))
    n = sum(len(p) for p in pts)
    assert got.is_valid() and len(got.pts) >= n
    assert np.array_equal(got.pts[:n], np.concatenate(pts)) and np.array_equal(got.responses[:n], np.concatenate(resp))
    pos, gdes = 0, got.descriptors
    for p, d in zip(pts, des):
        assert np.array_equal(gdes[pos:pos + len(p)][::7], d)
        pos += len(p)
    # the host-tile route gives the very same combined features
    old = TR.find_features(img, tile, ctx)
    assert np.array_equal(old.pts, got.pts) and np.array_equal(old.descriptors, got.descriptors)
    # ... and so does ma_feature_extract when its workspace holds two tiles at a time (several batches, running offsets)
    P = tile + 2 * FD.TILE_OVERLAP
    batched = FD.find_features_of_device_image(d_img, tile, ctx, workspace_bytes=2 * 128 * P * P)
    assert np.array_equal(batched.pts, got.pts) and np.array_equal(batched.responses, got.responses)
    assert np.array_equal(batched.descriptors, got.descriptors)
    # the 2-NN search takes the descriptors where they are
    idx_d, dist_d = ctx.knn2(got.descriptors_for_search, got.descriptors_for_search)
    idx_h, dist_h = ctx.knn2(old.descriptors, old.descriptors)
    assert np.array_equal(idx_d, idx_h) and np.array_equal(dist_d, dist_h)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
@pytest.mark.parametrize("case", ["rot", "scale", "shift", "far"])
def test_warp_affine_cv_bit_exact(ctx, dtype, case):
    rng = np.random.default_rng(7)
    img = (rng.random((203, 317)) * (60000 if dtype == np.uint16 else 255)).astype(dtype)
    th = np.deg2rad(7.3)
    M = {"rot": np.array([[np.cos(th), -np.sin(th), 10.4], [np.sin(th), np.cos(th), -21.7]]),
         "scale": np.array([[1.37, 0.02, -30.0], [-0.01, 0.81, 12.125]]),
         "shift": np.array([[1.0, 0.0, 0.484375], [0.0, 1.0, -3.0]]),
         "far": np.array([[1.0, 0.0, 5000.0], [0.0, 1.0, 0.0]])}[case]
    got = ctx.warp_affine_cv(ctx.asdevice(img), M).numpy()
    assert np.array_equal(got, O.warp_affine(img, M))
    got = ctx.warp_affine_cv(ctx.asdevice(img), M, dsize=(400, 150)).numpy()
    assert got.shape == (150, 400) and np.array_equal(got, O.warp_affine(img, M, dsize=(400, 150)))


@pytest.mark.gpu
@pytest.mark.parametrize("nq,nt,dim", [(150, 260, 200), (1, 2, 200), (700, 64, 8), (129, 1000, 198), (3000, 5000, 200)])
def test_device_knn2_is_the_exact_search(ctx, nq, nt, dim):
    """ma_knn2_l2 against the brute-force definition and against the host search of the sparse stage."""
    rng = np.random.default_rng(nq + nt)
    q, t = rng.random((nq, dim)).astype(np.float32), rng.random((nt, dim)).astype(np.float32)
    if nt > 10:
        t[7] = t[3]                      # an exact duplicate: the tie goes to the lower index
        q[0] = t[3]
    idx, dist = ctx.knn2(q, t)
    # brute force in float64, in query blocks
    o = np.empty((nq, 2), np.int64)
    od = np.empty((nq, 2), np.float64)
    t64 = t.astype(np.float64)
    for s0 in range(0, nq, 256):
        D = np.sqrt(((q[s0:s0 + 256, None, :].astype(np.float64) - t64[None]) ** 2).sum(-1))
        o[s0:s0 + 256] = np.argsort(D, 1, kind="stable")[:, :2]
        od[s0:s0 + 256] = np.take_along_axis(D, o[s0:s0 + 256], 1)
    assert np.allclose(dist, od, rtol=1e-5, atol=1e-6)
    # (the distances above pin the result; float32 accumulation may still order two near-equal neighbours differently
    # from float64)
    assert (idx != o).any(1).mean() < 0.01
    if nq * nt < 2e5:
        assert np.array_equal(idx, o)
    if nq * nt <= 3000 * 5000:
        si, sd = SP.knn2_sequential(q, t)          # the float32 definition, restated with numpy: bit for bit
        assert np.array_equal(idx, si) and np.array_equal(dist, sd)
    hi, hd = SP.knn2(q, t)     # the host search (|q|^2 + |t|^2 - 2 q.t in float32) agrees up to its own rounding,
    far = od[:, 0] > 0.1       # which is large for (near-)duplicates: cancellation
    assert (idx == hi).all(1).mean() > 0.98 and np.allclose(dist[far], hd[far], rtol=2e-3, atol=3e-3)
    if nt > 10:
        assert list(idx[0]) == [3, 7] and dist[0, 0] == 0 and dist[0, 1] == 0
    with pytest.raises(ValueError):
        ctx.knn2(q, t[:1])


@pytest.mark.gpu
@pytest.mark.parametrize("radius,P", [(9, 173), (21, 173), (21, 330), (36, 140), (60, 173)])
def test_device_daisy_smoothing_radii(ctx, radius, P):
    """ma_daisy_describe over the radius buckets of its smoothing kernels (register windows of 12, 24 and 40 taps a side,
    the tap-by-tap kernel beyond) and over window sizes that are no multiple of the 16 outputs a thread makes: descriptors
    identical to the oracle's scipy chain at points all over the tile, edges and corners included."""
    from oracle import feature_oracle as FO
    from microaligner_amd.feature_reg import feature_detection as FD
    from microaligner_amd.feature_reg.sparse_cpu import Daisy
    rng = np.random.default_rng(radius + P)
    tiles = np.stack([O.dog(synthetic.make_cells(P, P, seed=radius + k), True) for k in range(2)])
    pts = np.concatenate([rng.integers(0, P, (300, 2)), [[0, 0], [P - 1, P - 1], [0, P - 1], [P - 1, 0], [P // 2, 0]]]).astype(np.float64)
    halves, cos_sin, offsets = FD._daisy_tables(Daisy(radius=radius, q_radius=3, q_theta=8, q_hist=8))
    kp_tile = np.repeat(np.arange(2, dtype=np.int32), len(pts))
    got = ctx.daisy_describe(ctx.asdevice(np.ascontiguousarray(tiles)), kp_tile, np.concatenate([pts, pts]), halves, cos_sin, offsets)
    for k in range(2):
        assert np.array_equal(got[k * len(pts):(k + 1) * len(pts)], FO.daisy_describe(tiles[k], pts, radius=radius))


def _knn_case(kind, nq, nt, dim, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((nq, dim)).astype(np.float32), rng.random((nt, dim)).astype(np.float32)
    if kind == "histograms":       # DAISY-like: non-negative, unit L2 norm, most of the mass in a few bins
        def h(n):
            a = rng.gamma(0.3, 1.0, (n, dim)).astype(np.float32)
            return a / np.sqrt((a * a).sum(1, keepdims=True), dtype=np.float32)
        t = h(nt)
        q = t[rng.integers(0, nt, nq)] + 0.02 * h(nq)      # queries near train rows, as matching descriptors are
        return q.astype(np.float32), t
    if kind == "duplicates":       # a few distinct rows, each many times: every query ties among dozens of rows
        base = rng.random((16, dim)).astype(np.float32)
        return base[rng.integers(0, 16, nq)], base[rng.integers(0, 16, nt)]
    if kind == "near_ties":        # rows that differ in the last bits only
        base = rng.random((1, dim)).astype(np.float32)
        rows = np.repeat(base, nt, 0)
        t = np.where(rng.random((nt, dim)) < 0.5, rows, np.nextafter(rows, np.float32(2), dtype=np.float32))
        return rng.random((nq, dim)).astype(np.float32), t
    if kind == "wide_range":       # magnitudes over nine decades within every row: the split-float16 operands lose the small ones
        mag = np.exp(rng.uniform(np.log(1e-6), np.log(1e3), (nt, dim)))
        t = (rng.standard_normal((nt, dim)) * mag).astype(np.float32)
        q = (t[rng.integers(0, nt, nq)] * (1 + 1e-3 * rng.standard_normal((nq, dim)))).astype(np.float32)
        return q, t
    if kind == "tiny":             # everything around 1e-20: only the power-of-two scaling keeps it out of the float16 underflow
        return (rng.random((nq, dim)) * 1e-20).astype(np.float32), (rng.random((nt, dim)) * 1e-20).astype(np.float32)
    if kind == "outlier":          # one huge element sets the scale of the whole set
        q, t = rng.random((nq, dim)).astype(np.float32), rng.random((nt, dim)).astype(np.float32)
        t[nt // 2, 3] = 3e4
        q[0, 5] = 1e5
        return q, t
    if kind == "clusters":         # tight clusters: many train rows within 1e-4 relative of each query's neighbours
        centres = rng.random((40, dim)).astype(np.float32)
        t = (centres[rng.integers(0, 40, nt)] + 1e-4 * rng.standard_normal((nt, dim))).astype(np.float32)
        q = (centres[rng.integers(0, 40, nq)] + 1e-4 * rng.standard_normal((nq, dim))).astype(np.float32)
        return q, t
    raise KeyError(kind)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["filtered", "filtered_f32"])
@pytest.mark.parametrize("kind,nq,nt,dim", [
    ("uniform", 1000, 3000, 200), ("uniform", 129, 1000, 200), ("uniform", 1, 2, 200), ("uniform", 300, 5, 8),
    ("uniform", 257, 131, 216), ("uniform", 64, 700, 12), ("histograms", 2500, 4100, 200), ("duplicates", 500, 1500, 200),
    ("near_ties", 200, 900, 200), ("uniform", 6000, 9000, 200), ("wide_range", 700, 2100, 200), ("tiny", 400, 1300, 200),
    ("outlier", 500, 1500, 200), ("clusters", 900, 2600, 200), ("histograms", 333, 1111, 44), ("uniform", 200, 600, 208)])
def test_filtered_knn2_equals_the_exact_search(ctx, kind, nq, nt, dim, mode):
    """ma_knn2_l2_ex: the matrix-core shortlist (split-float16 operands on the FP16 cores, or round 3's FP32 one) + exact
    re-evaluation + certificate returns, bit for bit, what the exact kernel returns -- on easy data through the certificate, on
    ties, duplicates and data the split operands cannot resolve (nine decades of magnitude, tight clusters) through the exact
    fallback."""
    q, t = _knn_case(kind, nq, nt, dim, seed=nq + nt)
    dq, dt = ctx.asdevice(q), ctx.asdevice(t)
    stats = {}
    fi, fd = ctx.knn2(dq, dt, mode=mode, stats=stats)
    ei, ed = ctx.knn2(dq, dt, mode="exact")
    assert np.array_equal(fi, ei) and np.array_equal(fd, ed), (kind, stats)
    if nq * nt <= 3000 * 5000:
        si, sd = SP.knn2_sequential(q, t)
        assert np.array_equal(fi, si) and np.array_equal(fd, sd)
    if kind in ("uniform", "histograms") and nt > 16:
        assert stats["uncertified"] <= nq // 50, stats          # the certificate carries the typical case
    if kind in ("duplicates", "near_ties"):
        assert stats["uncertified"] > nq // 2, stats            # ... and refuses to guess where rows tie
    ai, ad = ctx.knn2(dq, dt)                                    # whatever "auto" picks
    assert np.array_equal(ai, ei) and np.array_equal(ad, ed)
    with pytest.raises(ValueError):
        ctx.knn2(dq, dt, mode="fast")


@pytest.mark.gpu
def test_filtered_knn2_limits(ctx):
    q = np.random.default_rng(0).random((40, 220)).astype(np.float32)
    ctx.knn2(q, q, mode="auto")                                  # longer descriptors: the exact kernel serves them
    with pytest.raises(ValueError, match="216"):
        ctx.knn2(q, q, mode="filtered")


@pytest.mark.gpu
@pytest.mark.parametrize("shape,tile,seed", [((450, 620), 300, 3), ((700, 700), 1000, 4), ((1300, 1100), 400, 5)])
def test_device_features_equal_the_host_features(ctx, shape, tile, seed):
    """ma_fast_nms + ma_daisy_describe (all tiles of a level in one batch) against the numpy / scipy code of the
    sparse stage tile by tile: same keypoints in the same order, same descriptors."""
    img = O.dog(synthetic.make_cells(*shape, seed=seed), True)
    host = TR.find_features(img, tile)
    dev = TR.find_features(img, tile, ctx)
    assert host.is_valid() and dev.is_valid() and len(dev.keypoints) == len(host.keypoints)
    assert [(k.pt, k.response) for k in dev.keypoints] == [(k.pt, k.response) for k in host.keypoints]
    assert dev.descriptors.shape == host.descriptors.shape and dev.descriptors.dtype == np.float32
    diff = np.abs(dev.descriptors - host.descriptors)
    print("descriptor max |d|", diff.max(), "differing", int((diff > 0).sum()), "/", diff.size)
    assert diff.max() <= 1e-6
    # an all-zero tile and a featureless image
    assert not TR.find_features(np.zeros((300, 300), np.uint8), 200, ctx).is_valid()
    with pytest.raises(ValueError):
        TR.find_features(img.astype(np.uint16), tile, ctx)


@pytest.mark.gpu
def test_feature_registrator_recovers_a_similarity_transform():
    from microaligner_amd import FeatureRegistrator, transform_img_with_tmat
    H, W = 1500, 1700
    ref = synthetic.make_cells(H, W, seed=6)
    th = np.deg2rad(0.7)
    M = np.array([[np.cos(th), -np.sin(th), 17.0], [np.sin(th), np.cos(th), -11.0]])
    mov = O.warp_affine(ref, M)
    freg = FeatureRegistrator()
    freg.verbose = False
    freg.num_pyr_lvl, freg.tile_size = 2, 500
    assert freg.num_iterations == 3 and freg.use_dog is True
    freg.ref_img, freg.mov_img = ref, mov
    assert freg.level_factors == [4, 2]
    T = freg.register()
    assert T.shape == (2, 3) and T.dtype == np.float64
    Mi = np.linalg.inv(np.vstack([M, [0, 0, 1]]))[:2]
    assert np.abs(T[:, :2] - Mi[:, :2]).max() < 2e-3 and np.abs(T[:, 2] - Mi[:, 2]).max() < 2.0
    aligned = transform_img_with_tmat(mov, (H, W), T)
    inner = (slice(100, -100), slice(100, -100))
    before = np.abs(mov[inner].astype(np.float64) - ref[inner]).mean()
    after = np.abs(aligned[inner].astype(np.float64) - ref[inner]).mean()
    assert after < 0.35 * before
    # reuse_ref_img keeps the reference features (the pipeline registers many cycles against one reference)
    feats = [lvl.features for lvl in freg._levels]
    freg.mov_img = mov
    T2 = freg.register(reuse_ref_img=True)
    assert all(lvl.features is f0 for lvl, f0 in zip(freg._levels, feats)) and np.allclose(T2, T)


@pytest.mark.gpu
def test_feature_registrator_helpers():
    from microaligner_amd import FeatureRegistrator
    f = FeatureRegistrator()
    a, b = np.array([[1.0, 0, 5], [0, 1, -2]]), np.array([[0.0, -1, 0], [1, 0, 3]])
    from microaligner_amd.feature_reg import affine_math as am
    assert np.allclose(am.compose([a, b]), (np.vstack([a, [0, 0, 1]]) @ np.vstack([b, [0, 0, 1]]))[:2])
    assert np.array_equal(am.with_translation_scaled(a, 4), np.array([[1.0, 0, 20], [0, 1, -8]]))
    assert am.scales_plausible(a) and not am.scales_plausible(a * np.array([[5, 5, 1]] * 2))
    assert not am.scales_plausible(np.zeros((2, 3)))
    assert am.scales_plausible(np.array([[0.0, 1.0, 0], [0.0, 2.0, 0]])) is False      # rank 1: zero area
    assert am.axis_scales(np.array([[0.0, -2.0, 3], [2.0, 0.0, 1]])) == (2.0, 2.0)     # rotation by 90 deg x 2
    assert am.centre_stays_inside(a, (100, 100)) and not am.centre_stays_inside(np.array([[1.0, 0, 500], [0, 1, 0]]), (100, 100))
    assert f.level_factors == [8, 4, 2]
    f.num_pyr_lvl, f.use_full_res_img = 1, True
    f.ref_img = np.zeros((300, 500), np.uint8)
    assert f.level_factors == [2, 1]
    f.num_pyr_lvl, f.use_full_res_img = 3, False
    with pytest.raises(ValueError):
        f.ref_img = np.zeros((3, 3, 3))
    f.num_pyr_lvl = 0
    f.ref_img = f.mov_img = np.zeros((300, 300), np.uint8)
    with pytest.raises(ValueError, match="use_full_res_img"):
        f.register()


def _lattice_case(rng, trial, big=False):
    """Matched keypoint pairs as FeatureRegistrator meets them: integer pixel positions, a similarity (every third trial a pure
    integer translation, under which residuals of EXACTLY the 3-px threshold occur) plus noise, a share of outliers."""
    n = int(rng.integers(9000, 12000)) if big else int(rng.integers(5, 2500))
    th, sc = rng.uniform(-0.05, 0.05), rng.uniform(0.95, 1.05)
    M = np.array([[sc * np.cos(th), -sc * np.sin(th), float(rng.integers(-30, 30))],
                  [sc * np.sin(th), sc * np.cos(th), float(rng.integers(-30, 30))]])
    if trial % 3 == 0:
        M = np.array([[1, 0, float(rng.integers(-30, 30))], [0, 1, float(rng.integers(-30, 30))]], float)
    src = rng.integers(0, 1500, (n, 2)).astype(np.float32)
    dst = np.rint(src @ M[:, :2].T + M[:, 2] + rng.normal(0, 1.2, (n, 2))).astype(np.float32)
    out = rng.random(n) < rng.uniform(0, 0.6)
    dst[out] = rng.integers(0, 1500, (int(out.sum()), 2)).astype(np.float32)
    return src, dst


def test_similarity_fit_is_the_least_squares_solution():
    """sparse_cpu._fit_similarity (closed form from centred integer sums, float64) against the same minimisation solved
    EXACTLY -- the normal equations of the uncentred problem in rational arithmetic -- and, loosely, against numpy's SVD-based
    lstsq on the 2n x 4 design matrix (which loses digits itself when the points sit far from the origin)."""
    from fractions import Fraction as F
    rng = np.random.default_rng(2)
    for trial in range(40):
        src, dst = _lattice_case(rng, trial)
        src, dst = src.astype(np.float64) + (4000 if trial % 5 == 0 else 0), dst.astype(np.float64)
        n = len(src)
        x, y, u, v = ([int(t) for t in col] for col in (src[:, 0], src[:, 1], dst[:, 0], dst[:, 1]))
        Sx, Sy, Su, Sv = sum(x), sum(y), sum(u), sum(v)
        Sxx = sum(p * p + q * q for p, q in zip(x, y))
        Sxu = sum(p * r + q * t for p, q, r, t in zip(x, y, u, v))
        Sxv = sum(p * t - q * r for p, q, r, t in zip(x, y, u, v))
        D = F(n * Sxx - (Sx * Sx + Sy * Sy))
        a, b = (n * Sxu - (Sx * Su + Sy * Sv)) / D, (n * Sxv - (Sx * Sv - Sy * Su)) / D
        tx, ty = (Su - (a * Sx - b * Sy)) / n, (Sv - (b * Sx + a * Sy)) / n
        exact = np.array([[float(a), float(-b), float(tx)], [float(b), float(a), float(ty)]])
        M = SP._fit_similarity(src, dst)
        assert np.allclose(M, exact, rtol=1e-13, atol=1e-12), (trial, np.abs(M - exact).max())
        A = np.zeros((2 * n, 4))
        A[0::2, 0], A[0::2, 1], A[0::2, 2] = src[:, 0], -src[:, 1], 1.0
        A[1::2, 0], A[1::2, 1], A[1::2, 3] = src[:, 1], src[:, 0], 1.0
        la, lb, ltx, lty = np.linalg.lstsq(A, dst.reshape(-1), rcond=None)[0]
        assert np.allclose(M, [[la, -lb, ltx], [lb, la, lty]], rtol=0, atol=1e-7), trial
    assert SP._fit_similarity(np.ones((5, 2)), rng.random((5, 2))) is None           # coinciding source points: rank < 4
    assert SP._fit_similarity(np.ones((1, 2)), np.ones((1, 2))) is None


def test_random_stream_of_the_device_ransac_is_numpys():
    """csrc/ransac.hip restates numpy's PCG64 stream, its Lemire bounded integers and Generator.choice(n, 2, replace=False)
    on the host side of ma_match_similarity: the very pairs numpy draws, for populations from 2 to 2^31; and its adaptive
    iteration count is sparse_cpu.ransac_iterations (both the C library's log)."""
    import ctypes as C
    from microaligner_amd import _lib as L
    lib = L.load()
    for seed in (0, 7):
        raw = np.random.PCG64(seed).state["state"]
        m64 = (1 << 64) - 1
        st = (C.c_ulonglong * 4)(raw["state"] >> 64, raw["state"] & m64, raw["inc"] >> 64, raw["inc"] & m64)
        for n in (2, 3, 4, 5, 7, 100, 4097, 45000, 65536, 1000003, 2 ** 31 - 5):
            cnt = 2000 if n < 100000 else 200
            out = np.zeros((cnt, 2), np.int32)
            L.check(lib.ma_host_pcg64_choice2(st, n, cnt, out.ctypes.data_as(C.POINTER(C.c_int))))
            rng = np.random.default_rng(seed)
            assert np.array_equal(out, np.array([rng.choice(n, 2, replace=False) for _ in range(cnt)])), (seed, n)
    # a NULL state means numpy.random.PCG64(0): the constants in csrc/ransac.hip are numpy's
    out = np.zeros((500, 2), np.int32)
    L.check(lib.ma_host_pcg64_choice2(None, 12345, 500, out.ctypes.data_as(C.POINTER(C.c_int))))
    rng0 = np.random.default_rng(0)
    assert np.array_equal(out, np.array([rng0.choice(12345, 2, replace=False) for _ in range(500)]))
    rng = np.random.default_rng(3)
    for _ in range(3000):
        n = int(rng.integers(2, 50000))
        count = int(rng.integers(0, n + 1))
        it = int(rng.integers(1, 2000))
        conf = float(rng.choice([0.99, 0.9, 0.999, rng.uniform(0.5, 0.9999)]))
        got = C.c_int(-1)
        L.check(lib.ma_host_ransac_iterations(count, n, conf, 2000, it, C.byref(got)))
        assert got.value == SP.ransac_iterations(count, n, conf, 2000, it), (count, n, conf, it)


@pytest.mark.gpu
def test_device_ratio_test_and_ransac_equal_the_host_statement(ctx):
    """ma_match_similarity against sparse_cpu.estimate_affine_partial_2d over the good matches, bit for bit: the same matrix
    (array_equal), the same number of good matches, None where the host returns None -- on pixel lattices with exact-threshold
    residuals, with outliers, duplicated points, and with too few matches."""
    rng = np.random.default_rng(1)
    for trial in range(150):
        # (every 25th case has more than 8192 good matches: several passes of the single-block kernels per thread)
        big = trial % 25 == 11
        mov, ref = _lattice_case(rng, trial, big)            # query (moving) points and where each one's match lies
        nq = len(mov)
        if trial % 10 == 4:
            mov[: nq // 2] = mov[0]                          # many coinciding points: degenerate samples
        if trial % 10 == 7:
            mov, ref = mov[:4], ref[:4]
            nq = 4
        # train set: the matched points shuffled among decoys; the search result: nearest = the match, second = anything
        nt = nq + int(rng.integers(0, 300))
        perm = rng.permutation(nt)[:nq]
        train = rng.integers(0, 1500, (nt, 2)).astype(np.float64)
        train[perm] = ref
        idx = np.stack([perm, rng.integers(0, nt, nq)], 1).astype(np.int32)
        d1 = rng.uniform(0.5, 2.0, nq).astype(np.float32)
        good_share = 1.0 if big else rng.uniform(0.0, 1.0) if trial % 6 else 0.002
        d0 = np.where(rng.random(nq) < good_share, d1 * rng.uniform(0.01, 0.2499, nq), d1 * rng.uniform(0.2501, 1.0, nq)).astype(np.float32)
        if trial % 4 == 1:
            d0[::7] = (d1[::7] * np.float32(0.25))           # sqrt(d0) == 0.5 sqrt(d1) up to float32 rounding: the strict test
        dist_sq = np.stack([d0, d1], 1).astype(np.float32)
        dist = np.sqrt(dist_sq)
        good = np.nonzero(dist[:, 0] < 0.5 * dist[:, 1])[0]
        mat, n_good, status = ctx.match_similarity(ctx._upload_raw(idx), ctx._upload_raw(dist_sq),
                                                   ctx._upload_raw(mov.astype(np.float64)), ctx._upload_raw(train))
        assert n_good == len(good), trial
        if len(good) < 3:
            assert status == 1 and mat is None
            continue
        exp, _ = SP.estimate_affine_partial_2d(mov[good], train[idx[good, 0]].astype(np.float32))
        if exp is None:
            assert status == 2 and mat is None, trial
        else:
            assert status == 0 and np.array_equal(mat, exp), (trial, mat, exp)
    # coordinates that are not integer-valued are left to the host statement
    mov = rng.random((50, 2)) * 100
    idx = np.stack([np.arange(50), np.arange(50)[::-1]], 1).astype(np.int32)
    dist_sq = np.stack([np.full(50, 0.01), np.ones(50)], 1).astype(np.float32)
    mat, n_good, status = ctx.match_similarity(ctx._upload_raw(idx), ctx._upload_raw(dist_sq), ctx._upload_raw(mov),
                                               ctx._upload_raw(mov + 3.0))
    assert (mat, n_good, status) == (None, 50, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
@pytest.mark.parametrize("zero", ["mov", "ref", "far"])
def test_feature_registrator_zero_images_fall_back_to_the_careful_path(zero, dtype):
    """register()'s fast path does not wait to learn whether a dog() input's max() is 0 (the reference's shortcut,
    feature_registrator.py:288-291): the flags are read at the gate's synchronisation points and a set one restarts the call
    in the careful mode.  An all-black moving image, an all-black reference, and a pair whose content the coarse estimate
    throws out of view: the same matrix and the same log as the careful mode alone, printed once."""
    import contextlib
    import io
    from microaligner_amd import FeatureRegistrator
    H, W = 900, 1000
    ref = synthetic.make_cells(H, W, seed=3, dtype=dtype)
    mov = O.warp_affine(ref, np.array([[1.0, 0, 9.0], [0, 1.0, -6.0]]))
    if zero == "mov":
        mov = np.zeros_like(ref)
    elif zero == "ref":
        ref = np.zeros_like(ref)
    else:
        mov = np.zeros_like(ref)
        mov[:40, :40] = ref[:40, :40]            # a corner of content: transforms may move it out of view

    def run(careful):
        f = FeatureRegistrator()
        f.num_pyr_lvl, f.num_iterations, f.tile_size = 2, 2, 500
        f._careful = careful
        f.ref_img, f.mov_img = ref, mov
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                T = f.register()
        except ValueError as e:
            # an all-zero image that is not uint8 comes out of dog() unchanged (:288-291) and FAST refuses it -- cv2.error in
            # the reference, ValueError here -- in either mode
            T = str(e)
        assert f._careful == careful
        return T, buf.getvalue()
    fast, log_fast = run(False)
    careful, log_careful = run(True)
    assert type(fast) is type(careful) and np.array_equal(fast, careful) and log_fast == log_careful
    assert log_fast.count("Pyramid factor") <= 2


@pytest.mark.gpu
def test_affine_init_then_optical_flow_refine():
    """BASELINE cfg5 in miniature: feature-based affine initialisation, then the optical-flow refinement on the
    affinely aligned image (the reference pipeline's two stages, __main__.py:257-286 then :398-433)."""
    from microaligner_amd import FeatureRegistrator, OptFlowRegistrator, Warper, transform_img_with_tmat
    H, W = 1200, 1300
    ref = synthetic.make_cells(H, W, seed=8)
    th = np.deg2rad(0.5)
    M = np.array([[np.cos(th), -np.sin(th), 11.0], [np.sin(th), np.cos(th), -8.0]])
    mov = O.warp_affine(ref, M)
    # a smooth non-linear residual on top of the affine part
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    m = np.stack([xs + 1.5 * np.sin(ys / 90.0), ys + 1.5 * np.cos(xs / 110.0)], -1)
    mov = O.remap(mov, m)
    freg = FeatureRegistrator()
    freg.verbose = False
    freg.num_pyr_lvl, freg.tile_size = 2, 500
    freg.ref_img, freg.mov_img = ref, mov
    T = freg.register()
    affine = transform_img_with_tmat(mov, (H, W), T)
    ofreg = OptFlowRegistrator()
    ofreg.verbose = False
    ofreg.num_pyr_lvl, ofreg.tile_size, ofreg.overlap, ofreg.use_full_res_img, ofreg.use_dog = 2, 400, 60, True, True
    ofreg.ref_img, ofreg.mov_img = ref, affine
    flow = ofreg.register()
    w = Warper()
    w.tile_size, w.overlap = 400, 60
    w.image, w.flow = affine, flow
    final = w.warp()
    # the sharded driver composes the same two stages
    from microaligner_amd import parallel
    (final2, T2, flow2), = parallel.align_pairs([(ref, mov)], dict(num_pyr_lvl=2, tile_size=500),
                                                dict(num_pyr_lvl=2, tile_size=400, overlap=60, use_full_res_img=True,
                                                     use_dog=True))
    assert np.array_equal(final2, final) and np.array_equal(T2, T) and np.array_equal(flow2, flow)
    # ... and, for more than one pair, runs them through the three-engine stream (device arrays end to end): same bits
    of_params = dict(num_pyr_lvl=2, tile_size=400, overlap=60, use_full_res_img=True, use_dog=True)
    streamed = parallel.align_pairs([(ref, mov), (ref, mov[::-1].copy()), (ref, mov)], dict(num_pyr_lvl=2, tile_size=500), of_params)
    assert len(streamed) == 3
    for final3, T3, flow3 in (streamed[0], streamed[2]):
        assert np.array_equal(final3, final) and np.array_equal(T3, T) and np.array_equal(flow3, flow)
    # ... and with three tiles IN FLIGHT (a context and a host thread per lane: what bench.py's variants.tile_lanesN times for
    # BASELINE cfg5; FeatureRegistrator's fused rounds, the cached descriptor tables and the pools are per context): same bits
    laned = parallel.align_pairs([(ref, mov), (ref, mov[::-1].copy()), (ref, mov), (ref, mov[::-1].copy()), (ref, mov)],
                                 dict(num_pyr_lvl=2, tile_size=500), of_params, lanes=3)
    assert len(laned) == 5
    for k in (0, 2, 4):
        assert np.array_equal(laned[k][0], final) and np.array_equal(laned[k][1], T) and np.array_equal(laned[k][2], flow)
    for k in (1, 3):
        assert all(np.array_equal(a, b) for a, b in zip(laned[k], streamed[1]))
    inner = (slice(120, -120), slice(120, -120))
    err = [np.abs(a[inner].astype(np.float64) - ref[inner]).mean() for a in (mov, affine, final)]
    assert err[1] < 0.5 * err[0] and err[2] < 0.8 * err[1]


FEATURE_INDEX = __import__("json").load(open(__import__("os").path.join(
    __import__("os").path.dirname(__file__), "golden", "feature_index.json")))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FEATURE_INDEX))
def test_feature_registrator_reproduces_the_reference_orchestration(name):
    """Fixtures made by the REFERENCE's FeatureRegistrator class driven over the oracle / sparse_cpu primitives
    (tests/golden/make_feature_golden.py): same levels, same match counts, same gate decisions, same matrix."""
    import contextlib
    import io
    import re
    from microaligner_amd import FeatureRegistrator
    case = FEATURE_INDEX[name]
    H, W = case["shape"]
    dt = np.dtype(case["dtype"])
    ref = synthetic.make_cells(H, W, seed=case["seed"], dtype=dt)
    if case["M"] is None:
        mov = synthetic.make_cells(H, W, seed=case["seed"] + 1000, dtype=dt)
    else:
        th, sc, tx, ty = case["M"]
        th = np.deg2rad(th)
        mov = O.warp_affine(ref, np.array([[sc * np.cos(th), -sc * np.sin(th), tx], [sc * np.sin(th), sc * np.cos(th), ty]]))
    freg = FeatureRegistrator()
    for k, v in case["params"].items():
        setattr(freg, k, v)
    freg.ref_img, freg.mov_img = ref, mov
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        T = freg.register()
    log = buf.getvalue()
    assert [int(f) for f in re.findall(r"Pyramid factor (\d+)", log)] == case["factors"]
    assert [[int(a), int(b)] for a, b in re.findall(r"Good matches (\d+) / (\d+)", log)] == case["good_matches"]
    assert [("Better" in ln) for ln in log.splitlines() if "alignment than before" in ln] == case["accepted"]
    mi = [(float(a), float(b)) for a, b in re.findall(r"MI score after: (\S+) \| MI score before: (\S+)", log)]
    assert np.allclose(mi, case["mi"], rtol=0, atol=1e-12)
    assert np.allclose(T, np.array(case["t_mat"]), rtol=0, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [301, 308, 317])
def test_feature_registrator_device_path_equals_the_host_statement(seed):
    """End to end: register() with the feature stage on the device (ma_feature_extract, ma_knn2_l2 with the split-float16
    shortlist, ma_match_similarity) against the same registrator with `features_on_host` -- FAST, DAISY, the sequential exact
    2-NN, ratio test and RANSAC of feature_reg/sparse_cpu.py, the definition the kernels reproduce: the same matrix, bit for bit,
    and the same log (levels, match counts, scores, decisions).  tools/soak_feature.py runs 30 such configurations."""
    import contextlib
    import io
    from microaligner_amd import FeatureRegistrator
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(420, 760)), int(rng.integers(420, 760))
    dtype = [np.uint8, np.uint16, np.float32][seed % 3]
    ref = synthetic.make_cells(H, W, seed=seed, dtype=dtype)
    th = np.deg2rad(rng.uniform(-1.0, 1.0))
    M = np.array([[np.cos(th), -np.sin(th), rng.uniform(-12, 12)], [np.sin(th), np.cos(th), rng.uniform(-12, 12)]])
    if seed % 2:
        M = np.array([[1.0, 0.0, 7.0], [0.0, 1.0, -5.0]])         # exact integer shift: residuals exactly on the RANSAC threshold
    mov = O.warp_affine(ref, M)
    out = []
    for host in (False, True):
        f = FeatureRegistrator()
        f.num_pyr_lvl, f.num_iterations, f.tile_size, f.use_full_res_img = 1, 2, 300, bool(seed % 2)
        f.features_on_host = host
        f.ref_img, f.mov_img = ref, mov
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            T = f.register()
        out.append((T, buf.getvalue()))
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    assert "Good matches" in out[0][1]


@pytest.mark.gpu
def test_a_plain_c_host_drives_a_matching_round(ctx, tmp_path):
    """tests/c_host/feature_host.c: a C program that knows only include/microaligner_hip.h (gcc, linked against
    libmicroaligner_hip.so) extracts the features of two images, searches the two nearest neighbours and fits the similarity --
    ma_feature_extract, ma_knn2_l2, ma_match_similarity with the built-in seed-0 random state -- and prints what the Python path
    (find_features_of_device_image, Context.knn2, Context.match_similarity) computes: the same counts, the same matrix."""
    import os
    import subprocess
    from microaligner_amd import _lib
    from microaligner_amd.feature_reg import feature_detection as FD
    from microaligner_amd.feature_reg.sparse_cpu import Daisy
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "feature_host"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c_host", "feature_host.c"), "-o", str(exe), "-L", libdir,
                    "-lmicroaligner_hip", f"-Wl,-rpath,{libdir}"], check=True, capture_output=True, text=True)
    H, W, tile = 700, 820, 300
    ref = synthetic.make_cells(H, W, seed=21)
    th = np.deg2rad(0.4)
    mov = O.warp_affine(ref, np.array([[np.cos(th), -np.sin(th), 6.0], [np.sin(th), np.cos(th), -4.0]]))
    dref, dmov = O.dog(ref, True), O.dog(mov, True)
    dref.tofile(tmp_path / "ref.bin")
    dmov.tofile(tmp_path / "mov.bin")
    halves, cos_sin, offsets = FD._daisy_tables(Daisy(radius=21, q_radius=3, q_theta=8, q_hist=8))
    with open(tmp_path / "tables.bin", "wb") as f:
        f.write(np.array([len(h) - 1 for h in halves], np.int32).tobytes())
        for h in halves:
            f.write(np.ascontiguousarray(h, np.float64).tobytes())
        f.write(np.ascontiguousarray(cos_sin, np.float64).tobytes())
        f.write(np.ascontiguousarray(offsets, np.float64).tobytes())
    r = subprocess.run([str(exe), str(H), str(W), str(tile), str(tmp_path / "ref.bin"), str(tmp_path / "mov.bin"),
                        str(tmp_path / "tables.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = r.stdout.split()
    fr = FD.find_features_of_device_image(ctx.asdevice(dref), tile, ctx)
    fm = FD.find_features_of_device_image(ctx.asdevice(dmov), tile, ctx)
    idx_d, dist_d = ctx.knn2(fm.descriptors_for_search, fr.descriptors_for_search, on_device=True)
    mat, n_good, status = ctx.match_similarity(idx_d, dist_d, fm.device_pts(ctx), fr.device_pts(ctx))
    assert [int(v) for v in got[:4]] == [len(fr.descriptors_for_search), len(fm.descriptors_for_search), n_good, status]
    assert status == 0 and n_good > 100
    assert np.array_equal(np.array([float(v) for v in got[4:]]).reshape(2, 3), mat)


@pytest.mark.gpu
@pytest.mark.parametrize("use_dog,dtype", [(True, np.uint16), (False, np.uint8), (True, np.float32)])
def test_fused_round_equals_the_step_by_step_entry_points(use_dog, dtype):
    """ma_feature_round chains the calls the level loop otherwise makes one by one: the same matrix and the same log with
    `fuse_rounds` on and off, with and without the DOG preprocess, for an accepted and for a rejected pair."""
    import contextlib
    import io
    from microaligner_amd import FeatureRegistrator
    H, W = 640, 700
    ref = synthetic.make_cells(H, W, seed=33, dtype=dtype)
    th = np.deg2rad(-0.5)
    pairs = [O.warp_affine(ref, np.array([[np.cos(th), -np.sin(th), -8.0], [np.sin(th), np.cos(th), 5.0]])),
             synthetic.make_cells(H, W, seed=1033, dtype=dtype)]
    for mov in pairs:
        out = []
        for fuse in (True, False):
            f = FeatureRegistrator()
            f.num_pyr_lvl, f.num_iterations, f.tile_size, f.use_dog, f.use_full_res_img = 1, 3, 300, use_dog, True
            f.fuse_rounds = fuse
            f.ref_img, f.mov_img = ref, mov
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                T = f.register()
            out.append((T, buf.getvalue()))
        assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_replayed_rounds_equal_recomputed_rounds(dtype):
    """A round after a rejected one sees the same images and the same reference features, and every step of a round is
    deterministic: FeatureRegistrator replays its outcome instead of recomputing it (`skip_repeated_rounds`).  The same matrix and
    the same log, line for line, as with the replay off (every round recomputed), with the reference side on its own stream and
    without, for pairs with accepted rounds, with rejected ones, and for an unrelated pair (every round rejected)."""
    import contextlib
    import io
    from microaligner_amd import FeatureRegistrator
    H, W = 900, 1000
    ref = synthetic.make_cells(H, W, seed=41, dtype=dtype)
    th = np.deg2rad(0.4)
    pairs = [O.warp_affine(ref, np.array([[np.cos(th), -np.sin(th), 11.0], [np.sin(th), np.cos(th), -7.0]])),
             O.warp_affine(ref, np.array([[1.0, 0.0, 3.0], [0.0, 1.0, -2.0]])),
             synthetic.make_cells(H, W, seed=1041, dtype=dtype)]
    replays = 0
    for mov in pairs:
        out = []
        for skip, overlap in ((True, True), (False, True), (False, False)):
            f = FeatureRegistrator()
            f.num_pyr_lvl, f.num_iterations, f.tile_size = 2, 4, 400
            f.skip_repeated_rounds, f.overlap_reference = skip, overlap
            f.ref_img, f.mov_img = ref, mov
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                T = f.register()
            out.append((T, buf.getvalue()))
        assert all(np.array_equal(out[0][0], o[0]) and out[0][1] == o[1] for o in out[1:])
        lines = out[0][1].splitlines()
        replays += sum(1 for a, b in zip(lines, lines[4:]) if a.startswith("    Good matches") and a == b)
    assert replays >= 3        # the cases do contain repeated rounds
