"""CPU oracle of the orchestration: OptFlowRegistrator.register() + Warper.warp() restated with
numpy on top of the C oracle primitives (oracle/oracle.py).

TEST INFRASTRUCTURE, not product code.  Each function cites the reference lines it follows.
Pinned by tests/golden/*.npz, which were produced by driving the reference's own classes
(imported from /root/reference in the build container) over the same C primitives through a
`cv2` stand-in -- see tests/golden/make_golden.py.  That pins tile geometry, level logic and the
quirks Q1-Q3; the OpenCV arithmetic itself stays unpinned (ma_oracle.c header).
"""
import time
from contextlib import contextmanager

import numpy as np

from . import oracle as O


@contextmanager
def _stage(acc, name):
    """Adds the wall time of the block to acc[name] (acc may be None): the per-stage seconds of the CPU baseline."""
    t0 = time.perf_counter()
    try:
        yield
    finally:
        if acc is not None:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0


# ---- slicer.py:23-118 / stitcher.py:25-118 --------------------------------------------------------
def split_tiles(arr, tile, overlap):
    """Zero-padded (tile+2*overlap)^2 windows in row-major tile order + (ny, nx)."""
    H, W = arr.shape[:2]
    ny, nx = -(-H // tile), -(-W // tile)
    P = tile + 2 * overlap
    tiles = []
    for ty in range(ny):
        for tx in range(nx):
            y0, x0 = ty * tile - overlap, tx * tile - overlap
            win = np.zeros((P, P) + arr.shape[2:], arr.dtype)
            ys, xs = max(y0, 0), max(x0, 0)
            ye, xe = min(y0 + P, H), min(x0 + P, W)
            win[ys - y0:ye - y0, xs - x0:xe - x0] = arr[ys:ye, xs:xe]
            tiles.append(win)
    return tiles, (ny, nx)


def stitch_tiles(tiles, grid, shape, tile, overlap):
    """Keep the centre [overlap, overlap+valid) of every window (stitcher.py:62-65,115)."""
    ny, nx = grid
    H, W = shape[:2]
    out = np.zeros(tuple(shape[:2]) + tiles[0].shape[2:], tiles[0].dtype)
    for ty in range(ny):
        for tx in range(nx):
            vh, vw = min(tile, H - ty * tile), min(tile, W - tx * tile)
            out[ty * tile:ty * tile + vh, tx * tile:tx * tile + vw] = \
                tiles[ty * nx + tx][overlap:overlap + vh, overlap:overlap + vw]
    return out


# ---- flow_calc.py:59-98 -------------------------------------------------------------------------------
def tile_flow(ref, mov, tile, overlap, win_size, num_iter, fused=False, nthreads=1):
    if ref.dtype != mov.dtype:
        # cv2.calcOpticalFlowFarneback converts prev and next to CV_32F separately (optflowgf.cpp): a mixed pair is the
        # float32 pair (exact for uint8 / uint16)
        ref, mov = ref.astype(np.float32), mov.astype(np.float32)
    if max(ref.shape) / tile < 2:
        return O.calc_optical_flow_farneback(mov, ref, win_size, num_iter, fused=fused)
    rt, grid = split_tiles(ref, tile, overlap)
    mt, _ = split_tiles(mov, tile, overlap)
    flows = O.farneback_batch(np.stack(mt), np.stack(rt), win_size, num_iter, fused=fused, nthreads=nthreads)
    return stitch_tiles(list(flows), grid, ref.shape, tile, overlap)


# ---- warper.py:37-76 ---------------------------------------------------------------------------------
def warp(img, flow, tile, overlap):
    it, grid = split_tiles(img, tile, overlap)
    ft, _ = split_tiles(flow, tile, overlap)
    out = []
    for im, fl in zip(it, ft):
        h, w = fl.shape[:2]
        m = np.negative(fl)
        m[:, :, 0] += np.arange(w)           # warper.py:58 (float32 += int64 -> via float64)
        m[:, :, 1] += np.arange(h).reshape(-1, 1)
        out.append(O.remap(im, m))
    return stitch_tiles(out, grid, img.shape, tile, overlap)


# ---- optflow_registrator.py:37-47, 217-240 -----------------------------------------------------------
def merge_two_flows(f1, f2):
    if f1.max() == 0:
        return f2
    if f2.max() == 0:
        return f1
    return f1 + O.remap(f2, -f1)


def merge_flows(f1, f2, tile, overlap):
    t1, grid = split_tiles(f1, tile, overlap)
    t2, _ = split_tiles(f2, tile, overlap)
    return stitch_tiles([merge_two_flows(a, b) for a, b in zip(t1, t2)], grid, f1.shape, tile, overlap)


# ---- similarity_scoring.py:27-68 ----------------------------------------------------------------------
def mi_tiled(a, b, tile):
    if a.dtype != np.uint8 or b.dtype != np.uint8:
        # dog() returned an image unchanged (max() == 0, optflow_registrator.py:256-257): the reference hands the raw
        # values to scikit-learn as labels (similarity_scoring.py:36,44); so does the oracle
        from sklearn.metrics import normalized_mutual_info_score
        fa, fb = np.ravel(a), np.ravel(b)
        if max(a.shape) / tile < 2:
            return normalized_mutual_info_score(fa, fb)
        chunk = tile * tile
        return np.mean([normalized_mutual_info_score(fa[i:i + chunk], fb[i:i + chunk]) for i in range(0, fa.size, chunk)])
    if max(a.shape) / tile < 2:
        return O.nmi_u8(a, b)
    return np.mean(O.nmi_u8_chunks(a, b, tile * tile))


# ---- optflow_registrator.py:175-215 -------------------------------------------------------------------
def image_pyramid(img, num_pyr_lvl, use_full_res_img):
    if num_pyr_lvl < 0 or (num_pyr_lvl == 0 and not use_full_res_img):
        raise ValueError("bad pyramid parameters")
    levels, factors, cur = [], [], img
    for lvl in range(num_pyr_lvl):
        f = 2 ** (lvl + 1)
        if img.shape[0] / f < 100 or img.shape[1] / f < 100:
            break
        cur = O.pyr_down(cur)
        levels.append(cur)
        factors.append(f)
    levels, factors = levels[::-1], factors[::-1]
    if use_full_res_img:
        levels.append(img)
        factors.append(1)
    return levels, factors


def upscale_to_full(flow, factor, full_shape):
    if abs(flow.shape[0] - full_shape[0]) <= 1 or factor < 2:
        return flow
    return O.pyr_up(flow, dstsize=full_shape[::-1])  # the ORIGINAL flow, once, no x2 (Q2)


# ---- optflow_registrator.py:93-173 --------------------------------------------------------------------
def register(ref, mov, num_pyr_lvl=4, num_iterations=3, tile_size=1000, overlap=100, use_full_res_img=False,
             use_dog=False, fused=False, nthreads=1, stage_seconds=None, dog_flags=0):
    """Returns (flow, reports); reports = [(factor, mi_after, mi_before, accepted), ...].
    stage_seconds: optional dict that receives the wall seconds per stage (pyramid, dog, farneback, warp, nmi,
    merge_pyrup).  fused: window blur of Farneback with FMA; dog_flags: rounding model of the dog() chain
    (oracle.DOG_FUSED_*)."""
    win = overlap - (1 - overlap % 2)
    O.set_threads(nthreads)  # rows / NMI chunks / Farneback windows fan out over host threads; same bits for any count
    T = stage_seconds
    with _stage(T, "pyramid"):
        ref_pyr, factors = image_pyramid(ref, num_pyr_lvl, use_full_res_img)
        mov_pyr, _ = image_pyramid(mov, num_pyr_lvl, use_full_res_img)
    n = len(factors)
    reports, m_flow = [], None
    for lvl, factor in enumerate(factors):
        last = lvl == n - 1
        mov_lvl = mov_pyr[lvl].copy()
        if lvl > 0:
            with _stage(T, "warp"):
                mov_lvl = warp(mov_lvl, m_flow, tile_size, overlap)
        with _stage(T, "dog"):
            fb_ref, fb_mov = O.dog(ref_pyr[lvl], use_dog, flags=dog_flags), O.dog(mov_lvl, use_dog, flags=dog_flags)
        with _stage(T, "farneback"):
            this_flow = tile_flow(fb_ref, fb_mov, tile_size, overlap, win, num_iterations, fused=fused,
                                  nthreads=nthreads)
        del fb_ref, fb_mov
        with _stage(T, "warp"):
            warped = warp(mov_lvl, this_flow, tile_size, overlap)
        with _stage(T, "dog"):
            ref_d, warped_d, raw_d = (O.dog(ref_pyr[lvl], True, flags=dog_flags), O.dog(warped, True, flags=dog_flags),
                                     O.dog(mov_pyr[lvl], True, flags=dog_flags))
        with _stage(T, "nmi"):
            after = mi_tiled(ref_d, warped_d, tile_size)
            before = mi_tiled(ref_d, raw_d, tile_size)
        del ref_d, warped_d, raw_d, warped
        ok = bool(after > before)
        reports.append((factor, float(after), float(before), ok))
        nxt = None if last else mov_pyr[lvl + 1].shape
        t_mp = time.perf_counter()
        if ok:
            if lvl == 0:
                m_flow = O.pyr_up(this_flow * 2, dstsize=nxt[::-1]) if not last else \
                    upscale_to_full(this_flow, factor, ref.shape)
            else:
                merged = merge_flows(m_flow, this_flow, tile_size, overlap)
                if last:
                    m_flow = merged if use_full_res_img else upscale_to_full(merged, factor, ref.shape)
                else:
                    m_flow = O.pyr_up(merged * 2, dstsize=nxt[::-1])
        else:
            if lvl == 0:
                m_flow = np.zeros(tuple(nxt if not last else ref.shape) + (2,), np.float32)
            elif last:
                if not use_full_res_img:
                    m_flow = O.pyr_up(m_flow * 2, dstsize=ref.shape[::-1])
            else:
                m_flow = O.pyr_up(m_flow * 4, dstsize=nxt[::-1])  # sic, Q3
        if T is not None:
            T["merge_pyrup"] = T.get("merge_pyrup", 0.0) + time.perf_counter() - t_mp
    return m_flow, reports
