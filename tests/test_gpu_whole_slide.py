"""-m gpu: whole-slide sizes.  The reference tiles anything (slicer.py:69-118, stitcher.py:72-118) and whole-slide cycles are
routinely 30-60 k pixels a side; BASELINE's largest configuration (16384^2) stays below every interesting limit.  Here:
a uint8 pair of 46300 x 46700 pixels -- both dimensions beyond 32767, 2.16 G pixels (more than 2^31: a 32-bit pixel
index overflows), a flow of 17.3 GB (beyond 4 GiB), 47 x 47 = 2209 windows of 1200^2 = 6 batches of the 48 GiB Farneback
workspace, 2163-chunk NMI gates -- through the tiled Farneback, the warp, the whole register() (C engine, 5 levels with
the full-resolution one) and the page-warp driver with uint16 pages of that size.

The oracle cannot run the whole image in test time, and does not have to: windows are independent, so the stitched
result restricted to one tile must equal the oracle run on that window alone.  Checked bit for bit on corner, interior,
edge and last windows (spread over all workspace batches) for the flow and for the warp; register() is held by
properties: every level accepted, the flow follows the synthetic displacement, the C engine and the Python loop (two
schedules of the same primitives with different buffer lifetimes) agree bit for bit, and a second run reproduces the
first.  Sizes can be reduced for a quick trial with MA_WHOLE_SLIDE="H,W".
"""
import hashlib
import os
import time

import numpy as np
import pytest

from oracle import register_oracle as RO
from microaligner_amd import OptFlowRegistrator, Warper, synthetic

H, W = (int(v) for v in os.environ.get("MA_WHOLE_SLIDE", "46300,46700").split(","))
TILE, OV = 1000, 100
def _mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 2 ** 20
    except OSError:
        pass
    return 0.0


# host arrays of the full-size case: images 2 x 2.2 GB, flow 17.3 GB (+ a page-locked twin), pages 4 x 4.3 GB, windows for the
# oracle: ~70 GB at the peak; never run where that could exhaust the machine
NEED_GB = 0.0 if "MA_WHOLE_SLIDE" in os.environ else 160.0
BIG_HOST = ((os.cpu_count() or 1) >= int(os.environ.get("MA_FULLSIZE_MIN_CORES", "64"))) and _mem_available_gb() >= NEED_GB
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not BIG_HOST, reason="the whole-slide case needs a host with >= 64 cores and >= 160 GB of free "
                                                      f"memory (this host: {os.cpu_count()} cores, {_mem_available_gb():.0f} GB)")]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def slide():
    """(ref, mov, base displacement) uint8: a 4 x 4 mosaic of one synthetic pair, every block rolled by its own offset so
    that no two blocks hold the same content (an indexing error by a whole block cannot hide), the same roll for ref and
    mov (the displacement field inside a block is the synthetic one)."""
    t0 = time.perf_counter()
    bh, bw = -(-H // 4), -(-W // 4)
    r, m = synthetic.make_pair(bh, bw, seed=11, dtype=np.uint8)
    ref, mov = np.empty((H, W), np.uint8), np.empty((H, W), np.uint8)
    for i in range(4):
        for j in range(4):
            ys, xs = slice(i * bh, min((i + 1) * bh, H)), slice(j * bw, min((j + 1) * bw, W))
            sh = (i * 1237 + j * 311, i * 173 + j * 911)
            ref[ys, xs] = np.roll(r, sh, (0, 1))[:ys.stop - ys.start, :xs.stop - xs.start]
            mov[ys, xs] = np.roll(m, sh, (0, 1))[:ys.stop - ys.start, :xs.stop - xs.start]
    print(f"\n[whole slide] {H} x {W} uint8 pair ({H * W / 1e9:.2f} Gpx) synthesised in {time.perf_counter() - t0:.1f} s")
    return ref, mov


def _window(arr, ty, tx):
    P = TILE + 2 * OV
    y0, x0 = ty * TILE - OV, tx * TILE - OV
    win = np.zeros((P, P) + arr.shape[2:], arr.dtype)
    ys, xs, ye, xe = max(y0, 0), max(x0, 0), min(y0 + P, arr.shape[0]), min(x0 + P, arr.shape[1])
    win[ys - y0:ye - y0, xs - x0:xe - x0] = arr[ys:ye, xs:xe]
    return win


def _picks():
    nty, ntx = -(-H // TILE), -(-W // TILE)
    n = nty * ntx
    # corner, interior windows spread over the workspace batches, right edge, bottom edge, last (ragged) window
    idx = sorted({0, n // 7, n // 3, n // 2, (2 * n) // 3 + 1, ntx - 1, (nty - 1) * ntx + ntx // 2, n - 1})
    return [(i // ntx, i % ntx) for i in idx]


def _valid(ty, tx):
    return min(TILE, H - ty * TILE), min(TILE, W - tx * TILE)


def test_whole_slide_farneback_and_warp_windows_equal_the_oracle(ctx, slide):
    ref, mov = slide
    assert H > 32767 and W > 32767 or "MA_WHOLE_SLIDE" in os.environ
    dref, dmov = ctx.asdevice(ref), ctx.asdevice(mov)
    ctx.sync()
    t0 = time.perf_counter()
    dflow = ctx.farneback(dmov, dref, 99, 3, tile=TILE, overlap=OV)
    ctx.sync()
    t1 = time.perf_counter()
    nwin = (-(-H // TILE)) * (-(-W // TILE))
    print(f"\n[whole slide] tiled Farneback, {nwin} windows: {t1 - t0:.2f} s; flow {dflow.nbytes / 2 ** 30:.1f} GiB")
    flow = dflow.numpy()
    assert flow.shape == (H, W, 2) and flow.nbytes == H * W * 8
    picks = _picks()
    prev = np.stack([_window(mov, ty, tx) for ty, tx in picks])
    nxt = np.stack([_window(ref, ty, tx) for ty, tx in picks])
    exp = RO.O.farneback_batch(prev, nxt, 99, 3, nthreads=len(picks))
    for (ty, tx), e in zip(picks, exp):
        vh, vw = _valid(ty, tx)
        got = flow[ty * TILE:ty * TILE + vh, tx * TILE:tx * TILE + vw]
        assert np.array_equal(got, e[OV:OV + vh, OV:OV + vw]), (ty, tx)
    t2 = time.perf_counter()
    dwarped = ctx.warp(dmov, dflow, TILE, OV)
    ctx.sync()
    print(f"[whole slide] warp: {time.perf_counter() - t2:.3f} s")
    warped = dwarped.numpy()
    for ty, tx in picks:
        im, fl = _window(mov, ty, tx), _window(flow, ty, tx)
        mp = np.negative(fl)
        mp[:, :, 0] += np.arange(mp.shape[1])
        mp[:, :, 1] += np.arange(mp.shape[0]).reshape(-1, 1)
        e = RO.O.remap(im, mp)
        vh, vw = _valid(ty, tx)
        assert np.array_equal(warped[ty * TILE:ty * TILE + vh, tx * TILE:tx * TILE + vw], e[OV:OV + vh, OV:OV + vw]), (ty, tx)
    # the page-warp driver over uint16 pages of this size, one resident flow (warp_and_save_pages, __main__.py:288-302)
    rng = np.random.default_rng(3)
    pages = [(mov.astype(np.uint16) * 257) ^ np.uint16(k * 4369) for k in range(2)]
    pages[1][::2] += rng.integers(0, 200, (1, W), dtype=np.uint16)
    w = Warper()
    w.tile_size, w.overlap = TILE, OV
    w.flow = dflow
    t3 = time.perf_counter()
    out = w.warp_pages(pages)
    dt = time.perf_counter() - t3
    print(f"[whole slide] page-warp driver: 2 uint16 pages of {pages[0].nbytes / 2 ** 30:.1f} GiB in {dt:.2f} s "
          f"({2 * H * W / dt / 1e9:.2f} Gpix/s host to host)")
    for page, o in zip(pages, out):
        assert o.dtype == np.uint16 and o.shape == (H, W)
        for ty, tx in picks[::2]:
            fl = _window(flow, ty, tx)
            mp = np.negative(fl)
            mp[:, :, 0] += np.arange(mp.shape[1])
            mp[:, :, 1] += np.arange(mp.shape[0]).reshape(-1, 1)
            e = RO.O.remap(_window(page, ty, tx), mp)
            vh, vw = _valid(ty, tx)
            assert np.array_equal(o[ty * TILE:ty * TILE + vh, tx * TILE:tx * TILE + vw], e[OV:OV + vh, OV:OV + vw]), (ty, tx)


def test_whole_slide_register_properties(ctx, slide):
    ref, mov = slide
    params = dict(num_pyr_lvl=4, use_full_res_img=True, use_dog=False)
    dref, dmov = ctx.asdevice(ref), ctx.asdevice(mov)
    out = {}
    for engine in ("c", "python", "c"):
        reg = OptFlowRegistrator()
        reg.verbose = False
        for k, v in dict(params, engine=engine).items():
            setattr(reg, k, v)
        reg.ref_img, reg.mov_img = dref, dmov
        ctx.sync()
        t0 = time.perf_counter()
        dflow = reg.register()
        ctx.sync()
        dt = time.perf_counter() - t0
        rep = [(r.factor, r.shape, r.mi_after, r.mi_before, r.accepted) for r in reg.level_reports]
        print(f"\n[whole slide] register() engine={engine}: {dt:.2f} s ({H * W / dt / 1e6:.0f} Mpix/s); levels "
              f"{[(r[0], r[4]) for r in rep]}")
        flow = dflow.numpy()
        del dflow
        out.setdefault(engine, []).append((sha(flow), rep))
        if engine == "c" and len(out["c"]) == 1:
            assert flow.shape == (H, W, 2)
            assert [r[0] for r in rep] == [16, 8, 4, 2, 1]
            if "MA_WHOLE_SLIDE" in os.environ:
                continue                     # a reduced trial size: the blocks are too small for the properties below
            assert all(r[4] for r in rep)
            # inside one block of the mosaic, away from its seams, the flow is the synthetic displacement of that block
            bh, bw = -(-H // 4), -(-W // 4)
            i, j = 2, 3
            sh = (i * 1237 + j * 311, i * 173 + j * 911)
            y0, x0 = i * bh + sh[0] + 600, j * bw + sh[1] + 600          # block-local row sh[0] + 600 ... is base row 600 ...
            dx, dy = synthetic.displacement(bh, bw)
            n = 1500
            blk = flow[y0:y0 + n, x0:x0 + n]
            exp = np.stack([dx[600:600 + n, 600:600 + n], dy[600:600 + n, 600:600 + n]], -1)
            err = float(np.abs(blk - exp).mean())
            print(f"[whole slide] mean |flow - synthetic displacement| inside block ({i}, {j}): {err:.3f} px")
            assert err < 0.35
        del flow
    assert out["c"][0] == out["c"][1], "a second run does not reproduce the first"
    assert out["c"][0] == out["python"][0], "the C engine and the Python level loop disagree at this size"
