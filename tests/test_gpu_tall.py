"""-m gpu: an image with more than 65535 rows.  gridDim.y ends at 65535; every kernel of the path that puts rows on the y axis
takes several rows per block, so the image height is bounded by 65535 x rows-per-block (524 280 for the warp), not by
65535 -- the reference has no bound at all (slicer.py:69-118 tiles anything).  A strip of 70 001 x 1 203 pixels (84 Mpx:
71 windows at the default tiling, the last 5 of them entirely beyond row 65535) through every primitive of the path and
through the whole register() + warp(), bit for bit against the oracle on the whole image."""
import os

import numpy as np
import pytest

from conftest import oracle_threads

from oracle import oracle as O
from oracle import register_oracle as RO
from microaligner_amd import OptFlowRegistrator, Warper, synthetic

H, W = 70001, 1203
TILE, OV = 1000, 100
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif((os.cpu_count() or 1) < 16, reason="the oracle side of the tall-strip case wants >= 16 cores")]


@pytest.fixture(scope="module")
def strip():
    """(ref, mov) uint8: a synthetic pair of 5 000 rows stacked 15 times, every copy rolled by its own offset (an indexing
    error by a whole copy cannot hide), the same roll for ref and mov."""
    bh = 5000
    r, m = synthetic.make_pair(bh, W, seed=23, dtype=np.uint8)
    ref, mov = np.empty((H, W), np.uint8), np.empty((H, W), np.uint8)
    for i in range(-(-H // bh)):
        ys = slice(i * bh, min((i + 1) * bh, H))
        sh = (i * 337, i * 59)
        ref[ys] = np.roll(r, sh, (0, 1))[:ys.stop - ys.start]
        mov[ys] = np.roll(m, sh, (0, 1))[:ys.stop - ys.start]
    return ref, mov


def test_primitives_on_a_strip_taller_than_the_grid_limit(ctx, strip):
    ref, mov = strip
    assert H > 65535
    dref, dmov = ctx.asdevice(ref), ctx.asdevice(mov)
    # pyramid down (u8) and up (flow), both crossing row 65535 on the large side
    down = ctx.pyr_down(dmov)
    assert np.array_equal(down.numpy(), O.pyr_down(mov))
    rng = np.random.default_rng(2)
    small = rng.standard_normal(((H + 1) // 2, (W + 1) // 2, 2)).astype(np.float32)
    up = ctx.pyr_up_flow(ctx.asdevice(small), (H, W), 2.0)
    assert np.array_equal(up.numpy(), O.pyr_up(small * np.float32(2.0), dstsize=(W, H)))
    # dog
    assert np.array_equal(ctx.dog_u8(dmov).numpy(), O.dog(mov))
    # tiled Farneback, warp, merge: whole image against the oracle
    flow = ctx.farneback(dmov, dref, 51, 2, tile=TILE, overlap=OV)
    exp_flow = RO.tile_flow(ref, mov, TILE, OV, 51, 2, nthreads=oracle_threads())
    got_flow = flow.numpy()
    assert np.array_equal(got_flow, exp_flow)
    assert np.abs(got_flow[65536:]).max() > 0                      # the rows beyond the old bound carry a real flow
    assert np.array_equal(ctx.warp(dmov, flow, TILE, OV).numpy(), RO.warp(mov, exp_flow, TILE, OV))
    other = ctx.asdevice((exp_flow[::-1] * np.float32(0.5)).copy())
    assert np.array_equal(ctx.merge_flows(flow, other, TILE, OV).numpy(),
                          RO.merge_flows(exp_flow, (exp_flow[::-1] * np.float32(0.5)).copy(), TILE, OV))
    # the page-warp driver with a band boundary beyond row 65535 (uint16 pages)
    pages = [(mov.astype(np.uint16) * 257) ^ np.uint16(k * 4369) for k in range(2)]
    w = Warper()
    w.tile_size, w.overlap = TILE, OV
    w.flow = flow
    for p, o in zip(pages, w.warp_pages(pages)):
        assert np.array_equal(o, RO.warp(p, exp_flow, TILE, OV))


def test_register_and_warp_on_a_strip_taller_than_the_grid_limit(ctx, strip):
    ref, mov = strip
    params = dict(num_pyr_lvl=2, use_full_res_img=True, use_dog=True, tile_size=TILE, overlap=OV, num_iterations=2)
    exp, reports = RO.register(ref, mov, nthreads=oracle_threads(), **params)
    got = {}
    for engine in ("c", "python"):
        reg = OptFlowRegistrator()
        reg.verbose = False
        for k, v in dict(params, engine=engine).items():
            setattr(reg, k, v)
        reg.ref_img, reg.mov_img = ref, mov
        got[engine] = reg.register()
        assert [r.accepted for r in reg.level_reports] == [r[3] for r in reports], engine
        assert np.array_equal(got[engine], exp), engine
    assert any(r[3] for r in reports)
    w = Warper()
    w.tile_size, w.overlap = TILE, OV
    w.image, w.flow = mov, got["c"]
    assert np.array_equal(w.warp(), RO.warp(mov, exp, TILE, OV))
