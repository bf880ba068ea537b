"""The oracle's restatement of register()/warp() (oracle/register_oracle.py) against fixtures produced by the
REFERENCE's own classes driven over the same C primitives (tests/golden/make_golden.py).  Bit-exact."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import register_oracle as RO
from microaligner_amd import synthetic

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
INDEX = json.load(open(os.path.join(GOLDEN, "index.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def golden_inputs(case):
    H, W = case["shape"]
    make = synthetic.make_unrelated_pair if case["unrelated"] else synthetic.make_pair
    return make(H, W, case["seed"], case["dtype"])


@pytest.mark.parametrize("name", sorted(INDEX))
def test_oracle_register_reproduces_reference_orchestration(name):
    case = INDEX[name]
    ref, mov = golden_inputs(case)
    flow, reports = RO.register(ref, mov, **case["params"])
    s5 = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert [r[0] for r in reports] == case["factors"]
    assert [r[3] for r in reports] == case["accepted"]
    np.testing.assert_allclose([(r[1], r[2]) for r in reports], case["mi"], rtol=0, atol=1e-15)
    assert flow.dtype == np.float32 and list(flow.shape) == case["flow_shape"]
    np.testing.assert_array_equal(flow[::5, ::5], s5["flow_s5"])
    assert sha(flow) == case["flow_sha256"]
    tile, ov = case["params"].get("tile_size", 1000), case["params"].get("overlap", 100)
    warped = RO.warp(mov, flow, tile, ov)
    np.testing.assert_array_equal(warped[::5, ::5], s5["warped_s5"])
    assert sha(warped) == case["warped_sha256"]
    mov16 = synthetic._cast(synthetic.make_pair(*case["shape"], case["seed"], np.float32)[1], np.uint16)
    assert sha(RO.warp(mov16, flow, tile, ov)) == case["warped_u16_sha256"]


def test_goldens_cover_every_branch():
    acc = [tuple(c["accepted"]) for c in INDEX.values()]
    assert any(a[0] is False for a in acc)                                   # level-0 reject -> zeros
    assert any(len(a) == 3 and a[0] and not a[1] for a in acc)               # middle reject (x4, Q3)
    assert any(len(a) == 3 and not a[2] for a in acc)                        # last reject, full res
    assert any(not c["params"].get("use_full_res_img", False) and not c["accepted"][-1] for c in INDEX.values())
    assert any(len(a) == 1 for a in acc)                                     # single level
    assert any(c["dtype"] == "uint8" for c in INDEX.values())


def test_split_stitch_roundtrip_and_info():
    rng = np.random.default_rng(0)
    a = rng.random((250, 230)).astype(np.float32)
    tiles, grid = RO.split_tiles(a, 100, 10)
    assert grid == (3, 3) and all(t.shape == (120, 120) for t in tiles)
    assert np.array_equal(RO.stitch_tiles(tiles, grid, a.shape, 100, 10), a)
    f = rng.random((130, 90, 2)).astype(np.float32)
    tiles, grid = RO.split_tiles(f, 64, 7)
    assert np.array_equal(RO.stitch_tiles(tiles, grid, f.shape, 64, 7), f)
