"""Host-side logic that needs no GPU: API surface, validation, tile geometry, synthetic inputs."""
import numpy as np
import pytest

import microaligner_amd as ma
from microaligner_amd.shared_modules.tiling import TileGrid, is_tiled
from microaligner_amd.shared_modules.utils import pad_to_shape
from microaligner_amd import synthetic


def test_api_surface_and_defaults_match_the_reference():
    # microaligner/__init__.py:18-20, optflow_registrator.py:54-59, warper.py:30-35, flow_calc.py:50-57
    reg = ma.OptFlowRegistrator()
    assert (reg.num_pyr_lvl, reg.num_iterations, reg.tile_size, reg.overlap) == (4, 3, 1000, 100)
    assert reg.use_full_res_img is False and reg.use_dog is False
    for name in ("register", "dog", "get_dog_sigmas", "ref_img", "mov_img"):
        assert hasattr(reg, name)
    w = ma.Warper()
    assert (w.tile_size, w.overlap) == (1000, 100) and hasattr(w, "warp")
    fc = ma.TileFlowCalc()
    assert (fc.num_iter, fc.win_size, fc.tile_size, fc.overlap) == (1, 51, 1000, 100)
    assert reg.get_dog_sigmas(1) == (5, 9) and reg.get_dog_sigmas(32) == (1, 2)


def test_validation_errors():
    reg = ma.OptFlowRegistrator()
    with pytest.raises(ValueError, match="2D grayscale"):
        reg.ref_img = np.zeros((4, 4, 3))
    with pytest.raises(ValueError, match="2D grayscale"):
        reg.mov_img = np.zeros((4, 4, 3))
    with pytest.raises(ValueError, match="No ref image"):
        reg.register()
    reg.ref_img = np.zeros((200, 200), np.float32)
    with pytest.raises(ValueError, match="No mov image"):
        reg.register()
    reg.mov_img = np.zeros((200, 201), np.float32)
    with pytest.raises(ValueError, match="different dimensions"):
        reg.register()
    with pytest.raises(ValueError, match="No image"):
        ma.Warper().warp()


def test_window_size_rule():
    reg = ma.OptFlowRegistrator()
    for ov, win in ((100, 99), (99, 99), (20, 19), (21, 21), (10, 9)):
        reg.overlap = ov
        reg._init_tile_flow_calc()
        assert reg._tile_flow_calc.win_size == win


def test_tile_grid_matches_slicer_info():
    g = TileGrid(2500, 2300, 1000, 100)
    assert g.slicer_info() == {"tile_shape": [1000, 1000], "ntiles": {"x": 3, "y": 3}, "overlap": 100,
                               "padding": {"left": 0, "right": 700, "top": 0, "bottom": 500}}
    assert g.window == 1200 and g.ntiles == 9 and g.padded_pixels == 9 * 1200 * 1200
    assert list(g.origins())[:4] == [(-100, -100), (-100, 900), (-100, 1900), (900, -100)]
    # SURVEY Appendix B
    for size, n in ((2048, 9), (4096, 25), (8192, 81), (16384, 289)):
        assert TileGrid(size, size, 1000, 100).ntiles == n
    assert not is_tiled((1024, 1024), 1000) and is_tiled((2048, 100), 1000) and not is_tiled((1999, 5), 1000)


def test_pad_to_shape():
    img = np.ones((5, 6), np.uint8)
    out, pad = pad_to_shape(img, (5, 6))
    assert out is img and pad == (0, 0, 0, 0)
    out, pad = pad_to_shape(img, (8, 7))
    assert out.shape == (8, 7) and pad == (0, 1, 1, 2) and out.sum() == 30


def test_synthetic_pair_is_seeded_and_displaced():
    a1, b1 = synthetic.make_pair(120, 130, 7)
    a2, b2 = synthetic.make_pair(120, 130, 7)
    assert np.array_equal(a1, a2) and np.array_equal(b1, b2)
    assert a1.dtype == np.float32 and a1.min() >= 0 and a1.max() <= 255
    a3, _ = synthetic.make_pair(120, 130, 8)
    assert not np.array_equal(a1, a3)
    u8, _ = synthetic.make_pair(64, 64, 1, np.uint8)
    u16, _ = synthetic.make_pair(64, 64, 1, np.uint16)
    assert u8.dtype == np.uint8 and u16.dtype == np.uint16
    # banded generation is seam-free
    _, m1 = synthetic.make_pair(100, 90, 3, band=2048)
    _, m2 = synthetic.make_pair(100, 90, 3, band=17)
    assert np.array_equal(m1, m2)


def test_lanes_run_every_unit_once_under_their_own_context(monkeypatch):
    """parallel.run_sharded(lanes=L): L threads, each with its own context installed as the thread's current one."""
    import threading
    from microaligner_amd import device, parallel

    class FakeCtx:
        made, synced = {}, []

        def __init__(self, k):
            self.k = k

        def sync(self):
            FakeCtx.synced.append(self.k)

    def lane_context(k):     # lane contexts are cached: the same lane index yields the same context
        return FakeCtx.made.setdefault(k, FakeCtx(k))

    monkeypatch.setattr(parallel, "_lane_context", lane_context)
    seen = {}

    def fn(u):
        seen[u] = (threading.current_thread().name, device.get_context())
        return u * u

    out = parallel.run_sharded(list(range(11)), fn, gather=True, lanes=3)
    assert out == [u * u for u in range(11)]
    assert sorted(FakeCtx.made) == [0, 1, 2] and sorted(FakeCtx.synced) == [0, 1, 2]
    assert all(c is FakeCtx.made[c.k] for _, c in seen.values())
    assert all(name.startswith("ma-lane-") and isinstance(c, FakeCtx) for name, c in seen.values())
    assert getattr(device._tls, "ctx", None) is None            # nothing leaks onto the calling thread

    def bad(u):
        if u == 4:
            raise RuntimeError("boom")
        return u

    with pytest.raises(RuntimeError, match="boom"):
        parallel.run_sharded(list(range(8)), bad, lanes=2)


def test_resident_cache_recognises_unmodified_host_arrays_only():
    """The host <-> device pair cache behind Context.asdevice() (no device needed: a stand-in device array): same memory
    and same bytes hit, edits and recycled or dead memory miss, small arrays are never recorded, the byte bound evicts
    the least recently used pair."""
    import gc
    import numpy as np
    from microaligner_amd.device import _ResidentCache

    class Dev:
        def __init__(self, a):
            self.ptr, self.shape, self.dtype, self.nbytes = 1, a.shape, a.dtype, a.nbytes

    c = _ResidentCache()
    a = np.random.default_rng(0).random((600, 600)).astype(np.float32)
    d = Dev(a)
    c.remember(a, d)
    assert c.lookup(a) is d and c.lookup(a[:]) is d and c.hits == 2      # a view of the same memory is the same array
    assert c.lookup(a[:300]) is None and c.lookup(a.copy()) is None
    a *= 2
    assert c.lookup(a) is None and not c.entries                          # edited: forgotten, will be uploaded again
    b = np.ones((600, 600), np.float32)
    c.remember(b, Dev(b))
    assert len(c.entries) == 1
    del b
    gc.collect()
    assert not c.entries and c.bytes == 0                                 # the pair dies with the host memory
    small = np.zeros(10, np.float32)
    c.remember(small, Dev(small))
    assert not c.entries
    c.limit = 3 * a.nbytes
    keep = [np.full((600, 600), i, np.float32) for i in range(5)]
    for k in keep:
        c.remember(k, Dev(k))
    assert len(c.entries) == 3 and c.lookup(keep[0]) is None and c.lookup(keep[4]) is not None
    d.ptr = None
    c.remember(a, d)
    assert c.lookup(a) is None                                            # a freed device array is never handed out


def test_nmi_of_raw_labels_is_scikit_learns():
    """The host path of the gate for label images that are not uint8 (a float image with max() == 0 leaves the reference's
    dog() unchanged and scikit-learn labels its distinct values): identical to normalized_mutual_info_score."""
    import warnings
    import numpy as np
    from sklearn.metrics import normalized_mutual_info_score as nmi
    from microaligner_amd.shared_modules.similarity_scoring import _nmi_of_labels
    rng = np.random.default_rng(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for n in (50, 5000):
            a = -rng.integers(0, 7, n).astype(np.float32) * 0.5
            b = (a * 2 + rng.integers(0, 3, n)).astype(np.float32)
            assert abs(_nmi_of_labels(a, b) - nmi(a, b)) < 1e-14
            assert abs(_nmi_of_labels(a, a) - 1.0) < 1e-14
        z = np.zeros(9)
        assert _nmi_of_labels(z, z) == 1.0 == nmi(z, z) and _nmi_of_labels(z, np.arange(9.0)) == 0.0 == nmi(z, np.arange(9.0))
