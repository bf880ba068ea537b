"""Host-side logic that needs no GPU: API surface, validation, tile geometry, synthetic inputs."""
import os
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


class _Dev:
    def __init__(self, a):
        self.ptr, self.shape, self.dtype, self.nbytes = 1, a.shape, a.dtype, a.nbytes


def test_resident_cache_only_trusts_memory_that_cannot_change(monkeypatch):
    """The host <-> device pair cache behind Context.asdevice() (no device needed: a stand-in device array).  Default
    mode "results": only read-only memory is ever recorded -- the arrays the library hands out (frozen down to the root
    of their base chain) and read-only arrays of the caller -- so an in-place edit is either impossible (it raises) or the
    array was never cached (it is uploaded again).  Recycled or dead memory misses, small arrays are never recorded, the
    byte bound evicts the least recently used pair."""
    import gc
    import numpy as np
    from microaligner_amd.device import _ResidentCache

    monkeypatch.delenv("MICROALIGNER_RESIDENT", raising=False)
    c = _ResidentCache()
    assert c.mode == "results"
    a = np.random.default_rng(0).random((600, 600)).astype(np.float32)
    c.remember(a, _Dev(a))
    assert not c.entries and c.lookup(a) is None              # a writable array of the caller is never recorded ...
    a[100:200, 100:200] = 0                                   # ... so a block edit between uploads cannot be missed
    assert c.lookup(a) is None

    # what DeviceArray.numpy() does with an array it allocated: frozen to the root, then recorded
    root = np.zeros(600 * 600 * 4, np.uint8)
    res = root.view(np.float32).reshape(600, 600)
    c.freeze(res)
    assert not res.flags.writeable and not root.flags.writeable
    d = _Dev(res)
    c.remember(res, d)
    assert c.lookup(res) is d and c.lookup(res[:]) is d and c.hits == 2      # a view of the same memory is the same array
    assert c.lookup(res[:300]) is None and c.lookup(res.copy()) is None
    with pytest.raises(ValueError, match="read-only"):
        res[100:200, 100:200] = 0                             # the advisor's sub-stride block edit: loud, not silent
    with pytest.raises(ValueError):
        res.flags.writeable = True                            # numpy refuses while the root is read-only
    root.flags.writeable = True                               # a caller who insists (root first, then the view) ...
    res.flags.writeable = True
    res[100:200, 100:200] = 1
    assert c.lookup(res) is None and not c.entries            # ... is no longer a hit, and the pair is forgotten

    # a read-only array of the caller is recorded; the pair dies with the host memory
    b = np.ones((600, 600), np.float32)
    b.flags.writeable = False
    c.remember(b, _Dev(b))
    assert len(c.entries) == 1 and c.lookup(b) is not None
    del b
    gc.collect()
    assert not c.entries and c.bytes == 0
    small = np.zeros(10, np.float32)
    small.flags.writeable = False
    c.remember(small, _Dev(small))
    assert not c.entries
    c.limit = 3 * a.nbytes
    keep = [np.full((600, 600), i, np.float32) for i in range(5)]
    for k in keep:
        k.flags.writeable = False
        c.remember(k, _Dev(k))
    assert len(c.entries) == 3 and c.lookup(keep[0]) is None and c.lookup(keep[4]) is not None
    dead = _Dev(keep[0])
    dead.ptr = None
    c.remember(keep[0], dead)
    assert c.lookup(keep[0]) is None                          # a freed device array is never handed out


def test_resident_cache_sampled_mode_is_opt_in_and_documented_unsound(monkeypatch):
    """MICROALIGNER_RESIDENT=sampled (the round-3 default, now opt-in): writable arrays are recorded behind a sampled
    CRC.  Whole-array edits are noticed; a block edit that falls between the samples is NOT -- which is why it is no
    longer the default."""
    import numpy as np
    from microaligner_amd.device import _ResidentCache

    monkeypatch.setenv("MICROALIGNER_RESIDENT", "sampled")
    c = _ResidentCache()
    a = np.random.default_rng(0).random((2048, 2048)).astype(np.float32)
    d = _Dev(a)
    c.remember(a, d)
    assert c.lookup(a) is d
    step = max(1, a.nbytes // c.SAMPLES) | 1                  # the guard samples bytes 0, step, 2 step, ...
    flat = a.reshape(-1)
    i = next(i for i in range(500_000, 600_000) if all((4 * i + b) % step for b in range(4)))
    flat[i] = -1.0                                            # an edit between two samples
    assert c.lookup(a) is d                                   # the documented hole
    a *= 2
    assert c.lookup(a) is None and not c.entries
    monkeypatch.setenv("MICROALIGNER_RESIDENT", "readonly")   # earlier name of today's default
    assert _ResidentCache().mode == "results"
    monkeypatch.setenv("MICROALIGNER_RESIDENT", "sometimes")
    with pytest.raises(ValueError):
        _ResidentCache()


class _FakeDev:
    def __init__(self, shape, dtype):
        import numpy as np
        self.shape, self.dtype, self.data = tuple(shape), np.dtype(dtype), np.zeros(shape, dtype)
        self.nbytes, self.ptr = self.data.nbytes, 1


class _FakeStreamCtx:
    """Stand-in for device.Context with just what parallel.stream_pairs touches; copies are numpy copies, events are
    counters, so the pipeline's threading, ordering, back-pressure and error handling run without a GPU."""

    def __init__(self):
        import threading
        self.lock = threading.Lock()
        self.up = self.down = 0
        self.events = 0
        self.log = []

    def sync(self): pass
    def engine_sync(self, e): pass
    def engine_wait(self, e, ev): pass
    def event_sync(self, ev): pass
    def elapsed_ms(self, a, b): return 1.0
    def transfer_stats(self): return self.up, self.down

    def event(self):
        self.events += 1
        return object()

    def event_destroy(self, ev):
        self.events -= 1

    def engine_record(self, e, ev):
        with self.lock:
            self.log.append(("record", e))

    def empty(self, shape, dtype):
        return _FakeDev(shape, dtype)

    def host_empty(self, shape, dtype, limit=None):
        import numpy as np
        return np.empty(shape, dtype)

    def host_reserve(self, shape, dtype, count):
        with self.lock:
            self.log.append(("reserve", tuple(shape), count))

    def engine_upload(self, dst, arr, engine=1):
        dst.data[...] = arr
        with self.lock:
            self.up += arr.nbytes

    def engine_download(self, src, out, engine=2):
        out[...] = src.data
        with self.lock:
            self.down += out.nbytes


def test_stream_pairs_pipeline_logic_without_a_gpu(monkeypatch):
    """parallel.stream_pairs over a stand-in context: results in input order and equal to the stage applied pair by pair,
    lazily evaluated input, shape changes mid-stream, caller-provided outputs, byte counts, engine busy times, an
    exception in any engine or in the input generator surfaces in the consumer, and early exit of the consumer stops the
    engine threads."""
    import threading
    import numpy as np
    from microaligner_amd import parallel
    from microaligner_amd.device import use_context

    def stage(ctx, dref, dmov):
        flow = _FakeDev(dref.shape + (2,), np.float32)
        flow.data[..., 0] = dref.data - dmov.data
        flow.data[..., 1] = dref.data + dmov.data
        warped = _FakeDev(dmov.shape, dmov.dtype)
        warped.data[...] = dmov.data[::-1]
        return [flow, warped], ["report"], {"sum": float(dref.data.sum())}

    rng = np.random.default_rng(0)
    shapes = [(30, 40)] * 4 + [(50, 20)] * 3 + [(30, 40)] * 2
    pairs = [(rng.random(s).astype(np.float32), rng.random(s).astype(np.float32)) for s in shapes]
    pulled = []

    def lazy():
        for k, p in enumerate(pairs):
            pulled.append(k)
            yield p

    fake = _FakeStreamCtx()
    stats = {}
    with use_context(fake):
        got = []
        for res in parallel.stream_pairs(lazy(), stage=stage, depth=2, stats=stats):
            assert len(pulled) <= res.index + 2 + 2 + 2 + 1      # bounded look-ahead: the queues between the engines
            got.append(res)
    assert [r.index for r in got] == list(range(len(pairs)))
    for r, (ref, mov) in zip(got, pairs):
        flow, warped = r
        assert np.array_equal(flow[..., 0], ref - mov) and np.array_equal(flow[..., 1], ref + mov)
        assert np.array_equal(warped, mov[::-1]) and r.reports == ["report"] and r.extra["sum"] == float(ref.sum())
    assert stats["pairs"] == len(pairs) and stats["h2d_bytes"] == sum(a.nbytes + b.nbytes for a, b in pairs)
    assert stats["d2h_bytes"] == sum(r.flow.nbytes + r.warped.nbytes for r in got)
    assert stats["compute_busy_ms"] == len(pairs) and stats["wall_ms"] > 0
    assert fake.events == 0                                        # every event destroyed
    assert not [t for t in threading.enumerate() if t.name.startswith("ma-engine-")]

    # caller-provided outputs (rows of a memmap in the pipeline); warp=False-like stages may return None entries
    dst_f = np.zeros((4, 30, 40, 2), np.float32)
    dst_w = np.zeros((4, 30, 40), np.float32)
    with use_context(fake):
        for res in parallel.stream_pairs(pairs[:4], stage=stage, out=lambda i: (dst_f[i], dst_w[i])):
            assert res.flow is not None and res.flow.base is dst_f
    assert np.array_equal(dst_f[2][..., 0], pairs[2][0] - pairs[2][1]) and np.array_equal(dst_w[3], pairs[3][1][::-1])

    # two compute lanes (each pair registered by one of two threads with a context of its own): results still leave in
    # input order although the lanes finish out of order
    import time as _time
    lanes_used = set()
    monkeypatch.setattr(parallel, "_lane_context", lambda k: _FakeStreamCtx())

    def uneven_stage(c, dref, dmov):
        lanes_used.add(id(c))
        _time.sleep(0.03 if int(dref.data[0, 0] * 1e6) % 2 else 0.0)      # some pairs take much longer than others
        return stage(c, dref, dmov)

    with use_context(fake):
        st2 = {}
        got2 = list(parallel.stream_pairs(iter(pairs), stage=uneven_stage, compute_lanes=2, stats=st2))
    assert [r.index for r in got2] == list(range(len(pairs))) and len(lanes_used) == 2 and st2["compute_lanes"] == 2
    for r, (ref, mov) in zip(got2, pairs):
        assert np.array_equal(r.flow[..., 0], ref - mov) and np.array_equal(r.warped, mov[::-1])
    assert {t.get("lane") for t in st2["timeline"]} == {0, 1}
    assert not [t for t in threading.enumerate() if t.name.startswith("ma-engine-")]

    # errors: in the stage, in the input, in a bad output array
    def bad_stage(ctx, dref, dmov):
        if dref.data[0, 0] == pairs[2][0][0, 0]:
            raise RuntimeError("boom in compute")
        return stage(ctx, dref, dmov)

    def bad_input():
        yield pairs[0]
        raise OSError("boom in the loader")

    with use_context(fake):
        with pytest.raises(RuntimeError, match="boom in compute"):
            list(parallel.stream_pairs(pairs, stage=bad_stage))
        with pytest.raises(OSError, match="boom in the loader"):
            list(parallel.stream_pairs(bad_input(), stage=stage))
        with pytest.raises(ValueError, match="output 0"):
            list(parallel.stream_pairs(pairs[:2], stage=stage, out=lambda i: (np.zeros((3, 3), np.float32), None)))
        with pytest.raises(ValueError, match="2-D"):
            list(parallel.stream_pairs([(np.zeros((2, 2, 2), np.float32), np.zeros((2, 2), np.float32))], stage=stage))
        # the consumer walks away after the first result: the generator's close() stops and joins the engines
        gen = parallel.stream_pairs(pairs, stage=stage)
        first = next(gen)
        assert first.index == 0
        gen.close()
    assert not [t for t in threading.enumerate() if t.name.startswith("ma-engine-")]
    assert fake.events == 0


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


def test_numa_binding_follows_the_devices_local_cpulist(tmp_path):
    """device.bind_to_device_numa: the CPUs of /sys/bus/pci/devices/<bdf>/local_cpulist, for every thread of the process;
    nothing happens when the topology is unknown, when the list is the whole machine, or when MICROALIGNER_BIND_NUMA=0."""
    import os
    import threading
    from microaligner_amd import device
    assert device._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and device._parse_cpulist("") == []
    before = os.sched_getaffinity(0)
    if len(before) < 2:
        pytest.skip("needs at least two CPUs")
    some = sorted(before)[:max(1, len(before) // 2)]
    bdf = tmp_path / "0000:c1:00.0"
    bdf.mkdir()
    (bdf / "local_cpulist").write_text(",".join(str(c) for c in some) + "\n")
    assert device.device_local_cpus(pci_bus_id="0000:C1:00.0", sysfs=str(tmp_path)) == some
    assert device.device_local_cpus(pci_bus_id="0000:ff:00.0", sysfs=str(tmp_path)) == []
    seen = {}
    stop = threading.Event()
    t = threading.Thread(target=lambda: (stop.wait(5), seen.update(mask=os.sched_getaffinity(0))))
    t.start()                                  # a thread that exists BEFORE the binding is moved as well
    try:
        os.environ["MICROALIGNER_BIND_NUMA"] = "0"
        assert device.bind_to_device_numa(pci_bus_id="0000:c1:00.0", sysfs=str(tmp_path)) == [] and os.sched_getaffinity(0) == before
        del os.environ["MICROALIGNER_BIND_NUMA"]
        assert device.bind_to_device_numa(pci_bus_id="0000:ff:00.0", sysfs=str(tmp_path)) == []
        assert device.bind_to_device_numa(pci_bus_id="0000:c1:00.0", sysfs=str(tmp_path)) == some
        assert os.sched_getaffinity(0) == set(some)
        stop.set()
        t.join()
        assert seen["mask"] == set(some)
        (bdf / "local_cpulist").write_text(",".join(str(c) for c in sorted(some)) + "\n")
        assert device.bind_to_device_numa(pci_bus_id="0000:c1:00.0", sysfs=str(tmp_path)) == []   # already there: nothing to do
    finally:
        stop.set()
        os.environ.pop("MICROALIGNER_BIND_NUMA", None)
        device.set_affinity(before)
    assert os.sched_getaffinity(0) == before


def test_warper_routes_large_host_pages_through_the_page_driver(monkeypatch):
    """Warper.warp() (warper.py:37-53 of the reference: image and flow in, warped image out, inputs consumed): a host page of
    HOST_BANDED_MIN bytes or more that is not resident goes through Context.warp_pages as ONE page (upload, kernel and
    download overlapped in bands); resident pages, device arrays and small pages take the plain device warp."""
    from microaligner_amd.optflow_reg import warper as warper_mod
    from microaligner_amd.device import DeviceArray
    calls = []

    class Dev:                                  # stands for a DeviceArray of the fake context
        def __init__(self, a):
            self.a, self.shape, self.ndim = a, a.shape, a.ndim

        def numpy(self):
            return self.a

        def __len__(self):
            return len(self.a)

    class Ctx:
        resident = set()

        def is_resident(self, a):
            return id(a) in self.resident

        def asdevice(self, a):
            return a if isinstance(a, Dev) else Dev(np.asarray(a))

        def host_empty(self, shape, dtype):
            return np.zeros(shape, dtype)

        def warp_pages(self, pages, flow, tile, overlap, out):
            calls.append(("pages", len(pages), tile, overlap))
            out[0][...] = pages[0] + 1
            return out

        def warp(self, img, flow, tile, overlap):
            calls.append(("device", tile, overlap))
            return Dev(img.a + 1)

    ctx = Ctx()
    monkeypatch.setattr(warper_mod, "get_context", lambda: ctx)
    monkeypatch.setattr(warper_mod, "DeviceArray", Dev)
    monkeypatch.setattr(warper_mod.Warper, "HOST_BANDED_MIN", 4096)
    big, small = np.ones((64, 64), np.uint16), np.ones((8, 8), np.uint16)
    flow_big, flow_small = np.zeros((64, 64, 2), np.float32), np.zeros((8, 8, 2), np.float32)

    def run(img, flow):
        w = warper_mod.Warper()
        w.tile_size, w.overlap = 30, 4
        w.image, w.flow = img, flow
        res = w.warp()
        assert len(w.image) == 0 and len(w.flow) == 0
        return res

    res = run(big, flow_big)
    assert calls == [("pages", 1, 30, 4)] and isinstance(res, np.ndarray) and (res == 2).all()
    calls.clear()
    assert (run(small, flow_small) == 2).all() and calls == [("device", 30, 4)]
    calls.clear()
    ctx.resident.add(id(big))
    assert (run(big, flow_big) == 2).all() and calls == [("device", 30, 4)]
    calls.clear()
    res = run(Dev(big), flow_big)
    assert isinstance(res, Dev) and calls == [("device", 30, 4)]
    with pytest.raises(ValueError):
        run(np.ones((4, 4, 3), np.uint8), np.zeros((4, 4, 2), np.float32))
    with pytest.raises(ValueError):
        run(np.array([]), flow_small)


def test_bench_counts_the_cpus_the_container_grants(tmp_path, monkeypatch):
    """bench.effective_cpus(): the smaller of the affinity mask and the cgroup CPU-time quota -- the GPU boxes of this pool show
    256 hardware threads and grant 16 CPUs of time, and `cpu_baseline.cores` must say 16."""
    import builtins
    import bench
    n, how = bench.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1) and how
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            p = tmp_path / "cpu.max"
            p.write_text("200000 100000\n")
            return real_open(p, *a, **k)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    if len(os.sched_getaffinity(0)) > 2:
        n, how = bench.effective_cpus()
        assert n == 2 and "quota" in how
