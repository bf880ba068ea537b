"""The C-ABI shared library: it loads, exports every symbol include/microaligner_hip.h declares,
and the product path fails loudly (no CPU fallback) when no HIP device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "microaligner_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ma_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from microaligner_amd import build, _lib
    build.build()
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes prototypes out of sync with the header"
    assert b"gfx950" in lib.ma_version()


def test_constants_of_the_header_match_the_bindings():
    """Enumerators and #defines that size caller-provided buffers or select rounding models exist three times (header,
    kernels via the header, ctypes binding): the binding must carry the header's values."""
    from microaligner_amd import _lib
    text = open(HEADER).read()
    vals = {k: int(v) for k, v in re.findall(r"\b(MA_[A-Z0-9_]+)\s*=\s*(-?\d+)", text)}
    vals.update({k: int(v) for k, v in re.findall(r"#define\s+(MA_[A-Z0-9_]+)\s+(\d+)", text)})
    for name in ("MA_U8", "MA_U16", "MA_F32", "MA_OK", "MA_EINVAL", "MA_ENOMEM", "MA_EHIP", "MA_ENODEV", "MA_FB_MULADD_FUSED",
                 "MA_DOG_FUSED_BLUR", "MA_DOG_FUSED_SCALE", "MA_FLOW_CELL_REPLICAS"):
        assert getattr(_lib, name) == vals[name], name
    for name, kid in _lib.KERNEL_IDS.items():
        assert vals["MA_K_" + name.upper()] == kid
    # struct layouts: field names and order of ma_params / ma_level_report
    for cname, cls in (("ma_params", _lib.MaParams), ("ma_level_report", _lib.MaLevelReport),
                       ("ma_feature_round_result", _lib.MaFeatureRoundResult)):
        body = re.search(r"typedef struct " + cname + r" \{(.*?)\} " + cname + ";", text, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                ctype, names = decl.split(None, 1)
                fields += [(n.strip(), ctype) for n in names.split(",")]
        assert [f[0] for f in cls._fields_] == [re.sub(r"\[\d+\]", "", f[0]) for f in fields], cname
        for (_, ct), (name, decl_t) in zip(cls._fields_, fields):
            base = {"int": C.c_int, "double": C.c_double}[decl_t]
            dim = re.search(r"\[(\d+)\]", name)
            assert (ct is base) if not dim else (ct._type_ is base and ct._length_ == int(dim.group(1)))


def test_numpy_mean_replica_is_bit_identical():
    """mi_tiled takes np.mean of the chunk scores (similarity_scoring.py:49); ma_optflow_register restates numpy's pairwise
    summation on the host so that the gate compares the very same doubles."""
    from microaligner_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    for n in list(range(1, 40)) + [63, 64, 65, 127, 128, 129, 130, 136, 137, 255, 256, 257, 269, 1000, 4097, 70000]:
        for scale in (1.0, 1e-3, 1e3):
            v = rng.random(n) * scale
            out = C.c_double()
            _lib.check(lib.ma_host_np_mean(v.ctypes.data_as(C.POINTER(C.c_double)), n, C.byref(out)))
            assert out.value == np.mean(v), n
    assert lib.ma_host_np_mean(None, 3, None) == _lib.MA_EINVAL


@pytest.mark.parametrize("fn", ["ma_host_parallel_copy", "ma_host_stream_copy"])
def test_host_copy_pool_copies_every_byte(fn):
    """The host threads that fill / drain the page-locked staging chunks of the transfer engines (ma_api.hip, CopyPool; the
    drain writes with non-temporal stores): sizes around the slice boundaries, odd sizes and offsets (unaligned heads and
    tails of the 16-byte stores), repeated calls from two threads at once."""
    import threading
    from microaligner_amd import _lib
    lib = _lib.load()
    copy = getattr(lib, fn)
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (40 << 20) + 977, dtype=np.uint8)
    for n in (0, 1, 4095, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, (8 << 20) + 13, (32 << 20), src.size - 21):
        for do, so in ((3, 5), (0, 0), (16, 1), (15, 16)):
            dst = np.zeros(n + 64, np.uint8)
            _lib.check(copy(dst.ctypes.data + do, src.ctypes.data + so, n))
            assert np.array_equal(dst[do:do + n], src[so:so + n]) and not dst[:do].any() and not dst[do + n:].any(), (n, do, so)
    errs = []

    def hammer(seed):
        r = np.random.default_rng(seed)
        for _ in range(6):
            n = int(r.integers(1 << 20, 24 << 20))
            off = int(r.integers(0, src.size - n))
            d = np.empty(n, np.uint8)
            copy(d.ctypes.data, src.ctypes.data + off, n)
            if not np.array_equal(d, src[off:off + n]):
                errs.append((seed, n, off))

    th = [threading.Thread(target=hammer, args=(s,)) for s in (1, 2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs
    assert copy(None, None, 8) == _lib.MA_EINVAL


def test_host_copy_pool_survives_a_fork():
    """A fork()ed child (multiprocessing's fork start method, dask workers) inherits the pool object but none of its worker
    threads: a copy of a few MiB must not wait for them (it used to wait forever).  The parent's pool keeps working."""
    import multiprocessing as mp
    from microaligner_amd import _lib
    lib = _lib.load()
    src = np.arange(6 << 20, dtype=np.uint8)
    dst = np.zeros_like(src)
    _lib.check(lib.ma_host_parallel_copy(dst.ctypes.data, src.ctypes.data, src.nbytes))      # the pool's threads exist now
    assert np.array_equal(dst, src)

    def child(q):
        d = np.zeros_like(src)
        rc1 = lib.ma_host_parallel_copy(d.ctypes.data, src.ctypes.data, src.nbytes)
        ok1 = bool(np.array_equal(d, src))
        d[:] = 0
        rc2 = lib.ma_host_stream_copy(d.ctypes.data, src.ctypes.data, src.nbytes)
        q.put((rc1, ok1, rc2, bool(np.array_equal(d, src))))

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=child, args=(q,))
    p.start()
    p.join(60)
    alive = p.is_alive()
    if alive:
        p.kill()
    assert not alive, "the forked child hung in the copy pool"
    assert q.get(timeout=10) == (0, True, 0, True)
    dst[:] = 0
    _lib.check(lib.ma_host_parallel_copy(dst.ctypes.data, src.ctypes.data, src.nbytes))
    assert np.array_equal(dst, src)


def test_register_entry_rejects_bad_parameters_without_a_device():
    from microaligner_amd import _lib
    lib = _lib.load()
    prm = _lib.MaParams()
    lib.ma_params_default(C.byref(prm))
    assert (prm.num_pyr_lvl, prm.num_iterations, prm.tile_size, prm.overlap, prm.use_full_res_img, prm.use_dog,
            prm.fb_flags, prm.dog_flags) == (4, 3, 1000, 100, 0, 0, 0, 0)          # optflow_registrator.py:54-59
    n = C.c_int()
    assert lib.ma_optflow_register(None, None, None, 2, 10, 10, C.byref(prm), None, None, 0, C.byref(n)) == _lib.MA_EINVAL


def test_code_object_targets_gfx950_only():
    from microaligner_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_80"):
        assert other not in blob


def _have_gpu():
    from microaligner_amd import device
    try:
        return device.device_count() > 0
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="checks the no-device behaviour")
def test_no_device_is_a_loud_error_not_a_fallback():
    from microaligner_amd import OptFlowRegistrator, Warper, _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.ma_ctx_create(0, C.byref(h)) == _lib.MA_ENODEV
    assert b"HIP device" in lib.ma_last_error()
    reg = OptFlowRegistrator()
    reg.ref_img = np.ones((200, 200), np.float32)
    reg.mov_img = np.ones((200, 200), np.float32)
    with pytest.raises(RuntimeError):
        reg.register()
    w = Warper()
    w.image = np.ones((50, 50), np.uint8)
    w.flow = np.zeros((50, 50, 2), np.float32)
    with pytest.raises(RuntimeError):
        w.warp()


def test_null_arguments_are_rejected_without_a_device():
    from microaligner_amd import _lib
    lib = _lib.load()
    assert lib.ma_sync(None) == _lib.MA_EINVAL
    assert lib.ma_farneback_tiled(None, None, None, 2, 10, 10, 0, 0, 9, 1, 1, 1.7, 0, None) == _lib.MA_EINVAL
    assert b"invalid argument" in lib.ma_last_error()
    with pytest.raises(ValueError):
        _lib.check(_lib.MA_EINVAL)


def test_the_product_never_reaches_for_the_oracle():
    """oracle/ is test infrastructure: nothing under microaligner_amd/ (Python or HIP) may import, load or mention it,
    and bench.py may only do so inside its cpu_baseline leg."""
    import ast
    pkg = os.path.join(ROOT, "microaligner_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                tree = ast.parse(open(path).read())
                for node in ast.walk(tree):
                    mods = []
                    if isinstance(node, ast.Import):
                        mods = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom):
                        mods = [node.module or ""]
                    assert not any(m == "oracle" or m.startswith("oracle.") for m in mods), path
                assert "libma_oracle" not in open(path).read(), path
    bench = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for node in ast.walk(bench):
        if isinstance(node, ast.FunctionDef):
            uses = [n for n in ast.walk(node) if isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle")]
            assert not uses or node.name in ("cpu_baseline", "cpu_baseline_opencv"), node.name
    top = [n for n in bench.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    assert not any(isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle") for n in top)


def test_a_missing_library_is_an_import_error(monkeypatch, tmp_path):
    from microaligner_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libmicroaligner_hip.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_page_warp_band_plan_never_needs_a_row_that_has_not_arrived():
    """ma_warp_pages_plan (the geometry of ma_warp_pages_host's bands): bands are whole tile rows and cover every output row
    once; the window of every output row of band b (slicer.py:69-118: tile row ty spans rows ty*T - ov ... (ty+1)*T + ov)
    ends at or before the source row the driver waits for before it launches the band; the band size is honoured."""
    import ctypes as C
    from microaligner_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(4)
    cases = [(16384, 16384, 1000, 100, 32 << 20, _lib.MA_U16), (46300, 46700, 1000, 100, 32 << 20, _lib.MA_U16),
             (333, 290, 100, 12, 1, _lib.MA_U8), (200, 300, 64, 0, 1, _lib.MA_F32), (96, 257, 0, 0, 1, _lib.MA_U8),
             (407, 130, 50, 40, 1, _lib.MA_U16), (65535, 11, 7, 7, 1000, _lib.MA_U8), (1, 1, 5, 2, 1, _lib.MA_F32)]
    for _ in range(300):
        H, W = int(rng.integers(1, 5000)), int(rng.integers(1, 5000))
        T = int(rng.integers(0, 1500))
        cases.append((H, W, T, int(rng.integers(0, T + 1)), int(rng.choice([1, 1 << 16, 1 << 20, 32 << 20, 1 << 40])),
                      int(rng.choice([_lib.MA_U8, _lib.MA_U16, _lib.MA_F32]))))
    for H, W, T, ov, band_bytes, dt in cases:
        rows, nb = C.c_int(), C.c_int()
        _lib.check(lib.ma_warp_pages_plan(dt, H, W, T, ov, band_bytes, C.byref(rows), C.byref(nb)))
        rows, nb = rows.value, nb.value
        esz = {_lib.MA_U8: 1, _lib.MA_U16: 2, _lib.MA_F32: 4}[dt]
        assert 1 <= rows <= H and nb == -(-H // rows), (H, W, T, ov, band_bytes)
        if T == 0:
            assert rows == H and nb == 1
            continue
        assert rows == H or rows % T == 0
        if nb > 1:
            assert rows * W * esz >= band_bytes                 # a band is at least the requested size ...
            assert (rows - T) * W * esz < band_bytes            # ... and no tile row more than that takes
        for b in range(nb):
            y_last = min(H, (b + 1) * rows) - 1                 # the band's last output row and the tile row it lies in
            window_end = min(H, (y_last // T + 1) * T + ov)     # one past the last source row that window reads
            waited_for = H if b == nb - 1 else min(H, (b + 1) * rows + ov)
            assert window_end <= waited_for, (H, W, T, ov, band_bytes, b)
    r, n = C.c_int(), C.c_int()
    assert lib.ma_warp_pages_plan(_lib.MA_U8, 0, 5, 5, 1, 1, C.byref(r), C.byref(n)) == _lib.MA_EINVAL
    assert lib.ma_warp_pages_plan(_lib.MA_U8, 5, 5, 5, 1, 0, C.byref(r), C.byref(n)) == _lib.MA_EINVAL
    assert lib.ma_warp_pages_plan(_lib.MA_U8, 5, 5, 5, 1, 1, None, C.byref(n)) == _lib.MA_EINVAL
