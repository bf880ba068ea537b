"""Device context, HBM-resident arrays and the primitive operations of the hot path.

Thin Python over the C-ABI (include/microaligner_hip.h).  One `Context` per HIP
device per process; `DeviceArray` is a dense row-major array in HBM.  All work is
stream ordered on the context's stream, so intermediate results never visit the host.
"""
import collections
import ctypes as C
import os
import threading
import weakref
import zlib

import numpy as np

from . import _lib as L

_DT = {np.dtype(np.uint8): L.MA_U8, np.dtype(np.uint16): L.MA_U16, np.dtype(np.float32): L.MA_F32}


def _dt(dtype):
    dtype = np.dtype(dtype)
    if dtype not in _DT:
        raise ValueError(f"unsupported image dtype {dtype}: the HIP path handles uint8, uint16 and float32")
    return _DT[dtype]


class DeviceArray:
    """Dense row-major array living in HBM.  Freed back to the context's pool on `free()`/GC."""

    __slots__ = ("ctx", "shape", "dtype", "ptr", "_nbytes_alloc", "_owner", "minmax", "cellkeys")

    def __init__(self, ctx, shape, dtype, ptr, nbytes_alloc, owner=True):
        self.ctx, self.shape, self.dtype, self.ptr = ctx, tuple(int(s) for s in shape), np.dtype(dtype), ptr
        self._nbytes_alloc, self._owner = nbytes_alloc, owner
        # optional DeviceArray of two floats (min, max of this array), left by the kernel that produced it
        # (Context.warp / pyr_down with minmax=True) for a following dog_u8; arrays are never modified in place
        self.minmax = None
        # optional (tile, overlap, DeviceArray of keys): per-cell maxima of this FLOW, left by a warp that read it
        # (Context.warp(flow_cells=True)) for a following merge_flows
        self.cellkeys = None

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def ndim(self):
        return len(self.shape)

    def __len__(self):
        return self.shape[0] if self.shape else 0

    def numpy(self, out=None):
        """Copy to the host.  Without `out` the result array comes from the context's pool of page-locked,
        already-faulted host buffers (Context.host_empty): an ordinary ndarray whose memory goes back to the pool
        when the last reference to it (or to a view of it) is dropped.  `out`: a C-contiguous array of the same
        shape and dtype to fill instead (e.g. a row of the caller's memmap)."""
        self._check_alive()
        own = out is None
        if own:
            out = self.ctx.host_empty(self.shape, self.dtype)
        elif (tuple(out.shape) != self.shape or out.dtype != self.dtype or not out.flags.c_contiguous
              or not out.flags.writeable):
            raise ValueError(f"out must be a writable C-contiguous {self.dtype} array of shape {self.shape}")
        if out.nbytes:
            L.check(self.ctx.lib.ma_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes))
        if own and self.ctx._resident.mode == "results":
            # results are handed out read-only (the flags of a caller's `out` array are never touched): that is what
            # lets asdevice(out) find this device array again instead of uploading the bytes it has just downloaded
            self.ctx._resident.freeze(out)
        self.ctx._resident.remember(out, self)
        return out

    def _check_alive(self):
        if self.ptr is None or self.ctx is None or self.ctx._closed:
            raise RuntimeError("this DeviceArray was freed (its context is closed or free() was called)")

    def copy(self):
        out = self.ctx.empty(self.shape, self.dtype)
        L.check(self.ctx.lib.ma_memcpy_d2d(self.ctx.handle, out.ptr, self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr and self._owner and self.ctx is not None:
            self.ctx._release(self.ptr, self._nbytes_alloc)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _HostBuffer:
    """Owner of one page-locked host buffer; exposes it through __array_interface__ so that numpy arrays built on
    it keep it alive (base chain), and hands the memory back to the pool when it dies."""

    __slots__ = ("ctx", "ptr", "bucket", "nbytes", "__weakref__")   # weakly referenced by the resident-pair cache

    def __init__(self, ctx, ptr, bucket, nbytes):
        self.ctx, self.ptr, self.bucket, self.nbytes = ctx, ptr, bucket, nbytes

    @property
    def __array_interface__(self):        # numpy (any Python): zero-copy view, base object = self
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        try:
            self.ctx._host_release(self.ptr, self.bucket)
        except Exception:
            pass


class _ResidentCache:
    """Host array <-> device array pairs that are known to hold the same bytes.

    The reference's callers move every array through numpy: register() returns the flow, the very next statement
    hands it to Warper (microaligner/__main__.py:418-433), and warp_and_save_pages sets the same flow on the warper
    for every page of a cycle (:296-301).  Through a drop-in API that is one 2.1 GB upload per call at 16384^2.  This
    cache lets Context.asdevice() recognise a host array whose bytes are already on the device and return the device
    array that is still alive instead.  An entry is only ever made for memory that CANNOT have changed since:

      * key: address, shape, strides and dtype of the host array (C-contiguous arrays of at least MIN_BYTES only);
      * liveness: a weak reference to the object that owns the memory (the end of the array's .base chain); when it
        dies the entry goes with it, so a recycled address can never match;
      * immutability (mode "results", the default): the arrays DeviceArray.numpy() hands out -- the flow of register(),
        the image of warp() -- are marked read-only down to the root of their base chain, so an in-place edit raises
        (numpy: "assignment destination is read-only") instead of going unnoticed; take a .copy() to edit one.  Arrays
        of the caller are recorded only when they are read-only themselves (np.memmap(mode="r"), arr.flags.writeable =
        False on the owner).  A hit requires every array of the base chain to be read-only still;
      * bound: least recently used pairs are dropped beyond MICROALIGNER_RESIDENT_GB (default 16) of device memory.

    MICROALIGNER_RESIDENT=off: no cache, results are ordinary writable arrays (everything is uploaded every time).
    MICROALIGNER_RESIDENT=sampled: results stay writable and ANY uploaded array is recorded, guarded by a CRC of head,
    tail and SAMPLES strided bytes checked on every hit -- cheap, but NOT sound: an in-place edit of a block that falls
    between the samples (stride ~256 KiB at 2 GB) goes unnoticed and the stale device copy is used.  Opt-in only, for
    callers that never edit arrays in place.

    Device arrays are never modified in place (every operation allocates its output), so a recorded pair stays valid
    for as long as the host side is untouched."""

    MIN_BYTES = 1 << 20
    SAMPLES = 8192
    EDGE = 4096

    def __init__(self):
        self.mode = os.environ.get("MICROALIGNER_RESIDENT", "results").lower()
        if self.mode == "readonly":      # earlier name of the default mode
            self.mode = "results"
        if self.mode not in ("results", "sampled", "off"):
            raise ValueError("MICROALIGNER_RESIDENT must be one of results, sampled, off")
        self.limit = int(float(os.environ.get("MICROALIGNER_RESIDENT_GB", "16")) * (1 << 30))
        self.entries = collections.OrderedDict()   # key -> [weakref(owner), DeviceArray, signature]
        self.bytes = 0
        self.hits = 0
        # weak-reference callbacks (_forget) run from the collector on whichever thread it picks, and the engines of
        # parallel.stream_pairs run on their own threads; re-entrant: a callback can fire inside remember() / lookup()
        self.lock = threading.RLock()

    @staticmethod
    def _key(arr):
        return (arr.__array_interface__["data"][0], arr.shape, arr.strides, arr.dtype.str)

    @staticmethod
    def _owner(arr):
        while isinstance(getattr(arr, "base", None), np.ndarray):
            arr = arr.base
        return arr.base if getattr(arr, "base", None) is not None else arr

    @staticmethod
    def _frozen(arr):
        """True when neither `arr` nor any array it is a view of can be written through numpy."""
        while isinstance(arr, np.ndarray):
            if arr.flags.writeable:
                return False
            arr = arr.base
        return True

    @staticmethod
    def freeze(arr):
        """Mark `arr` and every array of its base chain read-only (root first is not needed: clearing the flag is
        always allowed; setting it again is refused by numpy while the root is read-only)."""
        while isinstance(arr, np.ndarray):
            arr.flags.writeable = False
            arr = arr.base

    @classmethod
    def _signature(cls, arr):
        b = arr.reshape(-1).view(np.uint8)
        n = b.size
        step = max(1, n // cls.SAMPLES) | 1
        crc = zlib.crc32(b[:cls.EDGE].tobytes())
        crc = zlib.crc32(b[-cls.EDGE:].tobytes(), crc)
        return zlib.crc32(b[::step].tobytes(), crc)

    def _eligible(self, arr):
        if not (self.mode != "off" and isinstance(arr, np.ndarray) and arr.flags.c_contiguous
                and arr.nbytes >= self.MIN_BYTES):
            return False
        return self.mode == "sampled" or self._frozen(arr)

    def remember(self, arr, dev):
        if not self._eligible(arr) or dev.ptr is None:
            return
        key = self._key(arr)
        try:
            ref = weakref.ref(self._owner(arr), lambda _r, k=key: self._forget(k))
        except TypeError:       # the memory owner cannot be weakly referenced (mmap, bytes): liveness unknown
            return
        sig = self._signature(arr) if self.mode == "sampled" else None
        with self.lock:
            self._forget(key)
            self.entries[key] = [ref, dev, sig]
            self.bytes += dev.nbytes
            while self.bytes > self.limit and len(self.entries) > 1:
                self._forget(next(iter(self.entries)))

    def _forget(self, key):
        with self.lock:
            e = self.entries.pop(key, None)
            if e is not None:
                self.bytes -= e[1].nbytes if e[1].ptr is not None else 0

    def lookup(self, arr):
        if not self._eligible(arr):     # results mode: an array that has become writable again is never a hit
            if self.mode != "off" and isinstance(arr, np.ndarray):
                self._forget(self._key(arr))
            return None
        key = self._key(arr)
        with self.lock:
            e = self.entries.get(key)
            if e is None:
                return None
            ref, dev, sig = e
            if (ref() is None or dev.ptr is None or dev.shape != arr.shape or dev.dtype != arr.dtype
                    or (sig is not None and self._signature(arr) != sig)):
                self._forget(key)
                return None
            self.entries.move_to_end(key)
            self.hits += 1
            return dev

    def clear(self):
        with self.lock:
            self.entries.clear()
            self.bytes = 0


class Context:
    """One per HIP device.  Owns the C-side ma_ctx and a size-bucketed pool of HBM buffers."""

    def __init__(self, device=0):
        self.lib = L.load()
        h = C.c_void_p()
        L.check(self.lib.ma_ctx_create(int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)
        self._pool = {}
        self._live = {}        # ptr -> bucket of every buffer handed out and not yet released
        self._host_pool = {}   # bucket -> [pinned host pointers] (result arrays of the numpy-in / numpy-out API)
        self._host_owned = {}  # bucket -> page-locked buffers this context owns (handed out + pooled)
        self._host_lock = threading.RLock()   # re-entrant: a pinned result array finalised by a GC pass that starts inside
        #                                        _host_release() comes back here on the same thread
        self._resident = _ResidentCache()
        self._closed = False
        lim = os.environ.get("MICROALIGNER_WORKSPACE_GB")
        if lim:
            L.check(self.lib.ma_ctx_set_workspace_limit(self.handle, int(float(lim) * (1 << 30))))

    # -- memory --------------------------------------------------------------------------
    def empty(self, shape, dtype):
        shape = tuple(int(s) for s in shape)
        dtype = np.dtype(dtype)
        nbytes = max(int(np.prod(shape, dtype=np.int64)) * dtype.itemsize, 1)
        bucket = (nbytes + 0xFFFF) & ~0xFFFF
        with self._host_lock:    # _release() runs from __del__, i.e. from whichever thread the collector picks
            free = self._pool.get(bucket)
            ptr = free.pop() if free else None
        if ptr is None:
            p = C.c_void_p()
            rc = self.lib.ma_malloc(self.handle, bucket, C.byref(p))
            if rc == L.MA_ENOMEM:
                self.trim()
                rc = self.lib.ma_malloc(self.handle, bucket, C.byref(p))
            L.check(rc)
            ptr = p.value
        with self._host_lock:
            self._live[ptr] = bucket
        return DeviceArray(self, shape, dtype, ptr, bucket)

    def zeros(self, shape, dtype):
        a = self.empty(shape, dtype)
        L.check(self.lib.ma_memset(self.handle, a.ptr, 0, a.nbytes))
        return a

    def asdevice(self, arr):
        """numpy -> DeviceArray (H2D copy); DeviceArray passes through.  A host array this context has uploaded or
        downloaded before, whose memory is still alive and unmodified (_ResidentCache), maps to the device array that
        already holds it: no copy."""
        if isinstance(arr, DeviceArray):
            return arr
        arr = np.ascontiguousarray(arr)
        _dt(arr.dtype)
        d = self._resident.lookup(arr)
        if d is not None:
            return d
        d = self.empty(arr.shape, arr.dtype)
        if arr.nbytes:
            L.check(self.lib.ma_memcpy_h2d(self.handle, d.ptr, arr.ctypes.data, arr.nbytes))
        self._resident.remember(arr, d)
        return d

    def is_resident(self, arr):
        """Whether asdevice(arr) would find the host array in HBM already (no upload)."""
        return isinstance(arr, DeviceArray) or self._resident.lookup(np.ascontiguousarray(arr)) is not None

    def _release(self, ptr, bucket):
        with self._host_lock:
            if self._closed:
                return   # close() already returned every outstanding buffer to the driver
            self._live.pop(ptr, None)
            self._pool.setdefault(bucket, []).append(ptr)

    def trim(self):
        """Return every pooled buffer (device and host) to the driver."""
        self._resident.clear()
        self.lib.ma_ctx_trim(self.handle)   # the intermediates ma_optflow_register caches on the C side
        with self._host_lock:
            dev_pool, self._pool = self._pool, {}
        for free in dev_pool.values():
            for p in free:
                self.lib.ma_free(self.handle, p)
        with self._host_lock:
            pools, self._host_pool = self._host_pool, {}
            for bucket, free in pools.items():
                self._host_owned[bucket] = self._host_owned.get(bucket, 0) - len(free)
        for free in pools.values():
            for p in free:
                self.lib.ma_host_free(p)

    def _run(self, fn, *args):
        """Call a compute entry point; on MA_ENOMEM hand the cached buffers back to the driver and retry once
        (the pool never shrinks by itself and can hold tens of GB of flow-sized buffers)."""
        rc = fn(self.handle, *args)
        if rc == L.MA_ENOMEM:
            self.trim()
            rc = fn(self.handle, *args)
        L.check(rc)

    # page-locked result arrays ---------------------------------------------------------------
    # Page-locked buffers of one size that a context will own at any time (handed out or pooled).  Pinning is far
    # dearer than a page fault per 4 KiB (hipHostMalloc: ~200 ms per GiB; faulting a fresh pageable array: ~25 ms per GiB), so
    # it only pays for buffers that are reused: a caller that keeps accumulating results gets pageable arrays beyond
    # this number, a loop that drops each result before the next but one reuses the same two buffers forever.
    HOST_PINNED_PER_BUCKET = 2

    def host_empty(self, shape, dtype, limit=None):
        """ndarray on page-locked host memory drawn from a pool (`limit`: page-locked buffers of this size the context
        may own, default HOST_PINNED_PER_BUCKET; parallel.stream_pairs keeps more results in flight).  The memory returns to the pool when the array
        and all views of it are gone (the array's base object is the owner), so a loop that keeps calling
        register() / warp() reuses the same already-faulted, DMA-able buffers instead of paying a page fault per
        4 KiB of every fresh result."""
        shape = tuple(int(s) for s in shape)
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if nbytes < (1 << 20):          # small results: not worth a pinned allocation
            return np.empty(shape, dtype)
        bucket = (nbytes + 0x1FFFFF) & ~0x1FFFFF
        with self._host_lock:
            free = self._host_pool.get(bucket)
            ptr = free.pop() if free else None
            if ptr is None:
                if self._host_owned.get(bucket, 0) >= (limit or self.HOST_PINNED_PER_BUCKET):
                    return np.empty(shape, dtype)
                self._host_owned[bucket] = self._host_owned.get(bucket, 0) + 1
        if ptr is None:
            p = C.c_void_p()
            rc = self.lib.ma_host_alloc(bucket, C.byref(p))
            if rc != L.MA_OK:           # page-locked memory exhausted: fall back to a pageable array
                with self._host_lock:
                    self._host_owned[bucket] -= 1
                return np.empty(shape, dtype)
            ptr = p.value
        owner = _HostBuffer(self, ptr, bucket, nbytes)
        return np.asarray(owner).view(dtype).reshape(shape)   # base chain ends at `owner`

    def host_reserve(self, shape, dtype, count):
        """Make sure the pool can hand out `count` page-locked arrays of this shape at once (allocating what is missing
        now, in one go: hipHostMalloc takes ~0.2 s per GiB and stalls other HIP calls of the process while it runs, so a
        pipeline does this before its first pair rather than in the middle of the stream)."""
        held = [self.host_empty(shape, dtype, limit=count) for _ in range(count)]
        del held

    def _host_release(self, ptr, bucket):
        with self._host_lock:
            if not self._closed:
                self._host_pool.setdefault(bucket, []).append(ptr)
                return
        self.lib.ma_host_free(ptr)

    def sync(self):
        L.check(self.lib.ma_sync(self.handle))

    # -- options / transfer engines ------------------------------------------------------------
    def set_option(self, option, value):
        L.check(self.lib.ma_ctx_set_option(self.handle, int(option), int(value)))

    def get_option(self, option):
        v = C.c_longlong()
        L.check(self.lib.ma_ctx_get_option(self.handle, int(option), C.byref(v)))
        return v.value

    @property
    def companion_stream(self):
        """Whether ma_optflow_register runs the flow-independent dog() calls on the low-priority companion stream
        (MA_OPT_COMPANION_STREAM, default True).  False: one stream, every kernel alone on the chip (profiling)."""
        return bool(self.get_option(L.MA_OPT_COMPANION_STREAM))

    @companion_stream.setter
    def companion_stream(self, on):
        self.set_option(L.MA_OPT_COMPANION_STREAM, 1 if on else 0)

    def engine_upload(self, dst, arr, engine=L.MA_ENGINE_H2D):
        """Copy the C-contiguous host array `arr` into the DeviceArray `dst` on a transfer engine's stream; returns when
        this copy is complete (the compute stream is not involved: order it with engine_record / engine_wait)."""
        if arr.nbytes != dst.nbytes or not arr.flags.c_contiguous:
            raise ValueError("engine_upload needs a C-contiguous host array of the device array's size")
        _warn_if_staged_on_a_full_node(arr)
        L.check(self.lib.ma_engine_memcpy_h2d(self.handle, int(engine), dst.ptr, arr.ctypes.data, arr.nbytes))

    def engine_download(self, src, out, engine=L.MA_ENGINE_D2H):
        if out.nbytes != src.nbytes or not out.flags.c_contiguous:
            raise ValueError("engine_download needs a C-contiguous host array of the device array's size")
        _warn_if_staged_on_a_full_node(out)
        L.check(self.lib.ma_engine_memcpy_d2h(self.handle, int(engine), out.ctypes.data, src.ptr, out.nbytes))

    def engine_record(self, engine, ev):
        L.check(self.lib.ma_engine_record(self.handle, int(engine), ev))

    def engine_wait(self, engine, ev):
        L.check(self.lib.ma_engine_wait(self.handle, int(engine), ev))

    def engine_sync(self, engine):
        L.check(self.lib.ma_engine_sync(self.handle, int(engine)))

    def event_sync(self, ev):
        L.check(self.lib.ma_event_sync(self.handle, ev))

    def event_destroy(self, ev):
        if not self._closed:
            self.lib.ma_event_destroy(self.handle, ev)

    def clock_probe(self, milliseconds=20.0):
        """Sustained shader clock in GHz under a packed-FP32 load (ma_clock_probe)."""
        ghz = C.c_double()
        L.check(self.lib.ma_clock_probe(self.handle, float(milliseconds), C.byref(ghz)))
        return ghz.value

    def forget_host_arrays(self):
        """Drop every recorded host <-> device pair (Context.asdevice() uploads afresh); device memory held only by the
        pairs goes back to the pool."""
        self._resident.clear()

    def transfer_stats(self, reset=False):
        """(h2d_bytes, d2h_bytes) moved by this context's explicit host <-> device copies so far."""
        up, down = C.c_ulonglong(), C.c_ulonglong()
        L.check(self.lib.ma_ctx_transfer_stats(self.handle, C.byref(up), C.byref(down), int(bool(reset))))
        return up.value, down.value

    def close(self):
        if not self._closed:
            L.check(self.lib.ma_sync(self.handle))
            self.trim()
            # arrays that outlive the context: their HBM goes back to the driver now (they raise if used again)
            with self._host_lock:
                live, self._live = list(self._live), {}
                self._closed = True
            for p in live:
                self.lib.ma_free(self.handle, p)
            if getattr(self, "_zero_flags_ptr", None):
                self._zero_flags = None
                self.lib.ma_host_free(C.c_void_p(self._zero_flags_ptr))
                self._zero_flags_ptr = None
            if getattr(self, "_count_ptr", None):
                self._count_words = None
                self.lib.ma_host_free(C.c_void_p(self._count_ptr))
                self._count_ptr = None
            side, self._side = getattr(self, "_side", None), None
            if side is not None:
                side.close()
            self.lib.ma_ctx_destroy(self.handle)

    def side_context(self):
        """A second context on the same device, owned by this one (closed with it): an independent HIP stream, workspace and
        pools for work that may run beside this context's -- FeatureRegistrator puts the reference image's features of every
        level there while this one works through the moving image's coarse levels."""
        if getattr(self, "_side", None) is None:
            self._side = Context(self.device)
        return self._side

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- events / profiling --------------------------------------------------------------------
    def event(self):
        e = C.c_void_p()
        L.check(self.lib.ma_event_create(self.handle, C.byref(e)))
        return e

    def record(self, ev):
        L.check(self.lib.ma_event_record(self.handle, ev))

    def elapsed_ms(self, a, b):
        ms = C.c_float()
        L.check(self.lib.ma_event_elapsed_ms(self.handle, a, b, C.byref(ms)))
        return ms.value

    def profile(self, on=True):
        L.check(self.lib.ma_profile_enable(self.handle, int(on)))

    def profile_reset(self):
        L.check(self.lib.ma_profile_reset(self.handle))

    def profile_get(self):
        out = {}
        for name, kid in L.KERNEL_IDS.items():
            ms, n, px = C.c_double(), C.c_longlong(), C.c_double()
            L.check(self.lib.ma_profile_get(self.handle, kid, C.byref(ms), C.byref(n), C.byref(px)))
            out[name] = {"ms": ms.value, "launches": n.value, "px": px.value}
        return out

    # -- primitives (device in, device out) ---------------------------------------------------
    def farneback(self, prev, nxt, winsize, iterations, tile=0, overlap=0, poly_n=1, poly_sigma=1.7, fused=False):
        """cv2.calcOpticalFlowFarneback(prev, next, levels=0, GAUSSIAN) on the whole image (tile=0)
        or on TileFlowCalc's overlapping windows, stitched (flow_calc.py:59-98)."""
        if prev.shape != nxt.shape or prev.ndim != 2:
            raise ValueError("prev/next must be 2-D arrays of the same shape")
        if prev.dtype != nxt.dtype:
            # cv2.calcOpticalFlowFarneback converts each input to float32 on its own (exact for integers): a mixed pair
            # is the float32 pair
            prev, nxt = self.to_f32(prev), self.to_f32(nxt)
        H, W = prev.shape
        flow = self.empty((H, W, 2), np.float32)
        self._run(self.lib.ma_farneback_tiled, prev.ptr, nxt.ptr, _dt(prev.dtype), H, W, int(tile),
                                            int(overlap), int(winsize), int(iterations), int(poly_n),
                                            float(poly_sigma), L.MA_FB_MULADD_FUSED if fused else 0, flow.ptr)
        return flow

    def farneback_debug(self, prev, nxt, winsize, iterations, poly_sigma=1.7, fused=False):
        H, W = prev.shape
        flow = self.empty((H, W, 2), np.float32)
        r0, r1, m0 = (self.empty((5, H, W), np.float32) for _ in range(3))
        self._run(self.lib.ma_farneback_debug, prev.ptr, nxt.ptr, _dt(prev.dtype), H, W, int(winsize),
                                            int(iterations), float(poly_sigma),
                                            L.MA_FB_MULADD_FUSED if fused else 0, flow.ptr, r0.ptr, r1.ptr, m0.ptr)
        return flow, r0, r1, m0

    def optflow_register(self, ref, mov, num_pyr_lvl=4, num_iterations=3, tile_size=1000, overlap=100,
                         use_full_res_img=False, use_dog=False, fb_flags=0, dog_flags=0):
        """OptFlowRegistrator.register() as one C call (ma_optflow_register): device images in, (flow DeviceArray,
        [(factor, (h, w), mi_after, mi_before, accepted), ...]) out.  The flow is enqueued, not synchronised."""
        if ref.shape != mov.shape or ref.dtype != mov.dtype or ref.ndim != 2:
            raise ValueError("ref/mov must be 2-D images of equal shape and dtype")
        H, W = ref.shape
        prm = L.MaParams(int(num_pyr_lvl), int(num_iterations), int(tile_size), int(overlap), int(bool(use_full_res_img)),
                         int(bool(use_dog)), int(fb_flags), int(dog_flags))
        nrep_max = max(int(num_pyr_lvl), 0) + 1
        reps = (L.MaLevelReport * nrep_max)()
        n = C.c_int(0)
        flow = self.empty((H, W, 2), np.float32)
        self._run(self.lib.ma_optflow_register, ref.ptr, mov.ptr, _dt(ref.dtype), H, W, C.byref(prm), flow.ptr, reps,
                  nrep_max, C.byref(n))
        return flow, [(r.factor, (r.h, r.w), r.mi_after, r.mi_before, bool(r.accepted)) for r in reps[:n.value]]

    def remap(self, src, map_xy):
        """cv2.remap(src, map_xy, None, INTER_LINEAR)."""
        cn = 1 if src.ndim == 2 else src.shape[2]
        sh, sw = src.shape[:2]
        dh, dw = map_xy.shape[:2]
        if map_xy.dtype != np.float32 or map_xy.ndim != 3 or map_xy.shape[2] != 2:
            raise ValueError("map must be (h, w, 2) float32")
        dst = self.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), src.dtype)
        self._run(self.lib.ma_remap_bilinear, src.ptr, _dt(src.dtype), cn, sh, sw, map_xy.ptr, dh, dw,
                                           dst.ptr)
        return dst

    def warp(self, img, flow, tile, overlap, minmax=False, flow_cells=False):
        """Warper.warp() (warper.py:37-53).  minmax=True also leaves the output's (min, max) on the device
        (out.minmax) for a following dog_u8; flow_cells=True the per-cell maxima of `flow` (flow.cellkeys) for a
        following merge_flows (only where the tiling has cells: tile > 2*overlap > 0)."""
        H, W = img.shape
        if flow.shape != (H, W, 2) or flow.dtype != np.float32:
            raise ValueError(f"flow must be float32 of shape {(H, W, 2)}, got {flow.dtype} {flow.shape}")
        out = self.empty((H, W), img.dtype)
        if flow_cells and tile > 2 * overlap > 0:
            ncell = (2 * -(-W // tile) + 1) * (2 * -(-H // tile) + 1)
            keys = self._raw(L.MA_FLOW_CELL_REPLICAS * ncell * 4)
            if minmax:
                out.minmax = self.empty((2,), np.float32)
            self._run(self.lib.ma_warp_tiled_flowcells, img.ptr, _dt(img.dtype), H, W, flow.ptr, int(tile), int(overlap),
                      out.ptr, out.minmax.ptr if minmax else None, keys.ptr)
            flow.cellkeys = (int(tile), int(overlap), keys)
            return out
        if minmax:
            out.minmax = self.empty((2,), np.float32)
            self._run(self.lib.ma_warp_tiled_minmax, img.ptr, _dt(img.dtype), H, W, flow.ptr, int(tile),
                                                  int(overlap), out.ptr, out.minmax.ptr)
        else:
            self._run(self.lib.ma_warp_tiled, img.ptr, _dt(img.dtype), H, W, flow.ptr, int(tile),
                                           int(overlap), out.ptr)
        return out

    def warp_pages(self, pages, flow, tile, overlap, out=None):
        """warp_and_save_pages (__main__.py:288-302): warp every HOST page with one device-resident flow.
        pages: sequence of equal-shape C-contiguous numpy arrays; out: optional sequence of writable arrays of the
        same shape/dtype (e.g. memmap rows), allocated if None.  Returns `out`."""
        pages = [np.ascontiguousarray(p) for p in pages]
        if not pages:
            return []
        H, W = pages[0].shape
        dt = _dt(pages[0].dtype)
        if any(p.shape != (H, W) or p.dtype != pages[0].dtype for p in pages):
            raise ValueError("all pages must have the same shape and dtype")
        if flow.shape != (H, W, 2) or flow.dtype != np.float32:
            raise ValueError(f"flow must be float32 of shape {(H, W, 2)}, got {flow.dtype} {flow.shape}")
        if getattr(flow, "ctx", self) is not self:
            raise ValueError("flow belongs to another context")
        if out is None:
            out = [np.empty((H, W), pages[0].dtype) for _ in pages]
        if len(out) != len(pages) or any(o.shape != (H, W) or o.dtype != pages[0].dtype or not o.flags.c_contiguous
                                         or not o.flags.writeable for o in out):
            raise ValueError("out must hold one writable C-contiguous array of the page shape/dtype per page")
        n = len(pages)
        src = (C.c_void_p * n)(*[p.ctypes.data for p in pages])
        dst = (C.c_void_p * n)(*[o.ctypes.data for o in out])
        self._run(self.lib.ma_warp_pages_host, src, dst, n, dt, H, W, flow.ptr, int(tile), int(overlap))
        return out

    def merge_flows(self, flow1, flow2, tile, overlap):
        """_merge_flow_in_tiles (optflow_registrator.py:217-233)."""
        if flow1.shape != flow2.shape:
            raise ValueError("flows must have the same shape")
        H, W = flow1.shape[:2]
        out = self.empty((H, W, 2), np.float32)
        k1, k2 = flow1.cellkeys, flow2.cellkeys
        if k1 is not None and k2 is not None and k1[:2] == k2[:2] == (int(tile), int(overlap)):
            # both flows went through a warp that folded their cell maxima: no pass over the flows for the .max() tests
            self._run(self.lib.ma_merge_flows_tiled_cells, flow1.ptr, flow2.ptr, H, W, int(tile), int(overlap),
                      k1[2].ptr, k2[2].ptr, out.ptr)
        else:
            self._run(self.lib.ma_merge_flows_tiled, flow1.ptr, flow2.ptr, H, W, int(tile), int(overlap), out.ptr)
        return out

    def pyr_down(self, img, minmax=False):
        h, w = img.shape
        out = self.empty(((h + 1) // 2, (w + 1) // 2), img.dtype)
        if minmax:
            out.minmax = self.empty((2,), np.float32)
            self._run(self.lib.ma_pyr_down_minmax, img.ptr, _dt(img.dtype), h, w, out.ptr, out.minmax.ptr)
        else:
            self._run(self.lib.ma_pyr_down, img.ptr, _dt(img.dtype), h, w, out.ptr)
        return out

    def pyr_up_flow(self, flow, dst_hw, scale=1.0):
        """cv2.pyrUp(flow * scale, dstsize=(W, H)) with dst_hw = (H, W)."""
        h, w = flow.shape[:2]
        dh, dw = dst_hw
        out = self.empty((dh, dw, 2), np.float32)
        self._run(self.lib.ma_pyr_up_flow, flow.ptr, h, w, float(scale), out.ptr, int(dh), int(dw))
        return out

    def minmax(self, arr):
        mn, mx = C.c_double(), C.c_double()
        self._run(self.lib.ma_minmax, arr.ptr, _dt(arr.dtype), arr.size, C.byref(mn), C.byref(mx))
        return mn.value, mx.value

    def dog_u8(self, img, low_sigma=5, high_sigma=9, report_zero=False, flags=0):
        """Body of OptFlowRegistrator.dog -> uint8.  Stream ordered (no host sync) unless report_zero, in which
        case (out, src_max_is_zero) is returned; out is all zero when the input's max is 0.
        flags: rounding model of the chain (MA_DOG_FUSED_BLUR | MA_DOG_FUSED_SCALE, include/microaligner_hip.h)."""
        h, w = img.shape
        out = self.empty((h, w), np.uint8)
        if report_zero == "deferred":
            # no host sync: the flag lands in a page-locked word in stream order; any_deferred_zero() reads the words
            # handed out since its last call once the stream has been waited for (FeatureRegistrator's fast path)
            slot = self._zero_flag_slot()
            self._run(self.lib.ma_dog_u8_ex, img.ptr, _dt(img.dtype), h, w, int(low_sigma), int(high_sigma),
                      int(flags) | L.MA_DOG_REPORT_ASYNC, img.minmax.ptr if img.minmax is not None else None, out.ptr,
                      C.cast(C.c_void_p(self._zero_flags_ptr + 4 * slot), C.POINTER(C.c_int)))
            return out
        flag = C.c_int(0)
        fl = C.byref(flag) if report_zero else None
        # img.minmax: the producing kernel already reduced the image
        self._run(self.lib.ma_dog_u8_ex, img.ptr, _dt(img.dtype), h, w, int(low_sigma), int(high_sigma), int(flags),
                  img.minmax.ptr if img.minmax is not None else None, out.ptr, fl)
        return (out, bool(flag.value)) if report_zero else out

    _ZERO_FLAG_SLOTS = 1024

    def _zero_flag_slot(self):
        if getattr(self, "_zero_flags_ptr", None) is None:
            p = C.c_void_p()
            L.check(self.lib.ma_host_alloc(4 * self._ZERO_FLAG_SLOTS, C.byref(p)))
            self._zero_flags_ptr = p.value
            self._zero_flags = np.frombuffer((C.c_int * self._ZERO_FLAG_SLOTS).from_address(p.value), np.int32)
            self._zero_pending, self._zero_next, self._zero_carry = [], 0, False
        if len(self._zero_pending) >= self._ZERO_FLAG_SLOTS and self._settle_zero_flags():
            self._zero_carry = True             # ring full: what was pending is settled (synchronises) and remembered
        slot = self._zero_next
        self._zero_next = (slot + 1) % self._ZERO_FLAG_SLOTS
        self._zero_flags[slot] = 0              # not pending: no copy into it is in flight
        self._zero_pending.append(slot)
        return slot

    def _settle_zero_flags(self):
        pending = getattr(self, "_zero_pending", None)
        if not pending:
            return False
        self.sync()
        hit = bool(self._zero_flags[pending].any())
        pending.clear()
        return hit

    def any_deferred_zero(self):
        """True if any dog_u8(report_zero="deferred") since the last call saw an input whose max() is 0.  Waits for the
        stream (a no-op when the caller has just synchronised, e.g. by downloading scores)."""
        hit = self._settle_zero_flags() or getattr(self, "_zero_carry", False)
        self._zero_carry = False
        return hit

    def nmi_scores(self, a, b, chunk=0):
        if a.dtype != np.uint8 or b.dtype != np.uint8 or a.size != b.size:
            raise ValueError("NMI inputs must be uint8 arrays of equal size")
        n = a.size
        nch = 1 if (chunk <= 0 or chunk >= n) else (n + chunk - 1) // chunk
        scores = (C.c_double * nch)()
        got = C.c_int()
        self._run(self.lib.ma_nmi_u8, a.ptr, b.ptr, n, int(max(chunk, 0)), scores, nch, C.byref(got))
        return np.frombuffer(scores, dtype=np.float64, count=got.value).copy()

    def nmi_scores_pair(self, a, b0, b1, chunk=0):
        """Both halves of the gate in one call (ma_nmi_u8_pair): (scores of NMI(a, b0), scores of NMI(a, b1))."""
        for b in (b0, b1):
            if a.dtype != np.uint8 or b.dtype != np.uint8 or a.size != b.size:
                raise ValueError("NMI inputs must be uint8 arrays of equal size")
        n = a.size
        nch = 1 if (chunk <= 0 or chunk >= n) else (n + chunk - 1) // chunk
        s0, s1 = (C.c_double * nch)(), (C.c_double * nch)()
        got = C.c_int()
        self._run(self.lib.ma_nmi_u8_pair, a.ptr, b0.ptr, b1.ptr, n, int(max(chunk, 0)), s0, s1, nch, C.byref(got))
        return (np.frombuffer(s0, dtype=np.float64, count=got.value).copy(),
                np.frombuffer(s1, dtype=np.float64, count=got.value).copy())

    def max_project(self, stack):
        nz = stack.shape[0]
        out = self.empty(stack.shape[1:], stack.dtype)
        self._run(self.lib.ma_max_project, stack.ptr, _dt(stack.dtype), nz, out.size, out.ptr)
        return out

    def warp_affine(self, img, inverse_3x3):
        """skimage.transform.warp(img, AffineTransform(inverse_3x3), preserve_range=True).astype(img.dtype)."""
        h, w = img.shape
        m = (C.c_double * 9)(*[float(v) for v in np.asarray(inverse_3x3, dtype=np.float64).ravel()])
        out = self.empty((h, w), img.dtype)
        self._run(self.lib.ma_warp_affine, img.ptr, _dt(img.dtype), h, w, m, out.ptr)
        return out

    def warp_affine_cv(self, img, m2x3, dsize=None):
        """cv2.warpAffine(img, m2x3, dsize=(W, H)) with the default flags (bilinear, constant border 0)."""
        h, w = img.shape
        dw, dh = (w, h) if dsize is None else (int(dsize[0]), int(dsize[1]))
        m = np.asarray(m2x3, dtype=np.float64)
        if m.shape != (2, 3):
            raise ValueError("the transform must be a 2x3 matrix")
        mm = (C.c_double * 6)(*[float(v) for v in m.ravel()])
        out = self.empty((dh, dw), img.dtype)
        self._run(self.lib.ma_warp_affine_cv, img.ptr, _dt(img.dtype), h, w, mm, dh, dw, out.ptr)
        return out

    def knn2(self, query, train, mode="auto", stats=None, on_device=False):
        """Exact 2-NN (L2) of every row of `query` among the rows of `train`: (idx (n, 2) int64, dist (n, 2) float32)
        on the host, like feature_reg.sparse_cpu.knn2.  Either side may be a host array or a DeviceArray (descriptors
        that ma_daisy_describe left on the device are searched where they are).
        mode: "auto" | "exact" | "filtered" | "filtered_f32" (ma_knn2_l2_ex: matrix-core shortlist -- split-float16 or FP32 --
        + exact re-evaluation + certificate; the same result bit for bit); stats: a dict that receives {"uncertified": queries served by the exact fallback}.
        on_device: leave the results where they are -- (idx (n, 2) int32 bits, SQUARED dist (n, 2) float32) device buffers, what
        match_similarity takes."""
        modes = {"auto": L.MA_KNN_AUTO, "exact": L.MA_KNN_EXACT, "filtered": L.MA_KNN_FILTERED, "filtered_f32": L.MA_KNN_FILTERED_F32}
        if mode not in modes:
            raise ValueError(f"unknown search mode {mode!r}: auto, exact, filtered or filtered_f32")
        def prep(a):
            if isinstance(a, DeviceArray):
                if a.ndim != 2 or a.dtype != np.float32 or a.shape[1] % 4:
                    raise ValueError("device descriptors must be (n, dim) float32 with dim a multiple of 4")
                return a
            a = np.ascontiguousarray(a, np.float32)
            if a.ndim != 2:
                raise ValueError("query and train must be 2-D")
            pad = -a.shape[1] % 4
            return self.asdevice(np.pad(a, ((0, 0), (0, pad))) if pad else a)
        dq, dt = prep(query), prep(train)
        if dq.shape[1] != dt.shape[1]:
            raise ValueError("query and train must have the same descriptor length")
        nq = dq.shape[0]
        # (the image dtypes of asdevice() do not include int32: raw buffers for the results)
        idx, dist = self.empty((nq, 2), np.float32), self.empty((nq, 2), np.float32)
        unc = C.c_int(0)
        self._run(self.lib.ma_knn2_l2_ex, dq.ptr, nq, dt.ptr, dt.shape[0], dq.shape[1], idx.ptr, dist.ptr, modes[mode],
                  C.byref(unc) if stats is not None else None)
        if stats is not None:
            stats["uncertified"] = unc.value
        if on_device:
            return idx, dist
        out_i = np.empty((nq, 2), np.int32)
        L.check(self.lib.ma_memcpy_d2h(self.handle, out_i.ctypes.data, idx.ptr, out_i.nbytes))
        return out_i.astype(np.int64), np.sqrt(dist.numpy())   # the kernel returns squared distances

    _PCG64_STATES = {}

    def match_similarity(self, idx, dist_sq, query_pts, train_pts, ratio=0.5, confidence=0.99, reproj_threshold=3.0,
                         max_iters=2000, seed=0):
        """Ratio test + RANSAC similarity fit on the device (ma_match_similarity) over what knn2(on_device=True) left there:
        bit for bit feature_reg.sparse_cpu.estimate_affine_partial_2d(query points of the good matches, their train points).
        query_pts / train_pts: (n, 2) float64 device buffers (x, y).  Returns (matrix (2, 3) float64 or None, n_good, status);
        status as in include/microaligner_hip.h: 0 ok, 1 fewer than 3 good matches, 2 no model, 3 not computed (use the host)."""
        st = self._PCG64_STATES.get(seed)
        if st is None:     # numpy's seeding (SeedSequence -> PCG64) stays numpy's: only the stream is restated in C
            raw = np.random.PCG64(seed).state["state"]
            m64 = (1 << 64) - 1
            st = self._PCG64_STATES[seed] = (C.c_ulonglong * 4)(raw["state"] >> 64, raw["state"] & m64, raw["inc"] >> 64,
                                                                raw["inc"] & m64)
        nq = int(idx.nbytes // 8)             # (nq, 2) int32, typed (knn2) or raw (tests)
        mat = (C.c_double * 6)()
        n_good, status = C.c_int(0), C.c_int(0)
        self._run(self.lib.ma_match_similarity, idx.ptr, dist_sq.ptr, nq, query_pts.ptr, train_pts.ptr,
                  int(train_pts.nbytes // 16), float(ratio), float(confidence), float(reproj_threshold), int(max_iters), st, mat,
                  C.byref(n_good), C.byref(status))
        m = np.array(mat, np.float64).reshape(2, 3) if status.value == 0 else None
        return m, n_good.value, status.value

    def cut_tiles(self, img, tile, overlap, first_tile, n_tiles):
        """The zero-padded feature windows first_tile .. first_tile + n_tiles of a uint8 device image: (n, P, P)."""
        if img.dtype != np.uint8 or img.ndim != 2:
            raise ValueError("FAST works on uint8 images (the DOG output)")
        H, W = img.shape
        P = tile + 2 * overlap
        out = self.empty((n_tiles, P, P), np.uint8)
        self._run(self.lib.ma_cut_tiles_u8, img.ptr, H, W, int(tile), int(overlap), int(first_tile), int(n_tiles), out.ptr)
        return out

    def fast_keypoints(self, tiles, margin, limit, threshold=1):
        """FAST-9/16 corners of every tile interior, strongest `limit` first (row-major among equals), selected on the
        device: (counts (nt,), kp (nt, limit, 3) int32 = x, y, response; rows beyond counts[t] are undefined)."""
        nt, P, P2 = tiles.shape
        if P != P2 or tiles.dtype != np.uint8:
            raise ValueError("FAST works on square uint8 tiles (the DOG output)")
        out = self._raw(nt * limit * 3 * 4)
        counts = (C.c_int * nt)()
        self._run(self.lib.ma_fast_keypoints, tiles.ptr, nt, P, int(margin), int(threshold), int(limit), out.ptr, counts)
        host = np.empty((nt, limit, 3), np.int32)
        L.check(self.lib.ma_memcpy_d2h(self.handle, host.ctypes.data, out.ptr, host.nbytes))
        return np.frombuffer(counts, np.int32).copy(), host

    def _raw(self, nbytes):
        """Untyped HBM buffer from the pool (results that are not image dtypes: int32 scores, float64 points)."""
        return self.empty((max(int(nbytes), 1),), np.uint8)

    def _upload_raw(self, arr):
        arr = np.ascontiguousarray(arr)
        buf = self._raw(arr.nbytes)
        if arr.nbytes:
            L.check(self.lib.ma_memcpy_h2d(self.handle, buf.ptr, arr.ctypes.data, arr.nbytes))
        return buf

    def fast_nms(self, tiles, margin, threshold=1):
        """FAST-9/16 score at the 3x3 local maxima of every tile interior: (nt, P-2m, P-2m) int32 on the host."""
        nt, P, P2 = tiles.shape
        if P != P2 or tiles.dtype != np.uint8:
            raise ValueError("FAST works on square uint8 tiles (the DOG output)")
        Pi = P - 2 * margin
        out = self._raw(nt * Pi * Pi * 4)
        self._run(self.lib.ma_fast_nms, tiles.ptr, nt, P, int(margin), int(threshold), out.ptr)
        host = np.empty((nt, Pi, Pi), np.int32)
        L.check(self.lib.ma_memcpy_d2h(self.handle, host.ctypes.data, out.ptr, host.nbytes))
        return host

    def daisy_describe(self, tiles, kp_tile, kp_xy, weights, cos_sin, offsets, on_device=False):
        """DAISY descriptors at the given keypoints: (n, 200) float32 on the host, or left on the device (see
        ma_daisy_describe)."""
        nt, P, _ = tiles.shape
        n = len(kp_tile)
        d_tile = self._upload_raw(np.asarray(kp_tile, np.int32))
        d_xy = self._upload_raw(np.asarray(kp_xy, np.float64))
        desc = self.empty((n, 200), np.float32)
        halves = [np.ascontiguousarray(w, np.float64) for w in weights]
        wptr = (C.POINTER(C.c_double) * 3)(*[h.ctypes.data_as(C.POINTER(C.c_double)) for h in halves])
        radii = (C.c_int * 3)(*[len(h) - 1 for h in halves])
        cs = np.ascontiguousarray(cos_sin, np.float64)
        of = np.ascontiguousarray(offsets, np.float64)
        self._run(self.lib.ma_daisy_describe, tiles.ptr, _dt(tiles.dtype), nt, P, wptr, radii,
                  cs.ctypes.data_as(C.POINTER(C.c_double)), of.ctypes.data_as(C.POINTER(C.c_double)), d_tile.ptr, d_xy.ptr,
                  n, desc.ptr)
        return desc if on_device else desc.numpy()

    _COUNT_SLOTS = 256

    def _count_slot(self):
        """A page-locked int32 for a count that arrives in stream order (ma_feature_extract_enqueue): a view of one word of a
        small ring (a slot is read once, right after the event behind its call; 256 calls later it is handed out again)."""
        if getattr(self, "_count_ptr", None) is None:
            p = C.c_void_p()
            L.check(self.lib.ma_host_alloc(4 * self._COUNT_SLOTS, C.byref(p)))
            self._count_ptr = p.value
            self._count_words = np.frombuffer((C.c_int * self._COUNT_SLOTS).from_address(p.value), np.int32)
            self._count_next = 0
        k = self._count_next
        self._count_next = (k + 1) % self._COUNT_SLOTS
        self._count_words[k] = 0
        return self._count_words[k:k + 1], C.cast(C.c_void_p(self._count_ptr + 4 * k), C.POINTER(C.c_int))

    def feature_extract(self, img, tile, overlap, limit, weights, cos_sin, offsets, threshold=1, workspace_bytes=0, wait=True):
        """tile_registration.find_features of a uint8 device image in one call (ma_feature_extract): returns
        (descriptors (n, 200) float32 DeviceArray, points (n, 2) float64 raw device buffer, responses (n,) int32 raw device
        buffer, n); everything stays on the device, the keypoint count is the only thing that comes back.
        wait=False (ma_feature_extract_enqueue): nothing is waited for; n is a one-element int32 array on page-locked memory
        that holds the count once the stream has passed the call (the caller waits for an event recorded behind it), and
        the descriptor array keeps its capacity as its first dimension until then."""
        if img.dtype != np.uint8 or img.ndim != 2:
            raise ValueError("FAST works on uint8 images (the DOG output)")
        H, W = img.shape
        n_tiles = -(-H // tile) * -(-W // tile)
        cap = n_tiles * int(limit)
        desc = self.empty((cap, 200), np.float32)
        pts, resp = self._raw(cap * 16), self._raw(cap * 4)
        halves = [np.ascontiguousarray(w, np.float64) for w in weights]
        wptr = (C.POINTER(C.c_double) * 3)(*[h.ctypes.data_as(C.POINTER(C.c_double)) for h in halves])
        radii = (C.c_int * 3)(*[len(h) - 1 for h in halves])
        cs = np.ascontiguousarray(cos_sin, np.float64)
        of = np.ascontiguousarray(offsets, np.float64)
        if not wait:
            word, wp = self._count_slot()
            self._run(self.lib.ma_feature_extract_enqueue, img.ptr, H, W, int(tile), int(overlap), int(threshold), int(limit), wptr,
                      radii, cs.ctypes.data_as(C.POINTER(C.c_double)), of.ctypes.data_as(C.POINTER(C.c_double)),
                      int(workspace_bytes), cap, desc.ptr, pts.ptr, resp.ptr, wp)
            return desc, pts, resp, word
        n = C.c_int(0)
        self._run(self.lib.ma_feature_extract, img.ptr, H, W, int(tile), int(overlap), int(threshold), int(limit), wptr, radii,
                  cs.ctypes.data_as(C.POINTER(C.c_double)), of.ctypes.data_as(C.POINTER(C.c_double)), int(workspace_bytes), cap,
                  desc.ptr, pts.ptr, resp.ptr, C.byref(n))
        desc.shape = (n.value, 200)
        return desc, pts, resp, n.value

    def feature_round(self, current, current_gate, ref_gate, ref_features, tile, use_dog, nmi_chunk, weights, cos_sin, offsets,
                      workspace_bytes=0):
        """One round of FeatureRegistrator's level loop in one call (ma_feature_round).  current: the level's moving image
        (DeviceArray); current_gate: dog(current) or None (then it is computed and returned); ref_gate: dog(reference level);
        ref_features: (descriptors DeviceArray (n, 200), points raw buffer, n) of the reference level or None.  Returns a dict:
        estimate (2, 3), n_query, n_good, status, is_identity, zero_max, scores_after, scores_before (per chunk), current_gate,
        candidate, candidate_gate (the last two None for an identity estimate)."""
        H, W = current.shape
        made_gate = current_gate is None
        gate_out = self.empty((H, W), np.uint8) if made_gate else None
        cand, cand_gate = self.empty((H, W), current.dtype), self.empty((H, W), np.uint8)
        n = H * W
        nch = 1 if (nmi_chunk <= 0 or nmi_chunk >= n) else (n + nmi_chunk - 1) // nmi_chunk
        s_after, s_before = (C.c_double * nch)(), (C.c_double * nch)()
        halves = [np.ascontiguousarray(w, np.float64) for w in weights]
        wptr = (C.POINTER(C.c_double) * 3)(*[h.ctypes.data_as(C.POINTER(C.c_double)) for h in halves])
        radii = (C.c_int * 3)(*[len(h) - 1 for h in halves])
        cs = np.ascontiguousarray(cos_sin, np.float64)
        of = np.ascontiguousarray(offsets, np.float64)
        res = L.MaFeatureRoundResult()
        rdesc, rpts, rn = ref_features if ref_features is not None else (None, None, 0)
        self._run(self.lib.ma_feature_round, current.ptr, _dt(current.dtype), H, W,
                  None if made_gate else current_gate.ptr, gate_out.ptr if made_gate else None, ref_gate.ptr,
                  rdesc.ptr if rn else None, rpts.ptr if rn else None, int(rn), int(tile), 1 if use_dog else 0,
                  int(max(nmi_chunk, 0)), wptr, radii, cs.ctypes.data_as(C.POINTER(C.c_double)),
                  of.ctypes.data_as(C.POINTER(C.c_double)), int(workspace_bytes), cand.ptr, cand_gate.ptr, s_after, s_before, nch,
                  C.byref(res))
        ident = bool(res.is_identity)
        return dict(estimate=np.array(list(res.m2x3), np.float64).reshape(2, 3), n_query=res.n_query, n_good=res.n_good,
                    status=res.status, is_identity=ident, zero_max=res.zero_max,
                    scores_after=np.frombuffer(s_after, np.float64, res.n_scores).copy(),
                    scores_before=np.frombuffer(s_before, np.float64, res.n_scores).copy(),
                    current_gate=gate_out if made_gate else current_gate,
                    candidate=None if ident else cand, candidate_gate=None if ident else cand_gate)

    def download_raw(self, buf, shape, dtype):
        """A raw device buffer (_raw) as a host array of the given shape and dtype."""
        out = np.empty(shape, dtype)
        if out.nbytes:
            L.check(self.lib.ma_memcpy_d2h(self.handle, out.ctypes.data, buf.ptr, out.nbytes))
        return out

    def to_f32(self, arr):
        """Mat::convertTo(CV_32F): exact for the integer dtypes (ma_convert_f32); float32 arrays pass through."""
        if arr.dtype == np.float32:
            return arr
        out = self.empty(arr.shape, np.float32)
        self._run(self.lib.ma_convert_f32, arr.ptr, _dt(arr.dtype), arr.size, out.ptr)
        return out

    def normalize_minmax_u8(self, arr):
        out = self.empty(arr.shape, np.uint8)
        self._run(self.lib.ma_normalize_minmax_u8, arr.ptr, _dt(arr.dtype), arr.size, out.ptr)
        return out


_contexts = {}
_contexts_lock = threading.Lock()
_tls = threading.local()


class use_context:
    """`with use_context(ctx):` makes `ctx` the context get_context() returns on this thread.  Several contexts
    (each with its own HIP stream, workspace and buffer pool) can drive one device from different threads: that is
    how independent (ref, mov) pairs are kept in flight together (parallel.register_pairs(lanes=...))."""

    def __init__(self, ctx):
        self.ctx, self.prev = ctx, None

    def __enter__(self):
        self.prev = getattr(_tls, "ctx", None)
        _tls.ctx = self.ctx
        return self.ctx

    def __exit__(self, *exc):
        _tls.ctx = self.prev
        return False


def default_device():
    """Device index for this process: MICROALIGNER_DEVICE, else LOCAL_RANK (one process per GPU), else 0."""
    for var in ("MICROALIGNER_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(var)
        if v is not None and v != "":
            return int(v)
    return 0


def get_context(device=None):
    """Process-wide context of a device (created on first use).  Raises when no HIP device exists."""
    if device is None:
        cur = getattr(_tls, "ctx", None)
        if cur is not None:
            return cur
        device = default_device()
    with _contexts_lock:
        ctx = _contexts.get(device)
        if ctx is None:
            n = C.c_int()
            lib = L.load()
            lib.ma_device_count(C.byref(n))
            if n.value > 0 and not 0 <= device < n.value:
                # no silent wrap-around: a LOCAL_RANK beyond the visible devices would double-book a GPU
                raise ValueError(f"HIP device index {device} is out of range: {n.value} device(s) visible "
                                 "(MICROALIGNER_DEVICE / LOCAL_RANK select the device of this process)")
            ctx = _contexts[device] = Context(device)
    return ctx


def _parse_cpulist(text):
    """"0-63,128-191" -> sorted list of CPU indices (the format of sysfs cpulist files)."""
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return sorted(cpus)


def device_local_cpus(device=0, pci_bus_id=None, sysfs="/sys/bus/pci/devices"):
    """CPUs of the NUMA node a device hangs off (/sys/bus/pci/devices/<bdf>/local_cpulist), [] when unknown."""
    try:
        bdf = (pci_bus_id or device_info(device)["pci_bus_id"]).lower()
        with open(os.path.join(sysfs, bdf, "local_cpulist")) as f:
            return _parse_cpulist(f.read())
    except (OSError, ValueError, RuntimeError, KeyError):
        return []


def bind_to_device_numa(device=None, pci_bus_id=None, sysfs="/sys/bus/pci/devices"):
    """Pin this process (and every thread it starts afterwards) to the CPUs next to its GPU, so that the page-locked
    transfer buffers it allocates from now on and the pages its loaders first touch are on the memory the GPU reaches
    without crossing the socket interconnect: on a two-socket host a remote buffer costs a third of the PCIe rate
    (profiles/r04_notes.md).  One process per GPU: call it first thing in a rank.  Returns the CPU list it
    applied ([] when the topology is unknown or MICROALIGNER_BIND_NUMA=0: nothing changed); the previous mask is returned
    by os.sched_getaffinity before the call if the caller wants to restore it."""
    if os.environ.get("MICROALIGNER_BIND_NUMA", "1") == "0":
        return []
    cpus = device_local_cpus(default_device() if device is None else device, pci_bus_id, sysfs)
    allowed = os.sched_getaffinity(0)
    cpus = [c for c in cpus if c in allowed]
    if not cpus or set(cpus) == set(allowed):
        return []
    set_affinity(cpus)
    return cpus


def set_affinity(cpus):
    """Affinity of EVERY thread of this process (sched_setaffinity(0, ...) alone moves the calling thread only; the HIP
    runtime's helper threads, which carry the staged copies, exist already once a device has been queried)."""
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = [0]
    for tid in tids:
        try:
            # the library's staging-copy workers are pinned one per L3 domain (ma_api.hip CopyPool): they keep their
            # placement, restricted to the new set where the two intersect
            with open(f"/proc/self/task/{tid}/comm") as f:
                if f.read().startswith("ma-copy-"):
                    keep = os.sched_getaffinity(tid) & set(cpus)
                    if keep:
                        os.sched_setaffinity(tid, keep)
                        continue
        except OSError:
            pass
        try:
            os.sched_setaffinity(tid, cpus)
        except OSError:
            pass        # a thread that exited in the meantime
    os.sched_setaffinity(0, cpus)


def host_register(arr):
    """Page-lock `arr`'s memory in place (ma_host_register): transfers to and from it then go by DMA directly instead of
    through the staging chunks -- one pass over host DRAM per byte instead of three.  For arrays that live across many
    transfers (parallel.shared_array results, reused input buffers); registration itself costs about a first touch of
    every page.  Returns True when the range is page-locked now, False when nothing was registered: the runtime refused
    (no device in this process, pages that cannot be pinned -- a memmap of a file on disk), or `arr` is not an ndarray /
    memmap whose memory outlives the call (a list would be converted to a temporary, and a registration must never outlive
    its memory), or no finalizer can be attached to the object that owns the memory.  The array then goes through the
    staging path as before.  The registration ends with the array (weakref finalizer on its owner)."""
    import weakref
    if not isinstance(arr, np.ndarray) or arr.nbytes == 0 or not arr.flags.c_contiguous:
        return False
    base = arr
    while isinstance(getattr(base, "base", None), np.ndarray):     # the finalizer hangs on the object that owns the memory
        base = base.base
    lib = L.load()
    ptr = arr.ctypes.data
    if lib.ma_host_register(C.c_void_p(ptr), C.c_size_t(arr.nbytes)) != L.MA_OK:
        return False
    try:
        # (the finalizer must run BEFORE the memory is unmapped: it hangs on the owning ndarray / memmap, whose death precedes
        # the release of its buffer)
        weakref.finalize(base, lib.ma_host_unregister, C.c_void_p(ptr))
    except TypeError:          # not weak-referenceable: nothing would ever end the registration
        lib.ma_host_unregister(C.c_void_p(ptr))
        return False
    return True


def host_unregister(arr):
    """End a registration made by host_register(arr) now (its finalizer later finds nothing left to do)."""
    if isinstance(arr, np.ndarray) and arr.nbytes:
        return L.load().ma_host_unregister(C.c_void_p(arr.ctypes.data)) == L.MA_OK
    return False


def transfer_mode(arr):
    """How a blocking transfer to / from the whole of `arr` is carried out (ma_host_transfer_is_direct): "direct" -- page-locked
    over its full extent, DMA as it is; "transient" -- pageable, but the copy page-locks the range for its own duration (opt-in:
    MICROALIGNER_TRANSIENT_PIN=1, arrays of >= 64 MiB); "staged" -- through the page-locked ring."""
    a = np.asarray(arr)
    if a.nbytes == 0:
        return "staged"
    out = C.c_int(0)
    L.check(L.load().ma_host_transfer_is_direct(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), C.byref(out)))
    return {1: "direct", 2: "transient"}.get(out.value, "staged")


def transient_pin_stats():
    """dict(copies, slow_registrations, register_ms, gib, active) of the transient page-locking in this process
    (ma_transient_pin_stats)."""
    c, sl, ms, gib, act = C.c_longlong(), C.c_longlong(), C.c_double(), C.c_double(), C.c_int()
    L.check(L.load().ma_transient_pin_stats(C.byref(c), C.byref(sl), C.byref(ms), C.byref(gib), C.byref(act)))
    return dict(copies=c.value, slow_registrations=sl.value, register_ms=ms.value, gib=gib.value, active=bool(act.value))


def transfer_is_direct(arr):
    """Whether a transfer to / from the whole of `arr` goes by DMA as it is (page-locked over its full extent)."""
    return transfer_mode(arr) == "direct"


_STAGED_WARNED = [False]


def _warn_if_staged_on_a_full_node(arr):
    """beyond about three ranks per node the staged path (three to four passes over host DRAM per
    payload byte) cannot be the data plane.  Said once per process when a big pageable array reaches a transfer engine."""
    if _STAGED_WARNED[0] or arr.nbytes < (64 << 20):
        return
    try:
        ws = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    except ValueError:
        return
    if ws >= 3 and transfer_mode(arr) == "staged":
        import warnings
        _STAGED_WARNED[0] = True
        warnings.warn(f"microaligner_amd: a pageable {arr.nbytes >> 20} MiB array is going through the staged copy path with "
                      f"{ws} ranks on this node; the host's memory bandwidth will not carry that for every rank -- page-lock "
                      "buffers that are reused (device.host_register, parallel.shared_array, Context.host_empty)",
                      RuntimeWarning, stacklevel=3)


def device_count():
    n = C.c_int()
    L.load().ma_device_count(C.byref(n))
    return n.value


def device_info(device=0):
    """dict(name, pci_bus_id, mem_free, mem_total, compute_units) of a device (ma_device_info)."""
    name, pci = C.create_string_buffer(256), C.create_string_buffer(64)
    free, total, cus = C.c_size_t(), C.c_size_t(), C.c_int()
    L.check(L.load().ma_device_info(int(device), name, 256, pci, 64, C.byref(free), C.byref(total), C.byref(cus)))
    return dict(name=name.value.decode(), pci_bus_id=pci.value.decode(), mem_free=free.value, mem_total=total.value,
                compute_units=cus.value)
