"""ctypes binding of the CPU oracle (oracle/ma_oracle.c).

TEST INFRASTRUCTURE, not product code: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.  The functions carry
the names/argument meaning of the third-party calls the reference makes
(cv2.calcOpticalFlowFarneback, cv2.remap, ... -- see ma_oracle.c header), so a
`cv2` stand-in for driving the reference's own orchestration is a thin shim
(tests/golden/make_golden.py).

Parity status: unpinned for the OpenCV primitives, pinned (against the
installed scikit-learn) for NMI.  See DESIGN.md section "Oracle".
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libma_oracle.so")

U8, U16, F32 = 0, 1, 2
_DT = {np.dtype(np.uint8): U8, np.dtype(np.uint16): U16, np.dtype(np.float32): F32}


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "ma_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libma_oracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_farneback.restype = C.c_int
        _lib.orc_farneback.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
        _lib.orc_farneback_batch.restype = C.c_int
        _lib.orc_farneback_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                             C.c_int]
        _lib.orc_remap_bilinear.restype = C.c_int
        _lib.orc_remap_bilinear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_int, C.c_int, C.c_void_p]
        _lib.orc_pyr_down.restype = C.c_int
        _lib.orc_pyr_down.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib.orc_pyr_up_f32.restype = C.c_int
        _lib.orc_pyr_up_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        _lib.orc_normalize_minmax_to_f32.restype = C.c_int
        _lib.orc_normalize_minmax_to_f32.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_double, C.c_double,
                                                     C.c_void_p]
        _lib.orc_normalize_minmax_f32_to_u8.restype = C.c_int
        _lib.orc_normalize_minmax_f32_to_u8.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.orc_gaussian_blur_f32.restype = C.c_int
        _lib.orc_gaussian_blur_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
        _lib.orc_gaussian_kernel.restype = None
        _lib.orc_gaussian_kernel.argtypes = [C.c_int, C.c_double, C.c_void_p]
        _lib.orc_dog_u8.restype = C.c_int
        _lib.orc_dog_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib.orc_normalize_minmax_to_f32_ex.restype = C.c_int
        _lib.orc_normalize_minmax_to_f32_ex.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_double, C.c_double, C.c_int,
                                                        C.c_void_p]
        _lib.orc_normalize_minmax_f32_to_u8_ex.restype = C.c_int
        _lib.orc_normalize_minmax_f32_to_u8_ex.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        _lib.orc_gaussian_blur_f32_ex.restype = C.c_int
        _lib.orc_gaussian_blur_f32_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]
        _lib.orc_dog_u8_ex.restype = C.c_int
        _lib.orc_dog_u8_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib.orc_nmi_u8.restype = C.c_int
        _lib.orc_nmi_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_double)]
        _lib.orc_nmi_u8_chunks.restype = C.c_int
        _lib.orc_nmi_u8_chunks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int]
        _lib.orc_set_threads.restype = None
        _lib.orc_set_threads.argtypes = [C.c_int]
        _lib.orc_get_threads.restype = C.c_int
        _lib.orc_farneback_window_kernel.restype = None
        _lib.orc_farneback_window_kernel.argtypes = [C.c_int, C.c_void_p]
        _lib.orc_farneback_prepare_gaussian.restype = C.c_int
        _lib.orc_farneback_prepare_gaussian.argtypes = [C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                                        C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                        C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib.orc_remap_tables.restype = None
        _lib.orc_remap_tables.argtypes = [C.c_void_p, C.c_void_p]
        _lib.orc_warp_affine_cv.restype = C.c_int
        _lib.orc_warp_affine_cv.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                            C.c_void_p]
    return _lib


def set_threads(n):
    """Host threads for the row / chunk / window loops of the oracle (results do not depend on it)."""
    lib().orc_set_threads(int(n))


def get_threads():
    return lib().orc_get_threads()


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with status {rc}")


def _img(a):
    a = np.ascontiguousarray(a)
    if a.dtype not in _DT:
        raise ValueError(f"oracle supports uint8/uint16/float32 images, got {a.dtype}")
    return a, _DT[a.dtype]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# --- cv2.calcOpticalFlowFarneback(prev, next, None, 0.5, 0, win, iters, poly_n, poly_sigma, GAUSSIAN) ---
def calc_optical_flow_farneback(prev, nxt, winsize, iterations, poly_n=1, poly_sigma=1.7, fused=False,
                                dump=False):
    prev, dt = _img(prev)
    nxt, dt2 = _img(nxt)
    if dt != dt2:  # OpenCV converts each image to float32 on its own (convertTo): mixed dtypes are legal
        prev, nxt, dt = prev.astype(np.float32), nxt.astype(np.float32), F32
    if prev.shape != nxt.shape or prev.ndim != 2:
        raise ValueError("prev/next must be 2-D and of the same shape")
    h, w = prev.shape
    flow = np.empty((h, w, 2), np.float32)
    if dump:
        r0 = np.empty((h, w, 5), np.float32)
        r1 = np.empty((h, w, 5), np.float32)
        m0 = np.empty((h, w, 5), np.float32)
        _check(lib().orc_farneback(_p(prev), _p(nxt), dt, h, w, winsize, iterations, poly_n, poly_sigma,
                                   int(fused), _p(flow), _p(r0), _p(r1), _p(m0)), "farneback")
        return flow, r0, r1, m0
    _check(lib().orc_farneback(_p(prev), _p(nxt), dt, h, w, winsize, iterations, poly_n, poly_sigma,
                               int(fused), _p(flow), None, None, None), "farneback")
    return flow


def farneback_batch(prev_tiles, next_tiles, winsize, iterations, poly_n=1, poly_sigma=1.7, fused=False,
                    nthreads=1):
    """prev_tiles/next_tiles: (n, h, w) arrays.  Returns (n, h, w, 2) float32."""
    prev_tiles, dt = _img(prev_tiles)
    next_tiles, dt2 = _img(next_tiles)
    if dt != dt2:
        prev_tiles, next_tiles, dt = prev_tiles.astype(np.float32), next_tiles.astype(np.float32), F32
    n, h, w = prev_tiles.shape
    flow = np.empty((n, h, w, 2), np.float32)
    _check(lib().orc_farneback_batch(_p(prev_tiles), _p(next_tiles), dt, n, h, w, winsize, iterations,
                                     poly_n, poly_sigma, int(fused), _p(flow), nthreads), "farneback_batch")
    return flow


# --- cv2.remap(src, map_xy, None, INTER_LINEAR) ---
def remap(src, map_xy):
    src, dt = _img(src)
    map_xy = np.ascontiguousarray(map_xy, dtype=np.float32)
    cn = 1 if src.ndim == 2 else src.shape[2]
    sh, sw = src.shape[:2]
    dh, dw = map_xy.shape[:2]
    dst = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), src.dtype)
    _check(lib().orc_remap_bilinear(_p(src), dt, cn, sh, sw, _p(map_xy), dh, dw, _p(dst)), "remap")
    return dst


# --- cv2.warpAffine(src, M, dsize=(W, H)) with default flags (INTER_LINEAR, BORDER_CONSTANT 0) ---
def warp_affine(src, M, dsize=None):
    src, dt = _img(src)
    if src.ndim != 2:
        raise ValueError("warp_affine: 2-D images only")
    M = np.ascontiguousarray(M, dtype=np.float64)
    if M.shape != (2, 3):
        raise ValueError("M must be a 2x3 matrix")
    sh, sw = src.shape
    dw, dh = (sw, sh) if dsize is None else dsize
    dst = np.empty((dh, dw), src.dtype)
    _check(lib().orc_warp_affine_cv(_p(src), dt, sh, sw, _p(M), dh, dw, _p(dst)), "warp_affine")
    return dst


# --- cv2.pyrDown(img) / cv2.pyrUp(img, dstsize=(W, H)) ---
def pyr_down(img):
    img, dt = _img(img)
    h, w = img.shape
    dst = np.empty(((h + 1) // 2, (w + 1) // 2), img.dtype)
    _check(lib().orc_pyr_down(_p(img), dt, h, w, _p(dst)), "pyr_down")
    return dst


def pyr_up(img, dstsize=None):
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w = img.shape[:2]
    cn = 1 if img.ndim == 2 else img.shape[2]
    dw, dh = dstsize if dstsize is not None else (w * 2, h * 2)
    dst = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.float32)
    _check(lib().orc_pyr_up_f32(_p(img), cn, h, w, _p(dst), dh, dw), "pyr_up")
    return dst


# --- cv2.normalize(.., NORM_MINMAX, ..) / cv2.GaussianBlur ---
# Rounding models of the dog() chain (ma_oracle.c, ORC_DOG_*): 0 = multiply then add (OpenCV's SSE2 baseline objects),
# DOG_FUSED_BLUR | DOG_FUSED_SCALE = fused multiply-adds (the AVX2 + FMA3 objects an x86-64 host dispatches to).
DOG_FUSED_BLUR, DOG_FUSED_SCALE = 1, 2
DOG_FUSED = DOG_FUSED_BLUR | DOG_FUSED_SCALE


def normalize_minmax_f32(img, alpha=0.0, beta=1.0, fused=False):
    img, dt = _img(img)
    dst = np.empty(img.shape, np.float32)
    _check(lib().orc_normalize_minmax_to_f32_ex(_p(img), dt, img.size, alpha, beta, int(bool(fused)), _p(dst)), "normalize")
    return dst


def normalize_minmax_u8(img, fused=False):
    img = np.ascontiguousarray(img, dtype=np.float32)
    dst = np.empty(img.shape, np.uint8)
    _check(lib().orc_normalize_minmax_f32_to_u8_ex(_p(img), img.size, int(bool(fused)), _p(dst)), "normalize_u8")
    return dst


def gaussian_blur(img, ksize, sigma, fused=False):
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w = img.shape
    dst = np.empty_like(img)
    _check(lib().orc_gaussian_blur_f32_ex(_p(img), h, w, ksize, float(sigma), int(bool(fused)), _p(dst)), "gaussian_blur")
    return dst


def gaussian_kernel(ksize, sigma):
    k = np.empty(ksize, np.float32)
    lib().orc_gaussian_kernel(ksize, float(sigma), _p(k))
    return k


def dog(img, use_it=True, low_sigma=5, high_sigma=9, flags=0):
    """OptFlowRegistrator.dog (optflow_registrator.py:249-274) incl. the max()==0 shortcut.  flags: DOG_FUSED_*."""
    if not use_it:
        return img
    if img.max() == 0:
        return img
    img, dt = _img(img)
    h, w = img.shape
    dst = np.empty((h, w), np.uint8)
    _check(lib().orc_dog_u8_ex(_p(img), dt, h, w, low_sigma, high_sigma, int(flags), _p(dst)), "dog")
    return dst


# --- sklearn.metrics.normalized_mutual_info_score on u8 labels ---
def nmi_u8(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8).ravel()
    b = np.ascontiguousarray(b, dtype=np.uint8).ravel()
    if a.size != b.size:
        raise ValueError("size mismatch")
    out = C.c_double()
    _check(lib().orc_nmi_u8(_p(a), _p(b), a.size, C.byref(out)), "nmi")
    return out.value


def nmi_u8_chunks(a, b, chunk):
    """One NMI score per contiguous `chunk`-element piece of the flattened arrays (mi_tiled's loop)."""
    a = np.ascontiguousarray(a, dtype=np.uint8).ravel()
    b = np.ascontiguousarray(b, dtype=np.uint8).ravel()
    if a.size != b.size:
        raise ValueError("size mismatch")
    nch = -(-a.size // chunk)
    scores = np.empty(nch, np.float64)
    _check(lib().orc_nmi_u8_chunks(_p(a), _p(b), a.size, chunk, _p(scores), nch), "nmi_chunks")
    return scores


def farneback_window_kernel(winsize):
    k = np.empty(winsize // 2 + 1, np.float32)
    lib().orc_farneback_window_kernel(winsize, _p(k))
    return k


def farneback_polyexp_constants(n=1, sigma=1.7):
    buf = np.zeros((3, 2 * n + 1), np.float32)
    ig = [C.c_double() for _ in range(4)]
    base = buf.ctypes.data
    stride = (2 * n + 1) * 4
    _check(lib().orc_farneback_prepare_gaussian(n, sigma, C.c_void_p(base + n * 4),
                                                C.c_void_p(base + stride + n * 4),
                                                C.c_void_p(base + 2 * stride + n * 4),
                                                *[C.byref(v) for v in ig]), "prepare_gaussian")
    return buf[0], buf[1], buf[2], tuple(v.value for v in ig)


def remap_tables():
    tf = np.empty((1024, 4), np.float32)
    ti = np.empty((1024, 4), np.int16)
    lib().orc_remap_tables(_p(tf), _p(ti))
    return tf, ti
