"""The primitives oracle/register_oracle.py composes, served by a LIVE OpenCV instead of the C restatement.

TEST INFRASTRUCTURE (like everything under oracle/): used by tests/test_cv2_parity.py and by bench.py's `cpu_baseline`
leg ("plan A": the reference's orchestration over the real cv2, fanned out the way the reference
fans out, shared_modules/utils.py:117-119, optflow_reg/flow_calc.py:93-97).  Nothing under microaligner_amd/ imports it.

    import cv2                                   # raises ImportError where there is none (this image, the GPU pool)
    from oracle import cv2_backend
    flow, reports, warped = cv2_backend.register_over_cv2(ref, mov, workers=64, **params)

The cv2 calls are written exactly as the reference writes them (file:line beside each).
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import cv2

from . import oracle as O
from . import register_oracle as RO


def farneback(prev, nxt, winsize, iterations):
    # optflow_reg/flow_calc.py:33-44
    return cv2.calcOpticalFlowFarneback(prev, nxt, None, pyr_scale=0.5, levels=0, winsize=winsize, iterations=iterations,
                                        poly_n=1, poly_sigma=1.7, flags=cv2.OPTFLOW_FARNEBACK_GAUSSIAN)


def dog(img, low_sigma=5, high_sigma=9):
    # optflow_reg/optflow_registrator.py:249-274
    if img.max() == 0:
        return img
    fimg = cv2.normalize(img, None, 0, 1, cv2.NORM_MINMAX, cv2.CV_32F)
    ks = (low_sigma * 4 * 2 + 1, low_sigma * 4 * 2 + 1)
    ls = cv2.GaussianBlur(fimg, ks, sigmaX=low_sigma, dst=None, sigmaY=low_sigma)
    hs = cv2.GaussianBlur(fimg, ks, sigmaX=high_sigma, dst=None, sigmaY=high_sigma)
    return cv2.normalize(hs - ls, None, 0, 255, cv2.NORM_MINMAX, cv2.CV_8U)


class Cv2Prims:
    """Drop-in for the module `oracle.oracle` inside register_oracle (same names, same argument meaning)."""
    workers = 1          # windows of a level in flight at once (the reference: dask scheduler="processes")

    @staticmethod
    def set_threads(n):
        pass

    @staticmethod
    def calc_optical_flow_farneback(prev, nxt, winsize, iterations, fused=False):
        return farneback(prev, nxt, winsize, iterations)

    @classmethod
    def farneback_batch(cls, prev_tiles, next_tiles, winsize, iterations, fused=False, nthreads=1):
        # flow_calc.py:86-97: one task per window.  cv2 releases the GIL inside the call, so threads fan out as the
        # reference's processes do, without pickling 1200 x 1200 windows back and forth.
        n = max(1, min(cls.workers, len(prev_tiles)))
        if n == 1:
            return np.stack([farneback(p, q, winsize, iterations) for p, q in zip(prev_tiles, next_tiles)])
        with ThreadPoolExecutor(n) as ex:
            return np.stack(list(ex.map(lambda pq: farneback(pq[0], pq[1], winsize, iterations), zip(prev_tiles, next_tiles))))

    @staticmethod
    def remap(src, m):
        # optflow_reg/warper.py:65, optflow_registrator.py:45
        return cv2.remap(src, np.ascontiguousarray(m, dtype=np.float32), None, cv2.INTER_LINEAR)

    pyr_down = staticmethod(lambda img: cv2.pyrDown(img))                                  # optflow_registrator.py:194
    pyr_up = staticmethod(lambda img, dstsize=None: cv2.pyrUp(img, dstsize=dstsize))       # :140,150,164,169,212,214

    @staticmethod
    def dog(img, use_it=True, low_sigma=5, high_sigma=9, flags=0):
        return dog(img, low_sigma, high_sigma) if use_it else img

    @staticmethod
    def nmi_u8(a, b):
        # shared_modules/similarity_scoring.py:36,44
        try:
            from sklearn.metrics import normalized_mutual_info_score
            return float(normalized_mutual_info_score(np.ravel(a), np.ravel(b)))
        except ImportError:
            return O.nmi_u8(a, b)

    @classmethod
    def nmi_u8_chunks(cls, a, b, chunk):
        fa, fb = np.ravel(a), np.ravel(b)
        spans = range(0, fa.size, chunk)
        n = max(1, min(cls.workers, len(spans)))
        if n == 1:
            return np.array([cls.nmi_u8(fa[i:i + chunk], fb[i:i + chunk]) for i in spans])
        with ThreadPoolExecutor(n) as ex:
            return np.array(list(ex.map(lambda i: cls.nmi_u8(fa[i:i + chunk], fb[i:i + chunk]), spans)))


def register_over_cv2(ref, mov, workers=None, stage_seconds=None, **params):
    """register() + warp() of the oracle ORCHESTRATION (pinned by fixtures made with the reference's own classes,
    tests/golden/make_golden.py) over the live cv2.  Returns (flow, reports, warped)."""
    Cv2Prims.workers = int(workers or os.cpu_count() or 1)
    saved = RO.O
    RO.O = Cv2Prims
    try:
        flow, reports = RO.register(ref, mov, stage_seconds=stage_seconds, **params)
        with RO._stage(stage_seconds, "final_warp"):
            warped = RO.warp(mov, flow, params.get("tile_size", 1000), params.get("overlap", 100))
    finally:
        RO.O = saved
    return flow, reports, warped


def build_summary():
    """The lines of cv2.getBuildInformation() that decide the rounding model (dispatch, IPP) + version + thread count."""
    info = cv2.getBuildInformation()
    keep = [ln.strip() for ln in info.splitlines() if any(k in ln for k in ("Version", "CPU/HW", "Baseline", "Dispatched", "IPP", "Parallel framework"))]
    return {"version": cv2.__version__, "threads": cv2.getNumThreads(), "build": keep[:12],
            "standin": bool(getattr(cv2, "__microaligner_standin__", False))}
