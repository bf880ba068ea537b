"""THROW-AWAY stand-in for cv2 that forwards to the CPU oracle.  Its ONLY purpose is to exercise the plumbing of
tests/golden/make_cv2_golden.py and tests/test_cv2_golden.py in an image without OpenCV (tests/test_cv2_golden.py::
test_generator_and_consumers_work_end_to_end_with_a_standin puts this directory on PYTHONPATH for one subprocess).  A file
made with it pins nothing -- it IS the oracle -- and says so (meta["standin"] = true); never commit one as cv2_4.5.5.npz."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import oracle as _O  # noqa: E402

__microaligner_standin__ = True
__version__ = "4.5.5"
INTER_LINEAR, INTER_CUBIC, NORM_MINMAX, CV_32F, CV_8U, OPTFLOW_FARNEBACK_GAUSSIAN = 1, 2, 32, 5, 0, 256
CPU_SSE2, CPU_AVX2, CPU_FMA3 = 2, 11, 12


def getBuildInformation():
    return "General configuration for OpenCV 4.5.5 (STAND-IN forwarding to oracle/ma_oracle.c)\n  Version control: none\n"


def checkHardwareSupport(feature):
    return False


def getNumThreads():
    return 1


def calcOpticalFlowFarneback(prev, next, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags):
    assert flow is None and levels == 0 and flags == OPTFLOW_FARNEBACK_GAUSSIAN and pyr_scale == 0.5
    return _O.calc_optical_flow_farneback(prev, next, winsize, iterations, poly_n, poly_sigma)


def remap(src, map1, map2, interpolation):
    m = map1 if map2 is None else np.stack([map1, map2], -1)     # (cubic requests are served bilinearly: inputs only)
    return _O.remap(src, np.ascontiguousarray(m, dtype=np.float32))


def pyrDown(img):
    return _O.pyr_down(img)


def pyrUp(img, dstsize=None):
    return _O.pyr_up(img, dstsize)


def GaussianBlur(img, ksize, sigmaX, dst=None, sigmaY=0):
    k = ksize[0] if ksize[0] > 0 else (int(round(sigmaX * 4 * 2 + 1)) | 1)
    return _O.gaussian_blur(np.ascontiguousarray(img, dtype=np.float32), k, sigmaX)


def normalize(src, dst, alpha, beta, norm_type, dtype):
    assert dst is None and norm_type == NORM_MINMAX
    if dtype == CV_32F:
        return _O.normalize_minmax_f32(src, alpha, beta)
    assert (alpha, beta) == (0, 255)
    return _O.normalize_minmax_u8(src)


def warpAffine(src, M, dsize):
    return _O.warp_affine(src, M, dsize=dsize)
