"""Coarse-to-fine optical-flow registration on MI355X.

Counterpart of microaligner/optflow_reg/optflow_registrator.py (class OptFlowRegistrator :50,
register :93-173, _generate_img_pyr :175-202, _upscale_flow_to_full_res :204-215,
_merge_list_of_flows :235-240, dog :249-274, merge_two_flows :37-47).  Same attributes, same
defaults, same return value -- an (H, W, 2) float32 flow with mov(p) ~ ref(p + flow(p)) -- but every
array stays in HBM between the steps of a level and every step is a HIP kernel pipeline
(include/microaligner_hip.h).  The level loop reproduces the reference's accept/reject rules,
including its quirks (Q1 absolute-coordinate flow merge, Q2 no x2 when upscaling to
full resolution, Q3 x4 in the middle-level reject branch).
"""
from dataclasses import dataclass
from math import log2
from typing import List, Optional, Tuple

import numpy as np

from .. import _lib as L
from ..device import DeviceArray, get_context
from ..shared_modules.img_checks import check_img_dims_match, check_img_is_2d_grey, check_img_is_provided
from ..shared_modules.similarity_scoring import mi_tiled
from .flow_calc import TileFlowCalc
from .warper import Warper


def merge_two_flows(flow1, flow2):
    """optflow_registrator.py:37-47 for one pair of (window) flows:
    flow2 if flow1.max()==0, flow1 if flow2.max()==0, else flow1 + cv2.remap(flow2, -flow1)."""
    ctx = get_context()
    f1, f2 = ctx.asdevice(flow1), ctx.asdevice(flow2)
    out = ctx.merge_flows(f1, f2, 0, 0)  # tile=0: the arrays themselves are the window
    return out if isinstance(flow1, DeviceArray) else out.numpy()


@dataclass
class LevelReport:
    """What the reference only prints: one record per pyramid level."""
    factor: int
    shape: Tuple[int, int]
    mi_after: float
    mi_before: float
    accepted: bool


class OptFlowRegistrator:
    def __init__(self):
        self._ref_img = np.array([])
        self._mov_img = np.array([])
        self.num_pyr_lvl = 4
        self.num_iterations = 3
        self.tile_size = 1000
        self.overlap = 100
        self.use_full_res_img = False
        self.use_dog = False
        # additions (defaults keep the reference's behaviour)
        self.verbose = True            # the reference prints per-level progress
        self.muladd_fused = False      # window blur with FMA (see MA_FB_MULADD_FUSED)
        # dog() chain (GaussianBlur, normalize) with fused multiply-adds: what OpenCV's AVX2 + FMA3 objects compute
        # (MA_DOG_FUSED_BLUR | MA_DOG_FUSED_SCALE); default: the SSE2 baseline arithmetic
        self.dog_muladd_fused = False
        # "c": the level loop runs inside the library (ma_optflow_register, one C call); "python": the same loop stated
        # here over the primitive entry points (the second implementation the tests compare the first with, and the one
        # that serves the two input classes the C entry point does not model: reference and moving image of different
        # dtypes, and float images whose max() is 0 without being all zero)
        self.engine = "c"
        # True (default, the reference's behaviour): the mov_img GETTER returns the REFERENCE image, as the reference's does
        # (optflow_registrator.py:73-74, quirk Q4 -- nothing on the path reads it); False: it returns the moving image
        self.compat_mov_getter = True
        self.level_reports: List[LevelReport] = []
        self._warper = Warper()
        self._tile_flow_calc = TileFlowCalc()

    # -- inputs ------------------------------------------------------------------------------
    @property
    def ref_img(self):
        return self._ref_img

    @ref_img.setter
    def ref_img(self, img):
        check_img_is_2d_grey(img, "ref")
        self._ref_img = img

    @property
    def mov_img(self):
        # the reference's getter returns the *reference* image (optflow_registrator.py:73-74, quirk Q4): so does this one
        # unless compat_mov_getter is switched off
        return self._ref_img if self.compat_mov_getter else self._mov_img

    @mov_img.setter
    def mov_img(self, img):
        check_img_is_2d_grey(img, "mov")
        self._mov_img = img

    def _log(self, *args):
        if self.verbose:
            print(*args)

    def _init_warper(self):
        self._warper = Warper()
        self._warper.tile_size = self.tile_size
        self._warper.overlap = self.overlap

    def _init_tile_flow_calc(self):
        fc = self._tile_flow_calc = TileFlowCalc()
        fc.tile_size = self.tile_size
        fc.overlap = self.overlap
        fc.num_iter = self.num_iterations
        fc.win_size = self.overlap - (1 - self.overlap % 2)  # largest odd number <= overlap (:91)
        fc.muladd_fused = self.muladd_fused

    def _warp(self, img, flow):
        # same call as Warper.warp() on device arrays; the kernel also leaves the output's min / max on the device,
        # which the dog() of the warped image would otherwise have to reduce in a pass of its own
        # ... and the per-cell maxima of the flow it reads, which a following merge of that flow uses for its
        # per-window flow.max() == 0 tests instead of a pass of its own over both flows
        return self._ctx.warp(img, flow, self._warper.tile_size, self._warper.overlap, minmax=True, flow_cells=True)

    # -- the hot path -----------------------------------------------------------------------
    def register(self):
        check_img_is_provided(self._ref_img, "ref")
        check_img_is_provided(self._mov_img, "mov")
        check_img_dims_match(self._ref_img, self._mov_img)
        device_in = isinstance(self._ref_img, DeviceArray) and isinstance(self._mov_img, DeviceArray)
        ctx = get_context()
        self._ctx = ctx
        self._init_tile_flow_calc()
        self._init_warper()
        self.level_reports = []

        ref_full, mov_full = ctx.asdevice(self._ref_img), ctx.asdevice(self._mov_img)
        self._full_shape = ref_full.shape
        if self.engine not in ("c", "python"):
            raise ValueError(f"unknown engine {self.engine!r}: 'c' or 'python'")
        reports = None
        # images of different dtypes (each keeps its own pyramid arithmetic, cv2 converts per input): the Python loop
        if self.engine == "c" and ref_full.dtype == mov_full.dtype:
            try:
                result, reports = ctx.optflow_register(
                    ref_full, mov_full, self.num_pyr_lvl, self.num_iterations, self.tile_size, self.overlap,
                    self.use_full_res_img, self.use_dog, L.MA_FB_MULADD_FUSED if self.muladd_fused else 0,
                    self._dog_flags())
            except ValueError as e:
                if "max() == 0" not in str(e):
                    raise
                # a float image with max() == 0 that is not all zero: dog() returns it unchanged in the reference
                # (:256-257); the loop below follows it there
        if reports is not None:
            for factor, shape, after, before, accepted in reports:
                self._log("Pyramid factor", factor)
                self._log("    MI score after:", after, "| MI score before:", before)
                self._log("    Better alignment than before" if accepted else "    Worse alignment than before")
                self.level_reports.append(LevelReport(factor, tuple(shape), float(after), float(before), accepted))
            self._ctx = None
            return result if device_in else result.numpy()
        ref_pyr, factors = self._generate_img_pyr(ref_full)
        mov_pyr, _ = self._generate_img_pyr(mov_full)

        n_lvl = len(factors)
        if n_lvl == 0:
            # the reference dies here with UnboundLocalError (m_flow is never assigned, optflow_registrator.py:173)
            raise ValueError(
                f"image of shape {tuple(ref_full.shape)} is too small for num_pyr_lvl={self.num_pyr_lvl} "
                "(every pyramid level must keep >= 100 px per side) and use_full_res_img is False")
        m_flow: Optional[DeviceArray] = None
        for lvl, factor in enumerate(factors):  # smallest level first
            self._log("Pyramid factor", factor)
            last = lvl == n_lvl - 1
            ref_lvl, mov_raw = ref_pyr[lvl], mov_pyr[lvl]
            mov_lvl = mov_raw if lvl == 0 else self._warp(mov_raw, m_flow)

            ref_dog = self._dog_dev(ref_lvl)  # needed by the gate; doubles as Farneback input if use_dog
            fc = self._tile_flow_calc
            fc.ref_img = ref_dog if self.use_dog else ref_lvl
            fc.mov_img = self._dog_dev(mov_lvl) if self.use_dog else mov_lvl
            this_flow = fc.calc_flow()

            mov_warped = self._warp(mov_lvl, this_flow)
            # gate (optflow_registrator.py:127-132): "before" is the RAW level, not the pre-warped one
            after = mi_tiled(ref_dog, self._dog_dev(mov_warped), self.tile_size)
            before = mi_tiled(ref_dog, self._dog_dev(mov_raw), self.tile_size)
            self._log("    MI score after:", after, "| MI score before:", before)
            accepted = bool(after > before)
            self.level_reports.append(LevelReport(factor, tuple(ref_lvl.shape), float(after), float(before), accepted))

            nxt_hw = None if last else mov_pyr[lvl + 1].shape
            if accepted:
                self._log("    Better alignment than before")
                if lvl == 0:
                    m_flow = (ctx.pyr_up_flow(this_flow, nxt_hw, 2.0) if not last
                              else self._upscale_flow_to_full_res(this_flow, factor))
                else:
                    merged = ctx.merge_flows(m_flow, this_flow, self.tile_size, self.overlap)
                    if last:
                        m_flow = merged if self.use_full_res_img else self._upscale_flow_to_full_res(merged, factor)
                    else:
                        m_flow = ctx.pyr_up_flow(merged, nxt_hw, 2.0)
            else:
                self._log("    Worse alignment than before")
                if lvl == 0:
                    m_flow = ctx.zeros(tuple(nxt_hw if not last else self._full_shape) + (2,), np.float32)
                elif last:
                    if not self.use_full_res_img:
                        m_flow = ctx.pyr_up_flow(m_flow, self._full_shape, 2.0)
                else:
                    m_flow = ctx.pyr_up_flow(m_flow, nxt_hw, 4.0)  # sic: x4 (optflow_registrator.py:169)

        result = m_flow
        self._ctx = None
        return result if device_in else result.numpy()

    # -- pieces ---------------------------------------------------------------------------------
    def _generate_img_pyr(self, arr) -> Tuple[list, List[int]]:
        """Pyramid from the smallest level to the largest (optflow_registrator.py:175-202)."""
        if self.num_pyr_lvl < 0:
            raise ValueError("Number of pyramid levels cannot be less than 0")
        if self.num_pyr_lvl == 0 and not self.use_full_res_img:
            raise ValueError("Number of pyramid levels is 0 and use_full_res_img is False. "
                             "Please change one of the parameters")
        ctx = get_context()
        full = ctx.asdevice(arr)
        levels, factors = [], []
        cur = full
        for lvl in range(self.num_pyr_lvl):
            factor = 2 ** (lvl + 1)
            if full.shape[0] / factor < 100 or full.shape[1] / factor < 100:
                break
            cur = ctx.pyr_down(cur)
            levels.append(cur)
            factors.append(factor)
        levels.reverse()
        factors.reverse()
        if self.use_full_res_img:
            levels.append(full)
            factors.append(1)
        return levels, factors

    def _upscale_flow_to_full_res(self, flow, pyramid_factor: int):
        """optflow_registrator.py:204-215.  The reference's loop always upsamples the ORIGINAL flow and
        never doubles its magnitude (quirk Q2): one pyrUp to the full-resolution size, or nothing."""
        ctx = get_context()
        full = getattr(self, "_full_shape", None) or tuple(self._ref_img.shape)
        if abs(flow.shape[0] - full[0]) <= 1:
            return flow
        if int(log2(pyramid_factor)) < 1:
            return flow
        return ctx.pyr_up_flow(ctx.asdevice(flow), full, 1.0)

    def _merge_list_of_flows(self, flow_list):
        """optflow_registrator.py:235-240."""
        ctx = get_context()
        m_flow = ctx.asdevice(flow_list[0])
        for f in flow_list[1:]:
            m_flow = ctx.merge_flows(m_flow, ctx.asdevice(f), self.tile_size, self.overlap)
        return m_flow

    def get_dog_sigmas(self, pyr_factor: int) -> Tuple[int, int]:
        """Unused by register() in the reference as well (optflow_registrator.py:242-247)."""
        if pyr_factor > 16:
            return 1, 2
        return {1: (5, 9), 2: (4, 7), 4: (3, 5), 8: (2, 3), 16: (1, 2)}[pyr_factor]

    def _dog_dev(self, img: DeviceArray, low_sigma: int = 5, high_sigma: int = 9):
        """dog(img, True) for the Python level loop.  Where the reference returns an image whose max is 0 unchanged
        (:256-257): an all-zero image becomes the all-zero uint8 image, which every consumer on the path (Farneback's
        convertTo float, the NMI labels) treats identically; a float image with max() == 0 that is NOT all zero is
        returned unchanged like the reference's (Farneback converts a mixed pair per input, the gate labels the raw
        values: shared_modules/similarity_scoring.py).  Integer images cannot be negative: no check, no host sync."""
        ctx = get_context()
        if img.dtype != np.float32:
            return ctx.dog_u8(img, low_sigma, high_sigma, flags=self._dog_flags())
        out, src_max_is_zero = ctx.dog_u8(img, low_sigma, high_sigma, report_zero=True, flags=self._dog_flags())
        if src_max_is_zero and ctx.minmax(img)[0] < 0:
            return img
        return out

    def _dog_flags(self) -> int:
        f = self.dog_muladd_fused
        return int(f) if isinstance(f, int) and not isinstance(f, bool) else (L.MA_DOG_FUSED_BLUR | L.MA_DOG_FUSED_SCALE if f else 0)

    def dog(self, img, use_it: bool, low_sigma: int = 5, high_sigma: int = 9):
        """Difference of Gaussians -> uint8 (optflow_registrator.py:249-274)."""
        if not use_it:
            return img
        ctx = get_context()
        out, src_max_is_zero = ctx.dog_u8(ctx.asdevice(img), low_sigma, high_sigma, report_zero=True,
                                          flags=self._dog_flags())
        if src_max_is_zero:
            return img
        return out if isinstance(img, DeviceArray) else out.numpy()
