"""Feature-based (affine) registration on MI355X -- the API of microaligner's FeatureRegistrator
(microaligner/feature_reg/feature_registrator.py:35-312): same attributes and defaults, register() returns the
2x3 float64 matrix that maps the moving image onto the reference.

Division of labour: pyramid (cv2.pyrDown), dog(), the image transforms (cv2.warpAffine up to
32000 px, scikit-image's warp above), the mutual-information gate, the feature extraction of a level (windows, FAST,
corner selection, DAISY: ma_feature_extract), the 2-NN search, the ratio test and the RANSAC similarity fit
(ma_knn2_l2, ma_match_similarity) are HIP kernels behind the C-ABI; the host keeps the level loop and its decisions
(feature_detection.py; sparse_cpu.py is the host statement the kernels reproduce bit for bit).  The public surface mirrors the reference; the machinery below it
is this package's own: one `_Level` record per pyramid level, one `_register_level` pass per level.
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from ..device import DeviceArray, get_context, use_context
from ..shared_modules.img_checks import check_img_dims_match, check_img_is_2d_grey, check_img_is_provided
from ..shared_modules.similarity_scoring import check_if_higher_similarity
from . import affine_math
from .feature_detection import Features, find_features_of_device_image
from .tile_registration import find_features, register_img_pair


@dataclass
class _Level:
    """One pyramid level of the reference image: subsampling factor, the level (device), its features."""
    factor: int
    image: DeviceArray
    features: Optional[Features] = None
    gate: Optional[DeviceArray] = None      # dog(image): the reference half of the level's gate, kept for its rounds
    # the level's reference-side work ENQUEUED on the side context (register()'s fast attempt): the context, an event behind the
    # work, the side context's deferred max() == 0 report slots of the level's dog() calls
    side: object = None
    ready: object = None
    zero_slots: tuple = ()


class _ZeroMaxImage(Exception):
    """A dog() of the fast path met an input whose max() is 0: register() starts over in the careful mode."""


class FeatureRegistrator:
    def __init__(self):
        self._ref_img = np.array([])
        self._mov_img = np.array([])
        self.num_pyr_lvl = 3
        self.num_iterations = 3
        self.tile_size = 1000
        self.use_full_res_img = False
        self.use_dog = True
        self.verbose = True      # addition: the reference prints unconditionally
        self.compat_mov_getter = True   # the mov_img getter returns the REFERENCE image, as the reference's does; False: the moving image
        self._levels: List[_Level] = []   # reference side, coarsest first; kept for register(reuse_ref_img=True)
        self.fuse_rounds = True           # a round's device work in one C call (ma_feature_round) where that makes the same calls
        self.overlap_reference = True     # the reference features of all levels on a second stream under the moving image's coarse levels
        self.skip_repeated_rounds = True  # a round after a REJECTED one sees the same images: its outcome is replayed, not recomputed
        self._round_lines = None          # log lines of the round in progress (replayed by a repeated round)
        self.features_on_host = False     # True: features and matching by the HOST statement (sparse_cpu.py, the definition the
        #                                     kernels reproduce; slow) -- dense steps stay on the device.  For cross-checks
        self._careful = False             # True: register() runs in the careful mode only (see register())
        self._fast = False                # True while register()'s fast attempt runs: dog() defers its max() == 0 report
        self._log_buf = None              # log lines of the fast attempt that no synchronisation point has confirmed yet
        self._log_shown = 0               # lines of the fast attempt already printed (a restart does not repeat them)

    # -- inputs ---------------------------------------------------------------------------------------------
    @property
    def ref_img(self):
        return self._ref_img

    @ref_img.setter
    def ref_img(self, img):
        check_img_is_2d_grey(img, "ref")
        self._ref_img = img

    @property
    def mov_img(self):
        # the reference's getter returns the REFERENCE image (feature_registrator.py:59-60, an obvious slip that nothing on
        # the path reads): so does this one unless compat_mov_getter is switched off
        return self._ref_img if self.compat_mov_getter else self._mov_img

    @mov_img.setter
    def mov_img(self, img):
        check_img_is_2d_grey(img, "mov")
        self._mov_img = img

    def _log(self, *args):
        # In the fast attempt of register() a line is held back until the next synchronisation point has confirmed that no
        # dog() before it met an all-zero image (_check_deferred prints what it confirms: progress shows as it happens); the
        # careful re-run after a restart reproduces the confirmed lines and skips them.  Per object: no process-wide
        # redirection of stdout, which threads registering side by side would trip over.
        if self._round_lines is not None:
            self._round_lines.append(args)
        if self.verbose:
            if self._log_buf is not None:
                self._log_buf.append(" ".join(str(a) for a in args))
            elif self._log_shown > 0:
                self._log_shown -= 1
            else:
                print(*args)

    @property
    def level_factors(self) -> List[int]:
        """Subsampling factors of the levels register() will visit (coarsest first), e.g. [8, 4, 2]."""
        return [f for f, _ in self._pyramid_plan(np.shape(self._ref_img))] if np.size(self._ref_img) else [8, 4, 2]

    # -- registration ---------------------------------------------------------------------------------------
    def calc_ref_img_features(self):
        """:70-76: pyramid of the reference image and the features of every level."""
        self._levels = [_Level(factor, img) for factor, img in self._build_pyramid(self._ref_img)]
        for lvl in self._levels:
            lvl.features = self._features_of(lvl.image)

    def _ref_side_enqueue(self, ctx):
        """calc_ref_img_features() without waiting for anything: pyramid, dog() and feature extraction of every level of the
        reference image ENQUEUED on the side context (a second HIP stream), coarsest level first, an event behind each level.
        A level's rounds wait for ITS event only (_settle_level): the reference features of the finer levels -- three quarters
        of the reference side's work sit in the finest one -- are computed while this context works through the moving image's
        coarse levels, whose small kernels and synchronisations leave most of the chip idle.  Same kernels on the same inputs:
        the features do not depend on the stream they were computed on."""
        side = ctx.side_context()
        if isinstance(self._ref_img, DeviceArray):
            ctx.sync()                      # the caller's array may still be written on this context's stream
        with use_context(side):
            levels = [_Level(factor, img) for factor, img in self._build_pyramid(self._ref_img)]
            for lvl in levels:
                k0 = len(getattr(side, "_zero_pending", None) or ())
                lvl.gate = self.dog(lvl.image, True)
                pre = lvl.gate if self.use_dog else lvl.image
                lvl.features = find_features_of_device_image(pre, self.tile_size, side, wait=False)
                lvl.zero_slots = tuple((getattr(side, "_zero_pending", None) or ())[k0:])
                lvl.side, lvl.ready = side, side.event()
                side.record(lvl.ready)
        self._drop_levels()                 # (events of an attempt that ended before its levels were settled)
        self._levels = levels

    def _settle_level(self, lvl: _Level):
        """Wait for a level's reference-side work (if it was enqueued), take its keypoint count and its dog() reports."""
        if lvl.ready is None:
            return
        side, ev, slots = lvl.side, lvl.ready, lvl.zero_slots
        lvl.side, lvl.ready, lvl.zero_slots = None, None, ()
        side.event_sync(ev)
        side.event_destroy(ev)
        lvl.features.settle()
        if slots:
            hit = bool(side._zero_flags[list(slots)].any())
            pending = side._zero_pending
            for k in slots:
                if k in pending:
                    pending.remove(k)
            if hit:
                raise _ZeroMaxImage()

    def _drop_levels(self):
        for lvl in self._levels:
            if lvl.ready is not None:
                try:
                    lvl.side.event_destroy(lvl.ready)
                except Exception:   # noqa: BLE001
                    pass
                lvl.ready = None
        self._levels = []

    def _can_overlap_reference(self, reuse_ref_img: bool) -> bool:
        ref = self._ref_img
        return (self._fast and self.overlap_reference and self.fuse_rounds and not self.features_on_host
                and not (reuse_ref_img and self._levels) and np.ndim(ref) == 2
                and (self.use_dog or np.dtype(getattr(ref, "dtype", None) or np.asarray(ref).dtype) == np.uint8))

    def register(self, reuse_ref_img: bool = False) -> np.ndarray:
        """:78-119: coarse-to-fine.  Every level sees the moving level pre-transformed by what the coarser levels
        found (their matrices brought to this level's pixel size), refines it in `num_iterations` gated rounds and
        contributes one matrix; the result is the product of the per-level matrices at full resolution."""
        check_img_is_provided(self._ref_img, "ref")
        check_img_is_provided(self._mov_img, "mov")
        check_img_dims_match(self._ref_img, self._mov_img)
        # Fast path: dog() does not wait for the device to learn whether its input's max() is 0 (the reference's shortcut,
        # :288-291); the flags are collected at the synchronisation points the rounds have anyway.  If one turns out set --
        # an all-black level, a transform that moved everything out of view -- the call starts over in the careful mode,
        # which asks after every dog() as the reference does.  The deferred reports exist only between here and the end of
        # the attempt: the public dog() and calc_ref_img_features() outside register() are synchronous.
        if self._careful:
            return self._register(reuse_ref_img)
        get_context().any_deferred_zero()       # flags an aborted attempt (of any registrator on this context) left behind
        if getattr(get_context(), "_side", None) is not None:
            get_context()._side.any_deferred_zero()
        self._log_buf, self._log_shown, self._fast = [], 0, True
        try:
            result = self._register(reuse_ref_img)
        except _ZeroMaxImage:
            self._log_buf, self._fast = None, False
            self._drop_levels()
            self._careful = True
            try:
                return self._register(False)      # prints from the first line the fast attempt had not shown
            finally:
                self._careful = False
                self._log_shown = 0
        finally:
            self._fast = False
            held, self._log_buf = self._log_buf, None
            for line in held or ():
                print(line)
        return result

    def _register(self, reuse_ref_img: bool) -> np.ndarray:
        if self._can_overlap_reference(reuse_ref_img):
            self._ref_side_enqueue(get_context())
        elif not (reuse_ref_img and self._levels):
            self.calc_ref_img_features()
            self._check_deferred()
        moving = self._build_pyramid(self._mov_img)
        found: List[np.ndarray] = []            # one matrix per finished level, in full-resolution pixels
        for ref_level, (factor, mov_level) in zip(self._levels, moving):
            self._log("Pyramid factor", factor)
            if found:
                so_far = affine_math.compose([affine_math.with_translation_scaled(m, 1 / factor) for m in found])
                mov_level = self.transform_img(mov_level, so_far)
            level_mat = self._register_level(ref_level, mov_level)
            found.append(affine_math.with_translation_scaled(level_mat, factor))
        return affine_math.compose(found)

    # -- image transforms -----------------------------------------------------------------------------------
    def transform_big_img(self, img, t_mat):
        """:121-126: skimage.transform.warp(img, AffineTransform(pinv(M)), preserve_range=True).astype(dtype)."""
        ctx = get_context()
        out = ctx.warp_affine(ctx.asdevice(img), np.linalg.pinv(affine_math.homogeneous(t_mat)))
        return out if isinstance(img, DeviceArray) else out.numpy()

    def transform_img(self, img, t_mat):
        """:128-132: cv2.warpAffine(img, t_mat, dsize=img.shape[::-1]) up to 32000 px per side."""
        if max(img.shape) > 32000:
            return self.transform_big_img(img, t_mat)
        ctx = get_context()
        out = ctx.warp_affine_cv(ctx.asdevice(img), t_mat)
        return out if isinstance(img, DeviceArray) else out.numpy()

    # -- pyramid --------------------------------------------------------------------------------------------
    def _pyramid_plan(self, shape) -> List[Tuple[int, int]]:
        """[(factor, number of pyrDown steps)] coarsest first (:134-160): factors 2, 4, ... while both sides keep at
        least 100 px, at most num_pyr_lvl of them, plus the image itself if use_full_res_img."""
        if self.num_pyr_lvl < 0:
            raise ValueError("Number of pyramid levels cannot be less than 1")
        if self.num_pyr_lvl == 0 and not self.use_full_res_img:
            raise ValueError("Number of pyramid levels is 0 and use_full_res_img is False. "
                             "Please change one of the parameters")
        plan = []
        for steps in range(1, self.num_pyr_lvl + 1):
            if min(shape[0], shape[1]) / 2 ** steps < 100:
                break
            plan.append((2 ** steps, steps))
        plan.reverse()
        if self.use_full_res_img:
            plan.append((1, 0))
        return plan

    def _build_pyramid(self, arr) -> List[Tuple[int, DeviceArray]]:
        """[(factor, level on the device)] following _pyramid_plan; every level is pyrDown of the next finer one."""
        ctx = get_context()
        full = ctx.asdevice(arr)
        plan = self._pyramid_plan(full.shape)
        by_steps, cur = {0: full}, full
        for steps in range(1, max((n for _, n in plan), default=0) + 1):
            cur = by_steps[steps] = ctx.pyr_down(cur)
        return [(factor, by_steps[steps]) for factor, steps in plan]

    # -- one level ------------------------------------------------------------------------------------------
    def _features_of(self, img, pre=None) -> Features:
        """Features of dog(img) (or of img itself with use_dog off); `pre`: that image when the caller has it already.
        A uint8 device image goes through ma_feature_extract in one call, anything else is cut into tiles on the host."""
        ctx = get_context()
        if pre is None:
            pre = self.dog(img, self.use_dog)
        if isinstance(pre, DeviceArray) and pre.dtype == np.uint8 and not self.features_on_host:
            return find_features_of_device_image(pre, self.tile_size, ctx)
        host = pre.numpy() if isinstance(pre, DeviceArray) else np.asarray(pre)
        return find_features(host, self.tile_size, None if self.features_on_host else ctx)

    def _register_level(self, ref_level: _Level, mov_level) -> np.ndarray:
        """:162-207: `num_iterations` rounds on one level.  A round estimates the similarity that maps the current
        image onto the reference features; it is kept only if the mutual-information gate prefers the transformed
        image and the matrix is plausible (affine_math), otherwise the round contributes the identity.  After a kept
        round the ORIGINAL level is transformed by the product of the kept matrices (no resampling chain).
        dog(current) serves both the features (use_dog) and the gate; a kept FIRST round's candidate is the next current
        image (the original level transformed by that one matrix: the same call on the same inputs)."""
        if self.num_iterations < 1:
            raise ValueError("Number of iterations cannot be less than 1")
        ctx = get_context()
        self._round_lines = None
        self._settle_level(ref_level)
        ref_gate = ref_level.gate if ref_level.gate is not None else self.dog(ref_level.image, True)
        rounds: List[np.ndarray] = []
        current = mov_level
        current_gate = None
        rejected_lines = None        # the log lines of the last round if it was rejected (nothing has changed since)
        for it in range(self.num_iterations):
            self._log("    Iteration", it + 1, "/", self.num_iterations)
            if rejected_lines is not None and self.skip_repeated_rounds and not self.features_on_host:
                # The round before was rejected: `current`, its dog() and the reference features are the very objects it saw,
                # and every step of a round is a deterministic function of them (exact 2-NN search, RANSAC seeded per call,
                # integer histograms, fixed-order sums) -- it would find the same matches, the same estimate and the same
                # two scores, and be rejected again.  Its lines are replayed, its work is not repeated.  (The reference
                # recomputes; with OpenCV's randomised FLANN trees and free-running RANSAC generator its repeat may differ --
                # this package's definition of both steps is deterministic, oracle and device alike.)
                for args in rejected_lines:
                    self._log(*args)
                rounds.append(affine_math.IDENTITY.copy())
                continue
            self._round_lines = []
            fused = self._fused_round(ctx, ref_level, ref_gate, current, current_gate)
            if fused is not None:
                estimate, is_identity, candidate, candidate_gate, current_gate, improved = fused
            else:
                if current_gate is None:
                    current_gate = self.dog(current, True)
                # the exact 2-NN search over up to 45 000 x 45 000 descriptors, the ratio test and the RANSAC fit run on the
                # device (ma_knn2_l2, ma_match_similarity): the matrix and the match count come back
                mov_features = self._features_of(current, pre=current_gate if self.use_dog else None)
                if self.features_on_host:
                    from .sparse_cpu import knn2_sequential
                    estimate = register_img_pair(ref_level.features, mov_features, self.verbose, knn=knn2_sequential, log=self._log)
                else:
                    estimate = register_img_pair(ref_level.features, mov_features, self.verbose, log=self._log, ctx=ctx)
                is_identity = bool(np.array_equal(estimate, affine_math.IDENTITY))
                candidate = current if is_identity else self.transform_img(current, estimate)
                candidate_gate = current_gate if is_identity else self.dog(candidate, True)
                improved = check_if_higher_similarity(ref_gate, candidate_gate, current_gate,
                                                      self.tile_size, self.verbose, log=self._log)
            plausible = (affine_math.centre_stays_inside(estimate, mov_level.shape)
                         and affine_math.scales_plausible(estimate))
            self._check_deferred()       # the gate has just synchronised: the flags of this round's dog() calls are in
            if any(improved) and plausible:
                self._log("    Better alignment than before")
                rejected_lines, self._round_lines = None, None
                rounds.append(estimate)
                if len(rounds) == 1 and current is mov_level:
                    current, current_gate = candidate, candidate_gate
                else:
                    current, current_gate = self.transform_img(mov_level, affine_math.compose(rounds)), None
            else:
                self._log("    Worse alignment than before")
                rejected_lines, self._round_lines = self._round_lines, None
                rounds.append(affine_math.IDENTITY.copy())
        return affine_math.compose(rounds)

    def _fused_round(self, ctx, ref_level, ref_gate, current, current_gate):
        """The round's device work in ONE call (ma_feature_round: dog(current) if missing, features, 2-NN, ratio test, RANSAC,
        cv2.warpAffine, dog(candidate), both halves of the gate) where the step-by-step path would make the very same calls:
        the fast attempt of register(), device images, uint8 feature images, sides up to 32000 px (cv2.warpAffine's range).
        Returns None when the round has to go step by step, else (estimate, is_identity, candidate, candidate_gate,
        current_gate, improved) -- the same values and the same log lines."""
        from ..shared_modules.tiling import is_tiled
        from .feature_detection import DEVICE_WORKSPACE_BYTES, _daisy_tables
        from .sparse_cpu import Daisy
        ref_f = ref_level.features
        if not (self._fast and self.fuse_rounds and not self.features_on_host and isinstance(current, DeviceArray) and current.ndim == 2
                and isinstance(ref_gate, DeviceArray) and ref_gate.dtype == np.uint8 and ref_gate.shape == current.shape
                and max(current.shape) <= 32000 and (self.use_dog or current.dtype == np.uint8)
                and (current_gate is None or (isinstance(current_gate, DeviceArray) and current_gate.dtype == np.uint8))
                and (not ref_f.is_valid() or (ref_f._dev is not None and isinstance(ref_f.descriptors_for_search, DeviceArray)))):
            return None
        halves, cos_sin, offsets = _daisy_tables(Daisy(radius=21, q_radius=3, q_theta=8, q_hist=8))
        chunk = self.tile_size * self.tile_size if is_tiled(current.shape, self.tile_size) else 0
        ref_features = (ref_f.descriptors_for_search, ref_f._dev[1], ref_f._dev[3]) if ref_f.is_valid() else None
        r = ctx.feature_round(current, current_gate, ref_gate, ref_features, self.tile_size, self.use_dog, chunk, halves, cos_sin,
                              offsets, workspace_bytes=DEVICE_WORKSPACE_BYTES)
        if r["zero_max"]:
            raise _ZeroMaxImage()
        if r["status"] == 3:        # (integer keypoints never get here) -- the host statement decides, step by step
            return None
        if r["status"] != 4:
            self._log("    Good matches", r["n_good"], "/", r["n_query"])
        after, before = ((float(r["scores_after"][0]), float(r["scores_before"][0])) if chunk == 0 else
                         (np.mean(r["scores_after"]), np.mean(r["scores_before"])))
        self._log("    MI score after:", after, "| MI score before:", before)
        ident = r["is_identity"]
        gate = r["current_gate"]
        return (r["estimate"], ident, current if ident else r["candidate"], gate if ident else r["candidate_gate"], gate,
                [after > before])

    def get_dog_sigmas(self, pyr_factor: int) -> Tuple[int, int]:
        if pyr_factor > 16:
            return 1, 2
        return {1: (5, 9), 2: (4, 7), 4: (3, 5), 8: (2, 3), 16: (1, 2)}[pyr_factor]

    # -- dog -------------------------------------------------------------------------------------------------------
    def dog(self, img, use_it: bool, low_sigma: int = 5, high_sigma: int = 9):
        """:287-312, same body as OptFlowRegistrator.dog; stays on the device for device inputs."""
        if not use_it:
            return img
        ctx = get_context()
        if self._fast and isinstance(img, DeviceArray):
            return ctx.dog_u8(img, low_sigma, high_sigma, report_zero="deferred")
        out, src_max_is_zero = ctx.dog_u8(ctx.asdevice(img), low_sigma, high_sigma, report_zero=True)
        if src_max_is_zero:
            return img
        return out if isinstance(img, DeviceArray) else out.numpy()

    def _check_deferred(self):
        """At a synchronisation point of the fast attempt: restart if a dog() since the last one met an all-zero image,
        otherwise the lines held back so far stand and are printed."""
        if not self._fast:
            return
        if get_context().any_deferred_zero():
            raise _ZeroMaxImage()
        for line in self._log_buf or ():
            print(line)
        self._log_shown += len(self._log_buf or ())
        self._log_buf = []
