"""Feature-based (affine) registration on MI355X -- the API of microaligner's FeatureRegistrator
(microaligner/feature_reg/feature_registrator.py:35-312): same attributes and defaults, register() returns the
2x3 float64 matrix that maps the moving image onto the reference.

Division of labour: the dense steps -- pyramid (cv2.pyrDown), dog(), the image transforms
(cv2.warpAffine up to 32000 px, scikit-image's warp above) and the mutual-information gate -- are HIP kernels behind
the C-ABI; the sparse steps (FAST, DAISY, matching, RANSAC) run on the host (feature_detection.py / sparse_cpu.py).
"""
import gc
from typing import List, Tuple, Union

import numpy as np

from ..device import DeviceArray, get_context
from ..shared_modules.img_checks import check_img_dims_match, check_img_is_2d_grey, check_img_is_provided
from ..shared_modules.similarity_scoring import check_if_higher_similarity
from . import affine_math
from .feature_detection import Features
from .tile_registration import find_features, register_img_pair


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
        self._ref_pyr_features: List[Features] = []
        self._ref_img_pyr: list = []
        self._factors = [8, 4, 2]
        self._this_pyr_factor = 1

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
        return self._mov_img  # the reference's getter returns the reference image (:62-63), an obvious slip

    @mov_img.setter
    def mov_img(self, img):
        check_img_is_2d_grey(img, "mov")
        self._mov_img = img

    def _log(self, *args):
        if self.verbose:
            print(*args)

    # -- registration ---------------------------------------------------------------------------------------
    def calc_ref_img_features(self):
        """:70-76: pyramid of the reference image and the features of every level."""
        self._ref_img_pyr, self._factors = self._generate_img_pyr(self._ref_img)
        self._ref_pyr_features = [find_features(self._host(self.dog(lvl, self.use_dog)), self.tile_size, get_context())
                                  for lvl in self._ref_img_pyr]

    def register(self, reuse_ref_img: bool = False) -> np.ndarray:
        """:78-119: coarse-to-fine; every level aligns the moving level (pre-transformed by what the coarser
        levels found) in `num_iterations` rounds and contributes one matrix, rescaled to full resolution."""
        check_img_is_provided(self._ref_img, "ref")
        check_img_is_provided(self._mov_img, "mov")
        check_img_dims_match(self._ref_img, self._mov_img)
        if not reuse_ref_img or self._ref_pyr_features == []:
            self.calc_ref_img_features()
        mov_img_pyrs, _ = self._generate_img_pyr(self._mov_img)

        fullscale_t_mat_list = []
        for i, factor in enumerate(self._factors):
            self._log("Pyramid factor", factor)
            self._this_pyr_factor = factor
            mov_lvl = mov_img_pyrs[i]
            if i > 0:
                rescaled = [self._rescale_t_mat(m, 1 / factor) for m in fullscale_t_mat_list]
                mov_lvl = self.transform_img(mov_lvl, self._multiply_transform_matrices(rescaled))
            _, t_mat = self._iterative_alignment(self._ref_img_pyr[i], self._ref_pyr_features[i], mov_lvl)
            fullscale_t_mat_list.append(self._rescale_t_mat(t_mat, factor))
            gc.collect()
        return self._multiply_transform_matrices(fullscale_t_mat_list)

    # -- image transforms -----------------------------------------------------------------------------------
    def transform_big_img(self, img, t_mat):
        """:121-126: skimage.transform.warp(img, AffineTransform(pinv(M)), preserve_range=True).astype(dtype)."""
        ctx = get_context()
        inv = np.linalg.pinv(np.append(np.asarray(t_mat, np.float64), [[0, 0, 1]], axis=0))
        out = ctx.warp_affine(ctx.asdevice(img), inv)
        return out if isinstance(img, DeviceArray) else out.numpy()

    def transform_img(self, img, t_mat):
        """:128-132: cv2.warpAffine(img, t_mat, dsize=img.shape[::-1]) up to 32000 px per side."""
        if max(img.shape) > 32000:
            return self.transform_big_img(img, t_mat)
        ctx = get_context()
        out = ctx.warp_affine_cv(ctx.asdevice(img), t_mat)
        return out if isinstance(img, DeviceArray) else out.numpy()

    def _generate_img_pyr(self, arr) -> Tuple[list, List[int]]:
        """:134-160: levels from the smallest to the largest, each kept on the device."""
        if self.num_pyr_lvl < 0:
            raise ValueError("Number of pyramid levels cannot be less than 1")
        if self.num_pyr_lvl == 0 and not self.use_full_res_img:
            raise ValueError("Number of pyramid levels is 0 and use_full_res_img is False. "
                             "Please change one of the parameters")
        ctx = get_context()
        full = ctx.asdevice(arr)
        pyramid, factors, cur = [], [], full
        for lvl in range(self.num_pyr_lvl):
            factor = 2 ** (lvl + 1)
            if full.shape[0] / factor < 100 or full.shape[1] / factor < 100:
                break
            cur = ctx.pyr_down(cur)
            pyramid.append(cur)
            factors.append(factor)
        pyramid.reverse()
        factors.reverse()
        if self.use_full_res_img:
            pyramid.append(full)
            factors.append(1)
        return pyramid, factors

    # -- one level -------------------------------------------------------------------------------------------------
    def _iterative_alignment(self, ref_img, ref_features: Features, mov_img):
        """:162-193: estimate, gate on mutual information and plausibility, accumulate."""
        if self.num_iterations < 1:
            raise ValueError("Number of iterations cannot be less than 1")
        t_matrices = []
        aligned_img = mov_img
        ref_dog = self.dog(ref_img, True)
        for i in range(self.num_iterations):
            self._log("    Iteration", i + 1, "/", self.num_iterations)
            mov_img_aligned, est_t_mat_pyr = self._align_imgs(ref_features, aligned_img)
            is_more_similar = check_if_higher_similarity(ref_dog, self.dog(mov_img_aligned, True),
                                                         self.dog(aligned_img, True), self.tile_size, self.verbose)
            is_valid_transform = self._check_if_valid_transform(est_t_mat_pyr, mov_img.shape)
            if any(is_more_similar) and is_valid_transform:
                self._log("    Better alignment than before")
                t_matrices.append(est_t_mat_pyr)
                aligned_img = self._realign_img(mov_img, t_matrices)
            else:
                self._log("    Worse alignment than before")
                t_matrices.append(np.eye(2, 3))
        return aligned_img, self._multiply_transform_matrices(t_matrices)

    def _align_imgs(self, ref: Union[np.ndarray, DeviceArray, Features], mov_img):
        """:195-207."""
        if isinstance(ref, Features):
            ref_features = ref
        else:
            ref_features = find_features(self._host(self.dog(ref, self.use_dog)), self.tile_size, get_context())
        mov_features = find_features(self._host(self.dog(mov_img, self.use_dog)), self.tile_size, get_context())
        # the exact 2-NN search over up to 45 000 x 45 000 descriptors runs on the device (ma_knn2_l2)
        transform_mat = register_img_pair(ref_features, mov_features, self.verbose, knn=get_context().knn2)
        if np.equal(transform_mat, np.eye(2, 3)).all():
            return mov_img, np.eye(2, 3)
        return self.transform_img(mov_img, transform_mat), transform_mat

    def _realign_img(self, mov_img, mat_list):
        return self.transform_img(mov_img, self._multiply_transform_matrices(mat_list))

    # -- matrix bookkeeping (feature_reg/affine_math.py) ---------------------------------------------------------
    def _multiply_transform_matrices(self, mat_list):
        return affine_math.compose(mat_list)

    def _rescale_t_mat(self, t_mat, scale: float):
        return affine_math.with_translation_scaled(t_mat, scale)

    def _check_if_valid_transform(self, t_mat, img_shape) -> bool:
        """Gate of :224-279: the image centre stays inside the image and neither axis is scaled outside [0.3, 3]."""
        return affine_math.centre_stays_inside(t_mat, img_shape) and affine_math.scales_plausible(t_mat)

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
        out, src_max_is_zero = ctx.dog_u8(ctx.asdevice(img), low_sigma, high_sigma, report_zero=True)
        if src_max_is_zero:
            return img
        return out if isinstance(img, DeviceArray) else out.numpy()

    @staticmethod
    def _host(img):
        return img.numpy() if isinstance(img, DeviceArray) else np.asarray(img)
