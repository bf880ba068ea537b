"""Per-level dense optical flow (counterpart of microaligner/optflow_reg/flow_calc.py).

`farneback` and `TileFlowCalc` keep the reference's names, arguments and defaults; the work
is one batched HIP pipeline over all windows of the level (ma_farneback_tiled) instead of a
dask fan-out of cv2.calcOpticalFlowFarneback calls.
"""
import numpy as np

from ..device import DeviceArray, get_context
from ..shared_modules.tiling import is_tiled


def _result(flow_dev, like):
    """numpy in -> numpy out; DeviceArray in -> DeviceArray out."""
    return flow_dev if isinstance(like, DeviceArray) else flow_dev.numpy()


def farneback(mov_img, ref_img, pyr_size=0, win_size=51, num_iter=1, muladd_fused=False):
    """cv2.calcOpticalFlowFarneback(mov, ref, None, 0.5, pyr_size, win_size, num_iter, poly_n=1,
    poly_sigma=1.7, OPTFLOW_FARNEBACK_GAUSSIAN) -- flow_calc.py:30-47.  Only the single-scale
    form the reference uses (pyr_size == 0) exists on the device."""
    if pyr_size != 0:
        raise ValueError("only pyr_size == 0 (single scale, as microaligner calls it) is supported")
    ctx = get_context()
    prev, nxt = ctx.asdevice(mov_img), ctx.asdevice(ref_img)
    flow = ctx.farneback(prev, nxt, win_size, num_iter, tile=0, overlap=0, fused=muladd_fused)
    return _result(flow, mov_img)


class TileFlowCalc:
    """flow_calc.py:50-98.  `calc_flow()` consumes `ref_img`/`mov_img` like the reference (Q6)."""

    def __init__(self):
        self.ref_img = np.array([])
        self.mov_img = np.array([])
        self.num_iter = 1
        self.win_size = 51
        self.tile_size = 1000
        self.overlap = 100
        self.muladd_fused = False

    def calc_flow(self):
        ctx = get_context()
        like = self.ref_img
        ref, mov = ctx.asdevice(self.ref_img), ctx.asdevice(self.mov_img)
        if ref.shape != mov.shape or ref.ndim != 2:
            raise ValueError(f"ref/mov must be 2-D images of equal shape, got {ref.shape} and {mov.shape}")
        tiled = is_tiled(ref.shape, self.tile_size)
        flow = ctx.farneback(mov, ref, self.win_size, self.num_iter,
                             tile=self.tile_size if tiled else 0, overlap=self.overlap if tiled else 0,
                             fused=self.muladd_fused)
        self.ref_img = np.array([])
        self.mov_img = np.array([])
        return _result(flow, like)
