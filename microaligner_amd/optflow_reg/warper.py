"""Tiled backward-bilinear warp (counterpart of microaligner/optflow_reg/warper.py:29-76).

One HIP kernel evaluates, per output pixel, the window the reference would have cut
(tile_size + 2*overlap, zero padded), the window-local map float32(x_local - flow) and the
cv2.remap INTER_LINEAR arithmetic (fixed point for uint8, float for uint16/float32).
"""
import numpy as np

from ..device import DeviceArray, get_context


class Warper:
    HOST_BANDED_MIN = 64 << 20   # bytes; below this a page is a band or two and the plain upload / warp / download is as fast

    def __init__(self):
        self.image = np.array([])
        self.flow = np.array([])
        self.tile_size = 1000
        self.overlap = 100

    def warp(self):
        if len(self.image) == 0:
            raise ValueError("No image provided")
        if len(self.flow) == 0:
            raise ValueError("No flow provided")
        ctx = get_context()
        like = self.image
        if np.ndim(like) != 2:
            raise ValueError(f"Expected 2D grayscale image, got shape {np.shape(like)}")
        if isinstance(like, np.ndarray) and like.nbytes >= self.HOST_BANDED_MIN:
            like = np.ascontiguousarray(like)      # a strided view is gathered once, here
        if isinstance(like, np.ndarray) and like.nbytes >= self.HOST_BANDED_MIN and not ctx.is_resident(like):
            # a large host page that is not in HBM yet (the reference's own per-page loop, __main__.py:288-302, kept by a
            # caller who only swapped the import): the page-warp driver moves it in bands of tile rows, so upload,
            # kernel and download of the one page overlap
            flow = ctx.asdevice(self.flow)
            out = ctx.host_empty(like.shape, like.dtype)
            ctx.warp_pages([like], flow, self.tile_size, self.overlap, [out])
            self.image = np.array([])
            self.flow = np.array([])
            return out
        img, flow = ctx.asdevice(like), ctx.asdevice(self.flow)
        out = ctx.warp(img, flow, self.tile_size, self.overlap)
        # like the reference (warper.py:41,45) the inputs are consumed
        self.image = np.array([])
        self.flow = np.array([])
        return out if isinstance(like, DeviceArray) else out.numpy()

    def warp_pages(self, pages, out=None):
        """Apply `self.flow` to many pages (the channel x z pages of a cycle, __main__.py:288-302,427-433) with the
        flow uploaded once and the page transfers overlapped.  Unlike warp() this keeps `self.flow`."""
        if len(self.flow) == 0:
            raise ValueError("No flow provided")
        ctx = get_context()
        flow = ctx.asdevice(self.flow)
        self.flow = flow  # stays resident for further calls
        return ctx.warp_pages(pages, flow, self.tile_size, self.overlap, out)
