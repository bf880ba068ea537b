"""Tiled backward-bilinear warp (counterpart of microaligner/optflow_reg/warper.py:29-76).

One HIP kernel evaluates, per output pixel, the window the reference would have cut
(tile_size + 2*overlap, zero padded), the window-local map float32(x_local - flow) and the
cv2.remap INTER_LINEAR arithmetic (fixed point for uint8, float for uint16/float32).
"""
import numpy as np

from ..device import DeviceArray, get_context


class Warper:
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
        img, flow = ctx.asdevice(self.image), ctx.asdevice(self.flow)
        if img.ndim != 2:
            raise ValueError(f"Expected 2D grayscale image, got shape {img.shape}")
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
