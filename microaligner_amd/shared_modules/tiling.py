"""Tile geometry of the reference's slicer/stitcher (microaligner/shared_modules/slicer.py:69-118,
stitcher.py:25-118) as plain index arithmetic.

On the HIP path tiles are never materialised: the kernels evaluate the same geometry per pixel
(window origin = tile origin - overlap, zero padding outside the image, centre crop on output).
This module exposes that geometry to host code (work sizing, tests, byte accounting).
"""
from dataclasses import dataclass
from typing import Iterator, Tuple


@dataclass(frozen=True)
class TileGrid:
    height: int
    width: int
    tile: int
    overlap: int

    @property
    def ny(self) -> int:
        return -(-self.height // self.tile)

    @property
    def nx(self) -> int:
        return -(-self.width // self.tile)

    @property
    def ntiles(self) -> int:
        return self.ny * self.nx

    @property
    def window(self) -> int:
        """Edge of the zero-padded window handed to OpenCV in the reference: tile + 2*overlap."""
        return self.tile + 2 * self.overlap

    @property
    def padded_pixels(self) -> int:
        """Pixels actually processed (px_pad of SURVEY.md 8d)."""
        return self.ntiles * self.window * self.window

    def origins(self) -> Iterator[Tuple[int, int]]:
        """(y, x) image coordinate of every window's top-left corner, row-major like slicer.py."""
        for ty in range(self.ny):
            for tx in range(self.nx):
                yield ty * self.tile - self.overlap, tx * self.tile - self.overlap

    def slicer_info(self) -> dict:
        """The info dict split_image_into_tiles_of_size returns (slicer.py:106-117)."""
        pad_r = 0 if self.width % self.tile == 0 else self.tile - self.width % self.tile
        pad_b = 0 if self.height % self.tile == 0 else self.tile - self.height % self.tile
        return dict(tile_shape=[self.tile, self.tile], ntiles=dict(x=self.nx, y=self.ny), overlap=self.overlap,
                    padding=dict(left=0, right=pad_r, top=0, bottom=pad_b))


def is_tiled(shape, tile_size: int) -> bool:
    """TileFlowCalc / mi_tiled switch to tiles when max(shape)/tile_size >= 2 (flow_calc.py:60-61)."""
    return max(shape) / tile_size >= 2
