"""Host-side data path for the bundled FloodPlanet rasters: TIFF decoding, tile grid, dataset."""
from .assemble import assemble_tiles
from .floodplanet import FloodplanetTiles, collate_tiles
from .loader import TileLoader
from .resize import resize_image, resize_lanczos4, resize_nearest
from .tiff import TiffError, read_tiff, tiff_info, tiff_size
from .tiles import CropParams, generate_image_slice_object, get_crop_slices

__all__ = ["assemble_tiles", "FloodplanetTiles", "collate_tiles", "TileLoader", "resize_image", "resize_lanczos4", "resize_nearest", "TiffError",
           "read_tiff", "tiff_info", "tiff_size", "CropParams", "generate_image_slice_object", "get_crop_slices"]
