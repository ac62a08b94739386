"""FloodPlanet tile dataset over the bundled `CSDAP_complete/<region>/{S1,L8,PS,S2,labels}` rasters, without tifffile /
rasterio / cv2 / omegaconf.  Mirrors `Floodplanet_Dataset` (st_water_seg/datasets/floodplanet.py:20-658) and the parts of
`BaseDataset` it uses (base_dataset.py:18-113, 271-341): same constructor arguments, same example list (one example per
crop of `get_crop_slices(label_h, label_w, ..., mode="exact")`), same per-sensor scaling to [0, 1]
(S1 (x+50)/100 :347, S2 x/2^12 :406, PS x/2^16 for uint16 :468, L8 x/18607.72 :525), the label mapping
{0: ignore_index, 1: 0, 2: 1} (:586-596), normalisation modes None / 'local' (base_dataset.py:95-104), zero / ignore
padding of edge crops to the nominal tile size (:271-325), and the item dict {image f32 [C,h,w], target i64 [h,w],
mean, std [C,1,1]} (:644-648).

Differences, all at the edges of the path: rasters are decoded by `datasets.tiff` (uncompressed strips only) and
resampled by `datasets.resize` (Lanczos-4 restated, parity unpinned); `transforms` must be None -- flips / rotations
run on the GPU on whole batches (`floodplanet_code_amd.augment`, C ABI `fu_augment`) instead of per item on the CPU
(base_dataset.py:494-555); norm_mode 'global' needs a parameter file the reference does not ship and raises; dem /
slope inputs raise NotImplementedError exactly as the reference does (:107-115)."""
from __future__ import annotations

import os
import random
from glob import glob
from typing import Dict, List, Optional

import numpy as np
import torch

from .resize import lanczos4_axis_window, resize_image
from .tiff import read_tiff, tiff_size
from .tiles import CropParams, get_crop_slices

__all__ = ["FloodplanetTiles", "RawTileView", "collate_tiles", "collate_raw_tiles"]

_N_CHANNELS = {"S2": {"RGB": 3, "RGB_NIR": 4, "ALL": 10}, "PS": {"RGB": 3, "RGB_NIR": 4, "ALL": 4},
               "S1": {"ALL": 2}, "L8": {"ALL": 7}}


class FloodplanetTiles(torch.utils.data.Dataset):
    def __init__(self, root_dir, split, slice_params, eval_region=None, transforms=None, sensor="PS", channels=None,
                 dset_name="floodplanet", seed_num=0, output_metadata=False, norm_mode=None, dem=False, slope=False,
                 preflood=False, pre_post_difference=False, chirps=False, hand=False, ignore_index=-1,
                 train_split_pct=0.8):
        if transforms is not None:
            raise NotImplementedError("per-item CPU transforms are replaced by the batched GPU augmentation "
                                      "(floodplanet_code_amd.augment); pass transforms=None")
        if train_split_pct < 0 or train_split_pct > 1:
            raise ValueError(f"Train split pct must be between 0 and 1. Invalid value: {train_split_pct}")
        if norm_mode == "global":
            raise NotImplementedError("norm_mode 'global' needs the dataset's parameter file, which is not bundled")
        if norm_mode not in (None, "local"):
            raise NotImplementedError(f'Normalization mode "{norm_mode}" not implemented.')
        self.channels = "ALL" if channels is None else channels
        self.split, self.sensor, self.root_dir, self.seed_num = split, sensor, root_dir, seed_num
        self.dset_name, self.norm_mode, self.transforms = dset_name, norm_mode, None
        self.eval_region, self.ignore_index, self.slice_params = eval_region, ignore_index, slice_params
        self.train_split_pct, self.output_metadata = train_split_pct, output_metadata
        self.dem, self.slope, self.hand, self.chirps = dem, slope, hand, chirps
        self.preflood, self.pre_post_difference = preflood, pre_post_difference
        self.n_classes = 3
        self._raster_cache: Dict[tuple, np.ndarray] = {}
        if seed_num is not None:
            if type(seed_num) is not int:
                raise TypeError(f"Input seed value is not an int but type {type(seed_num)}")
            random.seed(seed_num)           # the random image split below draws from it (floodplanet.py:210-212)
            np.random.seed(seed_num)
        self._prepare_data(sensor)
        self.n_channels = self._get_n_channels()

    # ---- example list (floodplanet.py:72-139) ------------------------------------------------------------------
    def _prepare_data(self, sensor_name):
        region_dirs = sorted(glob(os.path.join(self.root_dir, "CSDAP_complete") + "/*/"))
        region_dirs_dict = {p.split("/")[-2]: p for p in region_dirs}
        image_paths = self._split_data(region_dirs_dict, sensor_name)
        self.dataset: List[dict] = []
        self.skipped_without_label: List[str] = []
        for image_path, region_name in image_paths:
            image_name = os.path.splitext(os.path.split(image_path)[1])[0]
            label_path = os.path.join("/".join(image_path.split("/")[:-3]), "labels", image_name + ".tif")
            if not os.path.exists(label_path):
                # the reference stops in a debugger here (floodplanet.py:97-99) and then fails in rasterio.open; the
                # bundled sample has 3 such images (US-Nebraska S1 x2, L8 x1): they are skipped and counted
                self.skipped_without_label.append(image_path)
                continue
            label_height, label_width = tiff_size(label_path)
            if self.dem:
                raise NotImplementedError(f'DEM finding not implemented for "{self.dset_name}" dataset.')
            if self.slope:
                raise NotImplementedError(f'SLOPE finding not implemented for "{self.dset_name}" dataset.')
            for crop in get_crop_slices(label_height, label_width, self.slice_params.height, self.slice_params.width,
                                        self.slice_params.stride, mode="exact"):
                self.dataset.append({"image_path": image_path, "label_path": label_path, "region_name": region_name,
                                     "crop_params": CropParams(*crop, label_height, label_width,
                                                               self.slice_params.height, self.slice_params.width)})
        self.image_paths = image_paths

    def _split_data(self, region_dirs: Dict[str, str], sensor_name: str):
        """floodplanet.py:140-229: by region when eval_region is given, else a seeded random split by image."""
        if len(region_dirs) == 0:
            raise ValueError(f'No regions found for dataset "{self.dset_name}" and sensor "{self.sensor}"')
        if self.eval_region is not None:
            if isinstance(self.eval_region, str):
                self.eval_region = [self.eval_region]
            if not isinstance(self.eval_region, (list, tuple)):
                raise ValueError(f"Eval regions variable is not a list but a {type(self.eval_region)}")
            names = list(region_dirs.keys())
            if self.split == "train":
                for r in self.eval_region:
                    if r not in names:
                        raise ValueError(f"Eval region {r} not found in avilable regions {names}")
                region_dirs = {k: v for k, v in region_dirs.items() if k not in self.eval_region}
            elif self.split in ("valid", "test"):
                region_dirs = {r: region_dirs[r] for r in self.eval_region}      # KeyError for an unknown region
            elif self.split != "all":
                raise ValueError(f'Cannot handle split "{self.split}" for splitting data by region.')
        image_paths = []
        for region_name, region_dir in region_dirs.items():
            for p in glob(region_dir + f"/{sensor_name}/*.tif"):
                image_paths.append([p, region_name])
        if self.eval_region is None:
            random.shuffle(image_paths)
            n_train = int(len(image_paths) * self.train_split_pct)
            image_paths = image_paths[:n_train] if self.split == "train" else image_paths[n_train:]
        if len(image_paths) == 0:
            raise ValueError(f'No images found for eval regions "{self.eval_region}" and sensor "{self.sensor}"')
        return image_paths

    def _get_n_channels(self):
        try:
            n = {"ms_image": _N_CHANNELS[self.sensor][self.channels]}
        except KeyError:
            raise NotImplementedError(f'Cannot get number of {self.sensor} channels for channel query '
                                      f'"{self.channels}"') from None
        if self.dem:
            n["dem"] = 1
        if self.slope:
            n["slope"] = 1
        return n

    # ---- loading (floodplanet.py:288-598) ------------------------------------------------------------------------
    @staticmethod
    def _crop(image, cp: CropParams):
        return image[..., cp.h0:cp.hE, cp.w0:cp.wE]

    def _load_crop_norm_image(self, image_path, crop_params, channels, resize_dims):
        # the crops of one raster arrive back to back: keep the last few decoded + resampled rasters (the reference
        # decodes and Lanczos-resamples the whole raster again for every crop, floodplanet.py:313-340)
        key = (image_path, channels, tuple(resize_dims))
        image = self._raster_cache.get(key)
        if image is None:
            image = self._load_norm_raster(image_path, channels, resize_dims)
            if len(self._raster_cache) >= 8:
                self._raster_cache.pop(next(iter(self._raster_cache)))
            self._raster_cache[key] = image
        if crop_params is not None:
            image = self._crop(image, crop_params)
        return np.ascontiguousarray(image, dtype=np.float32)

    def _load_raw_raster(self, image_path, channels):
        """TIFF decode + band selection only (what `_load_norm_raster` does before it resamples): -> (bands-first array in
        the raster's own resolution and dtype-derived float32, stored-as-uint16 flag).  Cached like the resampled rasters."""
        key = ("raw", image_path, channels)
        hit = self._raster_cache.get(key)
        if hit is not None:
            return hit
        image = read_tiff(image_path)
        s = self.sensor
        if s == "S1":
            if image.ndim == 3 and (image.shape[0] > image.shape[1] or image.shape[0] > image.shape[2]):
                image = np.transpose(image, (2, 0, 1))
            image = image[:2]
            if channels != "ALL":
                raise NotImplementedError(f'No method to subselect S1 images with "{channels}" channel query.')
        elif s == "PS":
            image = np.transpose(image, (2, 0, 1))[:4]
            sel = {"RGB": [2, 1, 0], "RGB_NIR": [2, 1, 0, 3], "ALL": None}
            if channels not in sel:
                raise NotImplementedError(f'No method to subselect PS images with "{channels}" channel query.')
            image = image if sel[channels] is None else image[sel[channels]]
        elif s == "S2":
            sel = {"RGB": [3, 2, 1], "RGB_NIR": [3, 2, 1, 7], "ALL": None}
            if channels not in sel:
                raise NotImplementedError(f'No method to subselect S2 images with "{channels}" channel query.')
            image = image if sel[channels] is None else image[sel[channels]]
        elif s == "L8":
            if channels != "ALL":
                raise NotImplementedError(f'No method to subselect L8 images with "{channels}" channel query.')
        else:
            raise NotImplementedError(f'No loader for sensor "{s}"')
        out = (np.ascontiguousarray(image, dtype=np.float32), image.dtype == np.uint16)
        if len(self._raster_cache) >= 8:
            self._raster_cache.pop(next(iter(self._raster_cache)))
        self._raster_cache[key] = out
        return out

    def _load_norm_raster(self, image_path, channels, resize_dims):
        crop_params = None
        image = read_tiff(image_path)
        s = self.sensor
        if s == "S1":
            if image.ndim == 3 and (image.shape[0] > image.shape[1] or image.shape[0] > image.shape[2]):
                image = np.transpose(image, (2, 0, 1))        # stored [H, W, C]
            image = image[:2]
            if channels != "ALL":
                raise NotImplementedError(f'No method to subselect S1 images with "{channels}" channel query.')
        elif s == "PS":
            image = np.transpose(image, (2, 0, 1))[:4]         # stored [H, W, C]
            sel = {"RGB": [2, 1, 0], "RGB_NIR": [2, 1, 0, 3], "ALL": None}
            if channels not in sel:
                raise NotImplementedError(f'No method to subselect PS images with "{channels}" channel query.')
            image = image if sel[channels] is None else image[sel[channels]]
        elif s == "S2":
            sel = {"RGB": [3, 2, 1], "RGB_NIR": [3, 2, 1, 7], "ALL": None}
            if channels not in sel:
                raise NotImplementedError(f'No method to subselect S2 images with "{channels}" channel query.')
            image = image if sel[channels] is None else image[sel[channels]]
        elif s == "L8":
            if channels != "ALL":
                raise NotImplementedError(f'No method to subselect L8 images with "{channels}" channel query.')
        else:
            raise NotImplementedError(f'No loader for sensor "{s}"')
        was_u16 = image.dtype == np.uint16
        if resize_dims[0] is not None and resize_dims[1] is not None:
            image = resize_image(image, resize_dims[0], resize_dims[1])
        if crop_params is not None:
            image = self._crop(image, crop_params)
        if s == "S1":
            image = np.nan_to_num(np.clip((image + 50) / 100, 0, 1))
        elif s == "S2":
            image = np.clip(image / 2 ** 12, 0, 1)
        elif s == "PS":
            image = image / 2 ** 16 if was_u16 else image
        else:
            image = np.clip(image, 0, 18607.72) / 18607.72
        return np.ascontiguousarray(image, dtype=np.float32)

    def _load_label_image(self, label_path, desired_height, desired_width, crop_params):
        # (cached like the image rasters: the reference decodes the whole 1024 x 1024 label raster again for every crop,
        #  floodplanet.py:557-583 -- with the resampling on the device this decode was what capped the loader)
        key = ("label", label_path, desired_height, desired_width)
        label = self._raster_cache.get(key)
        if label is None:
            label = read_tiff(label_path)
            if label.shape != (desired_height, desired_width):
                label = resize_image(label, desired_height, desired_width, resize_mode="nearest")
            if len(self._raster_cache) >= 8:
                self._raster_cache.pop(next(iter(self._raster_cache)))
            self._raster_cache[key] = label
        label = self._crop(label, crop_params)
        out = np.zeros(label.shape, dtype=np.uint8)          # 1 (no flood) -> 0
        out[label == 2] = 1                                  # flood
        # uint8 array, as in the reference: ignore_index -1 wraps to 255 there too; callers pass 0 (config.yaml:26)
        out[label == 0] = np.uint8(self.ignore_index % 256)  # no data
        return out

    def normalize(self, image):
        if self.norm_mode == "local":
            flat = image.reshape(image.shape[0], -1)
            mean, std = flat.mean(axis=1)[:, None, None], flat.std(axis=1)[:, None, None]
        else:
            mean = np.zeros([image.shape[0], 1, 1], dtype=image.dtype)
            std = np.ones([image.shape[0], 1, 1], dtype=image.dtype)
        return (image - mean) / std, mean, std

    @staticmethod
    def _add_buffer(image, height, width, constant_value=0):
        h, w = image.shape[-2], image.shape[-1]
        if h >= height and w >= width:
            return image
        canvas = np.ones(image.shape[:-2] + (height, width), dtype=image.dtype) * constant_value
        canvas[..., :h, :w] = image
        return canvas

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index, output_metadata: Optional[bool] = None):
        ex = self.dataset[index]
        cp: CropParams = ex["crop_params"]
        image = self._load_crop_norm_image(ex["image_path"], cp, self.channels, [cp.og_height, cp.og_width])
        target = self._load_label_image(ex["label_path"], cp.og_height, cp.og_width, cp)
        image, mean, std = self.normalize(image)
        image = self._add_buffer(image, cp.max_crop_height, cp.max_crop_width)
        target = self._add_buffer(target, cp.max_crop_height, cp.max_crop_width, constant_value=self.ignore_index)
        out = {"image": torch.from_numpy(np.ascontiguousarray(image)).float(),
               "target": torch.from_numpy(np.ascontiguousarray(target)).long(),
               "mean": mean, "std": std}
        if self.output_metadata if output_metadata is None else output_metadata:
            out["metadata"] = {"image_path": ex["image_path"], "crop_params": cp, "region_name": ex["region_name"]}
        return out


    def raw_item(self, index) -> dict:
        """The same example BEFORE `normalize` and `_add_buffer_to_image` (floodplanet.py:613-625): the scaled crop as it
        comes out of the raster, [C, h, w] with h <= max_crop_height, w <= max_crop_width at the raster's edge, and the
        padded target.  The per-tile normalisation, the zero padding to the nominal tile and the sensor concat then run on
        the whole batch in HBM (`datasets.assemble.assemble_tiles`, C ABI fu_assemble_tiles) -- TileLoader(device_assembly=True)."""
        ex = self.dataset[index]
        cp: CropParams = ex["crop_params"]
        image = self._load_crop_norm_image(ex["image_path"], cp, self.channels, [cp.og_height, cp.og_width])
        target = self._load_label_image(ex["label_path"], cp.og_height, cp.og_width, cp)
        target = self._add_buffer(target, cp.max_crop_height, cp.max_crop_width, constant_value=self.ignore_index)
        out = {"raw": torch.from_numpy(np.ascontiguousarray(image)).float(),
               "target": torch.from_numpy(np.ascontiguousarray(target)).long(),
               "tile_hw": (cp.max_crop_height, cp.max_crop_width)}
        if self.output_metadata:
            out["metadata"] = {"image_path": ex["image_path"], "crop_params": cp, "region_name": ex["region_name"]}
        return out


    def window_item(self, index) -> dict:
        """The example for the DEVICE resampling path (TileLoader(device_resize=True); C ABI fu_resize_lanczos4_tiles): the
        reference resamples the whole raster to the label raster's size for every item and then crops (floodplanet.py:338-341);
        here the worker only cuts the window of the SOURCE raster that the tile's rows / columns touch and makes the two
        8-tap tables -- resampling, sensor scaling, normalisation and padding run on the batch in HBM.  -> window [C, wh, ww],
        iy / wy [tile_h, 8], ix / wx [tile_w, 8], valid sizes, scale_mode, padded target."""
        ex = self.dataset[index]
        cp: CropParams = ex["crop_params"]
        raster, was_u16 = self._load_raw_raster(ex["image_path"], self.channels)
        iy, wy, (y0, y1) = lanczos4_axis_window(raster.shape[1], cp.og_height, cp.h0, cp.hE, cp.max_crop_height)
        ix, wx, (x0, x1) = lanczos4_axis_window(raster.shape[2], cp.og_width, cp.w0, cp.wE, cp.max_crop_width)
        target = self._load_label_image(ex["label_path"], cp.og_height, cp.og_width, cp)
        target = self._add_buffer(target, cp.max_crop_height, cp.max_crop_width, constant_value=self.ignore_index)
        mode = {"S1": 1, "S2": 2, "L8": 3, "PS": 4 if was_u16 else 0}[self.sensor]
        out = {"window": torch.from_numpy(np.ascontiguousarray(raster[:, y0:y1, x0:x1])),
               "iy": torch.from_numpy(iy), "wy": torch.from_numpy(wy), "ix": torch.from_numpy(ix), "wx": torch.from_numpy(wx),
               "valid_hw": (cp.hE - cp.h0, cp.wE - cp.w0), "scale_mode": mode,
               "target": torch.from_numpy(np.ascontiguousarray(target)).long(),
               "tile_hw": (cp.max_crop_height, cp.max_crop_width)}
        if self.output_metadata:
            out["metadata"] = {"image_path": ex["image_path"], "crop_params": cp, "region_name": ex["region_name"]}
        return out


class WindowTileView(torch.utils.data.Dataset):
    """FloodplanetTiles seen through window_item (what DataLoader workers produce for the device-side resampling)."""

    def __init__(self, tiles: "FloodplanetTiles"):
        self.tiles = tiles

    def __len__(self):
        return len(self.tiles)

    def __getitem__(self, index):
        return self.tiles.window_item(index)


def collate_window_tiles(items: List[dict]) -> dict:
    """Source windows of one batch in ONE zero-filled host buffer [B, C, wh_max, ww_max] (the tables index from the window's
    top-left corner, so the filler is never read) + the stacked tap tables, valid sizes and targets."""
    Cc = items[0]["window"].shape[0]
    wh = max(i["window"].shape[1] for i in items)
    ww = max(i["window"].shape[2] for i in items)
    win = torch.zeros(len(items), Cc, wh, ww, dtype=torch.float32)
    for b, it in enumerate(items):
        h, w = it["window"].shape[-2:]
        win[b, :, :h, :w] = it["window"]
    modes = {i["scale_mode"] for i in items}
    if len(modes) != 1:
        raise ValueError("one sensor scaling per batch")
    out = {"window": win, "scale_mode": modes.pop(), "tile_hw": items[0]["tile_hw"],
           "iy": torch.stack([i["iy"] for i in items]), "wy": torch.stack([i["wy"] for i in items]),
           "ix": torch.stack([i["ix"] for i in items]), "wx": torch.stack([i["wx"] for i in items]),
           "valid_h": torch.tensor([i["valid_hw"][0] for i in items], dtype=torch.int32),
           "valid_w": torch.tensor([i["valid_hw"][1] for i in items], dtype=torch.int32),
           "target": torch.stack([i["target"] for i in items])}
    if "metadata" in items[0]:
        out["metadata"] = [i["metadata"] for i in items]
    return out


class RawTileView(torch.utils.data.Dataset):
    """FloodplanetTiles seen through raw_item (what DataLoader workers produce for the device-side assembly)."""

    def __init__(self, tiles: FloodplanetTiles):
        self.tiles = tiles

    def __len__(self):
        return len(self.tiles)

    def __getitem__(self, index):
        return self.tiles.raw_item(index)


def collate_raw_tiles(items: List[dict]) -> dict:
    """Raw crops of one batch in ONE zero-filled host buffer [B, C, H, W] (crop in the top-left corner, as the device kernel
    expects it) + the valid sizes; the host neither normalises nor pads with meaning -- the filler is never read."""
    H, W = items[0]["tile_hw"]
    Cc = items[0]["raw"].shape[0]
    raw = torch.zeros(len(items), Cc, H, W, dtype=torch.float32)
    vh = torch.empty(len(items), dtype=torch.int32)
    vw = torch.empty(len(items), dtype=torch.int32)
    for b, it in enumerate(items):
        h, w = it["raw"].shape[-2:]
        raw[b, :, :h, :w] = it["raw"]
        vh[b], vw[b] = h, w
    out = {"raw": raw, "valid_h": vh, "valid_w": vw, "target": torch.stack([i["target"] for i in items])}
    if "metadata" in items[0]:
        out["metadata"] = [i["metadata"] for i in items]
    return out


def collate_tiles(items: List[dict]) -> dict:
    """Default-collate equivalent for the item dicts above (metadata kept as a list)."""
    out = {"image": torch.stack([i["image"] for i in items]), "target": torch.stack([i["target"] for i in items]),
           "mean": torch.from_numpy(np.stack([i["mean"] for i in items])),
           "std": torch.from_numpy(np.stack([i["std"] for i in items]))}
    if "metadata" in items[0]:
        out["metadata"] = [i["metadata"] for i in items]
    return out
