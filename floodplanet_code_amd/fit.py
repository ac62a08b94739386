"""Minimal trainer that drives the plugin exactly the way st_water_seg/fit.py:16-103 drives it through
pytorch_lightning -- for environments where Lightning / Hydra / the GeoTIFF stack are absent (this image, the GPU
box).  When those packages exist, use the reference's own fit.py with `floodplanet_code_amd.models.build_model`
(INTEGRATION.md); this file only restates the loop Lightning's automatic optimisation runs:

    per epoch:  for batch in train_loader:  opt.zero_grad(); loss = model.training_step(batch, i);
                                            loss.backward(); opt.step()                         (fit.py:95-97)
                for batch in valid_loader:  model.validation_step(batch, i)
                model.validation_epoch_end(outputs)  -> 'val_MulticlassJaccardIndex'
                keep the top-k checkpoints by that metric, named as fit.py:80-85 names them.

`cfg` is a plain nested dict with the reference's key names (conf/config.yaml): lr, batch_size, n_epochs,
crop_height, crop_width, ignore_index, save_topk_models, limit_train_batches, limit_val_batches,
model: {name, model_kwargs}, plus `n_channels` (dict) / `n_classes` that the reference reads off the dataset.
The tile stream is synthetic (SyntheticTiles) because the FloodPlanet rasters and their readers are out of scope.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional

import torch

from .models import build_model

DEFAULTS = dict(lr=1e-4, batch_size=10, n_epochs=11, crop_height=300, crop_width=300, ignore_index=0,
                save_topk_models=3, limit_train_batches=None, limit_val_batches=None, log_image_iter=200,
                seed_num=0, model=dict(name="ef_model", model_kwargs=dict(optimizer_name="adam")))


class SyntheticTiles:
    """An iterable of batches shaped like the default collate of Floodplanet_Dataset.__getitem__
    (datasets/floodplanet.py:644-648): image f32 [B,C,h,w] in [0,1), target i64 [B,h,w], mean/std [B,C,1,1]."""

    def __init__(self, n_batches: int, batch_size: int, n_channels: Dict[str, int], height: int, width: int,
                 device, seed: int = 0, extras: Iterable[str] = ()):
        self.n_batches, self.B, self.h, self.w = n_batches, batch_size, height, width
        self.n_channels, self.device, self.seed = n_channels, torch.device(device), seed
        self.C = n_channels.get("ms_image", next(iter(n_channels.values())))
        self.extras = [k for k in n_channels if k != "ms_image" and k in
                       ("dem", "slope", "preflood", "pre_post_difference", "hand")] or list(extras)

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        g = torch.Generator(device=self.device).manual_seed(self.seed)
        yy, xx = torch.meshgrid(torch.arange(self.h, device=self.device), torch.arange(self.w, device=self.device),
                                indexing="ij")
        for _ in range(self.n_batches):
            ph = torch.rand(self.B, 3, device=self.device, generator=g) * 6.28
            f = (torch.sin(yy[None] * 0.07 + ph[:, 0, None, None]) + torch.cos(xx[None] * 0.05 + ph[:, 1, None, None])
                 + torch.sin((xx + yy)[None] * 0.03 + ph[:, 2, None, None]))
            target = (f > 0.3).long()                      # {0 = not flood / ignored under ignore_index 0, 1 = flood}
            image = torch.rand(self.B, self.C, self.h, self.w, device=self.device, generator=g)
            image[:, 0] = 0.5 * image[:, 0] + 0.5 * (f > 0.3).float()   # a learnable signal in band 0
            batch = {"image": image, "target": target,
                     "mean": torch.zeros(self.B, self.C, 1, 1, device=self.device),
                     "std": torch.ones(self.B, self.C, 1, 1, device=self.device)}
            for k in self.extras:
                batch[k] = torch.rand(self.B, 1, self.h, self.w, device=self.device, generator=g)
            yield batch


def _limited(loader, limit: Optional[int]):
    for i, b in enumerate(loader):
        if limit is not None and i >= limit:
            break
        yield i, b


def fit_model(cfg: dict, train_loader, valid_loader, n_channels: Dict[str, int], n_classes: int = 3,
              exp_dir: Optional[str] = None, device="cuda:0", to_rgb_fcn=None) -> str:
    """Restatement of fit.py:16-103 without Lightning.  Returns the best checkpoint path ('' if exp_dir is None)."""
    c = dict(DEFAULTS)
    c.update(cfg or {})
    torch.manual_seed(c["seed_num"])                                         # pl.seed_everything(seed_num)
    model = build_model(c["model"]["name"], n_channels, n_classes, c["lr"], log_image_iter=c["log_image_iter"],
                        to_rgb_fcn=to_rgb_fcn, ignore_index=c["ignore_index"], **c["model"].get("model_kwargs", {}))
    model = model.to(device)
    opt = model.configure_optimizers()
    best: List = []                                                          # (metric, path), top-k
    ckpt_dir = os.path.join(exp_dir, "checkpoints") if exp_dir else None
    if ckpt_dir:
        os.makedirs(ckpt_dir, exist_ok=True)
    history = []
    for epoch in range(c["n_epochs"]):
        model.current_epoch = epoch
        for i, batch in _limited(train_loader, c["limit_train_batches"]):
            opt.zero_grad()
            loss = model.training_step(batch, i)
            loss.backward()
            opt.step()
            model.global_step += 1
        model.valid_metrics.reset()
        outs = []
        for i, batch in _limited(valid_loader, c["limit_val_batches"]):
            outs.append(model.validation_step(batch, i))
        model.validation_epoch_end(outs)
        miou = float(model.logged.get("val_MulticlassJaccardIndex", torch.zeros(())))
        history.append({"epoch": epoch, "train_loss": float(loss.detach()), "val_MulticlassJaccardIndex": miou})
        if ckpt_dir:
            path = os.path.join(ckpt_dir, f"model-epoch={epoch:02d}-val_MulticlassJaccardIndex={miou:.4f}.ckpt")
            torch.save({"state_dict": model.state_dict(), "epoch": epoch, "hyper_parameters": dict(c)}, path)
            best.append((miou, path))
            best.sort(key=lambda t: -t[0])
            for _, stale in best[c["save_topk_models"]:]:
                if os.path.exists(stale):
                    os.remove(stale)
            best = best[:c["save_topk_models"]]
    model.history = history
    fit_model.last_model = model
    return best[0][1] if best else ""
