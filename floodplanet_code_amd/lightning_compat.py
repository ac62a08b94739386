"""pytorch_lightning is absent from the build image (and may be absent on the GPU box).  When it can be
imported the plugin classes subclass the real ``pl.LightningModule``; otherwise they subclass this minimal
stand-in, which keeps the attribute / hook surface the reference's callers use
(st_water_seg/fit.py:86-97, predict.py:164-178,236-240):  log_dict / log, logger, current_epoch,
global_step, load_from_checkpoint(path, **ctor_kwargs), automatic_optimization."""
from __future__ import annotations

from typing import Any, Dict

import torch
import torch.nn as nn

try:  # pragma: no cover - not available in this image
    import pytorch_lightning as pl  # type: ignore
    LightningModule = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # ModuleNotFoundError in this image
    pl = None
    HAVE_LIGHTNING = False

    class LightningModule(nn.Module):  # type: ignore[no-redef]
        automatic_optimization = True

        def __init__(self):
            super().__init__()
            self.logged: Dict[str, Any] = {}
            self.logger = None
            self.current_epoch = 0
            self.global_step = 0
            self.trainer = None

        def log(self, name, value, **kwargs):
            self.logged[name] = value

        def log_dict(self, metrics, **kwargs):
            for k, v in dict(metrics).items():
                self.logged[k] = v

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, **kwargs):
            ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=False)
            model = cls(**kwargs)
            model.load_state_dict(ckpt["state_dict"] if "state_dict" in ckpt else ckpt)
            return model

        @property
        def device(self):
            for p in self.parameters():
                return p.device
            return torch.device("cpu")
