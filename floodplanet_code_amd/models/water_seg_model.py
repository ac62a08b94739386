"""WaterSegmentationModel -- MI355X drop-in for st_water_seg/models/water_seg_model.py:14-240.

Same constructor, same Lightning hook set, same state_dict keys (``model.<unet key>``), same logged metric
names; the network, the loss, the backward pass and the metric counters run in libfloodunet's HIP kernels
(floodplanet_code_amd.unet.HipUNet).  Differences, all deliberate:
  * ``training_step`` uses the fused forward+CrossEntropy kernel path; the returned loss is a torch scalar
    whose ``backward()`` (issued by Lightning's automatic optimisation, fit.py:95-97) runs the HIP backward.
  * metrics come from the confusion counts the loss kernel emits (floodplanet_code_amd.metrics), not from
    torchmetrics (absent; parity unpinned).
  * ``ignore_index=None`` is accepted and means "ignore nothing" (-100), where the reference's
    nn.CrossEntropyLoss(ignore_index=None) fails at call time.
Extra keyword arguments (not in the reference): ``precision`` ('fp32' | 'bf16' | 'fp16'), ``base_channels``.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from ..lightning_compat import LightningModule
from ..metrics import SegmentationMetrics
from ..unet import HipAdam, HipUNet


class WaterSegmentationModel(LightningModule):

    def __init__(self, in_channels, n_classes, lr, log_image_iter=50, to_rgb_fcn=None, ignore_index=None,
                 optimizer_name='adam', precision='fp32', base_channels=64):
        super().__init__()
        self.lr = lr
        self.n_classes = n_classes
        self.in_channels = in_channels
        self.ignore_index = ignore_index
        self.optimizer_name = optimizer_name
        self.precision = precision
        self.base_channels = base_channels

        self._build_model()

        if self.ignore_index == -1:                      # water_seg_model.py:35-36
            self.ignore_index = self.n_classes - 1
        self.tracked_metrics = self._get_tracked_metrics()

        self._loss_ignore = -100 if self.ignore_index is None else int(self.ignore_index)
        self.loss_func = nn.CrossEntropyLoss(ignore_index=self._loss_ignore)   # kept for API parity (:40)

        self.to_rgb_fcn = to_rgb_fcn
        self.log_image_iter = log_image_iter

    # ------------------------------------------------------------------ construction
    def _get_tracked_metrics(self, average_mode='micro'):
        metrics = SegmentationMetrics(self.n_classes, self.ignore_index)
        self.train_metrics = metrics.clone(prefix='train_')
        self.valid_metrics = metrics.clone(prefix='val_')
        self.test_metrics = metrics.clone(prefix='test_')
        # the reference returns None here (:46-63), leaving tracked_metrics = None

    def _n_input_channels(self):
        if type(self.in_channels) is dict:
            return sum(self.in_channels.values())
        # water_seg_model.py:81-85 leaves n_in_channels unbound for a non-dict argument
        raise UnboundLocalError("local variable 'n_in_channels' referenced before assignment")

    def _build_model(self):
        self.model = HipUNet(self._n_input_channels(), self.n_classes, bilinear=True,
                             base_channels=self.base_channels, precision=self.precision)

    # ------------------------------------------------------------------ forward
    def _gather_input(self, batch):
        return batch['image']                            # water_seg_model.py:88

    def _gather_sources(self, batch):
        """What the network is fed: a tensor, or a list of tensors it sees side by side along C (EarlyFusionModel)."""
        return self._gather_input(batch)

    def forward(self, batch):
        return self.model(self._gather_sources(batch))

    def _set_model_to_train(self):
        self.model.train()

    def _set_model_to_eval(self):
        self.model.eval()

    def _fused_loss(self, batch, want_logits=False):
        """forward + CE(ignore_index) + NaN guard + argmax + confusion counts in the fused kernels.  The fp32 NCHW logits
        are only written when someone needs them: training_step's image logging is disabled in the reference
        (`if False:`, water_seg_model.py:116), so the training path never does; the counts stay on the device."""
        images = self._gather_sources(batch)
        out = self.model.loss(images, batch['target'], self._loss_ignore, return_logits=want_logits)
        loss, output = out if want_logits else (out, None)
        counts = self.model.pop_confusion()
        return loss, output, counts

    # ------------------------------------------------------------------ Lightning hooks
    def training_step(self, batch, batch_idx):
        self._set_model_to_train()
        loss, output, counts = self._fused_loss(batch)   # CE + NaN guard (:103-106) live in the kernel
        metric_output = self.train_metrics.update_from_counts(counts)
        self.log_dict(metric_output, prog_bar=True, on_step=True, on_epoch=True)
        return loss

    def validation_step(self, batch, batch_idx):
        self._set_model_to_eval()
        with torch.no_grad():
            loss, output, counts = self._fused_loss(batch)
        metric_output = self.valid_metrics.update_from_counts(counts)
        self.valid_metrics.update_from_counts(counts)    # the reference counts each batch twice (:150-151)
        metric_output['valid_loss'] = loss
        self.log_dict(metric_output, prog_bar=True, on_step=True, on_epoch=True)

    def test_step(self, batch, batch_idx):
        self._set_model_to_eval()
        with torch.no_grad():
            loss, output, counts = self._fused_loss(batch)
        self.test_metrics.update_from_counts(counts)
        self.log_dict({'test_loss': loss}, prog_bar=True, on_step=True, on_epoch=True)

    def configure_optimizers(self):
        if self.optimizer_name == 'adam':
            # optim.Adam(self.parameters(), lr=self.lr) (water_seg_model.py:200) as ONE fused kernel launch on the
            # network's flat buffers; same hyper-parameters, param_groups and state_dict layout (unet.HipAdam).
            # FU_TORCH_ADAM=1 keeps torch's own optimiser (it works on the same parameters).
            import os
            if os.environ.get("FU_TORCH_ADAM") == "1":
                optimizer = optim.Adam(self.parameters(), lr=self.lr)
            else:
                optimizer = HipAdam(self.model, lr=self.lr)
        else:
            raise NotImplementedError(f'No implementation for optimizer of name: {self.optimizer_name}')
        return optimizer

    def validation_epoch_end(self, validation_step_outputs):
        if len(validation_step_outputs) == 0:
            self.test_f1_score = 0
            self.test_iou = 0
            self.test_acc = 0
        else:
            metric_output = self.valid_metrics.compute()
            self.log_dict(metric_output)

    def test_epoch_end(self, test_step_outputs) -> None:
        if len(test_step_outputs) == 0:
            return
        metric_output = self.test_metrics.compute()
        self.log_dict(metric_output)
        self.f1_score = metric_output['test_MulticlassF1Score'].item()
        self.acc = metric_output['test_MulticlassAccuracy'].item()
        self.iou = metric_output['test_MulticlassJaccardIndex'].item()

    def log_image_to_tensorflow(self, str_title, rgb_image, cm_image):
        """rgb_image, cm_image: np.array [height, width, 3]; stacked vertically and logged CHW (:227-240)."""
        log_image = np.concatenate((rgb_image, cm_image), axis=0).transpose((2, 0, 1))
        self.logger.experiment.add_image(str_title, log_image, self.global_step)
