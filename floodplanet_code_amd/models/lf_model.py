"""LateFusionModel -- drop-in for st_water_seg/models/lf_model.py:9-92 (feat_fusion='concat_conv'): one UNetEncoder per
input, the features concatenated level by level and fused by a 1x1 conv, one UNetDecoder; all of it runs in
libfloodunet.so (floodplanet_code_amd.latefusion.HipLateFusion).

state_dict keys are the reference's -- ``encoders.<name>. ...``, ``decoder. ...``, ``concat_convs.<level>. ...`` at the
top level of the module (lf_model.py:31-45 registers them on the LightningModule itself): the network lives in
``self.model`` like in the other plugins and the ``model.`` prefix is stripped / added by state-dict hooks."""
from __future__ import annotations

import torch

from ..latefusion import HipLateFusion
from .water_seg_model import WaterSegmentationModel

_TOP = ("encoders.", "decoder.", "concat_convs.")
_BATCH_KEY = {"ms_image": "image"}          # lf_model.py:56: encoders['ms_image'](batch['image']); others by their name


class LateFusionModel(WaterSegmentationModel):

    def __init__(self, in_channels, n_classes, lr, log_image_iter=50, to_rgb_fcn=None, ignore_index=None,
                 optimizer_name='adam', feat_fusion='concat_conv', precision='fp32', base_channels=64):
        self.feat_fusion = feat_fusion
        super().__init__(in_channels, n_classes, lr, log_image_iter=log_image_iter, to_rgb_fcn=to_rgb_fcn,
                         ignore_index=ignore_index, optimizer_name=optimizer_name, precision=precision,
                         base_channels=base_channels)

    def _build_model(self):
        if self.feat_fusion != 'concat_conv':
            raise NotImplementedError                 # lf_model.py:84-85 (raised in forward there)
        self.model = HipLateFusion(self.in_channels, self.n_classes, base_channels=self.base_channels,
                                   precision=self.precision)
        self._register_state_dict_hook(self._strip_model_prefix)
        self._register_load_state_dict_pre_hook(self._add_model_prefix)

    @staticmethod
    def _strip_model_prefix(module, state_dict, prefix, local_metadata):
        for k in [k for k in state_dict if k.startswith(prefix + "model.")]:
            state_dict[prefix + k[len(prefix) + 6:]] = state_dict.pop(k)
        return state_dict

    @staticmethod
    def _add_model_prefix(state_dict, prefix, *args):
        for k in [k for k in state_dict if k[len(prefix):].startswith(_TOP) and k.startswith(prefix)]:
            state_dict[prefix + "model." + k[len(prefix):]] = state_dict.pop(k)

    def _gather_input(self, batch):
        # the inputs side by side, in the order lf_model.py:56-76 runs the encoders and concatenates their features
        return torch.concat([batch[_BATCH_KEY.get(k, k)] for k in self.model.encoder_names], dim=1)
